#!/bin/bash
# the round's tables in one call: ISA profile of the shipping kernels, every BASELINE config on one GPU, scenes of more than 256 spheres,
# the shards of an 8-way split     usage: gpurun -- bash tools/gpu_round_tables.sh     -> gpurun_out/tables/*
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/tables; mkdir -p $O
bash tools/gpu_profile.sh > $O/k_instruction_counts.txt 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
timeout -k 10 500 python3 tools/run_configs.py > $O/i_all_configs_one_gpu.md 2>> $O/err.log || { tail -5 $O/err.log; exit 1; }
timeout -k 10 600 python3 tools/big_scene.py 256 300 512 700 1024 --both > $O/i_big_scenes.md 2>> $O/err.log || { tail -5 $O/err.log; exit 1; }
timeout -k 10 300 python3 tools/shard_bench.py > $O/j_shards.txt 2>> $O/err.log || { tail -5 $O/err.log; exit 1; }
tail -12 $O/i_all_configs_one_gpu.md; cat $O/i_big_scenes.md; tail -12 $O/j_shards.txt
