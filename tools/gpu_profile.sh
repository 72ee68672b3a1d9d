#!/bin/bash
# ISA profile of the shipping kernels on the GPU box (build/isa_profile.so from tools/build_isa_profile.sh): config 3 plain and
# decoupled (a 251-register profile build fits no 1024-thread workgroup: build with `tools/build_isa_profile.sh -DTRT_COMPACT_BLOCK=256` for that
# one), config 5.   usage: gpurun -- bash tools/gpu_profile.sh [lib]      -> gpurun_out/isa_profile/*.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
lib=${1:-build/isa_profile.so}
out=gpurun_out/isa_profile; mkdir -p $out
tag=$(basename $lib .so)
TRT_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 tools/isa_profile.py 1920 1080 64 8 0 > $out/${tag}_c3_plain.txt 2>$out/err.log || { tail -20 $out/err.log; exit 1; }
TRT_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 tools/isa_profile.py 1920 1080 64 8 1 > $out/${tag}_c3_decoupled.txt 2>$out/err.log || { tail -20 $out/err.log; exit 1; }
TRT_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 tools/isa_profile.py 1920 1080 256 12 0 > $out/${tag}_c5.txt 2>$out/err.log || { tail -20 $out/err.log; exit 1; }
cat $out/${tag}_c3_plain.txt $out/${tag}_c3_decoupled.txt $out/${tag}_c5.txt
