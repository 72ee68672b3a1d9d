#!/bin/bash
# What the separate ordered-mean pass costs each instantiation in the pipelined loop: builds with and without it (-DTRT_AB_SKIP_REDUCE=1;
# frames NOT verified).   usage: gpurun -- bash tools/gpu_noreduce.sh with.so without.so
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/noreduce; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-verify --no-configs --no-moving-camera"
for r in 1 2; do
for lib in "$@"; do
  for c in 0 1; do
    for mode in "" "--animation 60"; do
    [ -n "$mode" ] && [ $c = 1 ] && continue
    TRT_HIP_LIB=$PWD/$lib TRT_COMPACTION=$c timeout -k 10 200 $B $mode 2> $O/err.log | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-26s compaction $c %-14s %.3f G  ms/step %.4f  %s'%('$lib', '$mode', j['value']/1e9, j['ms_per_step'], j['roofline']['kernel']))" || { tail $O/err.log; exit 1; }
    done
  done
done
done
