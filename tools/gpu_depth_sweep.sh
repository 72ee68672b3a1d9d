#!/bin/bash
# Config 3, plain and decoupled instantiation, 1 ... 6 frames in flight.   usage: gpurun -- bash tools/gpu_depth_sweep.sh [lib]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/depth; mkdir -p $O
[ -n "$1" ] && export TRT_HIP_LIB=$PWD/$1
B="python3 bench.py --no-cpu-baseline --no-verify --no-configs --no-moving-camera"
for c in 0 1; do
  for depth in 1 2 3 4 6; do
    TRT_COMPACTION=$c timeout -k 10 200 $B --depth $depth 2> $O/err.log | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('compaction $c depth $depth  %.3f G  ms/step %.4f  device %.4f  %s'%(j['value']/1e9, j['ms_per_step'], j.get('device_ms_per_step') or 0, j['roofline']['kernel']))" || { tail $O/err.log; exit 1; }
  done
done
