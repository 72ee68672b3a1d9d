#!/bin/bash
# What the sample scratch costs in the pipelined steady state: the timed loop of bench.py with (a) the
# ordered-mean kernel skipped, (b) the render kernel's sample stores redirected to 48 KB, (c) both -- diagnostic builds
# (make lib LIB=build/ab_*.so TUNE=-DTRT_AB_...), frames not verified (they are wrong by construction).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/scratch_ab
for r in 1 2 3; do
for lib in terminalraytracer_amd/libtrt_hip.so build/ab_noreduce.so build/ab_dummy.so build/ab_both.so; do
  for depth in 3 1; do
  TRT_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-verify --no-configs --steps 40 --depth $depth 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib depth $depth round $r', 'ms/step %.4f  render kernel one at a time %.4f ms  reduce %.4f ms'%(d['ms_per_step'], d['one_frame_at_a_time']['render_kernel_ms'], d['one_frame_at_a_time']['reduce_kernel_ms']))"
  done
done
done | tee gpurun_out/scratch_ab/ab.txt
