#!/bin/bash
# A/B of library builds on the GPU box: bench.py (200 frames, three in flight; frames verified) on config 3 and config 5, interleaved
# rounds.  usage: gpurun -- bash tools/gpu_ab.sh [-r ROUNDS] lib [lib ...]     (paths relative to the repo root)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rounds=2
if [ "$1" = "-r" ]; then rounds=$2; shift 2; fi
out=gpurun_out/ab; mkdir -p $out
for r in $(seq $rounds); do
for lib in "$@"; do
  for mode in "" "--animation 60"; do
    TRT_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-configs --no-moving-camera $mode 2> $out/err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['diagnostics']
print('%-28s %-14s %7.3f G  ms/step %.4f  d1 %.4f ms  verified %s  vgprs %d  iters %s act %s'%('$lib','$mode' or 'c3', d['value']/1e9, d['ms_per_step'], d['one_frame_at_a_time']['render_kernel_ms'], d['verified'], d['kernel_info']['vgprs'], {k: round(v/max(1,g['wave_loop_trips']),2) for k,v in g.get('exact_loop_iterations',{}).items()}, g.get('exact_loop_lane_activity')))" || { tail -5 $out/err.log; exit 1; }
  done
done
done
