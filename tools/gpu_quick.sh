#!/bin/bash
# Quick loop on the GPU box: a -k selection of the parity tests (or "none"), then bench.py on config 3 and config 5 for the shipped
# library and any alternative builds given (make lib LIB=build/x.so TUNE=...), one line each with the loop diagnostics.
# usage: gpurun -- bash tools/gpu_quick.sh "<pytest -k expression | none | all>" [lib ...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/quick; mkdir -p $out
sel="$1"; shift
if [ "$sel" != none ]; then
  if [ "$sel" = all ]; then k=""; else k="-k"; fi
  timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -m gpu -x -q $k ${k:+"$sel"} > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
  tail -3 $out/tests.log
fi
for lib in terminalraytracer_amd/libtrt_hip.so "$@"; do
  for mode in "" "--animation 60"; do
    TRT_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-configs $mode 2> $out/err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['diagnostics']
print('$lib $mode', 'ms/step %.3f  d1 kernel %.3f ms  %.2f G path rays/s  verified %s  vgprs %d  iters %s  activity %s  fallbacks %s swept %d'%(d['ms_per_step'], d['one_frame_at_a_time']['render_kernel_ms'], d['value']/1e9, d['verified'], d['kernel_info']['vgprs'], {k: round(v/max(1,g['wave_loop_trips']),2) for k,v in g.get('exact_loop_iterations',{}).items()}, g.get('exact_loop_lane_activity'), g.get('point_light_closest_hit_fallbacks'), g['swept_traces']))" || { tail -5 $out/err.log; exit 1; }
  done
done
