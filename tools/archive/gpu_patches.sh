#!/bin/bash
# Sub-families of the spheres (trt_raygrid.h patches): parity tests of the tables, then config 5 / config 3 with the tables'
# resolution and the number of patches from the environment (TRT_PATHGRID="eye,sphere,m").  Output under gpurun_out/patches/.
# usage: gpurun -- bash tools/archive/gpu_patches.sh [tests|notests]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/patches; mkdir -p $out
if [ "${1:-tests}" = tests ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q \
    -k "single_ray_vectors or never_change_a_frame or host_reference_builder or baseline_configs or serve_nearly" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
  tail -3 $out/tests.log
fi
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', 'ms/step %.3f' % d['ms_per_step'], 'G path rays/s %.2f' % (d['value']/1e9), 'kernel 1-at-a-time %.3f ms' % d['one_frame_at_a_time']['render_kernel_ms'], 'verified', d['verified'], 'exact rounds/trace %.2f' % d['diagnostics']['exact_test_rounds_per_trace'], 'swept', d['diagnostics']['swept_traces'], d['roofline']['kernel'])
"; }
for pg in ${C5_GRIDS:-64,32,0 64,32,2 64,16,2 64,24,2 64,16,3 64,16,4 64,32,1}; do
  TRT_PATHGRID=$pg timeout -k 10 300 python bench.py --animation 60 --no-cpu-baseline 2> $out/c5_$pg.err | tee $out/c5_$pg.json | line "C5 $pg" || { tail -5 $out/c5_$pg.err; exit 1; }
done
for pg in ${C3_GRIDS:-64,32,0 64,32,2 64,16,2 64,32,1}; do
  TRT_PATHGRID=$pg timeout -k 10 300 python bench.py --no-cpu-baseline 2> $out/c3_$pg.err | tee $out/c3_$pg.json | line "C3 $pg" || { tail -5 $out/c3_$pg.err; exit 1; }
done
