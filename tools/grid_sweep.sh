#!/bin/bash
# light-table resolution sweep: north-star config (bench.py) and the 256-sphere config (tools/stamp_config.py prints nothing
# without the stamped build, so time it through run_one below)
for g in "64,32" "128,64" "256,128" "512,256"; do
  TRT_LIGHTGRID=$g python bench.py --no-cpu-baseline --depth 1 --steps 15 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C3 grids $g', 'ms %.4f' % d['ms_per_step'], 'phase2/trace %.3f' % d['diagnostics']['exact_test_rounds_per_trace'])"
  TRT_LIGHTGRID=$g python - <<PY
import time, torch, sys
sys.path.insert(0, '.')
from terminalraytracer_amd import hip, scenes as S
w, h, n, b = 1920, 1080, 256, 12
scene = S.synth_scene(n, S.synth_sky(256), S.orbit_camera(1.0, w, h))
with hip.Context(0) as ctx:
    ctx.set_scene(scene)
    fb = torch.zeros(w * h * 3, dtype=torch.float64, device='cuda:0')
    rs = hip.RowSet.whole(w, h)
    for i in range(13):
        if i == 3:
            torch.cuda.synchronize(); ctx.synchronize(); t0 = time.perf_counter()
        ctx.render_device(scene.camera, rs, b, 10, fb.data_ptr(), fb.numel() * 8)
    ctx.synchronize()
    print('C5 grids $g ms %.3f' % ((time.perf_counter() - t0) / 10 * 1e3))
PY
done
