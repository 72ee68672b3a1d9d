"""Parameter extremes against the oracle on the GPU box (ad-hoc companion of tests/test_gpu_parity.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import support as T
from terminalraytracer_amd import hip, scenes as S
import test_gpu_parity as P

cases = [(1, 1, 6, 4, 10), (1, 300, 17, 3, 10), (4000, 1, 64, 5, 3), (64, 64, 64, 64, 2), (48, 27, 64, 3, 257), (16, 9, 500, 6, 4),
         (33, 19, 1000, 4, 3), (20, 10, 0, 8, 10), (2, 2, 1, 1, 1)]
bad = 0
with hip.Context(0) as ctx:
    for (w, h, n, b, spp) in cases:
        spheres = S.demo_spheres() if n == 6 else S.synth_spheres(n, seed=7)
        scene = S.synth_scene(max(n, 1), T.sky("synth"), T.bench_camera(w, h, 2.5), seed=7).with_spheres(spheres)
        try:
            got = P.render(ctx, scene, w, h, b, spp)
        except Exception as e:
            print(f"{w}x{h} N={n} B={b} spp={spp}: library said: {e}")
            continue
        want, st = T.oracle_render(scene, w, h, b, spp)
        ok = np.array_equal(P.bits(got), P.bits(want))
        bad += not ok
        print(f"{w}x{h} N={n} B={b} spp={spp}: {'equal' if ok else 'MISMATCH'}  ({st.path_rays} path rays)")
sys.exit(1 if bad else 0)
