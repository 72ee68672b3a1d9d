"""When do the spheres' patches pay?  Render kernel time (one frame at a time, HIP events) of SYNTH-v0 scenes of n spheres at
1920x1080, 12 bounces, with one family per sphere (m = 0; the shading decoupled where the rings fit, as the library would) and with
24 patches (m = 2).  usage (GPU box): python tools/archive/patch_policy.py [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from terminalraytracer_amd import hip, scenes as S

ns = [int(x) for x in sys.argv[1:]] or [64, 96, 128, 160, 192, 256]
w, h, b = 1920, 1080, 12
fb = torch.zeros(h * w * 3, dtype=torch.float64, device="cuda:0")
with hip.Context(0) as ctx:
    for n in ns:
        scene = S.synth_scene(n, S.synth_sky(256), S.orbit_camera(1.0, w, h))
        row = []
        for m in (0, 2):
            ctx.set_path_patches(m)
            ctx.set_scene(scene)
            for _ in range(7):
                ctx.render_device(scene.camera, hip.RowSet.whole(w, h), b, 10, fb.data_ptr(), fb.numel() * 8)
                ctx.synchronize()
            ms = float(np.mean(ctx.render_kernel_times(5)[0]))
            row.append((m, ms, ctx.render_variant()["decoupled"]))
        print(f"n {n:4d}: " + "   ".join(f"m {m}: {ms:.3f} ms{' (decoupled)' if dec else ''}" for m, ms, dec in row) +
              f"   patches/none {row[1][1] / row[0][1]:.3f}", flush=True)
