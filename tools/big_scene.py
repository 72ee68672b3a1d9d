"""Scenes of more than 256 spheres at 1920x1080, 12 bounces, 10 rays per pixel through the calls bench.py makes (three frames in
flight): ms per frame, path rays/s, the share of wave-level traces that swept, workgroups per CU.  Frames are checked against the
all-core oracle when --check is given (slow: minutes per frame at 1080p).
usage: python tools/big_scene.py SPHERES [SPHERES ...] [--no-path-tables] [--small]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from terminalraytracer_amd import hip, scenes as S

args = [a for a in sys.argv[1:] if not a.startswith("--")]
small = "--small" in sys.argv
w, h, b = (480, 270, 12) if small else (1920, 1080, 12)
print("| spheres | path tables | ms/frame | G path rays/s | swept traces / round | list entries | workgroups per CU | table MB | build s |")
print("|---|---|---|---|---|---|---|---|---|")
for n in (int(a) for a in args):
    scene = S.synth_scene(n, S.synth_sky(256), S.orbit_camera(1.0, w, h))
    for tables in ((True, False) if "--both" in sys.argv else (("--no-path-tables" not in sys.argv),)):
        d = hip.Dist(0, scene, None, 0, 1, w, h, tile_rows=8, frames_in_flight=3)
        ctxs = [d.context(i) for i in range(3)]
        if not tables:
            d.close()
            os.environ["TRT_PATHGRID"] = "0,0"
            d = hip.Dist(0, scene, None, 0, 1, w, h, tile_rows=8, frames_in_flight=3)
            ctxs = [d.context(i) for i in range(3)]
        c0 = ctxs[0]
        import torch
        fb = torch.zeros(h * w * 3, dtype=torch.float64, device="cuda:0")
        c0.enable_counters(True)
        c0.render_device(scene.camera, hip.RowSet.whole(w, h), b, 10, fb.data_ptr(), fb.numel() * 8)
        path, shadow = c0.read_counters()
        diag = c0.read_diagnostics()
        c0.enable_counters(False)
        for _ in range(3):
            d.render(scene.camera, b, 10)
        d.synchronize()
        frames = 12
        t0 = time.perf_counter()
        for _ in range(frames):
            d.render(scene.camera, b, 10)
        d.synchronize()
        dt = (time.perf_counter() - t0) / frames
        info = c0.scene_info()
        ki = c0.kernel_info()
        print("| %d | %s | %.3f | %.2f | %.3f | %s bits | %d | %.0f | %.2f |" % (n, "on" if tables else "off (every path ray sweeps, as in round 4)", dt * 1e3, path / dt / 1e9,
              diag["swept_traces"] / max(1, diag["wave_loop_trips"]), 16 if n > 256 else 8, ki["max_blocks_per_cu"], info["table_bytes"] / 1e6, info["build_seconds"]))
        d.close()
        os.environ.pop("TRT_PATHGRID", None)
