"""A/B of the shading compaction (trt_set_compaction): frames with it on must equal frames with it off, bit for bit; time both."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import support as T
from terminalraytracer_amd import hip
import test_gpu_parity as P

import torch

MODES = {0: P.PLAIN, 1: P.COMPACT}

def timed(ctx, scene, w, h, b, spp, mode, reps=20):
    """median duration of the render kernel over whole frames rendered into device memory (no bands, no copies)"""
    ctx.set_scene(scene)
    ctx.set_compaction(mode)
    buf = torch.empty(w * h * 3, dtype=torch.float64, device="cuda:0")
    rows = hip.RowSet.whole(w, h)
    ts = []
    for _ in range(reps + 3):
        ctx.render_device(scene.camera, rows, b, spp, buf.data_ptr(), buf.numel() * 8)
        ctx.synchronize()
        ts.append(ctx.render_kernel_times()[0][-1])
    return float(np.median(ts[3:]))

def stats(ctx, scene, w, h, b, spp, mode):
    ctx.enable_counters(True)
    try:
        P.render(ctx, scene, w, h, b, spp, MODES[mode])
        path, shadow = ctx.read_counters()
        d = ctx.read_diagnostics()
    finally:
        ctx.enable_counters(False)
    lights = len(scene.dir_lights) + len(scene.point_lights)
    hits = shadow / max(lights, 1)
    return (f"path {path} hits {hits:.0f} rounds {d['wave_loop_trips']} passes {d['shading_passes']} "
            f"P activity {path / (64.0 * d['wave_loop_trips']):.3f} S activity {hits / (64.0 * max(d['shading_passes'], 1)):.3f}")

bad = 0
with hip.Context(0) as ctx:
    ctx.set_path_grids_min_spheres(0)
    for seed in range(100, 160):
        rng = np.random.default_rng(seed)
        w, h = int(rng.integers(8, 160)), int(rng.integers(4, 90))
        b, spp = int(rng.integers(1, 13)), int(rng.choice([1, 3, 10]))
        scene = P._fuzz_scene(rng, w, h)
        with np.errstate(all="ignore"):
            want, st = T.oracle_render(scene, w, h, b, spp)
        for mode in (0, 1):
            got = P.render(ctx, scene, w, h, b, spp, MODES[mode])
            finite = np.isfinite(want)
            ok = np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(P.bits(got[finite]), P.bits(want[finite]))
            if not ok:
                bad += 1
                print(f"seed {seed} mode {mode}: MISMATCH {int((P.bits(got) != P.bits(want)).sum())} values ({w}x{h}, {len(scene.spheres)} spheres, B{b}, spp{spp})", flush=True)
    print("fuzz 60 scenes x 2 modes:", bad, "mismatches", flush=True)
    ctx.set_path_grids_min_spheres(12)
    from terminalraytracer_amd import scenes as S
    for name in ("c3_1080p_64sph_b8", "c3_1080p_64sph_b8+1lights", "c3_1080p_64sph_b8+2lights", "c3_1080p_64sph_b8+4lights", "c3_1080p_64sph_b8+6lights",
                 "c2_1080p_8sph_b4", "c2_1080p_8sph_b4+2lights"):
        case = T.golden_full()[name.split("+")[0]]
        w, h, b, spp = case["width"], case["height"], case["bounce_limit"], case["rays_per_pixel"]
        scene = T.full_scene(case)
        if "+" in name:
            extra = int(name.split("+")[1].replace("lights", ""))
            rng = np.random.default_rng(5)
            nd, npt = extra // 2, extra - extra // 2
            dl = np.concatenate([rng.normal(size=(nd, 3)) - [0, 1.5, 0], rng.uniform(0.1, 0.5, (nd, 3))], axis=1)
            pl = np.concatenate([rng.normal(size=(npt, 3)) * 4 + [0, 4, 0], rng.uniform(0.1, 0.5, (npt, 3)), rng.uniform(5, 30, (npt, 1))], axis=1)
            scene = S.SceneData(scene.spheres, scene.ground, np.concatenate([scene.dir_lights, dl]), np.concatenate([scene.point_lights, pl]),
                                scene.camera, scene.sky)
        frames = {}
        for mode in (0, 1):
            frames[mode] = P.render(ctx, scene, w, h, b, spp, MODES[mode])
            t = timed(ctx, scene, w, h, b, spp, mode)
            print(f"{name} compaction {mode}: render kernel {t:.4f} ms   {stats(ctx, scene, w, h, b, spp, mode)}", flush=True)
        same = np.array_equal(P.bits(frames[0]), P.bits(frames[1]))
        print(f"{name}: on == off: {same}", flush=True)
        bad += 0 if same else 1
sys.exit(1 if bad else 0)
