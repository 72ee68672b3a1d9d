"""Measure the five BASELINE.json configs on ONE MI355X (C1..C5) and print a markdown table.
Three frames in flight through trt_dist_* as in bench.py; ray counts from the counting kernel variant in an untimed pass."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from terminalraytracer_amd import hip, scenes as S
from terminalraytracer_amd.distributed import HipShardRenderer


def measure(name, scene, w, h, b, frames, cameras=None):
    """the calls bench.py makes: trt_dist_* with one rank, three frames in flight"""
    d = hip.Dist(0, scene, None, 0, 1, w, h, tile_rows=8, frames_in_flight=3)
    ctx0 = d.context(0)
    fb = torch.zeros(h * w * 3, dtype=torch.float64, device="cuda:0")
    cams = cameras or [scene.camera] * frames
    ctx0.enable_counters(True)
    counts = []
    for c in (cameras or [scene.camera]):  # the same basis as bench.py: every camera of an orbit has its own ray count
        ctx0.render_device(c, hip.RowSet.whole(w, h), b, 10, fb.data_ptr(), fb.numel() * 8)
        counts.append(ctx0.read_counters())
    path = sum(counts[i % len(counts)][0] for i in range(len(cams))) / len(cams)
    shadow = sum(counts[i % len(counts)][1] for i in range(len(cams))) / len(cams)
    ctx0.enable_counters(False)
    del fb
    for c in cams[:3]:
        d.render(c, b, 10)
    d.synchronize()
    t0 = time.perf_counter()
    for c in cams:
        d.render(c, b, 10)
    d.synchronize()
    dt = (time.perf_counter() - t0) / len(cams)
    variant = ctx0.render_variant()
    kernel = "decoupled" if variant["decoupled"] else ("plain, %d patches" % ctx0.path_patches()[1] if ctx0.path_patches()[0] else "plain")
    d.close()
    print(f"| {name} | {w}x{h} | {scene.num_spheres} | {b} | {dt * 1e3:.3f} | {1 / dt:.0f} | {path / dt / 1e9:.2f} | {(path + shadow) / dt / 1e9:.2f} | {path / 1e6:.1f} M | {kernel} |")


def main():
    print("| config | frame | spheres | bounces | ms/frame | frames/s | G path rays/s | G all rays/s | path rays/frame | rounds |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    sky = S.synth_sky(256)
    cam = lambda w, h, t=1.0: S.orbit_camera(t, w, h)
    measure("C1 demo scene (as BASELINE words it)", S.demo_scene(sky, S.orbit_camera(1.0, 160, 48, reference_aspect=True), 3), 160, 48, 4, 200)
    measure("C1' demo scene as written in main()", S.demo_scene(sky, S.orbit_camera(1.0, 480, 280, reference_aspect=True)), 480, 280, 10, 200)
    measure("C2", S.synth_scene(8, sky, cam(1920, 1080)), 1920, 1080, 4, 60)
    measure("C3 (north star)", S.synth_scene(64, sky, cam(1920, 1080)), 1920, 1080, 8, 60)
    measure("C3 harsher: 25 % of the spheres are perfect mirrors", S.synth_scene(64, sky, cam(1920, 1080), mirror_fraction=0.25), 1920, 1080, 8, 60)
    measure("C4 on one GPU", S.synth_scene(64, sky, cam(3840, 2160)), 3840, 2160, 8, 20)
    anim = [cam(1920, 1080, f / 60.0) for f in range(60)]
    measure("C5 60-frame orbit on one GPU", S.synth_scene(256, sky, cam(1920, 1080)), 1920, 1080, 12, 60, anim)


if __name__ == "__main__":
    main()
