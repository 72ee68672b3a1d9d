"""PCIe-inclusive rate of the drop-in boundary: trt_render_frame with host Scene* in, host Screen* out (1920x1080,
north-star scene, 8 bounces, 10 rays per pixel), i.e. what a caller of project_scene pays per frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from terminalraytracer_amd import hip
scene = bench.build_scene()
import ctypes as C
from terminalraytracer_amd import scenes as S
sc = scene.as_scene()
screen, px = S.new_screen(bench.W, bench.H)  # one caller-owned (pageable) framebuffer, reused like main()'s
call = lambda: hip._check(hip.lib().trt_render_frame(C.byref(sc), C.byref(screen), bench.BOUNCES, bench.SPP))
for _ in range(3):
    call()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    call()
dt = (time.perf_counter() - t0) / n
print(f"trt_render_frame host-in/host-out: {dt * 1e3:.2f} ms/frame  ({px.nbytes / 1e6:.1f} MB framebuffer copied back per frame)")

# the same with a sphere moving every frame: the library has to rebuild its light-space tables per call
t0 = time.perf_counter()
for i in range(n):
    scene.spheres[0, 1] += 1e-3
    sc = scene.as_scene()
    call()
dt = (time.perf_counter() - t0) / n
print(f"... with one sphere moved before every call (tables rebuilt): {dt * 1e3:.2f} ms/frame")

# the 256-sphere scene of config 5 (24 patches per sphere: ~0.1 s of table build): a sphere moved before every call.  From the
# second changed call on the drop-in layer treats the scene as moving and builds the cheap tables per call (trt_set_scene_policy)
scene = bench.build_scene("c5")
sc = scene.as_scene()
call5 = lambda: hip._check(hip.lib().trt_render_frame(C.byref(sc), C.byref(screen), 12, bench.SPP))
for policy, label in (((2, 3), "moving-scene policy on (default)"), ((0, 3), "policy off: every change builds the full tables")):
    hip._check(hip.lib().trt_set_scene_policy(*policy))
    for _ in range(6):  # a new scene, then still for three calls: the full tables are in place before anything is timed
        call5()
    t0 = time.perf_counter()
    for _ in range(10):
        call5()
    still = (time.perf_counter() - t0) / 10
    per = []
    for i in range(12):
        scene.spheres[0, 1] += 1e-3
        sc = scene.as_scene()
        t0 = time.perf_counter()
        call5()
        per.append(time.perf_counter() - t0)
    print(f"256 spheres, 1080p, 12 bounces, {label}: unchanged scene {still * 1e3:.2f} ms/call; a sphere moved before every call: "
          f"first {per[0] * 1e3:.1f} ms, then median {np.median(per[2:]) * 1e3:.2f} ms/call")
hip._check(hip.lib().trt_set_scene_policy(2, 3))
