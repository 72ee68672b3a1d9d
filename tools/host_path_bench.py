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
