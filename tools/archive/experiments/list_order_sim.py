"""EXPERIMENT (round 4): candidate lists ordered by distance from the family's apex, lanes stopping early -- a host simulation on
the oracle's path rays (tools/archive/experiments/list_order_sim.c).  Usage: python tools/archive/experiments/list_order_sim.py [spheres] [g_sph] [m]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import support as T  # noqa: E402
from terminalraytracer_amd import scenes as S  # noqa: E402
from test_filter import traced_rays  # noqa: E402


class OrderStats(C.Structure):
    _fields_ = [(k, C.c_ulonglong) for k in ("rays", "members", "tests_now", "tests_sorted", "tests_sorted_q", "wave_now", "wave_sorted",
                                             "wave_sorted_q", "groups")] + [("hist_now", C.c_ulonglong * 33), ("hist_sorted", C.c_ulonglong * 33)]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    g_sph = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    w, h, b = (96, 54, 12) if n > 64 else (240, 135, 8)
    so = "/tmp/liblistordersim.so"
    inc = os.path.join(ROOT, "terminalraytracer_amd", "csrc")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-I" + inc, "-o", so,
                           os.path.join(ROOT, "tools", "archive", "experiments", "list_order_sim.c"), "-lm"])
    lib = C.CDLL(so)
    lib.order_sim.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                              C.c_double, C.c_int, C.POINTER(OrderStats)]
    lib.order_sim.restype = None
    scene = S.synth_scene(n, T.sky("synth"), T.bench_camera(w, h))
    rays, kinds = traced_rays(scene, w, h, b, 10)
    sph = np.ascontiguousarray(scene.spheres, dtype=np.float64)
    ground = np.ascontiguousarray(scene.ground, dtype=np.float64)
    eye = np.ascontiguousarray(scene.camera[9:12], dtype=np.float64)
    rays = np.ascontiguousarray(rays)
    kinds = np.ascontiguousarray(kinds)
    for plane_first in (0, 1):
        st = OrderStats()
        lib.order_sim(sph.ctypes.data, n, ground.ctypes.data, eye.ctypes.data, rays.ctypes.data, kinds.ctypes.data, rays.shape[0], 64, g_sph, m,
                      0.0625, plane_first, C.byref(st))
        mm, gg = max(st.members, 1), max(st.groups, 1)
        print(f"n {n} g_sph {g_sph} m {m} plane_first {plane_first}: rays {st.rays} members {st.members}\n"
              f"  tests/ray  now {st.tests_now / mm:.2f}  sorted {st.tests_sorted / mm:.2f}  sorted, keys of 1/16 {st.tests_sorted_q / mm:.2f}\n"
              f"  max of 64  now {st.wave_now / gg:.2f}  sorted {st.wave_sorted / gg:.2f}  sorted, keys of 1/16 {st.wave_sorted_q / gg:.2f}\n"
              f"  hist now    {list(st.hist_now)[:20]}\n  hist sorted {list(st.hist_sorted)[:20]}")


if __name__ == "__main__":
    main()
