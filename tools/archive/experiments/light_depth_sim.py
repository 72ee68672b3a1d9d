"""EXPERIMENT (round 4): light tables with a depth coordinate, any-hit search for point lights -- host simulation on the oracle's
shadow rays (tools/archive/experiments/light_depth_sim.c).  Usage: python tools/archive/experiments/light_depth_sim.py [spheres]"""
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


class DepthStats(C.Structure):
    _fields_ = [(k, C.c_ulonglong) for k in ("rays", "cand_now", "cand_slab", "tests_now", "tests_any", "tests_slab_any", "wave_now", "wave_any",
                                             "wave_slab_any", "wave_cand_slab", "groups")]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    w, h, b = (96, 54, 12) if n > 64 else (240, 135, 8)
    so = "/tmp/liblightdepthsim.so"
    inc = os.path.join(ROOT, "terminalraytracer_amd", "csrc")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-I" + inc, "-o", so,
                           os.path.join(ROOT, "tools", "archive", "experiments", "light_depth_sim.c"), "-lm"])
    lib = C.CDLL(so)
    lib.depth_sim.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(DepthStats)]
    lib.depth_sim.restype = None
    scene = S.synth_scene(n, T.sky("synth"), T.bench_camera(w, h))
    rays, kinds = traced_rays(scene, w, h, b, 10)
    sph = np.ascontiguousarray(scene.spheres, dtype=np.float64)
    light = np.ascontiguousarray(scene.point_lights[0][:3], dtype=np.float64)
    for kind, g in ((1, 128), (2, 64)):
        for slabs in (1, 8, 16, 32):
            st = DepthStats()
            lib.depth_sim(sph.ctypes.data, n, rays.ctypes.data, kinds.ctypes.data, rays.shape[0], kind, light.ctypes.data, g, slabs, C.byref(st))
            r, gg = max(st.rays, 1), max(st.groups, 1)
            print(f"n {n} kind {kind} g {g} slabs {slabs}: rays {st.rays}  candidates/ray {st.cand_now / r:.2f} -> {st.cand_slab / r:.2f}   "
                  f"tests/ray now {st.tests_now / r:.2f} any-hit {st.tests_any / r:.2f} slab+any-hit {st.tests_slab_any / r:.2f}   "
                  f"max of 64: now {st.wave_now / gg:.2f} any-hit {st.wave_any / gg:.2f} slab+any-hit {st.wave_slab_any / gg:.2f} (slab list {st.wave_cand_slab / gg:.2f})")


if __name__ == "__main__":
    main()
