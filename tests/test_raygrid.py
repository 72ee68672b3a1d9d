"""The path-ray candidate tables of the production kernel (csrc/trt_raygrid.h) must be CONSERVATIVE: for every path ray
that passes the run-time membership test of its family, the list of the ray's cell must hold every sphere the exact
reference test (FP64, TRT.c:638-672) hits.  The very header the kernel compiles is compiled for the host and driven with
(a) every path ray the oracle traces in real frames, walked in trace order exactly as the kernel assigns families, (b) the
same rays against EVERY family whose membership test they pass, (c) adversarial scenes: tilted and non-unit ground
normals, scenes far from the origin, huge and tiny spheres, touching and nested spheres, the eye inside a sphere."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import support as T
from terminalraytracer_amd import scenes as S
from test_filter import traced_rays


class Stats(C.Structure):
    _fields_ = [(k, C.c_ulonglong) for k in ("rays", "members", "non_members", "exact_hits", "candidates", "violations", "brute_pairs",
                                             "brute_violations", "wave_max_cand", "wave_groups", "pooled_cells", "none_cells",
                                             "pool_words", "cells", "bits_set")] + \
               [("by_family", C.c_ulonglong * 4), ("cand_hist", C.c_ulonglong * 17), ("first_violation", C.c_double * 8)]


def build_checker():
    so = T.checker_so("libraygridcheck")
    src = os.path.join(T.ROOT, "tests", "raygrid_check.c")
    inc = os.path.join(T.ROOT, "terminalraytracer_amd", "csrc")
    newest = max(os.path.getmtime(p) for p in (src, os.path.join(inc, "trt_raygrid.h"), os.path.join(inc, "trt_lightgrid.h"),
                                               os.path.join(inc, "trt_filter.h")))
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC", "-shared"] + T.CHECKER_FLAGS + ["-I" + inc, "-o", so, src, "-lm"])
    lib = C.CDLL(so)
    lib.raygrid_check.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.POINTER(Stats)]
    lib.raygrid_check.restype = None
    lib.raygrid_host_cells.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_long]
    lib.raygrid_host_cells.restype = C.c_long
    return lib


@pytest.fixture(scope="module")
def checker():
    return build_checker()


def run(checker, scene, rays, kinds, g_eye, g_sph, brute=False, patch_m=0):
    sph = np.ascontiguousarray(scene.spheres, dtype=np.float64)
    ground = np.ascontiguousarray(scene.ground, dtype=np.float64)
    eye = np.ascontiguousarray(scene.camera[9:12], dtype=np.float64)
    rays = np.ascontiguousarray(rays, dtype=np.float64)
    kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
    st = Stats()
    checker.raygrid_check(sph.ctypes.data, sph.shape[0], ground.ctypes.data, eye.ctypes.data, rays.ctypes.data, kinds.ctypes.data,
                          rays.shape[0], g_eye, g_sph, patch_m, int(brute), C.byref(st))
    return st


def describe(st):
    m = max(st.members, 1)
    return (f"path rays {st.rays} members {st.members} (eye {st.by_family[0]}, mirror eye {st.by_family[1]}, sphere {st.by_family[2]}, "
            f"mirror sphere {st.by_family[3]}) non-members {st.non_members} exact hits/ray {st.exact_hits / m:.3f} "
            f"candidates/ray {st.candidates / m:.3f} max per 64 rays {st.wave_max_cand / max(st.wave_groups, 1):.2f} "
            f"hist {list(st.cand_hist)} cells {st.cells} pooled {st.pooled_cells} none {st.none_cells} pool words {st.pool_words} "
            f"brute pairs {st.brute_pairs}")


FRAMES = [("north-star scene, 64 spheres", lambda: S.synth_scene(64, T.sky("synth"), T.bench_camera(240, 135)), 240, 135, 8),
          ("demo scene", lambda: S.demo_scene(T.sky("synth"), T.bench_camera(160, 90)), 160, 90, 10),
          ("256 spheres", lambda: S.synth_scene(256, T.sky("synth"), T.bench_camera(96, 54)), 96, 54, 12),
          ("mirror-heavy", lambda: S.synth_scene(64, T.sky("synth"), T.bench_camera(96, 54, 10.0), mirror_fraction=0.5), 96, 54, 8),
          # more than 256 spheres: 16-bit list entries (3 inline, 4 per pool word; round 5: such scenes swept their path rays before)
          ("300 spheres", lambda: S.synth_scene(300, T.sky("synth"), T.bench_camera(64, 36), seed=11), 64, 36, 8)]


@pytest.mark.parametrize("name,make,w,h,b", FRAMES, ids=[f[0] for f in FRAMES])
def test_path_tables_hold_every_exact_hit_on_real_frames(checker, name, make, w, h, b):
    scene = make()
    rays, kinds = traced_rays(scene, w, h, b, 10)
    # the library's default resolution (64 / 32; 64 / 16 for the 256-sphere scene, whose host build would take a minute), a very
    # coarse one, and the spheres' surfaces cut into 24 patches with a family each (coarser cells: the host build is slow) and into 6
    dense = len(scene.spheres) > 64
    whole = None
    wide = len(scene.spheres) > 256  # the host build of 24 patches x 600 families would take minutes: one family per sphere and 6 patches
    for g_eye, g_sph, m in ((64, 16, 0), (7, 3, 0), (64, 8, 1)) if wide else ((64, 16 if dense else 32, 0), (7, 3, 0), (64, 8 if dense else 16, 2), (9, 4, 1)):
        st = run(checker, scene, rays, kinds, g_eye, g_sph, patch_m=m)
        print(f"\n{name} g {g_eye}/{g_sph} patches m {m}: {describe(st)}")
        assert st.rays == int((kinds == 0).sum()) and st.violations == 0, list(st.first_violation)
        if g_eye == 64:  # the library's resolutions: no list is lost to the pool's capacity, nearly every path ray is served
            assert st.none_cells == 0 and st.members > 0.995 * st.rays
            if m == 0:
                whole = st
            else:  # the patches serve the rays the whole-sphere families served, with fewer candidates although their cells are coarser
                assert st.members >= whole.members - 8 and (wide or st.candidates < whole.candidates)  # (wide: 6 patches on cells twice as coarse)


@pytest.mark.parametrize("m", [0, 1, 2, 3, 4])
def test_every_point_of_a_patch_lies_in_the_patch_s_ball(checker, m):
    """trt_patchset_init (csrc/trt_raygrid.h): a ray that starts on patch k of a sphere is a member of the patch's family because its
    origin lies within |r| rho_k of the apex c + |r| t_k.  Two million points of the unit sphere -- a fifth of them pushed onto the
    edges between patches, a fifth onto the cube map's edges, a fifth next to its corners -- looked up as the kernel looks them up
    (trt_patch_of, FP32): none is farther than rho_k from t_k, every patch occurs, and rho is what DESIGN.md says it is."""
    checker.raygrid_patch_cover.argtypes = [C.c_int, C.c_long, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    checker.raygrid_patch_cover.restype = C.c_double
    rho, seen = C.c_double(), C.c_int()
    worst = checker.raygrid_patch_cover(m, 2_000_000, C.byref(rho), C.byref(seen))
    print(f"\nm {m}: largest |u - t_k| / rho_k {worst:.6f}, largest rho {rho.value:.4f}, patches seen {seen.value}")
    assert worst <= 1.0 and seen.value == (6 * m * m if m else 1)
    assert abs(rho.value - {0: 1.0, 1: 0.8168, 2: 0.5164, 3: 0.4402, 4: 0.3281}[m]) < 2e-3


def test_any_family_a_ray_is_a_member_of_is_conservative_for_it(checker):
    """the structural assignment of families is a convenience: the membership test alone must make a table safe"""
    scene = S.synth_scene(64, T.sky("synth"), T.bench_camera(64, 36, 2.5))
    rays, kinds = traced_rays(scene, 64, 36, 8, 10)
    for m in (0, 1):
        st = run(checker, scene, rays, kinds, 32, 8, brute=True, patch_m=m)
        print("\n" + describe(st))
        assert st.violations == 0 and st.brute_violations == 0 and st.brute_pairs >= st.members


def _odd_scenes():
    base = S.synth_scene(40, T.sky("synth"), T.bench_camera(64, 36, 10.0), seed=5)
    out = [("base", base)]
    g = base.ground.copy()
    g[0:6] = [0.3, -1.25, 0.2, 0.1, 2.0, -0.2]  # tilted, offset ground with a non-unit normal
    out.append(("tilted ground", S.SceneData(base.spheres, g, base.dir_lights, base.point_lights, base.camera, base.sky)))
    g2 = base.ground.copy()
    g2[9] = g2[14] = 1.0  # a perfect mirror floor: long chains of ground families
    out.append(("mirror floor", S.SceneData(base.spheres, g2, base.dir_lights, base.point_lights, base.camera, base.sky)))
    far = base.spheres.copy()
    far[:, :3] += [3.0e4, -2.0e4, 1.0e4]  # the whole scene far from the origin: hit points lose digits
    gf = base.ground.copy()
    gf[0:3] += [3.0e4, -2.0e4, 1.0e4]
    cam = base.camera.copy()
    cam[9:12] += [3.0e4, -2.0e4, 1.0e4]
    out.append(("far from the origin", S.SceneData(far, gf, base.dir_lights, base.point_lights, cam, base.sky)))
    odd = base.spheres.copy()
    odd[0, 3] = 30.0      # a huge sphere
    odd[1, 3] = 1e-4      # a tiny one
    odd[2, :3] = odd[3, :3] + [odd[3, 3] + odd[2, 3], 0.0, 0.0]  # touching spheres
    odd[4, :4] = odd[5, :4]                                       # duplicates
    odd[6, :3] = odd[7, :3]
    odd[6, 3] = odd[7, 3] * 0.5                                   # nested
    odd[:, 7] = np.where(np.arange(len(odd)) % 2 == 0, 1.0, odd[:, 7])
    out.append(("odd spheres", base.with_spheres(odd)))
    inside = base.camera.copy()
    inside[9:12] = base.spheres[8, :3] + 0.3 * base.spheres[8, 3]  # the eye inside a sphere
    out.append(("eye inside a sphere", base.with_camera(inside)))
    return out


@pytest.mark.parametrize("name,scene", _odd_scenes(), ids=[s[0] for s in _odd_scenes()])
def test_path_tables_on_adversarial_scenes(checker, name, scene):
    with np.errstate(all="ignore"):
        rays, kinds = traced_rays(scene, 64, 36, 8, 10)
    for m, brute in ((0, name != "base"), (2, False), (1, name in ("tilted ground", "odd spheres"))):
        st = run(checker, scene, rays, kinds, 32, 8, brute=brute, patch_m=m)
        print(f"\n{name}, patches m {m}: {describe(st)}")
        assert st.violations == 0 and st.brute_violations == 0, list(st.first_violation)
