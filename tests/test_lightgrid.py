"""The light-space candidate masks of the production kernel (csrc/trt_lightgrid.h) must be CONSERVATIVE:
a shadow ray's cell must hold every sphere the exact reference test (FP64, TRT.c:638-672) hits -- for a
point light every hit that can matter to TRT.c:936-946, and the lit/dark decision taken from the cell alone
must equal the decision taken from all spheres.  The very header the kernel compiles is compiled for the host
and driven with (a) every shadow ray the oracle traces in real frames and (b) adversarial origins: grazing
offsets of +-1e-16..1e-3, origins far away, at and around the light, on cube-map edges and corners, lights
inside and next to spheres."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import support as T
from terminalraytracer_amd import scenes as S
from test_filter import traced_rays


class Stats(C.Structure):
    _fields_ = [("rays", C.c_ulonglong), ("far", C.c_ulonglong), ("exact_hits", C.c_ulonglong), ("candidates", C.c_ulonglong),
                ("violations", C.c_ulonglong), ("decision_mismatches", C.c_ulonglong), ("bits_set", C.c_ulonglong),
                ("cells", C.c_ulonglong), ("wave_max_cand", C.c_ulonglong), ("wave_groups", C.c_ulonglong),
                ("cand_hist", C.c_ulonglong * 17), ("first_violation", C.c_double * 8)]


class AnyHitStats(C.Structure):
    _fields_ = [(k, C.c_ulonglong) for k in ("rays", "far", "dark", "lit", "unsure", "wrong_dark", "wrong_lit", "tests", "tests_closest")] + \
               [("first_wrong", C.c_double * 6)]


@pytest.fixture(scope="module")
def checker():
    so = T.checker_so("liblightgridcheck")
    src = os.path.join(T.ROOT, "tests", "lightgrid_check.c")
    inc = os.path.join(T.ROOT, "terminalraytracer_amd", "csrc")
    newest = max(os.path.getmtime(p) for p in (src, os.path.join(inc, "trt_lightgrid.h"), os.path.join(inc, "trt_filter.h")))
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared"] + T.CHECKER_FLAGS + ["-I" + inc, "-o", so, src, "-lm"])
    lib = C.CDLL(so)
    lib.dirgrid_check.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(Stats)]
    lib.dirgrid_check.restype = None
    lib.pointgrid_check.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(Stats)]
    lib.pointgrid_check.restype = None
    lib.pointgrid_anyhit_check.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(AnyHitStats)]
    lib.pointgrid_anyhit_check.restype = None
    return lib


def run_anyhit(checker, spheres, ground, light, rays, g, shells=16):
    spheres = np.ascontiguousarray(spheres, dtype=np.float64).reshape(-1, 9)
    rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
    light = np.ascontiguousarray(light, dtype=np.float64)
    ground = None if ground is None else np.ascontiguousarray(ground, dtype=np.float64)
    st = AnyHitStats()
    checker.pointgrid_anyhit_check(spheres.ctypes.data, spheres.shape[0], None if ground is None else ground.ctypes.data, light.ctypes.data,
                                   rays.ctypes.data, rays.shape[0], g, shells, C.byref(st))
    return st


def run_dir(checker, spheres, rays, g, slabs=16):
    spheres = np.ascontiguousarray(spheres, dtype=np.float64).reshape(-1, 9)
    rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
    st = Stats()
    checker.dirgrid_check(spheres.ctypes.data, spheres.shape[0], rays.ctypes.data, rays.shape[0], g, slabs, C.byref(st))
    return st


def run_point(checker, spheres, light, rays, g, shells=16):
    spheres = np.ascontiguousarray(spheres, dtype=np.float64).reshape(-1, 9)
    rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
    light = np.ascontiguousarray(light, dtype=np.float64)
    st = Stats()
    checker.pointgrid_check(spheres.ctypes.data, spheres.shape[0], light.ctypes.data, rays.ctypes.data, rays.shape[0], g, shells, C.byref(st))
    return st


def describe(st):
    r = max(st.rays - st.far, 1)
    return (f"rays {st.rays} far {st.far} exact hits/ray {st.exact_hits / r:.3f} candidates/ray {st.candidates / r:.3f} "
            f"max-candidates per 64 rays {st.wave_max_cand / max(st.wave_groups, 1):.2f} bits/cell {st.bits_set / max(st.cells, 1):.2f} "
            f"hist {list(st.cand_hist)[:8]}")


FRAMES = [("north-star scene, 64 spheres", lambda: S.synth_scene(64, T.sky("synth"), T.bench_camera(240, 135)), 240, 135, 8),
          ("demo scene", lambda: S.demo_scene(T.sky("synth"), T.bench_camera(160, 90)), 160, 90, 10),
          ("256 spheres", lambda: S.synth_scene(256, T.sky("synth"), T.bench_camera(96, 54)), 96, 54, 12)]


@pytest.mark.parametrize("g,depth", [(16, 1), (128, 16), (32, 7), (64, 64)])
@pytest.mark.parametrize("name,make,w,h,b", FRAMES, ids=[f[0] for f in FRAMES])
def test_masks_hold_every_hit_of_real_frames(checker, name, make, w, h, b, g, depth):
    scene = make()
    rays, kinds = traced_rays(scene, w, h, b, 10)
    shadow = rays[kinds == 1]
    assert len(shadow) > 10000 and np.all(shadow[:, 3:] == shadow[0, 3:])
    st = run_dir(checker, scene.spheres, shadow, g, depth)
    print(name, "directional", g, depth, describe(st))
    assert st.violations == 0, list(st.first_violation)
    assert st.exact_hits > 100 and st.far < 0.02 * st.rays
    assert st.candidates < 0.25 * (st.rays - st.far) * len(scene.spheres)
    point = rays[kinds == 2]
    assert len(point) > 10000
    st = run_point(checker, scene.spheres, scene.point_lights[0, :3], point, g, depth)
    print(name, "point", g, depth, describe(st))
    assert st.violations == 0, list(st.first_violation)
    assert st.decision_mismatches == 0
    assert st.exact_hits > 100 and st.far < 0.02 * st.rays
    assert st.candidates < 0.25 * (st.rays - st.far) * len(scene.spheres)


def _random_scene(rng, n_s):
    scale = 10.0 ** rng.uniform(-1, 3)
    shift = rng.normal(size=3) * 10.0 ** rng.uniform(-2, 4)
    sph = np.zeros((n_s, 9))
    sph[:, :3] = rng.normal(size=(n_s, 3)) * scale + shift
    sph[:, 3] = 10.0 ** rng.uniform(-3, 0.5, n_s) * scale * 0.1
    sph[:, 4:] = 0.5
    return sph, scale, shift


OFFSETS = [0.0, 1e-16, -1e-16, 1e-12, -1e-12, 1e-9, -1e-9, 1e-6, -1e-6, 1e-3, -1e-3, -0.5]


def _unit(v):
    # the reference's normalisation: three divisions by the same length (TRT.c:247-251)
    return v / np.sqrt((v * v).sum(axis=-1, keepdims=True))


def test_directional_masks_are_conservative_on_adversarial_origins(checker):
    rng = np.random.default_rng(21)
    hits = 0
    for trial in range(12):
        n_s = int(rng.integers(1, 150))
        sph, scale, shift = _random_scene(rng, n_s)
        d = _unit(rng.normal(size=3))
        if trial % 4 == 0:
            d = _unit(np.array([0.0, 1.0, 0.0]) + rng.normal(size=3) * 1e-9)  # almost along an axis
        m = 30000
        k = rng.integers(0, n_s, m)
        perp = np.cross(rng.normal(size=(m, 3)), d)
        perp /= np.linalg.norm(perp, axis=1, keepdims=True)
        offs = rng.choice(OFFSETS, m)
        back = rng.uniform(0.1, 100.0, (m, 1)) * scale * rng.choice([1.0, 1.0, 30.0], (m, 1))  # some origins beyond the admissible radius
        o = sph[k, :3] + perp * (sph[k, 3] * (1 + offs))[:, None] - d * back
        rays = np.concatenate([o, np.broadcast_to(d, (m, 3))], axis=1)
        for g, slabs in ((8, 3), (64, 16), (256, 1), (32, 64)):
            st = run_dir(checker, sph, rays, g, slabs)
            assert st.violations == 0, (trial, g, slabs, list(st.first_violation))
        hits += st.exact_hits
    assert hits > 20000


def test_point_masks_are_conservative_on_adversarial_origins(checker):
    rng = np.random.default_rng(22)
    hits = 0
    for trial in range(14):
        n_s = int(rng.integers(1, 150))
        sph, scale, shift = _random_scene(rng, n_s)
        light = shift + rng.normal(size=3) * scale * 10.0 ** rng.uniform(-1, 1)
        if trial % 5 == 1:   # the light inside a sphere
            light = sph[0, :3] + _unit(rng.normal(size=3)) * sph[0, 3] * 0.5
        if trial % 5 == 2:   # the light a hair outside a sphere
            light = sph[0, :3] + _unit(rng.normal(size=3)) * sph[0, 3] * (1 + 10.0 ** rng.uniform(-9, -2))
        m = 30000
        # origins such that the ray to the light grazes sphere k: pick the tangent direction from the light's side
        k = rng.integers(0, n_s, m)
        c, r = sph[k, :3], sph[k, 3]
        to_c = c - light
        dist = np.linalg.norm(to_c, axis=1, keepdims=True)
        axis = np.cross(to_c, rng.normal(size=(m, 3)))
        axis /= np.linalg.norm(axis, axis=1, keepdims=True)
        rr = (r * (1 + rng.choice(OFFSETS, m)))[:, None]
        sin_t = np.clip(rr / np.maximum(dist, 1e-300), -1, 1)
        w = to_c / np.maximum(dist, 1e-300) * np.sqrt(1 - sin_t ** 2) + axis * sin_t   # unit, from the light past the sphere
        far_side = rng.choice([1.0, 1.0, 1.0, -1.0], (m, 1))                             # some origins on the other side of the light
        o = light + w * far_side * (dist + r[:, None]) * rng.uniform(0.2, 5.0, (m, 1)) * rng.choice([1.0, 1.0, 200.0], (m, 1))
        # directions towards cube-map edges and corners, and origins next to the light
        e = rng.choice([-1.0, 1.0], (3000, 3)) * np.where(rng.random((3000, 3)) < 0.3, rng.random((3000, 3)), 1.0)
        e = e * (1 + rng.normal(size=(3000, 3)) * 1e-7)
        o = np.concatenate([o, light + e * scale * rng.uniform(0.01, 3.0, (3000, 1)),
                            light + rng.normal(size=(2000, 3)) * scale * 10.0 ** rng.uniform(-12, -1, (2000, 1)), light[None, :]])
        with np.errstate(invalid="ignore", divide="ignore"):
            d = _unit(light - o)
        rays = np.concatenate([o, d], axis=1)
        for g, shells in ((4, 16), (32, 1), (128, 5), (16, 64)):
            st = run_point(checker, sph, light, rays, g, shells)
            assert st.violations == 0, (trial, g, shells, list(st.first_violation))
            assert st.decision_mismatches == 0, (trial, g, shells)
        hits += st.exact_hits
    assert hits > 20000


# ---- the any-hit search of a point light's shadow rays (trt_lightgrid.h (4)) ----

@pytest.mark.parametrize("name,make,w,h,b", FRAMES, ids=[f[0] for f in FRAMES])
def test_any_hit_search_decides_as_the_reference_on_real_frames(checker, name, make, w, h, b):
    """Every "dark" and every "lit" of the division-free any-hit classification the kernel runs for point lights must be the answer
    of the reference's closest-hit decision (all spheres and the ground, nudged blocker point, TRT.c:937-942); "unsure" -- which
    sends the wave through the closest-hit search -- must be rare."""
    scene = make()
    rays, kinds = traced_rays(scene, w, h, b, 10)
    point = rays[kinds == 2]
    for g, shells in ((16, 1), (64, 16)):
        st = run_anyhit(checker, scene.spheres, scene.ground, scene.point_lights[0, :3], point, g, shells)
        print(f"\n{name} g {g} shells {shells}: rays {st.rays} far {st.far} dark {st.dark} lit {st.lit} unsure {st.unsure} "
              f"tests any-hit {st.tests / max(1, st.rays - st.far):.3f} closest-hit {st.tests_closest / max(1, st.rays - st.far):.3f}")
        assert st.wrong_dark == 0 and st.wrong_lit == 0, list(st.first_wrong)
        assert st.dark > 1000 and st.lit > 1000 and st.far < 0.02 * st.rays
        assert st.unsure <= 1e-3 * st.rays


def test_any_hit_search_on_adversarial_origins(checker):
    """Blockers at (almost) exactly the light's distance, origins a hair off a sphere (the ray re-enters it after 1e-9 ... 1e-4),
    lights inside / next to spheres, scenes far from the origin (coordinates up to 1e9: the dark_floor / e2 bounds scale with them,
    and beyond 2^28 every hit must come back unsure), grounds that block: never a wrong "dark" or "lit"."""
    rng = np.random.default_rng(31)
    decided = unsure = 0
    for trial in range(16):
        n_s = int(rng.integers(1, 120))
        sph, scale, shift = _random_scene(rng, n_s)
        if trial % 4 == 3:
            shift = shift + rng.normal(size=3) * 10.0 ** rng.uniform(5, 9)
            sph[:, :3] += shift
        light = sph[:, :3].mean(axis=0) + rng.normal(size=3) * scale * 10.0 ** rng.uniform(-1, 1)
        if trial % 5 == 1:
            light = sph[0, :3] + _unit(rng.normal(size=3)) * sph[0, 3] * 0.5
        ground = np.zeros(16)
        ground[0:3] = sph[:, :3].mean(axis=0) - [0, scale, 0]
        ground[3:6] = _unit(np.array([0.0, 1.0, 0.0]) + rng.normal(size=3) * 0.2) * 10.0 ** rng.uniform(-1, 1)
        m = 20000
        k = rng.integers(0, n_s, m)
        c, r = sph[k, :3], np.abs(sph[k, 3])
        # (a) origins on the far side of sphere k from the light, at offsets that put the NEAR hit at the light's distance +- tiny
        u = _unit(rng.normal(size=(m, 3)))
        o_a = light + u * np.linalg.norm(c - light, axis=1, keepdims=True) * rng.uniform(1.0, 3.0, (m, 1))
        # (b) origins a hair outside sphere k, the light behind the surface: the ray re-enters its own sphere at once
        nrm = _unit(rng.normal(size=(m, 3)))
        o_b = c + nrm * (r * (1 + 10.0 ** rng.uniform(-12, -3, m)))[:, None]
        # (c) origins such that the blocker's near hit is at the light's distance times 1 +- 1e-16 .. 1e-3
        tow = _unit(c - light)
        o_c = light + tow * (np.linalg.norm(c - light, axis=1) - r)[:, None] * (1 + rng.choice(OFFSETS, m))[:, None] * rng.choice([1.0, 2.0], (m, 1))
        # (d) ground points (shadow rays that graze or hit the ground)
        gn = _unit(ground[3:6])
        t1 = np.cross(gn, [1.0, 0.3, 0.2])
        t1 /= np.linalg.norm(t1)
        t2 = np.cross(gn, t1)
        o_d = ground[0:3] + (rng.normal(size=(m, 1)) * t1 + rng.normal(size=(m, 1)) * t2) * scale * 3 + gn * (10.0 ** rng.uniform(-9, 0, (m, 1))) * scale * rng.choice([-1.0, 1.0], (m, 1))
        o = np.concatenate([o_a, o_b, o_c, o_d])
        with np.errstate(invalid="ignore", divide="ignore"):
            d = _unit(light - o)
        rays = np.concatenate([o, d], axis=1)
        for g, shells in ((8, 1), (64, 16), (16, 40)):
            st = run_anyhit(checker, sph, ground, light, rays, g, shells)
            assert st.wrong_dark == 0 and st.wrong_lit == 0, (trial, g, shells, list(st.first_wrong))
        decided += st.dark + st.lit
        unsure += st.unsure
        if np.abs(light).max() + 2 * 256 * 3 * scale > 2.0 ** 28 * 4:
            pass
    print(f"\nany-hit on adversarial origins: decided {decided}, unsure {unsure}")
    assert decided > 200000


def test_any_hit_search_is_switched_off_for_astronomic_coordinates(checker):
    """Mg > 2^28: the bounds of trt_lightgrid.h (4) do not hold, so no hit may be taken as proof of "dark" (every hit is unsure)."""
    rng = np.random.default_rng(32)
    sph = np.zeros((20, 9))
    sph[:, :3] = rng.normal(size=(20, 3)) + 3e9
    sph[:, 3] = 0.3
    light = sph[:, :3].mean(axis=0) + [0.0, 4.0, 0.0]
    o = sph[rng.integers(0, 20, 5000), :3] - [0.0, 2.0, 0.0] + rng.normal(size=(5000, 3)) * 0.2
    with np.errstate(invalid="ignore", divide="ignore"):
        d = _unit(light - o)
    st = run_anyhit(checker, sph, None, light, np.concatenate([o, d], axis=1), 32)
    print(f"\nastronomic: rays {st.rays} far {st.far} dark {st.dark} lit {st.lit} unsure {st.unsure}")
    assert st.dark == 0 and st.wrong_lit == 0
