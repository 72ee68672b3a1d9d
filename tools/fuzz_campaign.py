"""Randomised parity campaign on the GPU box: many fuzzed scenes (tests/test_gpu_parity._fuzz_scene: 0..100 spheres, exact ties,
degenerate radii, lights on sphere surfaces, cameras inside spheres, tilted grounds) rendered by the production kernel and
compared bit for bit with the oracle.  usage: python tools/fuzz_campaign.py [first_seed] [count] [mode]
modes: "" generic | wide | lights | compact (decoupled shading forced on) | refract (extension, against its own restatement) |
dense: 128..256 spheres packed tightly, the library's default patches (or 6 / 54 / 96) |
patches / patches_refract: a family per PATCH of a sphere's surface (trt_set_path_patches 1..4, random table resolutions, the
three scene generators in turn), tables for every scene. |
many: 257..1100 spheres (round 5: 16-bit list entries in every table, the wide family builder, 1024-thread workgroups above ~290
spheres; beyond 1024 the path rays sweep), the library's defaults or one family per sphere / 6 patches / coarse tables. |
sky: few small spheres, cubemaps of any side (1 ... 1031 texels, powers of two and not, procedural: every texel its own colour), the
camera turned anywhere and placed anywhere (the skybox look-up's FP32 estimate and its FP64 fall-back, csrc/trt_device.hpp)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import support as T
from terminalraytracer_amd import hip
import test_gpu_parity as P

from terminalraytracer_amd import scenes as S


def wide_scene(rng, w, h):
    """Scenes of any size anywhere: spheres spread over 0.1..100 units, the whole scene up to 1e3 from the origin, lights near,
    far and inside the cloud, the camera outside looking at the cloud (so that ground points near the horizon, far beyond the
    light tables' range, are in view)."""
    n = int(rng.choice([1, 3, 17, 64, 65, 130]))
    scale = 10.0 ** rng.uniform(-1, 2)
    shift = rng.normal(size=3) * 10.0 ** rng.uniform(-1, 3)
    sph = np.zeros((n, 9))
    sph[:, :3] = rng.normal(size=(n, 3)) * scale + shift
    sph[:, 3] = rng.uniform(0.02, 0.4, n) * scale
    sph[:, 4:7] = rng.uniform(0, 1, (n, 3))
    sph[:, 7] = rng.choice([0.0, 0.5, 1.0], n)
    sph[:, 8] = 100.0
    ground = S.demo_ground().copy()
    ground[0:3] = shift + np.array([0.0, -2.0 * scale, 0.0])
    ground[9] = rng.choice([0.0, 0.5])
    nd, npt = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    dl = np.concatenate([rng.normal(size=(nd, 3)) * [1, 1, 1] - [0, 1.0, 0], rng.uniform(0, 1.2, (nd, 3))], axis=1)
    pl = np.concatenate([shift + rng.normal(size=(npt, 3)) * scale * 10.0 ** rng.uniform(-1, 1.5, (npt, 1)), rng.uniform(0, 1.2, (npt, 3)),
                         rng.uniform(0.1, 30, (npt, 1)) * scale * scale], axis=1)
    cam = T.bench_camera(w, h, float(rng.choice([0.0, 0.5, 2.5, 10.0])))
    cam[9:12] = cam[9:12] * scale * rng.uniform(0.3, 3.0) + shift   # the stored orbit looks at the origin from ~10 units away
    return S.SceneData(sph, ground, dl.reshape(-1, 6), pl.reshape(-1, 7), cam, T.sky("synth"))


def light_heavy_scene(rng, w, h):
    """Up to six lights of each kind: directions of any magnitude (1e-7 .. 1e3: below 1e-4 the reference leaves them
    un-normalised, TRT.c:444), duplicated lights, lights inside spheres, far and near, zero and negative colours."""
    scene = P._fuzz_scene(rng, w, h)
    nd, npt = int(rng.integers(0, 7)), int(rng.integers(0, 7))
    dl = np.concatenate([rng.normal(size=(nd, 3)) * 10.0 ** rng.uniform(-7, 3, (nd, 1)), rng.uniform(-0.3, 1.2, (nd, 3))], axis=1)
    pl = np.concatenate([rng.normal(size=(npt, 3)) * 10.0 ** rng.uniform(-1, 2, (npt, 1)), rng.uniform(-0.3, 1.2, (npt, 3)),
                         rng.uniform(0.0, 50.0, (npt, 1))], axis=1)
    if nd >= 2:
        dl[1] = dl[0]
    if npt >= 2 and len(scene.spheres):
        pl[1, :3] = scene.spheres[0, :3]            # a light at a sphere's centre
    return S.SceneData(scene.spheres, scene.ground, dl.reshape(-1, 6), pl.reshape(-1, 7), scene.camera, scene.sky)


def dense_scene(rng, w, h):
    """128..256 spheres packed like SYNTH-v0 or tighter (the config-5 regime: long candidate lists, every table with patches),
    mirrors among them, a tilted or a plain ground, the camera inside or outside the cloud."""
    n = int(rng.integers(128, 257))
    box = rng.uniform(1.5, 5.0)
    sph = np.zeros((n, 9))
    sph[:, :3] = rng.uniform(-1, 1, (n, 3)) * [box, box * 0.5, box] + [0.0, 0.5, 0.0]
    sph[:, 3] = rng.uniform(0.05, 0.5, n) * rng.choice([0.5, 1.0, 1.5])
    sph[:, 4:7] = rng.uniform(0, 1, (n, 3))
    sph[:, 7] = rng.choice([0.0, 0.3, 0.8, 1.0], n)
    sph[:, 8] = 100.0
    ground = S.demo_ground().copy()
    if rng.integers(0, 3) == 0:
        ground[3:6] = [rng.normal() * 0.2, 1.0, rng.normal() * 0.2]
    ground[9] = rng.choice([0.0, 0.2, 0.9])
    d, p = S.demo_lights()
    if rng.integers(0, 2):
        p = np.concatenate([p, [[box, 2.0, -box, 1.0, 0.8, 0.6, 20.0]]])
    cam = T.bench_camera(w, h, float(rng.choice([0.0, 0.5, 1.0, 2.5, 10.0])))
    cam[9:12] *= rng.uniform(0.2, 2.0)
    return S.SceneData(sph, ground, d, p, cam, T.sky("synth"))


def many_scene(rng, w, h):
    """257..1100 spheres in a box like SYNTH-v0's or larger, duplicates and a few huge spheres among them."""
    n = int(rng.choice([257, 258, 300, 320, 511, 512, 513, 700, 1023, 1024, 1025, 1100]))
    box = rng.uniform(3.0, 12.0)
    sph = np.zeros((n, 9))
    sph[:, :3] = rng.uniform(-1, 1, (n, 3)) * [box, box * 0.4, box] + [0.0, 0.5, 0.0]
    sph[:, 3] = rng.uniform(0.05, 0.5, n) * rng.choice([0.5, 1.0, 2.0])
    sph[:, 4:7] = rng.uniform(0, 1, (n, 3))
    sph[:, 7] = rng.choice([0.0, 0.3, 0.8, 1.0], n)
    sph[:, 8] = 100.0
    if rng.integers(0, 3) == 0:
        sph[n - 1] = sph[0]          # exact twins at the two ends of the index range: the first index must win the tie
        sph[n // 2, 3] = box * 0.3   # one sphere that is in nearly every list
    ground = S.demo_ground().copy()
    if rng.integers(0, 3) == 0:
        ground[3:6] = [rng.normal() * 0.2, 1.0, rng.normal() * 0.2]
    ground[9] = rng.choice([0.0, 0.2, 0.9])
    d, p = S.demo_lights()
    cam = T.bench_camera(w, h, float(rng.choice([0.0, 0.5, 1.0, 2.5, 10.0])))
    cam[9:12] *= rng.uniform(0.3, 2.0)
    return S.SceneData(sph, ground, d, p, cam, T.sky("synth"))


def sky_scene(rng, w, h):
    """Most rays end on the sky: 0..4 small spheres, a ground that may face away, a cubemap of a random side whose texels all
    differ, a camera with a random orthonormal basis somewhere near or far from the origin."""
    n = int(rng.integers(0, 5))
    sph = np.zeros((n, 9))
    sph[:, :3] = rng.normal(size=(n, 3)) * 3.0
    sph[:, 3] = rng.uniform(0.05, 0.6, n)
    sph[:, 4:7] = rng.uniform(0, 1, (n, 3))
    sph[:, 7] = rng.choice([0.0, 0.9, 1.0], n)
    sph[:, 8] = 100.0
    ground = S.demo_ground().copy()
    ground[0:3] = [0.0, -float(rng.choice([1.0, 50.0, 1e4])), 0.0]
    ground[9] = rng.choice([0.0, 1.0])
    d, p = S.demo_lights()
    cam = T.bench_camera(w, h, float(rng.choice([0.0, 0.5, 2.5, 10.0])))
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))  # a random rotation of the camera's basis: every face of the cubemap gets looked at
    cam[0:9] = (cam[0:9].reshape(3, 3) @ q.T).reshape(9)
    cam[9:12] = rng.normal(size=3) * 10.0 ** rng.uniform(-2, 2)
    dim = int(rng.choice([1, 2, 3, 5, 8, 31, 64, 100, 255, 256, 257, 513, 1031]))
    return S.SceneData(sph, ground, d, p, cam, S.synth_sky(dim))


first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 5000), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
mode = sys.argv[3] if len(sys.argv) > 3 else ""
make = {"wide": wide_scene, "lights": light_heavy_scene, "compact": light_heavy_scene, "dense": dense_scene, "sky": sky_scene, "many": many_scene}.get(mode, P._fuzz_scene)
patches = mode.startswith("patches")
refract = mode in ("refract", "patches_refract")  # EXTENSION, parity unpinned: against the oracle's restatement of the extension, not the reference
bad = 0
with hip.Context(0) as ctx:
    kernel = P.COMPACT if mode == "compact" else hip.Context.PRODUCTION  # "compact": the decoupled shading (trt_set_compaction) forced on
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        w, h = int(rng.integers(8, 160)), int(rng.integers(4, 90))
        b, spp = int(rng.integers(1, 13)), int(rng.choice([1, 3, 10]))
        if mode == "dense":  # the library's defaults (24 patches from 128 spheres up) two times in three, else 6 / 54 / 96 patches
            w, h, spp = min(w, 96), min(h, 54), int(rng.choice([1, 3]))
            ctx.set_path_patches(-1 if seed % 3 else int(rng.choice([1, 3, 4])))
            ctx.set_path_grids(64, 32 if seed % 3 else 8)
        if mode == "many":  # small frames (the oracle tests every sphere): the library's defaults one time in two, else other tables
            w, h, b, spp = min(w, 64), min(h, 36), min(b, 6), int(rng.choice([1, 3]))
            ctx.set_path_patches(-1 if seed % 2 else int(rng.choice([0, 1])))
            ctx.set_path_grids(64 if seed % 2 else int(rng.choice([9, 64])), 32 if seed % 2 else int(rng.choice([3, 8, 16])))
        if seed % 2 == 0:  # round 4: the light tables' depth coordinate (slabs along a directional light, shells about a point light)
            ctx.set_light_slabs(int(rng.choice([1, 3, 16, 64])), int(rng.choice([1, 5, 16, 64])))
        if mode == "compact" and seed % 3 == 0:
            b = 1  # every hit ends its sample: the ring is flushed in every round
        if patches:
            scene = (P._fuzz_scene, wide_scene, light_heavy_scene)[seed % 3](rng, w, h)
            ctx.set_path_grids_min_spheres(0)
            ctx.set_path_patches(int(rng.integers(1, 5)))
            ctx.set_path_grids(int(rng.choice([16, 64])), int(rng.choice([3, 8, 16])))
        else:
            scene = make(rng, w, h)
        ior = None
        if refract and len(scene.spheres):
            ior = rng.choice([0.0, 0.0, 1.5, 1.33, 2.4, 1.0, 0.7], len(scene.spheres))
        ctx.set_refraction(ior)
        with np.errstate(all="ignore"):
            if ior is not None:
                want, st = T.oracle_render_refractive(scene, ior, w, h, b, spp)
            else:
                want, st = T.oracle_render(scene, w, h, b, spp)
        got = P.render(ctx, scene, w, h, b, spp, kernel)
        finite = np.isfinite(want)
        ok = np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(P.bits(got[finite]), P.bits(want[finite]))
        if not ok:
            bad += 1
            diff = int((P.bits(got) != P.bits(want)).sum())
            print(f"seed {seed}: MISMATCH {diff} values differ  ({w}x{h}, {len(scene.spheres)} spheres, B{b}, spp{spp}, "
                  f"{len(scene.dir_lights)}+{len(scene.point_lights)} lights)", flush=True)
        if (seed - first) % 50 == 49:
            print(f"... {seed - first + 1} scenes, {bad} mismatches", flush=True)
print(f"fuzz campaign: {count} scenes from seed {first}: {bad} mismatches")
sys.exit(1 if bad else 0)
