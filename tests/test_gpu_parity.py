"""GPU parity: the HIP frame producer (through the C-ABI of include/trt_hip.h) against
 (a) golden vectors taken from the genuine reference (tests/golden), and
 (b) the CPU oracle on the same inputs.
Bit-exact: the double framebuffer must be identical (which implies the 1e-5 tolerance of the
north star and the exact (int)(c*255) colour indices)."""
import ctypes as C

import numpy as np
import pytest

import support as T
from terminalraytracer_amd import hip
from terminalraytracer_amd import scenes as S

pytestmark = pytest.mark.gpu

SMALL = T.golden_cases(("small", "medium"))
LARGE = T.golden_cases(("large",))
# the production kernel as it ships (the shading decoupled from the owning lane for scenes of three lights or more,
# trt_set_compaction(-1)), the same with the decoupling forced on, and the reference-order kernel -- an independent HIP
# implementation of the path
COMPACT = "production_rounds_compact"
PLAIN = "production_rounds_plain"  # the decoupling forced off (what ships for scenes of one or two lights)
KERNELS = [hip.Context.PRODUCTION, COMPACT, hip.Context.REFERENCE_ORDER]
KERNEL_IDS = ["production_rounds", COMPACT, "reference_order"]


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    c.set_path_grids_min_spheres(0)  # by default scenes of fewer than 12 spheres sweep: here the small goldens use the tables too
    yield c
    c.close()


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def render(ctx, scene, w, h, b, s, kernel=hip.Context.PRODUCTION, rows=None):
    ctx.set_kernel(hip.Context.PRODUCTION if kernel in (COMPACT, PLAIN) else kernel)
    ctx.set_compaction({COMPACT: 1, PLAIN: 0}.get(kernel, -1))
    ctx.set_scene(scene)
    return ctx.render_host(scene.camera, rows or hip.RowSet.whole(w, h), b, s)


def test_device_division_and_sqrt_are_correctly_rounded(ctx):
    rng = np.random.default_rng(1)
    n = 1 << 20
    a = rng.uniform(0.0, 1.0, n) * 10.0 ** rng.uniform(-30, 30, n)
    b = (rng.uniform(0.5, 1.0, n) * 10.0 ** rng.uniform(-30, 30, n)) * rng.choice([-1.0, 1.0], n)
    a[:8] = [0.0, 1.0, 2.0, 1e-308, 1e308, 3.0, 0.1, 4.9e-324]
    b[:8] = [1.0, 3.0, 1e-300, 7.0, 1e-10, 0.0, 0.3, 2.0]
    q, r = ctx.selftest_div_sqrt(a, b)
    wq, wr = np.empty_like(a), np.empty_like(a)
    with np.errstate(all="ignore"):
        T.oracle().trt_oracle_div_sqrt(a.ctypes.data, b.ctypes.data, n, wq.ctypes.data, wr.ctypes.data)
    assert np.array_equal(bits(q), bits(wq)), int((bits(q) != bits(wq)).sum())
    assert np.array_equal(bits(r), bits(wr)), int((bits(r) != bits(wr)).sum())


def test_single_rays_match_reference_vectors(ctx):
    d = np.load(T.GOLDEN + "/rays.npz")
    scene = S.SceneData.from_arrays(d, T.sky("uv_checker"), prefix="scene/")
    ctx.set_scene(scene)
    obj, point, normal, material, lit = ctx.probe_rays(d["rays"])
    assert np.array_equal(obj, d["obj"])
    assert np.array_equal(bits(point), bits(d["point"]))
    assert np.array_equal(bits(normal), bits(d["normal"]))
    assert np.array_equal(bits(material), bits(d["material"]))
    hit = obj != 0
    assert np.array_equal(bits(lit[hit]), bits(d["lit"][hit]))


def _same_probe(got, want, hit_only_lit=True):
    obj, point, normal, material, lit = got
    assert np.array_equal(obj, want["obj"])
    assert np.array_equal(bits(point), bits(want["point"]))
    assert np.array_equal(bits(normal), bits(want["normal"]))
    assert np.array_equal(bits(material), bits(want["material"]))
    hit = obj != 0
    assert np.array_equal(bits(lit[hit]), bits(want["lit"][hit]))


@pytest.mark.parametrize("grids", [(64, 32, 0), (64, 16, 2), (40, 8, 3), (5, 3, 1), (0, 0, 0)],
                         ids=["default_tables", "24_patches", "54_patches", "coarse_tables_6_patches", "sweep_only"])
def test_production_stages_match_reference_single_ray_vectors(ctx, grids):
    """trace_ray, ray_intersects_sphere/plane, get_skybox_color and apply_lighting (TRT.c:638-963), each through the code that
    SHIPS: the probe runs the render kernel's own path_stage / shadow_stage.  (a) the 600 arbitrary rays of rays.npz (no
    family: every wave sweeps; non-unit directions among them); (b) chains of path rays as project_scene produces them, every
    ray looked up in the table of its family -- eye, mirror eye, spheres (one family per sphere, or 6 / 24 / 54 patches per sphere
    with a family each), their mirror images -- 64 spheres with mirrors and 256."""
    try:
        ctx.set_path_patches(grids[2])
        ctx.set_path_grids(*grids[:2])
        d = np.load(T.GOLDEN + "/rays.npz")
        scene = S.SceneData.from_arrays(d, T.sky("uv_checker"), prefix="scene/")
        ctx.set_scene(scene)
        _same_probe(ctx.probe_rays_production(scene.camera, d["rays"]), d)
        fam = np.load(T.GOLDEN + "/rays_families.npz")
        for tag in ("a", "b"):
            scene = S.SceneData.from_arrays(fam, T.sky("uv_checker"), prefix=tag + "/scene/")
            want = {k: fam[f"{tag}/{k}"] for k in ("obj", "point", "normal", "material", "lit")}
            ctx.set_scene(scene)
            codes = ctx.family_codes(fam[tag + "/families"], fam[tag + "/rays"], len(scene.spheres)) if grids[0] else fam[tag + "/families"]
            assert ctx.path_patches() == ((grids[2], max(1, 6 * grids[2] ** 2)) if grids[0] else (0, 0))
            _same_probe(ctx.probe_rays_production(scene.camera, fam[tag + "/rays"], codes), want)
            # a WRONG family must not matter either: the membership test sends such rays to the sweep
            wrong = np.roll(codes, 7)
            _same_probe(ctx.probe_rays_production(scene.camera, fam[tag + "/rays"], wrong), want)
            # nor must codes the tables have no family for (the numbering of an older header, a patch number beyond the
            # patches, a sphere beyond the scene, garbage): the library maps them to "no family" before the kernel sees them
            n_s = len(scene.spheres)
            garbage = np.array(codes, dtype=np.int64)
            garbage[0::4] = 2 + n_s + ((np.arange(len(garbage[0::4])) % max(n_s, 1)) << 7 | 127)  # patch 127 of a sphere
            garbage[1::4] = 2 + n_s + ((n_s + 5) << 7)                                          # a sphere beyond the scene
            garbage[2::4] = 2 ** 31 - 1
            garbage[3::4] = -7
            _same_probe(ctx.probe_rays_production(scene.camera, fam[tag + "/rays"], garbage.astype(np.int32)), want)
            # one to four rays of a wave without a family (every 17th ray; every 64th ray and its three neighbours): their waves sweep
            for lone in (np.arange(len(codes)) % 17 == 5, np.arange(len(codes)) % 64 < 4, np.arange(len(codes)) % 64 == 63):
                few = np.array(codes, dtype=np.int32)
                few[lone] = -1
                _same_probe(ctx.probe_rays_production(scene.camera, fam[tag + "/rays"], few), want)
            _same_probe(ctx.probe_rays(fam[tag + "/rays"]), want)  # and the reference-order kernel's probe
    finally:
        ctx.set_path_patches(-1)
        ctx.set_path_grids(64, 32)


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
@pytest.mark.parametrize("case", SMALL, ids=[c["name"] for c in SMALL])
def test_frame_matches_reference_golden(ctx, case, kernel):
    scene = T.golden_scene(case)
    px = render(ctx, scene, case["width"], case["height"], case["bounce_limit"], case["rays_per_pixel"], kernel)
    fb = T.golden_fb(case)
    if fb is not None and not np.array_equal(bits(px), bits(fb)):
        bad = np.argwhere((bits(px) != bits(fb)).any(axis=2))
        pytest.fail(f"{len(bad)} pixels differ, first {bad[:5].tolist()}")
    assert T.fnv(px) == case["fb_fnv"]
    assert T.fnv(T.oracle_rgb8(px)) == case["rgb8_fnv"]


@pytest.mark.parametrize("case", LARGE, ids=[c["name"] for c in LARGE])
def test_full_hd_frame_matches_reference_golden(ctx, case):
    scene = T.golden_scene(case)
    px = render(ctx, scene, case["width"], case["height"], case["bounce_limit"], case["rays_per_pixel"])
    assert T.fnv(px) == case["fb_fnv"]


def test_drop_in_project_scene_symbol():
    """void project_scene(Scene*, Screen*) itself, at the reference's B=10 / 10 rays per pixel."""
    case = next(c for c in SMALL if c["name"] == "demo_480x280_b10")
    px = hip.project_scene(T.golden_scene(case), 480, 280)
    assert T.fnv(px) == case["fb_fnv"] == "453219f388ade6f2"
    case = next(c for c in SMALL if c["name"] == "demo_160x48_b4")
    px = hip.render_frame(T.golden_scene(case), 160, 48, 4, 10)
    assert T.fnv(px) == case["fb_fnv"]
    # the scene may change between calls (main() rewrites the camera every frame): new camera, same skybox pointers
    case2 = next(c for c in SMALL if c["name"] == "demo_160x48_b10")
    px = hip.render_frame(T.golden_scene(case2), 160, 48, 10, 10)
    assert T.fnv(px) == case2["fb_fnv"]
    # the entry under the name BASELINE.json's north_star gives it
    px = hip.render_frame(T.golden_scene(case), 160, 48, 4, 10, symbol="render_frame")
    assert T.fnv(px) == case["fb_fnv"]
    assert hip.lib().trt_shutdown() == 0


def test_counters_equal_reference_trace_ray_counts(ctx):
    case = next(c for c in SMALL if c["name"] == "synth64_480x270_b8")
    ctx.enable_counters(True)
    try:
        for kernel in KERNELS:
            px = render(ctx, T.golden_scene(case), 480, 270, 8, 10, kernel)
            assert T.fnv(px) == case["fb_fnv"]
            assert ctx.read_counters() == (2966024, 3341736)  # path, shadow
    finally:
        ctx.enable_counters(False)


FULL = ["c3_1080p_64sph_b8", "c2_1080p_8sph_b4", "c4_2160p_64sph_b8", "c5_1080p_256sph_b12_f0", "c5_1080p_256sph_b12_f59",
        "c3_1080p_64sph_b8_f19"]  # the last: the headline's scene seen from the reference's moving camera (bench.py's value_moving_camera)


@pytest.mark.parametrize("name", FULL)
def test_baseline_configs_whole_frames_equal_reference_hash_and_oracle(ctx, name):
    """BASELINE configs 2-5 at their FULL sizes (c3 is the frame bench.py times): every pixel of the production kernel's
    frame equals the all-core CPU oracle's frame bit for bit, the frame's FNV equals the hash the GENUINE reference
    produced for it (tests/golden/golden_full.json), the (int)(c*255) bytes likewise, the reference-order kernel -- an
    independent HIP implementation -- agrees, and so do the trace_ray call counts.  The counting instantiation of the kernel is
    another binary than the one that ships (other register allocation): the frame is rendered once more WITHOUT counters -- the
    instantiation bench.py times: <false, false, true> (shading decoupled) for c3 / c4, the plain rounds for c2 (8 spheres: its
    path rays sweep, and a scene that sweeps is not decoupled), <false, false, false, true> (a family per patch of a sphere) for
    c5 -- and must have the same bits."""
    import os
    case = T.golden_full()[name]
    w, h, b, spp = case["width"], case["height"], case["bounce_limit"], case["rays_per_pixel"]
    scene = T.full_scene(case)
    ctx.enable_counters(True)
    try:
        fast = render(ctx, scene, w, h, b, spp, hip.Context.PRODUCTION)
        counts = ctx.read_counters()
    finally:
        ctx.enable_counters(False)
    assert T.fnv(fast) == case["fb_fnv"]
    assert T.fnv(T.oracle_rgb8(fast)) == case["rgb8_fnv"]
    assert counts == (case["path_rays"], case["shadow_rays"])
    # counters off, the whole frame in ONE launch into device memory as bench.py renders it (trt_render_host splits a large frame
    # into bands, which are too small for the decoupled kernel): the instantiation that is timed
    import torch
    fb = torch.zeros(h * w * 3, dtype=torch.float64, device="cuda:0")
    ctx.set_kernel(hip.Context.PRODUCTION)
    ctx.set_compaction(-1)
    ctx.set_path_grids_min_spheres(12)  # the library's default (this module's context otherwise builds path tables for every scene)
    try:
        ctx.render_device(scene.camera, hip.RowSet.whole(w, h), b, spp, fb.data_ptr(), fb.numel() * 8)
        ctx.synchronize()
        shipped = fb.cpu().numpy().reshape(h, w, 3)
        del fb
        variant = ctx.render_variant()
        n = len(scene.spheres)
        assert variant["decoupled"] == (12 <= n < 128) and ctx.path_patches() == ((2, 24) if n == 256 else ((0, 1) if n >= 12 else (0, 0))), (variant, n)
    finally:
        ctx.set_path_grids_min_spheres(0)
    assert np.array_equal(bits(shipped), bits(fast)) and T.fnv(shipped) == case["fb_fnv"]
    want, st = T.oracle_render(scene, w, h, b, spp, threads=min(os.cpu_count() or 1, 64))
    if not np.array_equal(bits(fast), bits(want)):
        bad = np.argwhere((bits(fast) != bits(want)).any(axis=2))
        pytest.fail(f"{len(bad)} pixels differ from the oracle, first {bad[:5].tolist()}")
    assert (st.path_rays, st.shadow_rays) == counts
    if name in ("c3_1080p_64sph_b8", "c2_1080p_8sph_b4"):  # the slow kernel takes 13 ms / 100+ ms on the larger ones: two are enough
        slow = render(ctx, scene, w, h, b, spp, hip.Context.REFERENCE_ORDER)
        assert np.array_equal(bits(fast), bits(slow))


@pytest.mark.parametrize("name", ["sky1024_240x135_64sph_b8", "sky2048_240x135_64sph_b8", "sky2048_480x270_8sph_b4"])
def test_large_cubemaps_match_the_reference(ctx, name):
    """Cubemap faces of 1024^2 and 2048^2 texels (the reference's main() loads `milky_way`, TRT.c:1244; the face size is
    data, TRT.c:388-427): 25 / 100 MB as packed texels on the device, no longer L2-resident.  Frames against the genuine
    reference's hash and the oracle, both kernels."""
    case = T.golden_full()[name]
    w, h, b, spp = case["width"], case["height"], case["bounce_limit"], case["rays_per_pixel"]
    scene = T.full_scene(case)
    want, _ = T.oracle_render(scene, w, h, b, spp)
    for kernel in KERNELS:
        got = render(ctx, scene, w, h, b, spp, kernel)
        assert np.array_equal(bits(got), bits(want)), kernel
        assert T.fnv(got) == case["fb_fnv"]


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
@pytest.mark.parametrize("world,tile", [(2, 8), (8, 8), (3, 5)])
def test_row_tile_shards_reassemble_to_the_whole_frame(ctx, world, tile, kernel):
    case = next(c for c in SMALL if c["name"] == "synth64_128x72_b8")
    scene = T.golden_scene(case)
    w, h = 128, 72
    whole = render(ctx, scene, w, h, 8, 10)
    out = np.zeros_like(whole)
    for rank in range(world):
        rs = hip.RowSet.shard(w, h, rank, world, tile)
        part = render(ctx, scene, w, h, 8, 10, kernel, rows=rs)
        for i in range(part.shape[0]):
            out[hip.lib().trt_rowset_frame_row(C.byref(rs), i)] = part[i]
    assert np.array_equal(bits(out), bits(whole))
    assert T.fnv(out) == case["fb_fnv"]


def test_device_resident_render_and_rgb8_quantisation(ctx):
    import torch
    case = next(c for c in SMALL if c["name"] == "demo_160x48_b4")
    scene = T.golden_scene(case)
    ctx.set_kernel(hip.Context.PRODUCTION)
    ctx.set_scene(scene)
    fb = torch.zeros(48 * 160 * 3, dtype=torch.float64, device="cuda:0")
    rgb = torch.zeros(48 * 160 * 3, dtype=torch.uint8, device="cuda:0")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        ctx.render_device(scene.camera, hip.RowSet.whole(160, 48), 4, 10, fb.data_ptr(), fb.numel() * 8)
        ctx.quantize_device(fb.data_ptr(), 48 * 160, rgb.data_ptr())
        torch.cuda.synchronize()
        assert T.fnv(fb.cpu().numpy()) == case["fb_fnv"]
        assert T.fnv(rgb.cpu().numpy()) == case["rgb8_fnv"]
        times = ctx.kernel_times(4)
        assert len(times) >= 1 and all(t > 0 for t in times)
        with pytest.raises(hip.TrtError):  # framebuffer too small
            ctx.render_device(scene.camera, hip.RowSet.whole(160, 48), 4, 10, fb.data_ptr(), 100)
    finally:
        ctx.set_stream(None)


def test_errors_are_reported_not_swallowed():
    with hip.Context(0) as c:
        with pytest.raises(hip.TrtError) as e:
            c.render_host(np.zeros(15), hip.RowSet.whole(4, 4), 4, 10)
        assert e.value.code == -3  # TRT_ERR_NO_SCENE
        case = SMALL[0]
        c.set_scene(T.golden_scene(case))
        with pytest.raises(hip.TrtError) as e:
            c.render_host(T.golden_scene(case).camera, hip.RowSet.whole(4, 4), 0, 10)
        assert e.value.code == -2
    with pytest.raises(hip.TrtError):
        hip.Context(99)


def test_gpu_frame_through_the_host_emitter_matches_reference_bytes(ctx):
    """GPU framebuffer -> device quantisation -> host emitter == the reference's screenbuffer bytes."""
    import zlib
    import torch
    from terminalraytracer_amd import host
    case = next(c for c in SMALL if c["name"] == "demo_160x48_b4")
    scene = T.golden_scene(case)
    ctx.set_kernel(hip.Context.PRODUCTION)
    ctx.set_scene(scene)
    fb = torch.zeros(48 * 160 * 3, dtype=torch.float64, device="cuda:0")
    rgb = torch.zeros(48 * 160 * 3, dtype=torch.uint8, device="cuda:0")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        ctx.render_device(scene.camera, hip.RowSet.whole(160, 48), 4, 10, fb.data_ptr(), fb.numel() * 8)
        ctx.quantize_device(fb.data_ptr(), 48 * 160, rgb.data_ptr())
        torch.cuda.synchronize()
    finally:
        ctx.set_stream(None)
    em = host.Emitter(160, 48)
    em.patch_rgb8(rgb.cpu().numpy())
    want = zlib.decompress(open(T.GOLDEN + "/emit_demo_160x48_b4.bin.z", "rb").read())
    assert em.bytes() == want


def test_host_rgb8_entries_feed_the_emitter_the_reference_bytes(ctx):
    """trt_render_host_rgb8 / trt_render_frame_rgb8 (3 bytes per pixel across PCIe): the bytes are (int)(c*255) of the very
    framebuffer the f64 path produces -- the reference's golden frame -- shards included, and the host emitter turns them into
    the reference's screenbuffer."""
    import zlib
    from terminalraytracer_amd import host
    case = next(c for c in SMALL if c["name"] == "demo_160x48_b4")
    scene = T.golden_scene(case)
    ctx.set_kernel(hip.Context.PRODUCTION)
    ctx.set_scene(scene)
    want_rgb = T.oracle_rgb8(T.golden_fb(case))
    whole = ctx.render_host_rgb8(scene.camera, hip.RowSet.whole(160, 48), 4, 10)
    assert whole.dtype == np.uint8 and np.array_equal(whole.reshape(-1), np.asarray(want_rgb).reshape(-1))
    rows = hip.RowSet.shard(160, 48, 1, 3, 5)
    part = ctx.render_host_rgb8(scene.camera, rows, 4, 10)
    mine = [hip.lib().trt_rowset_frame_row(C.byref(rows), i) for i in range(part.shape[0])]
    assert np.array_equal(part, whole[mine])
    through_default = hip.render_frame_rgb8(scene, 160, 48, 4, 10)
    assert np.array_equal(through_default, whole)
    em = host.Emitter(160, 48)
    em.patch_rgb8(through_default)
    assert em.bytes() == zlib.decompress(open(T.GOLDEN + "/emit_demo_160x48_b4.bin.z", "rb").read())
    big = T.golden_full()["c3_1080p_64sph_b8"]  # a frame large enough for the banded f64 path: same bytes either way
    scene = T.full_scene(big)
    ctx.set_scene(scene)
    rgb = ctx.render_host_rgb8(scene.camera, hip.RowSet.whole(1920, 1080), 8, 10)
    assert T.fnv(rgb) == big["rgb8_fnv"]


@pytest.mark.parametrize("through_rccl", [False, True], ids=["plain", "one_rank_rccl_communicator"])
def test_c_abi_dist_renderer_world_of_one(ctx, through_rccl):
    """trt_dist_* (include/trt_hip.h section 3) with one rank: the row tiles, the frame pipeline (three frames in flight,
    compute units reserved) and -- with an id -- a one-rank RCCL communicator with the whole gather path (group, assembly
    kernel) inside the library.  Every frame of a short orbit must equal the oracle's, in order."""
    case = next(c for c in SMALL if c["name"] == "synth64_128x72_b8")
    scene = T.golden_scene(case)
    uid = hip.dist_unique_id() if through_rccl else None
    with hip.Dist(0, scene, uid, 0, 1, 128, 72, tile_rows=8, frames_in_flight=3, reserved_cus=8) as d:
        assert (d.local_rows, d.max_rows) == (72, 72)
        frames = [d.render(scene.camera, 8, 10) for _ in range(7)]  # every slot reused at least twice
        assert T.fnv(d.fetch(frames[-1])) == case["fb_fnv"]
        for t in (0.0, 2.5, 33.3):
            cam = T.bench_camera(128, 72, t)
            got = d.fetch(d.render(cam, 8, 10))
            want, _ = T.oracle_render(scene.with_camera(cam), 128, 72, 8, 10)
            assert np.array_equal(bits(got), bits(want)), t
        assert d.context(0).kernel_info()["compute_units"] >= 8
        with pytest.raises(IndexError):
            d.context(3)
        # the same frames as the emitter's bytes (trt_dist_render_rgb8): not before the byte buffers exist, then interleaved with
        # f64 frames on the same slots; bad arguments are refused before anything is enqueued and leave the trt_dist usable
        with pytest.raises(hip.TrtError):
            d.render_rgb8(scene.camera, 8, 10)
        d.enable_rgb8()
        d.enable_rgb8()  # idempotent
        with pytest.raises(hip.TrtError):
            d.render(scene.camera, 0, 10)
        for t in (0.0, 2.5):
            cam = T.bench_camera(128, 72, t)
            rgb = d.fetch_rgb8(d.render_rgb8(cam, 8, 10))
            f64 = d.fetch(d.render(cam, 8, 10))
            want, _ = T.oracle_render(scene.with_camera(cam), 128, 72, 8, 10)
            assert np.array_equal(bits(f64), bits(want)) and np.array_equal(rgb, T.oracle_rgb8(want)), t
    with pytest.raises(hip.TrtError):
        hip.Dist(0, scene, None, 0, 2, 128, 72)  # more than one rank needs the communicator's id


def test_dist_renderer_on_a_scene_of_more_than_256_spheres(ctx):
    """The frame slots of a trt_dist share ONE copy of the scene's tables and keep the eye's two tables in slots of their own -- here
    with everything round 5 added for large scenes: 16-bit list entries, the wide family builder, eye parts of the pool sized for the
    longest lists, the plain rounds in 1024-thread workgroups.  Cameras in turn over three slots (every slot's eye tables rebuilt),
    then a new scene of another size on the same renderer: every frame the oracle's."""
    scene = S.synth_scene(320, T.sky("synth"), T.bench_camera(96, 54, 2.5), seed=21)
    with hip.Dist(0, scene, None, 0, 1, 96, 54, tile_rows=8, frames_in_flight=3) as d:
        cams = [T.bench_camera(96, 54, t) for t in (0.0, 0.5, 1.0, 2.5, 10.0, 33.3, 0.5)]
        frames = [d.fetch(d.render(cam, 6, 3)) for cam in cams]
        for cam, got in zip(cams, frames):
            want, _ = T.oracle_render(scene.with_camera(cam), 96, 54, 6, 3)
            assert np.array_equal(bits(got), bits(want))
        assert d.context(0).render_variant()["workgroup_threads"] == 1024 and d.context(0).scene_info()["sharers"] == 3
        smaller = S.synth_scene(200, T.sky("synth"), T.bench_camera(96, 54, 1.0), seed=22)
        d.set_scene(smaller)
        for t in (1.0, 10.0, 2.5, 0.0):
            cam = T.bench_camera(96, 54, t)
            want, _ = T.oracle_render(smaller.with_camera(cam), 96, 54, 6, 3)
            assert np.array_equal(bits(d.fetch(d.render(cam, 6, 3))), bits(want)), t
        assert d.context(1).render_variant()["workgroup_threads"] == 256


def test_c_host_drives_the_dist_renderer(tmp_path):
    """examples/trt_dist_demo: a C host, one process per GPU, the communicator id carried through a file -- here one rank, so
    communicator, group and assembly all run on the one GPU.  Its last frame must be the frame the single-GPU C demo's
    project_scene produces for the same camera (frame 2 of the orbit at 60 frames per second of scene time)."""
    import os
    import subprocess
    exe = os.path.join(T.ROOT, "examples", "trt_dist_demo")
    if not os.path.exists(exe):
        pytest.skip("examples/trt_dist_demo not built")
    sky = tmp_path / "colors"
    sky.mkdir()
    for f in T.FACES:
        (sky / (f + ".ppm")).write_bytes(T.golden_ppm_raw("colors", f))
    out = subprocess.run([exe, str(sky), "0", "1", str(tmp_path / "id"), "3", "160", "48"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    assert "3 frames 160x48 on 1 GPU(s)" in out.stdout
    from terminalraytracer_amd import host
    cam = host.orbit_camera(2 / 60.0, 160, 48)
    scene = S.demo_scene(T.sky("colors"), cam)
    want, _ = T.oracle_render(scene, 160, 48, 10, 10)
    assert T.fnv(want) in out.stdout, out.stdout


def build_rccl_stub():
    """tests/rccl_stub.cpp -> tests/_build/librccl_stub.so (TEST INFRASTRUCTURE: several ranks on one GPU); None if it cannot be built"""
    import os
    import subprocess
    build = os.path.join(T.ROOT, "tests", "_build")
    os.makedirs(build, exist_ok=True)
    so, src = os.path.join(build, "librccl_stub.so"), os.path.join(T.ROOT, "tests", "rccl_stub.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        rc = subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", so, src,
                             "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"], capture_output=True, text=True)
        if rc.returncode != 0:
            return None
    return so


@pytest.mark.parametrize("world,width,height,tile,depth,frames,rgb8",
                         [(2, 160, 48, 8, 3, 7, 0), (3, 67, 13, 4, 3, 8, 0), (3, 1920, 1080, 8, 3, 7, 0), (2, 1920, 1080, 8, 2, 5, 1), (3, 160, 48, 4, 3, 7, 1),
                          (2, 64, 9, 1, 1, 4, 0), (3, 40, 5, 8, 2, 5, 0), (3, 40, 5, 8, 2, 5, 1)],
                         ids=["2_ranks", "3_ranks_unequal_shards", "3_ranks_1080p", "2_ranks_1080p_rgb8", "3_ranks_rgb8",
                              "2_ranks_single_row_tiles_one_frame_in_flight", "3_ranks_two_without_rows", "3_ranks_two_without_rows_rgb8"])
def test_trt_dist_with_several_ranks_on_one_gpu(ctx, tmp_path, world, width, height, tile, depth, frames, rgb8):
    """The C-ABI multi-GPU path (csrc/trt_dist.hip) EXECUTED by several ranks: `world` processes of the C host
    examples/trt_dist_demo share this box's one GPU; the eight RCCL entry points the library binds by name come from the tests'
    stand-in (tests/rccl_stub.cpp, selected by TRT_RCCL_LIB; real RCCL refuses two ranks on one device).  What runs here and
    nowhere else on a one-GPU box: the peers' ncclSend branch, the root's ncclRecv offsets for ranks >= 1, shards of unequal
    height (13 rows in tiles of 4 over 3 ranks: 5 / 4 / 4), ranks that own NO row (5 rows in tiles of 8 over 3 ranks: nothing is
    sent or received for them), slots re-used while a gather is outstanding (more frames than slots), messages larger than the
    stand-in's staging slot, and the gather of the emitter's bytes (trt_dist_render_rgb8).
    Rank 0's last frame must be the frame one renderer produces for that camera."""
    import os
    import subprocess
    exe = os.path.join(T.ROOT, "examples", "trt_dist_demo")
    stub = build_rccl_stub()
    if not os.path.exists(exe) or stub is None:
        pytest.skip("examples/trt_dist_demo or the RCCL stand-in not built")
    sky = tmp_path / "colors"
    sky.mkdir()
    for f in T.FACES:
        (sky / (f + ".ppm")).write_bytes(T.golden_ppm_raw("colors", f))
    env = dict(os.environ, TRT_RCCL_LIB=stub, TRT_RCCL_STUB_SLOT_MB="8", TRT_RCCL_STUB_DEADLINE="60", GPU_MAX_HW_QUEUES="8")
    procs = [subprocess.Popen([exe, str(sky), str(r), str(world), str(tmp_path / "id"), str(frames), str(width), str(height), str(tile), str(depth),
                               "0", str(rgb8), "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, (r, err[-800:])
    from terminalraytracer_amd import host
    cam = host.orbit_camera((frames - 1) / 60.0, width, height)
    scene = S.demo_scene(T.sky("colors"), cam)
    if width * height <= 160 * 48:
        want, _ = T.oracle_render(scene, width, height, 10, 10)
    else:  # the single renderer's frame (itself checked against the oracle and the reference in the tests above)
        want = render(ctx, scene, width, height, 10, 10)
    fingerprint = T.fnv(T.oracle_rgb8(want)) if rgb8 else T.fnv(want)
    assert f"{frames} frames {width}x{height} on {world} GPU(s)" in outs[0][0] and fingerprint in outs[0][0], (outs[0][0], fingerprint)
    assert "STAND-IN (TRT_RCCL_LIB)" in outs[0][0], outs[0][0]  # the library says which library it bound in RCCL's place


def _write_scene_file(path, scene, cameras):
    """the input of tests/dist_ranks.c: int32 counts, then doubles (spheres, ground, lights, cameras), then the cubemap's texels"""
    sky = np.ascontiguousarray(scene.sky, dtype=np.uint8)
    with open(path, "wb") as fh:
        np.array([len(scene.spheres), len(scene.dir_lights), len(scene.point_lights), sky.shape[1], len(cameras)], dtype=np.int32).tofile(fh)
        for a in (scene.spheres, scene.ground, scene.dir_lights, scene.point_lights, np.stack(cameras)):
            np.ascontiguousarray(a, dtype=np.float64).tofile(fh)
        sky.tofile(fh)


@pytest.mark.parametrize("config", ["c4_2160p", "c5_orbit_f0_f59", "c3_rgb8"])
def test_the_eight_way_split_of_the_baseline_configs_on_one_gpu(tmp_path, config):
    """BASELINE configs[3] (3840x2160, 64 spheres, 8 bounces) and configs[4] (256 spheres, 12 bounces, the orbit) NAME 8 GPUs.  No
    8-GPU node is at hand, but the product path can still be split 8 ways and EXECUTED: eight ranks of trt_dist_* on this box's one
    GPU -- three processes (a box admits six: rank 0 alone, ranks 1-4 and 5-7 as threads of tests/dist_ranks.c), interleaved 8-row
    tiles, one gather per frame through the tests' stand-in for RCCL -- and rank 0's assembled frames must carry the hashes the
    GENUINE reference produced for the whole frames (tests/golden/golden_full.json): 2160p shards with 24.9 MB messages, the
    patch instantiation of the kernel on 1/8 shards (fewer than 16 M samples each), every one of the 8 receive offsets, three
    frames in flight with slots re-used.  Correctness only: nothing here says anything about xGMI."""
    import os
    import subprocess
    exe = os.path.join(T.ROOT, "tests", "_build", "dist_ranks")
    stub = build_rccl_stub()
    if stub is None or subprocess.run(["make", "-C", T.ROOT, "tests/_build/dist_ranks"], capture_output=True).returncode != 0 or not os.path.exists(exe):
        pytest.skip("tests/_build/dist_ranks or the RCCL stand-in could not be built")
    full = T.golden_full()
    if config == "c4_2160p":
        cases, rgb8 = [full["c4_2160p_64sph_b8"]] * 4, 0                    # four frames through three slots
    elif config == "c5_orbit_f0_f59":
        cases, rgb8 = [full[n] for n in ("c5_1080p_256sph_b12_f0", "c5_1080p_256sph_b12_f59", "c5_1080p_256sph_b12_f0", "c5_1080p_256sph_b12_f59")], 0
    else:
        cases, rgb8 = [full[n] for n in ("c3_1080p_64sph_b8_f0", "c3_1080p_64sph_b8_f19", "c3_1080p_64sph_b8_f59", "c3_1080p_64sph_b8")], 1
    first = cases[0]
    scene = T.full_scene(first)
    cams = [np.array(c["camera"], dtype=np.float64) for c in cases]
    _write_scene_file(tmp_path / "scene.bin", scene, cams)
    w, h, b = first["width"], first["height"], first["bounce_limit"]
    # staging: one slot holds a whole shard (2160p: 24.9 MB) and the ring holds every frame of the run, so that no sender ever waits
    env = dict(os.environ, TRT_RCCL_LIB=stub, TRT_RCCL_STUB_SLOT_MB="32", TRT_RCCL_STUB_RING="4", TRT_RCCL_STUB_DEADLINE="150", GPU_MAX_HW_QUEUES="8")
    procs = [subprocess.Popen([exe, str(tmp_path / "scene.bin"), ranks, "8", str(tmp_path / "id"), str(w), str(h), str(b), "8", "3", str(rgb8)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for ranks in ("0", "1,2,3,4", "5,6,7")]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=400))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-1200:]
    got = [line.split()[-1] for line in outs[0][0].splitlines() if line.startswith("frame ")]
    want = [c["rgb8_fnv" if rgb8 else "fb_fnv"] for c in cases]
    assert got == want, (got, want)
    assert all("STAND-IN (TRT_RCCL_LIB)" in err for _, err in outs), outs[0][1][-300:]
    print("\n" + "".join(err for _, err in outs))


def test_bench_two_ranks_through_the_c_abi_on_one_gpu():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), both ranks on this box's one
    GPU: torch.distributed (gloo) only carries the communicator id and the timing; the frame goes through trt_dist_* over the
    tests' stand-in for RCCL.  --check compares the gathered frame with a single renderer's, the timed frame is verified against
    the reference's hash."""
    import json
    import os
    import socket
    import subprocess
    import sys
    stub = build_rccl_stub()
    if stub is None:
        pytest.skip("the RCCL stand-in could not be built")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, TRT_RCCL_LIB=stub, TRT_RCCL_STUB_DEADLINE="120")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(T.ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--rccl-stand-in", "--check",
                          "--steps", "5", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=T.ROOT)
    assert out.returncode == 0, out.stderr[-1500:]
    assert "CHECK sharded(2) == single: True" in out.stderr
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["verified"] is True and "trt_dist_*" in line["config"]["multi_gpu_path"] and \
        "STAND-IN" in line["config"]["multi_gpu_path"], line["config"]
    assert line["rccl_ranks"] == 2 and "STAND-IN" in line["rccl_library"], (line["rccl_ranks"], line["rccl_library"])  # the communicator's own count

    # a product path that fails must fail the bench: no line, a non-zero exit code and the reason on stderr ...
    def run(extra, fail):
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                               "--master-port", str(port), os.path.join(T.ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--rccl-stand-in",
                               "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-verify"] + extra, capture_output=True, text=True, timeout=600,
                              env=dict(env, TRT_BENCH_FAIL_DIST="1") if fail else env, cwd=T.ROOT)
    out = run([], True)
    assert out.returncode != 0, out.stdout[-500:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")], out.stdout[-500:]
    assert "TRT_BENCH_FAIL_DIST is set" in out.stderr and "no fallback was allowed" in out.stderr, out.stderr[-1500:]
    # ... unless the PyTorch-level gather was explicitly allowed: then the line names the fallback and its reason
    out = run(["--allow-fallback"], True)
    assert out.returncode == 0, out.stderr[-1500:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["multi_gpu_path"].startswith("FALLBACK") and line["rccl_ranks"] is None, line["config"]


def test_sharded_renderer_world_of_one(ctx):
    from terminalraytracer_amd.distributed import HipShardRenderer
    case = next(c for c in SMALL if c["name"] == "synth64_128x72_b8")
    scene = T.golden_scene(case)
    r = HipShardRenderer(scene, 128, 72, 0, 1, 0, 8, 10)
    try:
        frame = r.render(scene.camera)
        import torch
        torch.cuda.synchronize()
        assert T.fnv(frame.cpu().numpy()) == case["fb_fnv"]
    finally:
        r.close()


def test_sharded_renderer_with_reserved_compute_units_and_three_frames_in_flight(ctx):
    """The multi-GPU defaults of bench.py on one rank: CU-masked render streams (trt_reserve_cus), three slots, the assembly
    on a stream of its own.  Every frame of a short orbit must come out as the reference's, in order."""
    import torch
    from terminalraytracer_amd.distributed import HipShardRenderer
    case = next(c for c in SMALL if c["name"] == "synth64_128x72_b8")
    scene = T.golden_scene(case)
    r = HipShardRenderer(scene, 128, 72, 0, 1, 0, 8, 10, depth=3, reserve_cus=8)
    try:
        assert r.external_streams and r.main is not None
        frames = []
        for _ in range(7):  # every slot reused at least twice
            frames.append(r.render(scene.camera))
            torch.cuda.synchronize()
            assert T.fnv(frames[-1].cpu().numpy()) == case["fb_fnv"]
        assert r.ctx.kernel_info()["compute_units"] >= 8
    finally:
        r.close()


def test_c_demo_driver_runs_the_reference_frame_loop(tmp_path):
    """examples/trt_demo: host C (scene literals, camera orbit, PPM loader, emitter) + GPU project_scene."""
    import os
    import subprocess
    exe = os.path.join(T.ROOT, "examples", "trt_demo")
    if not os.path.exists(exe):
        pytest.skip("examples/trt_demo not built")
    sky = tmp_path / "colors"
    sky.mkdir()
    for f in T.FACES:
        (sky / (f + ".ppm")).write_bytes(T.golden_ppm_raw("colors", f))
    out = subprocess.run([exe, str(sky), "3", "160", "48"], capture_output=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    assert b"3 frames 160x48" in out.stderr
    # three full emitter buffers went to stdout: 8 + (25*160+1)*48 + 1 bytes each plus the fps lines
    assert out.stdout.count(b"\033[48;2;") == 3 * 160 * 48


def test_orbit_animation_frames_match_oracle(ctx):
    """Config 5's moving camera: frames at several orbit times, pipelined over two contexts like the bench does."""
    from terminalraytracer_amd.distributed import HipShardRenderer
    import torch
    w, h = 240, 135
    scene = S.synth_scene(64, S.synth_sky(64), T.bench_camera(w, h), seed=1234)
    r = HipShardRenderer(scene, w, h, 0, 1, 0, 8, 10, depth=2)
    try:
        for t in (0.0, 0.5, 2.5, 10.0, 33.3):
            cam = T.bench_camera(w, h, t)
            frame = r.render(cam).clone()
            torch.cuda.synchronize()
            want, _ = T.oracle_render(scene.with_camera(cam), w, h, 8, 10)
            assert np.array_equal(bits(frame.cpu().numpy()), bits(want)), t
    finally:
        r.close()


def test_point_light_at_blocker_distance_forces_the_exact_branch(ctx):
    """The rounds kernel decides "is the blocker farther than the light?" (TRT.c:939-942) from bounds and forms the exact
    nudged blocker point only in a band of relative width ~1e-6.  Lights sitting on / within micrometres of a sphere's
    surface put whole regions of the image inside and just outside that band (both sides), the blocker being the very
    sphere the light touches; every kernel must still equal the oracle bit for bit."""
    w, h = 96, 54
    base = S.demo_scene(T.sky("synth"), T.bench_camera(w, h))
    centre, radius = base.spheres[4, :3], base.spheres[4, 3]  # the sphere at (0,-1,0), just above the ground
    lights = []
    for k, eps in enumerate([0.0, 1e-6, -1e-6, 2e-6, -2e-6, 1.0000001e-6, 0.5e-6, -0.5e-6, 1e-5, -1e-5]):
        n = np.array([np.cos(0.7 * k), -0.8, np.sin(0.7 * k)])
        n /= np.linalg.norm(n)
        lights.append(list(centre + n * (radius + eps)) + [1.0, 0.9, 0.8, 3.0])
    scene = S.SceneData(base.spheres, base.ground, base.dir_lights, np.array(lights), base.camera, base.sky)
    want, st = T.oracle_render(scene, w, h, 4, 10)
    assert st.shadow_rays > 10 * st.path_rays * 0.5  # eleven lights per hit
    for kernel in KERNELS:
        got = render(ctx, scene, w, h, 4, 10, kernel)
        assert np.array_equal(bits(got), bits(want)), kernel


@pytest.mark.parametrize("w,h,n,b,spp", [(33, 17, 6, 1, 10), (64, 36, 64, 3, 1), (50, 20, 17, 5, 7), (40, 12, 64, 8, 64),
                                           (257, 3, 100, 4, 10), (3, 257, 33, 2, 5), (16, 9, 64, 3, 500)])
def test_odd_parameters_against_the_oracle(ctx, w, h, n, b, spp):
    """No goldens here: the oracle (itself pinned to the reference) is the checker.  Bounce limit 1, one and many
    rays per pixel (more than a wave's worth of samples per pixel; 500: the jitter table pushes the shading rings out of LDS and
    the decoupled kernel must step aside), sphere counts that are not multiples of 8/32/64, frames narrower than a wave."""
    spheres = S.demo_spheres() if n == 6 else S.synth_spheres(n, seed=99)
    scene = S.synth_scene(n, T.sky("synth"), T.bench_camera(w, h, 2.5), seed=99).with_spheres(spheres)
    want, st = T.oracle_render(scene, w, h, b, spp)
    ctx.enable_counters(True)
    try:
        for kernel in KERNELS:
            got = render(ctx, scene, w, h, b, spp, kernel)
            assert np.array_equal(bits(got), bits(want)), kernel
            assert ctx.read_counters() == (st.path_rays, st.shadow_rays), kernel
    finally:
        ctx.enable_counters(False)


def _fuzz_scene(rng, w, h):
    n = int(rng.choice([0, 1, 2, 5, 9, 31, 32, 33, 64, 65, 100]))
    sph = np.zeros((n, 9))
    if n:
        sph[:, :3] = rng.normal(size=(n, 3)) * rng.choice([0.5, 2.0, 6.0])
        sph[:, 3] = rng.uniform(0.05, 1.2, n) * rng.choice([0.3, 1.0, 2.0])
        sph[:, 4:7] = rng.uniform(0, 1, (n, 3))
        sph[:, 7] = rng.choice([0.0, 0.3, 0.9, 1.0], n)
        sph[:, 8] = 100.0
        if n >= 5:  # exact ties: duplicated spheres with different materials (first index must win), nested and degenerate ones
            sph[n // 2, :4] = sph[1, :4]
            sph[n - 1, :4] = sph[1, :4]
            sph[2, :3] = sph[3, :3]
            sph[2, 3] = sph[3, 3] * 0.5  # concentric, smaller: always hidden from outside
            sph[4, 3] = 0.0               # zero radius
    ground = S.demo_ground().copy()
    if rng.random() < 0.5:
        ground[0:3] = rng.normal(size=3)
        ground[3:6] = rng.normal(size=3) * rng.choice([1.0, 0.01, 30.0])  # non-unit normals too
    ground[9] = rng.choice([0.0, 0.2, 1.0])
    nd, npt = int(rng.integers(0, 3)), int(rng.integers(0, 4))
    dl = np.concatenate([rng.normal(size=(nd, 3)), rng.uniform(0, 1.2, (nd, 3))], axis=1)
    pl = np.concatenate([rng.normal(size=(npt, 3)) * 3, rng.uniform(0, 1.2, (npt, 3)), rng.uniform(0.1, 30, (npt, 1))], axis=1)
    if npt and n:
        pl[0, :3] = sph[0, :3] + np.array([0.0, sph[0, 3], 0.0])  # a light exactly on a sphere's surface
    cam = T.bench_camera(w, h, float(rng.choice([0.0, 0.5, 2.5, 10.0, 33.3])))
    if rng.random() < 0.3 and n:
        cam[9:12] = sph[0, :3] + 0.3 * sph[0, 3]  # camera inside a sphere
    if rng.random() < 0.3:
        cam[9:12] = rng.normal(size=3) * 20
    return S.SceneData(sph, ground, dl.reshape(-1, 6), pl.reshape(-1, 7), cam, T.sky("synth"))


@pytest.mark.parametrize("seed", range(12))
def test_fuzzed_scenes_match_the_oracle(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(8, 72)), int(rng.integers(4, 40))
    b, spp = int(rng.integers(1, 9)), int(rng.choice([1, 3, 10]))
    scene = _fuzz_scene(rng, w, h)
    with np.errstate(all="ignore"):
        want, st = T.oracle_render(scene, w, h, b, spp)
    finite = np.isfinite(want).all()
    for kernel in KERNELS:
        got = render(ctx, scene, w, h, b, spp, kernel)
        if finite:
            assert np.array_equal(bits(got), bits(want)), (seed, kernel)
        else:  # NaNs (e.g. a camera exactly on a light) must at least sit in the same pixels with equal finite neighbours
            assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(bits(got[np.isfinite(want)]), bits(want[np.isfinite(want)]))


def test_rgb8_sharded_renderer_feeds_the_emitter(ctx):
    """SURVEY f-3: quantise on the device, move 3 bytes per pixel, patch the emitter on the host."""
    import zlib
    import torch
    from terminalraytracer_amd import host
    from terminalraytracer_amd.distributed import HipShardRenderer
    case = next(c for c in SMALL if c["name"] == "demo_160x48_b4")
    scene = T.golden_scene(case)
    r = HipShardRenderer(scene, 160, 48, 0, 1, 0, 4, 10, depth=2, rgb8=True)
    try:
        for _ in range(3):
            frame = r.render(scene.camera)
        torch.cuda.synchronize()
        rgb = frame.cpu().numpy()
    finally:
        r.close()
    assert rgb.dtype == np.uint8 and T.fnv(rgb) == case["rgb8_fnv"]
    em = host.Emitter(160, 48)
    em.patch_rgb8(rgb)
    assert em.bytes() == zlib.decompress(open(T.GOLDEN + "/emit_demo_160x48_b4.bin.z", "rb").read())


@pytest.mark.parametrize("cells", [(0, 0, 16, 16), (8, 2, 1, 1), (33, 7, 5, 3), (512, 256, 16, 16), (128, 64, 64, 64), (128, 64, 1, 1)],
                         ids=["off", "coarsest", "odd", "fine", "64 slabs", "no depth"])
def test_light_space_tables_never_change_a_frame(ctx, cells):
    """The production kernel reads a shadow ray's candidate spheres from per-light tables (csrc/trt_lightgrid.h).  Whatever
    their resolution -- or with the tables off, every shadow ray sweeping -- frames must equal the oracle bit for bit:
    64 and 256 spheres (one and four mask words per cell), lights inside / on / a hair outside spheres, many lights."""
    cases = [(S.synth_scene(64, T.sky("synth"), T.bench_camera(96, 54)), 96, 54, 8, 10),
             (S.synth_scene(256, T.sky("synth"), T.bench_camera(64, 36, 2.5)), 64, 36, 6, 4)]
    base = S.synth_scene(40, T.sky("synth"), T.bench_camera(80, 45, 10.0), seed=5)
    c, r = base.spheres[3, :3], base.spheres[3, 3]
    lights = [list(c) + [1.0, 1.0, 1.0, 5.0], list(c + [0.0, r * 0.999, 0.0]) + [0.2, 1.0, 0.3, 9.0],
              list(c + [r * (1 + 1e-9), 0.0, 0.0]) + [1.0, 0.3, 0.2, 9.0], list(c + [0.0, 0.0, r + 0.03]) + [0.5, 0.5, 1.0, 20.0],
              [0.0, 300.0, 0.0, 1.0, 1.0, 1.0, 9e4]]
    dirs = np.array([[0.0, -1.0, 0.0, 0.5, 0.5, 0.5], [1.0, -1e-9, 0.0, 0.3, 0.2, 0.1], [-0.3, -0.8, 0.55, 0.2, 0.3, 0.4]])
    cases.append((S.SceneData(base.spheres, base.ground, dirs, np.array(lights), base.camera, base.sky), 80, 45, 5, 5))
    try:
        for scene, w, h, b, spp in cases:
            want, _ = T.oracle_render(scene, w, h, b, spp)
            ctx.set_light_slabs(*cells[2:])  # the tables' depth coordinate (trt_lightgrid.h (5))
            ctx.set_light_grids(*cells[:2])
            got = render(ctx, scene, w, h, b, spp)
            assert np.array_equal(bits(got), bits(want)), (cells, len(scene.spheres))
    finally:
        ctx.set_light_slabs(16, 16)
        ctx.set_light_grids(128, 64)


def _decode_cells(cells, pool, bits=8):
    """list cells (csrc/trt_raygrid.h) -> list of tuples of sphere indices; None for a cell without a list.  bits: 8 per entry
    up to 256 spheres (7 inline, 8 per pool word), 16 above (3 inline, 4 per pool word)"""
    out = []
    per, mask = 64 // bits, (1 << bits) - 1
    for c in cells.tolist():
        ctl = c >> 56
        if ctl == 0xFF:
            out.append(None)
        elif ctl & 0x80:
            count, at = (c >> 32) & 0xFFFF, c & 0xFFFFFFFF
            out.append(tuple((int(pool[at + k // per]) >> (bits * (k % per))) & mask for k in range(count)))
        else:
            out.append(tuple((c >> (bits * k)) & mask for k in range(ctl)))
    return out


@pytest.mark.parametrize("cells", [(0, 0, 0), (2, 2, 0), (9, 5, 0), (64, 32, 0), (128, 48, 0), (64, 32, -1), (64, 16, 2), (7, 3, 1), (32, 8, 4), (48, 12, 3)],
                         ids=["off", "coarsest", "odd", "one_family_per_sphere", "fine", "default", "24_patches", "coarse_6_patches", "96_patches", "54_patches"])
def test_path_ray_tables_never_change_a_frame(ctx, cells):
    """A path ray's candidate spheres come from the direction table of its family (csrc/trt_raygrid.h): the eye, its mirror
    image in the ground, the sphere it starts on (the whole sphere or one of 6 m^2 patches of it), that family's mirror image.
    Whatever the tables' resolution and the number of patches -- or with the tables
    off, every path ray sweeping -- frames must equal the oracle bit for bit and the trace counts the reference's: 64 and 256
    spheres, mirror-heavy materials (long chains of families), a tilted ground with a non-unit normal, a perfect-mirror
    floor, the eye inside a sphere, touching / nested / duplicated / huge / tiny spheres, a moving eye (tables rebuilt)."""
    import test_raygrid as R
    cases = [(S.synth_scene(64, T.sky("synth"), T.bench_camera(96, 54)), 96, 54, 8, 10),
             (S.synth_scene(256, T.sky("synth"), T.bench_camera(64, 36, 2.5)), 64, 36, 12, 4),
             (S.synth_scene(64, T.sky("synth"), T.bench_camera(64, 36, 10.0), mirror_fraction=0.5), 64, 36, 8, 5)]
    cases += [(scene, 64, 36, 8, 3) for _, scene in R._odd_scenes()[1:]]
    base = S.synth_scene(64, T.sky("synth"), T.bench_camera(48, 27))
    cases += [(base.with_camera(T.bench_camera(48, 27, t)), 48, 27, 6, 2) for t in (0.0, 0.5, 33.3)]  # same spheres, the eye moves
    try:
        ctx.set_path_patches(cells[2])
        ctx.set_path_grids(*cells[:2])
        ctx.enable_counters(True)
        for scene, w, h, b, spp in cases:
            with np.errstate(all="ignore"):
                want, st = T.oracle_render(scene, w, h, b, spp)
            got = render(ctx, scene, w, h, b, spp)
            assert np.array_equal(bits(got), bits(want)), (cells, len(scene.spheres))
            assert ctx.read_counters() == (st.path_rays, st.shadow_rays)
            if cells[2] > 0 and cells[0] >= 32 and len(scene.spheres) >= 64 and w >= 64:
                # the patches must SERVE the rays, not send them to the sweep: a wrong patch would still be bit-exact
                diag = ctx.read_diagnostics()
                assert diag["swept_traces"] < 0.05 * 3 * diag["wave_loop_trips"], (cells, diag)
    finally:
        ctx.enable_counters(False)
        ctx.set_path_patches(-1)
        ctx.set_path_grids(64, 32)


@pytest.mark.parametrize("n", [257, 300, 700, 1030])
def test_scenes_of_more_than_256_spheres_keep_their_tables(ctx, n):
    """List cells index spheres with one byte up to 256 spheres; larger scenes use 16-bit entries -- for the light tables (up to
    65535 spheres) and, since round 5, for the path rays' family tables too (up to 1024 spheres; beyond that the path rays
    sweep).  Frames and trace counts against the oracle: every table on, one family per sphere / 6 patches, coarse tables, tables off."""
    scene = S.synth_scene(n, T.sky("synth"), T.bench_camera(72, 40, 2.5), seed=11)
    want, st = T.oracle_render(scene, 72, 40, 6, 4)
    ctx.enable_counters(True)
    try:
        for light_cells, path_cells, m in (((128, 64), (64, 32), 0), ((128, 64), (64, 16), 1), ((9, 3), (7, 3), 0), ((0, 0), (0, 0), 0)):
            ctx.set_light_grids(*light_cells)
            ctx.set_path_patches(m)
            ctx.set_path_grids(*path_cells)
            got = render(ctx, scene, 72, 40, 6, 4)
            assert np.array_equal(bits(got), bits(want)), (n, light_cells, path_cells, m)
            assert ctx.read_counters() == (st.path_rays, st.shadow_rays)
            diag = ctx.read_diagnostics()
            info = ctx.read_path_tables(scene.camera)[0]
            assert info["enabled"] == (1 if path_cells[0] and n <= 1024 else 0), info
            if light_cells == (128, 64) and n <= 1024:  # every trace reads a list: only rays beyond a table's range make their wave sweep
                assert diag["swept_traces"] < 0.05 * 3 * diag["wave_loop_trips"], (n, path_cells, m, diag)
            elif light_cells == (128, 64):  # beyond 1024 spheres the path rays sweep, the shadow rays read lists
                assert diag["wave_loop_trips"] <= diag["swept_traces"] < 2 * diag["wave_loop_trips"], diag
        # The library's own settings, no counters: scenes whose LDS image no longer fits four times per CU (from ~290 spheres) run the
        # plain rounds in 1024-thread workgroups, one image for sixteen waves (render_rounds_kernel<.., BIG>) -- the same frame.
        ctx.enable_counters(False)
        ctx.set_light_grids(128, 64)
        ctx.set_path_patches(-1)
        ctx.set_path_grids(64, 32)
        got = render(ctx, scene, 72, 40, 6, 4)
        assert np.array_equal(bits(got), bits(want)), n
        assert ctx.render_variant()["workgroup_threads"] == (1024 if 290 < n <= 1024 else 256), (n, ctx.render_variant())
    finally:
        ctx.enable_counters(False)
        ctx.set_light_grids(128, 64)
        ctx.set_path_patches(-1)
        ctx.set_path_grids(64, 32)


def _degenerate_scenes():
    base = S.synth_scene(24, T.sky("synth"), T.bench_camera(40, 24, 2.5), seed=3)

    def with_ground(point=None, normal=None, refl=None):
        g = base.ground.copy()
        if point is not None:
            g[0:3] = point
        if normal is not None:
            g[3:6] = normal
        if refl is not None:
            g[9] = g[14] = refl
        return S.SceneData(base.spheres, g, base.dir_lights, base.point_lights, base.camera, base.sky)

    out = [("ground without a normal", with_ground(normal=[0.0, 0.0, 0.0])),
           ("ground with a vanishing normal", with_ground(normal=[0.0, 1e-200, 0.0])),
           ("ground with a huge normal", with_ground(normal=[0.0, 1e150, 1e150], refl=1.0)),
           ("vertical mirror ground through the scene", with_ground(point=[0.3, 0.0, 0.0], normal=[1.0, 0.0, 0.0], refl=1.0))]
    cam = base.camera.copy()
    cam[10] = -2.0  # the eye exactly on the ground plane
    out.append(("eye on the ground plane", base.with_camera(cam)))
    cam = base.camera.copy()
    cam[9:12] = base.spheres[5, :3]  # the eye at a sphere's centre
    out.append(("eye at a sphere's centre", base.with_camera(cam)))
    sph = base.spheres.copy()
    sph[0, 3] = -0.4   # a negative radius (r*r is what the reference uses)
    sph[1, 3] = 0.0
    sph[2, :3] = [1e7, -3e6, 2e6]  # one sphere very far away: the tables' range explodes
    sph[3, :3] = sph[4, :3]        # concentric twins with equal radii: exact ties
    sph[3, 3] = sph[4, 3]
    out.append(("odd radii, a far sphere, exact twins", base.with_spheres(sph)))
    far = base.spheres.copy()
    far[:, :3] = far[:, :3] * 1e5  # an enormous scene: hit points lose digits against the 1e-6 nudge
    far[:, 3] *= 1e5
    gf = base.ground.copy()
    gf[1] *= 1e5
    cf = base.camera.copy()
    cf[9:12] *= 1e5
    out.append(("a scene 1e5 times larger", S.SceneData(far, gf, base.dir_lights, base.point_lights * np.array([1e5, 1e5, 1e5, 1, 1, 1, 1e10]), cf, base.sky)))
    pl = base.point_lights.copy()
    pl[0, :3] = base.camera[9:12]  # a light at the eye
    out.append(("a light at the eye", S.SceneData(base.spheres, base.ground, base.dir_lights, pl, base.camera, base.sky)))
    return out


@pytest.mark.parametrize("name,scene", _degenerate_scenes(), ids=[n for n, _ in _degenerate_scenes()])
def test_degenerate_scenes_match_the_oracle(ctx, name, scene):
    """Inputs the candidate tables must survive: grounds whose mirror images are NaN or astronomically far, the eye on the
    ground or at a sphere's centre, negative / zero radii, one sphere 1e7 away, exact twins, a scene 1e5 times larger, a
    light at the eye.  NaN pixels must sit where the oracle's do; everything else bit for bit; counts equal."""
    w, h, b, spp = 40, 24, 6, 3
    with np.errstate(all="ignore"):
        want, st = T.oracle_render(scene, w, h, b, spp)
    ctx.enable_counters(True)
    try:
        for kernel in KERNELS:
            got = render(ctx, scene, w, h, b, spp, kernel)
            finite = np.isfinite(want)
            assert np.array_equal(np.isnan(got), np.isnan(want)), (name, kernel)
            assert np.array_equal(bits(got[finite]), bits(want[finite])), (name, kernel)
            assert ctx.read_counters() == (st.path_rays, st.shadow_rays), (name, kernel)
    finally:
        ctx.enable_counters(False)


@pytest.mark.parametrize("tables", [(64, 32), (0, 0)], ids=["tables", "sweep"])
def test_refraction_extension_matches_its_cpu_restatement(ctx, tables):
    """EXTENSION, PARITY UNPINNED: the reference has no refraction (TRT.c:114-119, :657), so there is nothing of the reference's
    to compare with.  trt_set_refraction selects another instantiation of the production kernel; its frames must equal
    oracle/trt_oracle.c's restatement of the same semantics bit for bit (glass of index 1.5, an index below 1 with total
    reflection from outside, index 1, nested and touching refractors, a refractor on the floor), the trace counts too;
    with every index 0, and with the extension switched off again, the frame must be the reference path's."""
    base = S.synth_scene(24, T.sky("synth"), T.bench_camera(96, 54, 2.5), seed=3)
    sph = base.spheres.copy()
    sph[5, :3] = sph[4, :3]
    sph[5, 3] = sph[4, 3] * 0.5                     # a sphere nested in a refractor
    sph[7, :3] = sph[6, :3] + [sph[6, 3] + sph[7, 3], 0.0, 0.0]  # touching
    sph[9, 1] = -2.0 + sph[9, 3] * 0.8              # one that dips into the floor
    scene = base.with_spheres(sph)
    ior = np.where(np.arange(24) % 3 == 0, 1.5, 0.0)
    ior[1], ior[2], ior[4], ior[7] = 0.7, 1.0, 1.33, 2.4
    w, h, b, spp = 96, 54, 10, 4
    try:
        ctx.set_path_grids(*tables)
        plain, _ = T.oracle_render(scene, w, h, b, spp)
        want, st = T.oracle_render_refractive(scene, ior, w, h, b, spp)
        assert (want != plain).any()
        ctx.enable_counters(True)
        ctx.set_refraction(ior)
        got = render(ctx, scene, w, h, b, spp)
        assert np.array_equal(bits(got), bits(want))
        assert ctx.read_counters() == (st.path_rays, st.shadow_rays)
        ctx.set_refraction(np.zeros(24))
        assert np.array_equal(bits(render(ctx, scene, w, h, b, spp)), bits(plain))
        ctx.set_refraction(ior[:5])  # the wrong number of indices for this scene
        with pytest.raises(hip.TrtError):
            render(ctx, scene, w, h, b, spp)
        ctx.set_refraction(None)
        assert np.array_equal(bits(render(ctx, scene, w, h, b, spp)), bits(plain))
    finally:
        ctx.enable_counters(False)
        ctx.set_refraction(None)
        ctx.set_path_grids(64, 32)


def test_device_built_path_tables_equal_the_host_reference_builder(ctx):
    """The library forms the families' cones and marks and packs the cells on the GPU; tests/test_raygrid.py proves the HOST
    builder conservative.  Both run the same predicates (+ - * / sqrt only), so every cell must list the same spheres."""
    import test_raygrid as R
    lib = R.build_checker()
    base = S.synth_scene(40, T.sky("synth"), T.bench_camera(32, 18, 10.0), seed=5)
    g = base.ground.copy()
    g[0:6] = [0.3, -1.25, 0.2, 0.1, 2.0, -0.2]
    scenes = [S.synth_scene(64, T.sky("synth"), T.bench_camera(32, 18)), S.synth_scene(256, T.sky("synth"), T.bench_camera(32, 18, 2.5)),
              S.SceneData(base.spheres, g, base.dir_lights, base.point_lights, base.camera, base.sky), R._odd_scenes()[4][1],
              S.synth_scene(300, T.sky("synth"), T.bench_camera(32, 18, 2.5), seed=11)]  # 16-bit entries, the wide builder
    try:
        for scene in scenes:
            wide = len(scene.spheres) > 256
            for ge, gs, m in ((64, 16, 0), (11, 3, 0), (16, 8, 1)) if wide else ((64, 32, 0), (11, 3, 0), (64, 8, 2), (16, 6, 1)):
                ctx.set_path_patches(m)
                ctx.set_path_grids(ge, gs)
                ctx.set_scene(scene)
                info, cells, pool = ctx.read_path_tables(scene.camera)
                n = len(scene.spheres)
                P = 6 * m * m if m else 1
                assert ctx.path_patches() == (m, P)
                assert info["enabled"] == 1 and info["cells"] == 2 * 6 * ge * ge + 2 * n * P * 6 * gs * gs == len(cells)
                sph = np.ascontiguousarray(scene.spheres, dtype=np.float64)
                ground = np.ascontiguousarray(scene.ground, dtype=np.float64)
                eye = np.ascontiguousarray(scene.camera[9:12], dtype=np.float64)
                want_cells = np.zeros(len(cells), dtype=np.uint64)
                want_pool = np.zeros((2 if wide else 1) * len(cells) + 16, dtype=np.uint64)
                used = lib.raygrid_host_cells(sph.ctypes.data, n, ground.ctypes.data, eye.ctypes.data, ge, gs, m, want_cells.ctypes.data,
                                              want_pool.ctypes.data, len(want_pool))
                assert 0 <= used <= len(want_pool)
                got, want = _decode_cells(cells, pool, 16 if wide else 8), _decode_cells(want_cells, want_pool, 16 if wide else 8)
                assert all(e is None or (list(e) == sorted(set(e)) and all(0 <= i < n for i in e)) for e in got)  # ascending sphere indices
                # a pool too small for the long lists of a very coarse table leaves some cells without a list (their rays
                # sweep); WHICH cells depends on the order the cells reserved their words in: compare the others
                bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b and a is not None and b is not None]
                assert not bad, (n, ge, gs, len(bad), bad[:3], [got[i] for i in bad[:3]], [want[i] for i in bad[:3]])
                if ge == 64:  # the library's resolutions: every cell has its list
                    assert None not in got and None not in want
    finally:
        ctx.set_path_patches(-1)
        ctx.set_path_grids(64, 32)


def test_the_tables_serve_nearly_every_trace(ctx):
    """The sweep over all spheres is only the fall-back: on the north-star scene fewer than 2 % of the wave-level traces may
    take it (rays beyond a table's range: ground points near the horizon)."""
    scene = S.synth_scene(64, T.sky("synth"), T.bench_camera(480, 270))
    ctx.enable_counters(True)
    try:
        render(ctx, scene, 480, 270, 8, 10)
        path, shadow = ctx.read_counters()
        diag = ctx.read_diagnostics()
    finally:
        ctx.enable_counters(False)
    traces = diag["wave_loop_trips"] * 3  # one path trace and two shadow stages per round, at most
    assert 0 <= diag["swept_traces"] < 0.02 * traces, diag


def test_device_built_light_tables_equal_the_host_reference_builder(ctx):
    """The library marks the cells of the light-space tables on the GPU (one thread per cell); tests/test_lightgrid.py proves
    the HOST builder conservative.  Both run the same predicates (+ - * / sqrt only), so the tables must agree word for word."""
    import os
    import subprocess
    build = os.path.join(T.ROOT, "tests", "_build")
    os.makedirs(build, exist_ok=True)
    so = os.path.join(build, "liblightgridcheck.so")
    inc = os.path.join(T.ROOT, "terminalraytracer_amd", "csrc")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-I" + inc, "-o", so,
                           os.path.join(T.ROOT, "tests", "lightgrid_check.c"), "-lm"])
    lib = C.CDLL(so)
    lib.lightgrid_host_table.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.lightgrid_host_table.restype = C.c_long
    base = S.synth_scene(40, T.sky("synth"), T.bench_camera(32, 18), seed=5)
    c, r = base.spheres[3, :3], base.spheres[3, 3]
    lights = np.array([list(c + [0.0, r * 0.999, 0.0]) + [0.2, 1.0, 0.3, 9.0], list(c + [r * (1 + 1e-9), 0.0, 0.0]) + [1.0, 0.3, 0.2, 9.0],
                       [0.3, 7.0, -2.0, 1.0, 1.0, 1.0, 50.0], [40.0, 3.0, 11.0, 1.0, 1.0, 1.0, 900.0]])
    dirs = np.array([[0.0, -1.0, 0.0, 0.5, 0.5, 0.5], [1.0, -1e-9, 0.0, 0.3, 0.2, 0.1], [-0.3, -0.8, 0.55, 0.2, 0.3, 0.4]])
    scenes = [S.synth_scene(64, T.sky("synth"), T.bench_camera(32, 18)), S.synth_scene(256, T.sky("synth"), T.bench_camera(32, 18)),
              S.SceneData(base.spheres, base.ground, dirs, lights, base.camera, base.sky)]
    try:
        for scene in scenes:
            sph = np.ascontiguousarray(scene.spheres, dtype=np.float64)
            n, words = len(sph), max(1, (len(sph) + 63) // 64)
            # cells per side, and the tables' depth coordinate (slabs along a directional light, shells about a point light)
            for gd, gp, sd, sp in ((128, 64, 16, 16), (19, 5, 1, 1), (33, 9, 7, 3)):
                ctx.set_scene(scene)
                ctx.set_light_slabs(sd, sp)
                ctx.set_light_grids(gd, gp)
                for kind, count, g, depth, cells in ((0, len(scene.dir_lights), gd, sd, sd * gd * gd), (1, len(scene.point_lights), gp, sp, sp * 6 * gp * gp)):
                    for i in range(count):
                        v = np.ascontiguousarray(-scene.dir_lights[i, :3] if kind == 0 else scene.point_lights[i, :3], dtype=np.float64)
                        want = np.zeros(cells * words, dtype=np.uint64)
                        bits = lib.lightgrid_host_table(sph.ctypes.data, n, kind, v.ctypes.data, g, depth, want.ctypes.data)
                        got = ctx.read_light_grid(kind, i, cells * words)
                        assert bits > 0 and len(got) == cells * words
                        assert np.array_equal(got, want), (n, kind, i, g, int((got != want).sum()))
    finally:
        ctx.set_light_slabs(16, 16)
        ctx.set_light_grids(128, 64)


def test_lean_normalisation_and_sqrt_equal_the_plain_operators(ctx):
    """unit() shares one reciprocal refinement between its three divisions and takes a square root without the compiler's
    range handling when a whole wave is mid-range, and falls back to the plain operators otherwise (csrc/trt_device.hpp).
    Both must give the bits of `/` and sqrt: millions of vectors of every kind, laid out so that some waves are uniformly
    mid-range (short path taken) and others mix in zeros, denormals, huge, tiny, inf and NaN (fallback taken)."""
    rng = np.random.default_rng(3)
    n = 1 << 21
    v = rng.normal(size=(n, 4)) * 10.0 ** rng.uniform(-3, 3, (n, 1))
    v[:, 3] = np.abs(v[:, 3])
    blocks = v.reshape(-1, 64, 4)                          # one wave per block of 64 records
    kinds = rng.integers(0, 8, blocks.shape[0])
    blocks[kinds == 1, :, 0] = 0.0                         # exact zeros in a component, both signs
    blocks[kinds == 1, ::2, 1] = -0.0
    blocks[kinds == 2] *= 1e-160                           # squared length underflows towards denormals
    blocks[kinds == 3] *= 1e160                            # squared length overflows
    blocks[kinds == 4, :, 2] *= 1e-120                     # one component far below the others: outside the window
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 4.9e-324, 1e-310, 1.7e308, 1e-4, 1.0000000001e-4, 0.0001])
    blocks[kinds == 5, ::7, :] = rng.choice(special, (int((kinds == 5).sum()), len(range(0, 64, 7)), 4))
    blocks[kinds == 6] = np.round(blocks[kinds == 6] * 4) / 4        # many exactly representable quotients and ties
    blocks[kinds == 7, :, :3] *= 10.0 ** rng.uniform(-89, 89, (int((kinds == 7).sum()), 1, 1))  # around the window's edges (2^+-300)
    with np.errstate(all="ignore"):
        fast, ref = ctx.selftest_unit(blocks.reshape(-1, 4))
        # the plain operators themselves against the host's IEEE arithmetic
        x = blocks.reshape(-1, 4)
        length = np.sqrt(x[:, 0] * x[:, 0] + x[:, 1] * x[:, 1] + x[:, 2] * x[:, 2])
        host = np.where((length > 0.0001)[:, None], x[:, :3] / length[:, None], x[:, :3])
        host_root = np.sqrt(x[:, 3])
    same = (bits(fast) == bits(ref)) | (np.isnan(fast) & np.isnan(ref))
    assert same.all(), (int((~same).sum()), blocks.reshape(-1, 4)[np.argwhere(~same)[0][0]])
    ok = (bits(ref[:, :3]) == bits(host)) | (np.isnan(ref[:, :3]) & np.isnan(host))
    assert ok.all(), int((~ok).sum())
    ok = (bits(ref[:, 3]) == bits(host_root)) | (np.isnan(ref[:, 3]) & np.isnan(host_root))
    assert ok.all(), int((~ok).sum())


def test_drop_in_symbol_may_be_called_from_several_threads(tmp_path):
    """The reference's project_scene is a pure function; a host may call it from several threads.  The drop-in layer shares
    one device context behind a lock: four threads x five calls must all produce the same frame."""
    import os
    import subprocess
    exe = str(tmp_path / "mt_drop_in")
    libdir = os.path.join(T.ROOT, "terminalraytracer_amd")
    subprocess.check_call(["gcc", "-O1", "-I" + os.path.join(T.ROOT, "include"), "-o", exe, os.path.join(T.ROOT, "tests", "mt_drop_in.c"),
                           "-L" + libdir, "-ltrt_hip", "-Wl,-rpath," + libdir, "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "identical" in out.stdout, out.stdout + out.stderr


def _shadow_lane_activity(ctx, scene, w, h, b, spp, kernel):
    ctx.enable_counters(True)
    try:
        frame = render(ctx, scene, w, h, b, spp, kernel)
        path, shadow = ctx.read_counters()
        passes = ctx.read_diagnostics()["shading_passes"]
    finally:
        ctx.enable_counters(False)
    hits = shadow / (len(scene.dir_lights) + len(scene.point_lights))
    return frame, hits / (64.0 * passes)


def test_compaction_fills_the_shadow_lanes_of_the_bench_frame(ctx):
    """BASELINE config 3 at full size with the shading decoupled from the owning lane (what ships for it) and with the
    decoupling forced off: both frames are the one the genuine reference produced (FNV of tests/golden/golden_full.json), and
    the shadow stage runs with more than 90 % of its lanes busy (61 % without: the share of path rays that hit something)."""
    case = T.golden_full()["c3_1080p_64sph_b8"]
    w, h, b, spp = case["width"], case["height"], case["bounce_limit"], case["rays_per_pixel"]
    scene = T.full_scene(case)
    plain, idle = _shadow_lane_activity(ctx, scene, w, h, b, spp, PLAIN)
    packed, busy = _shadow_lane_activity(ctx, scene, w, h, b, spp, COMPACT)
    shipped, default = _shadow_lane_activity(ctx, scene, w, h, b, spp, hip.Context.PRODUCTION)
    assert T.fnv(plain) == case["fb_fnv"] and T.fnv(packed) == case["fb_fnv"] and T.fnv(shipped) == case["fb_fnv"]
    assert idle < 0.7 and busy > 0.9 and default > 0.9, (idle, busy, default)


def test_compaction_default_policy(ctx):
    """trt_set_compaction(-1): on for launches of 16 M samples or more of scenes with two lights or more whose rings fit in LDS
    (the bench frame: the test above); off for small launches (pipelined shards are faster on 256-thread workgroups), for a
    single light (a wash) and for a 256-sphere scene (no room: forcing it on changes nothing either).  Every frame equals the
    oracle's."""
    w, h = 480, 270
    base = S.synth_scene(64, T.sky("synth"), T.bench_camera(w, h, 2.5), seed=7)
    big = S.synth_scene(256, T.sky("synth"), T.bench_camera(w, h, 2.5), seed=7)
    assert len(base.dir_lights) + len(base.point_lights) == 2
    for scene, kernel, decoupled in ((base, hip.Context.PRODUCTION, False), (base, COMPACT, True), (big, hip.Context.PRODUCTION, False),
                                     (big, COMPACT, False)):
        frame, activity = _shadow_lane_activity(ctx, scene, w, h, 8, 10, kernel)
        assert (activity > 0.85) == decoupled and (decoupled or activity < 0.8), (len(scene.spheres), kernel, activity)
        assert ctx.render_variant()["decoupled"] == decoupled
        want, _ = T.oracle_render(scene, w, h, 8, 10)
        assert np.array_equal(bits(frame), bits(want))
    # at the bench frame's size: two lights decoupled (the test above), one light not
    case = T.golden_full()["c3_1080p_64sph_b8"]
    full = T.full_scene(case)
    one = S.SceneData(full.spheres, full.ground, full.dir_lights, full.point_lights[:0], full.camera, full.sky)
    _, activity = _shadow_lane_activity(ctx, one, case["width"], case["height"], 8, 10, hip.Context.PRODUCTION)
    assert activity < 0.8 and not ctx.render_variant()["decoupled"]


@pytest.mark.parametrize("seed", range(16))
def test_compaction_on_fuzzed_scenes_with_many_lights(ctx, seed):
    """The decoupled shading against the oracle where its bookkeeping is busiest: up to six lights of each kind, bounce
    limits from 1 (every hit ends its sample: the ring is flushed every round) to 12, 1 / 3 / 10 rays per pixel, frames
    narrower than a wave, scenes without spheres."""
    rng = np.random.default_rng(7000 + seed)
    w, h = int(rng.integers(8, 96)), int(rng.integers(4, 54))
    b, spp = int(rng.choice([1, 1, 2, 3, 8, 12])), int(rng.choice([1, 3, 10]))
    scene = _fuzz_scene(rng, w, h)
    nd, npt = int(rng.integers(0, 7)), int(rng.integers(0, 7))
    dl = np.concatenate([rng.normal(size=(nd, 3)) - [0, 1.0, 0], rng.uniform(0, 0.6, (nd, 3))], axis=1)
    pl = np.concatenate([rng.normal(size=(npt, 3)) * 10.0 ** rng.uniform(-1, 1.5, (npt, 1)), rng.uniform(0, 0.6, (npt, 3)),
                         rng.uniform(0.0, 50.0, (npt, 1))], axis=1)
    scene = S.SceneData(scene.spheres, scene.ground, dl.reshape(-1, 6), pl.reshape(-1, 7), scene.camera, scene.sky)
    with np.errstate(all="ignore"):
        want, st = T.oracle_render(scene, w, h, b, spp)
    ctx.enable_counters(True)
    try:
        got = render(ctx, scene, w, h, b, spp, COMPACT)
        counts = ctx.read_counters()
    finally:
        ctx.enable_counters(False)
    finite = np.isfinite(want)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(bits(got[finite]), bits(want[finite]))
    assert counts == (st.path_rays, st.shadow_rays)
    assert np.array_equal(bits(render(ctx, scene, w, h, b, spp, COMPACT)), bits(got))  # and without the counting variant


def test_render_kernels_keep_four_waves_per_simd(ctx):
    """Occupancy is part of the design: the plain rounds at <= 128 VGPRs in 256-thread workgroups, four to
    a CU; the decoupled kernel at <= 128 in one 1024-thread workgroup per CU.  hipFuncGetAttributes / the occupancy query say so
    for the kernel that the next frame runs."""
    case = T.golden_full()["c3_1080p_64sph_b8"]
    scene = T.full_scene(case)
    for kernel, threads, blocks in ((PLAIN, 256, 4), (COMPACT, 1024, 1)):
        render(ctx, scene, 64, 36, 2, 1, kernel)
        info, variant = ctx.kernel_info(), ctx.render_variant()
        assert variant == {"decoupled": kernel == COMPACT, "workgroup_threads": threads}
        assert info["vgprs"] <= 128 and info["max_blocks_per_cu"] == blocks, (kernel, info)


# ---- one copy of the scene's tables per device (trt_share_scene), the pool of long lists ----

def test_contexts_share_one_copy_of_the_scene_tables(ctx):
    """trt_share_scene: several contexts render ONE scene from one copy of its primitives, cubemap and candidate tables (the frame
    slots of a trt_dist: TRT.c:1296-1306 builds the scene once, TRT.c:1327-1339 moves the camera per frame); only the eye's two tables
    -- a slot each in the shared allocation -- are a context's own.  Three sharers render three cameras AT THE SAME TIME (their
    streams overlap), each frame must be the oracle's; the table setters refuse while shared; trt_set_scene gives a sharer tables
    of its own again; the tables outlive their first owner."""
    import torch
    scene = S.synth_scene(256, T.sky("synth"), T.bench_camera(96, 54))  # patches: the large tables
    cams = [T.bench_camera(96, 54, t) for t in (1.0, 2.5, 10.0)]
    want = [T.oracle_render(scene.with_camera(c), 96, 54, 6, 4)[0] for c in cams]
    owner = hip.Context(0)
    owner.set_scene(scene)
    alone = owner.scene_info()
    sharers = [hip.Context(0) for _ in range(2)]
    try:
        for c in sharers:
            c.share_scene(owner)
        info = owner.scene_info()
        assert info["sharers"] == 3 and info["table_bytes"] == alone["table_bytes"] and sharers[0].scene_info() == info
        bufs = [torch.zeros(54 * 96 * 3, dtype=torch.float64, device="cuda:0") for _ in cams]
        for _ in range(3):  # frames in flight on three streams, the eye's tables rebuilt per context
            for c, cam, buf in zip([owner] + sharers, cams, bufs):
                c.render_device(cam, hip.RowSet.whole(96, 54), 6, 4, buf.data_ptr(), buf.numel() * 8)
        for c in [owner] + sharers:
            c.synchronize()
        for buf, w in zip(bufs, want):
            assert np.array_equal(bits(buf.cpu().numpy().reshape(54, 96, 3)), bits(w))
        with pytest.raises(hip.TrtError):
            sharers[0].set_path_grids(64, 16)
        with pytest.raises(hip.TrtError):
            owner.set_light_grids(64, 32)
        # a sharer that is given another scene builds tables of its own and leaves the others alone
        other = S.synth_scene(64, T.sky("synth"), T.bench_camera(96, 54), seed=9)
        sharers[1].set_scene(other)
        assert sharers[1].scene_info()["sharers"] == 1 and owner.scene_info()["sharers"] == 2
        got = sharers[1].render_host(other.camera, hip.RowSet.whole(96, 54), 6, 4)
        assert np.array_equal(bits(got), bits(T.oracle_render(other, 96, 54, 6, 4)[0]))
        # the first owner goes; the remaining sharer keeps rendering from the tables
        owner.close()
        owner = None
        assert sharers[0].scene_info()["sharers"] == 1
        got = sharers[0].render_host(cams[2], hip.RowSet.whole(96, 54), 6, 4)
        assert np.array_equal(bits(got), bits(want[2]))
        sharers[0].set_path_grids(64, 16)  # alone again: the setters work
        got = sharers[0].render_host(cams[0], hip.RowSet.whole(96, 54), 6, 4)
        assert np.array_equal(bits(got), bits(want[0]))
    finally:
        for c in sharers + ([owner] if owner else []):
            c.close()


def test_nine_contexts_cannot_share_one_scene(ctx):
    scene = S.synth_scene(16, T.sky("synth"), T.bench_camera(32, 18))
    owner = hip.Context(0)
    owner.set_scene(scene)
    others = [hip.Context(0) for _ in range(8)]
    try:
        for c in others[:7]:
            c.share_scene(owner)
        with pytest.raises(hip.TrtError):
            others[7].share_scene(owner)
        others[3].close()  # a slot comes free
        others[7].share_scene(owner)
        got = others[7].render_host(scene.camera, hip.RowSet.whole(32, 18), 4, 2)
        assert np.array_equal(bits(got), bits(T.oracle_render(scene, 32, 18, 4, 2)[0]))
    finally:
        for i, c in enumerate(others):
            if i != 3:
                c.close()
        owner.close()


def test_an_exhausted_list_pool_only_costs_sweeps(ctx):
    """Lists longer than seven entries live in a pool; a list that finds no room leaves its cell without one and the cell's rays
    sweep (csrc/trt_tables.hip pack_cell).  The pool's counter keeps counting after exhaustion -- with 64 bits, so that it cannot wrap
    and hand out words that earlier cells point to.  A dense scene with the pool capped at 64 / 4096 words: nearly every long list
    is lost, frames and trace counts stay the oracle's, and more traces sweep than with the pool the library would have chosen."""
    scene = S.synth_scene(256, T.sky("synth"), T.bench_camera(64, 36))
    big = scene.spheres.copy()
    big[:, 3] *= 1.6  # fat, overlapping spheres: long lists everywhere
    scene = scene.with_spheres(big)
    want, st = T.oracle_render(scene, 64, 36, 6, 4)
    c = hip.Context(0)
    try:
        c.enable_counters(True)
        swept = {}
        for cap in (0, 4096, 64):
            c.set_list_pool_words(cap)
            c.set_scene(scene)
            got = c.render_host(scene.camera, hip.RowSet.whole(64, 36), 6, 4)
            assert np.array_equal(bits(got), bits(want)), cap
            assert c.read_counters() == (st.path_rays, st.shadow_rays)
            swept[cap] = c.read_diagnostics()["swept_traces"]
            info, cells, pool = c.read_path_tables(scene.camera)
            if cap:
                assert info["pool_used_scene"] > cap, info  # the counter went on counting past the cap ...
                lost = sum(1 for x in cells if (int(x) >> 56) == 0xFF)
                assert lost > 1000, lost    # ... and the lists that found no room say so
        assert swept[64] > swept[4096] >= swept[0], swept
    finally:
        c.close()


def test_an_exhausted_eye_table_pool_only_costs_sweeps(ctx):
    """The eye's two tables are rebuilt for every camera on the frame's stream, their long lists into a part of the pool of their
    own whose counter nobody reads back.  With that part capped at 16 words nearly every long list of the eye's tables finds no
    room: pack_cell's limit turns those cells into "no list" (never into a list at an offset beyond the part), the primary rays
    and the ground's reflections of them sweep, and frames and trace counts of a MOVING camera stay the oracle's.  (Round 4's
    "nopack" experiment faulted in the render kernel on cells that no builder had written; cells now start out as "no list".)"""
    scene = S.synth_scene(256, T.sky("synth"), T.bench_camera(64, 36))
    big = scene.spheres.copy()
    big[:, 3] *= 1.6  # fat, overlapping spheres: long lists everywhere
    scene = scene.with_spheres(big)
    c = hip.Context(0)
    try:
        c.enable_counters(True)
        swept = {}
        for cap in (0, 16):
            c.set_list_pool_words(cap)
            c.set_path_patches(0)  # the scene's own part is capped alike: one family per sphere keeps the comparison about the eye
            c.set_scene(scene)
            swept[cap] = 0
            for t in (1.0, 2.5, 10.0):  # a new eye, new eye tables, the same capped part
                cam = T.bench_camera(64, 36, t)
                moved = scene.with_camera(cam)
                want, st = T.oracle_render(moved, 64, 36, 4, 2)
                got = c.render_host(cam, hip.RowSet.whole(64, 36), 4, 2)
                assert np.array_equal(bits(got), bits(want)), (cap, t)
                assert c.read_counters() == (st.path_rays, st.shadow_rays)
                swept[cap] += c.read_diagnostics()["swept_traces"]
                info, cells, pool = c.read_path_tables(cam)
                eye_cells = cells[:2 * 6 * info["eye_cells"] ** 2]
                lost = sum(1 for x in eye_cells if (int(x) >> 56) == 0xFF)
                if cap:
                    assert info["pool_used_eye"] > cap and lost > 1000, (info, lost)  # the counter went on counting; the lists say so
                    at = [int(x) & 0xFFFFFFFF for x in eye_cells if ((int(x) >> 56) & 0x80) and (int(x) >> 56) != 0xFF]
                    assert all(a < info["pool_capacity"] for a in at)  # what lists there are lie inside the pool
                else:
                    assert lost == 0, lost
        assert swept[16] > swept[0], swept
    finally:
        c.close()


def test_a_scene_that_changes_with_every_call_of_the_drop_in_entry(ctx):
    """project_scene is a pure function of *scene (TRT.c:966): a caller may move a sphere before EVERY call.  The drop-in layer then
    stops building the full tables (24 patches per sphere at 256 spheres: ~0.1 s) and builds the cheap ones (one family per sphere)
    per call; once the scene has been still for three calls it builds the full ones again.  20 calls with a sphere moved before
    each, then 5 without: every frame is the oracle's, the layer says when it treats the scene as moving, and a moving call is
    several times cheaper than a full build."""
    import time
    scene = S.synth_scene(256, T.sky("synth"), T.bench_camera(96, 54))
    lib = hip.lib()
    hip._check(lib.trt_shutdown())  # a fresh default context: earlier tests have handed the drop-in entries other scenes
    hip._check(lib.trt_set_scene_policy(2, 3))
    try:
        took, moving = [], []
        for i in range(25):
            if i < 20:
                sph = scene.spheres.copy()
                sph[i % 256, 1] += 0.01 * (i + 1)
                scene = scene.with_spheres(sph)
            t0 = time.perf_counter()
            got = hip.render_frame(scene, 96, 54, 6, 4)
            took.append(time.perf_counter() - t0)
            moving.append(lib.trt_scene_is_moving())
            if i in (0, 1, 2, 7, 19, 20, 22, 23, 24):
                want, _ = T.oracle_render(scene, 96, 54, 6, 4)
                assert np.array_equal(bits(got), bits(want)), i
        # call 0: the first scene (full tables); call 1: the second change in a row -> moving from here on; calls 20, 21 unchanged
        # but not yet still for three calls; call 22: still -> promoted (full build); 23, 24: nothing to build
        assert moving == [0] + [1] * 21 + [0] * 3, moving
        full, cheap, idle = took[22], float(np.median(took[3:20])), float(np.median(took[23:]))
        print(f"\nper call at 256 spheres, 96x54: full tables {full * 1e3:.1f} ms, moving scene {cheap * 1e3:.1f} ms, unchanged {idle * 1e3:.1f} ms")
        assert cheap < 0.5 * full and idle < cheap
        # the policy can be switched off: every change builds the full tables
        hip._check(lib.trt_set_scene_policy(0, 3))
        sph = scene.spheres.copy()
        for i in range(3):
            sph[5, 0] += 0.02
            scene = scene.with_spheres(sph.copy())
            got = hip.render_frame(scene, 96, 54, 6, 4)
            assert lib.trt_scene_is_moving() == 0
        want, _ = T.oracle_render(scene, 96, 54, 6, 4)
        assert np.array_equal(bits(got), bits(want))
    finally:
        hip._check(lib.trt_set_scene_policy(2, 3))


def test_the_automatic_patch_policy_steps_down_instead_of_failing(ctx):
    """24 patches per sphere are the default from 128 spheres up; with a finer direction grid than the default the tables
    they ask for do not fit (256 spheres at 128 cells per side: 1.2e9 cells).  The automatic policy then steps down (2 -> 1 -> 0
    cells per side of the origin's cube map) until cells and pool fit its budget (4 GB); an m asked for by number is taken as it is."""
    scene = S.synth_scene(256, T.sky("synth"), T.bench_camera(64, 36))
    want, _ = T.oracle_render(scene, 64, 36, 6, 4)
    c = hip.Context(0)
    try:
        c.set_path_grids(64, 128)
        c.set_scene(scene)
        assert c.path_patches() == (1, 6), c.path_patches()       # 3e8 cells instead of 1.2e9
        assert c.scene_info()["table_bytes"] < 5 << 30
        got = c.render_host(scene.camera, hip.RowSet.whole(64, 36), 6, 4)
        assert np.array_equal(bits(got), bits(want))
        c.set_path_grids(64, 32)
        assert c.path_patches() == (2, 24)                          # the default grid: the default patches
        c.set_path_grids(64, 48)
        c.set_path_patches(3)                                       # asked for by number: taken as it is (54 patches, 3.8e8 cells)
        assert c.path_patches() == (3, 54)
    finally:
        c.close()


def test_the_cube_instructions_equal_their_c_restatement(ctx):
    """Since round 4 the cube-map look-ups of the candidate tables are gfx9's v_cubeid / v_cubesc / v_cubetc / v_cubema on the device
    and a C restatement of the ISA manual's pseudo-code on the host (builders, conservativeness checkers: csrc/trt_lightgrid.h,
    trt_cube_lookup).  Face, both face coordinates and the doubled major component must agree bit for bit: random directions of all
    magnitudes, every kind of tie (|x| = |y|, |y| = |z|, all three, with every sign pattern), zeros of both signs, denormals, inf."""
    rng = np.random.default_rng(41)
    v = (rng.normal(size=(400000, 3)) * 10.0 ** rng.uniform(-30, 30, (400000, 1))).astype(np.float32)
    ties = []
    for a in (1.0, 0.3, 1e-20, 3e15, 0.0):
        for b in (a, a * 0.5, 0.0):
            for sx in (1, -1):
                for sy in (1, -1):
                    for sz in (1, -1):
                        ties += [[sx * a, sy * a, sz * b], [sx * a, sy * b, sz * a], [sx * b, sy * a, sz * a], [sx * a, sy * a, sz * a]]
    special = [[0.0, 0.0, 0.0], [-0.0, 0.0, -0.0], [1e-42, 2e-42, -1e-42], [np.inf, 1.0, 2.0], [1.0, -np.inf, np.inf], [3.0, 3.0, -3.0],
               [1.0, 1e-40, 0.5], [-1e-39, 2.0, 3e-41], [1e-40, -1e-41, -7.0], [2e-38, 1e-39, -1e-38], [-1e-40, 1e-38, 0.0]]  # denormals among normals
    v = np.concatenate([v, np.array(ties, dtype=np.float32), np.array(special, dtype=np.float32)])
    dev, host = ctx.selftest_cube(v)
    null = (np.abs(v) < 1.1754944e-38).all(axis=1)  # no direction (denormals count as zeros): the hardware goes by sign BITS; such a vector is
    assert np.isin(dev[null, 0], (4.0, 5.0)).all() and (dev[null, 3] == 0).all()    # never looked up (|o - l|^2 > 0 / unit direction)
    same = dev.view(np.uint32)[~null] == host.view(np.uint32)[~null]
    bad = (~same).any(axis=1)
    assert same.all(), (int(bad.sum()), v[~null][bad][:5], dev[~null][bad][:5], host[~null][bad][:5])


def test_a_trt_dist_is_given_another_scene(ctx):
    """trt_dist_set_scene: the first frame slot builds the new scene's tables, the other slots let go of the old ones and share the
    new (one copy per device before and after); frames of both scenes, rendered through all three slots, are the oracle's."""
    first = S.synth_scene(256, T.sky("synth"), T.bench_camera(96, 54))
    second = S.synth_scene(40, T.sky("synth"), T.bench_camera(96, 54, 2.5), seed=3)
    d = hip.Dist(0, first, None, 0, 1, 96, 54, tile_rows=8, frames_in_flight=3)
    try:
        for turn, scene in enumerate((first, second, first)):
            if turn:
                d.set_scene(scene)
            info = [d.context(i).scene_info() for i in range(3)]
            assert all(i["sharers"] == 3 and i["table_bytes"] == info[0]["table_bytes"] for i in info), info
            want = {t: T.oracle_render(scene.with_camera(T.bench_camera(96, 54, t)), 96, 54, 6, 4)[0] for t in (1.0, 2.5, 10.0)}
            frames = []
            for t in (1.0, 2.5, 10.0, 1.0, 2.5):  # five frames through three slots: every slot, re-used
                frames.append((t, d.render(T.bench_camera(96, 54, t), 6, 4)))
                if len(frames) >= 3:
                    t0, f0 = frames.pop(0)
                    assert np.array_equal(bits(d.fetch(f0)), bits(want[t0])), (len(scene.spheres), t0)
            for t0, f0 in frames[-1:]:
                assert np.array_equal(bits(d.fetch(f0)), bits(want[t0]))
    finally:
        d.close()


@pytest.mark.gpu
def test_the_sky_estimate_vouches_only_for_the_reference_index(ctx):
    """get_skybox_color (TRT.c:700-789): the kernel takes the texel's index from an FP32 estimate where that provably truncates as
    the reference's FP64 value does (csrc/trt_device.hpp: sky_index_estimate).  Wherever the estimate is not ambiguous it must
    equal the FP64 form: random directions, directions ON texel edges, on face edges and corners, axis directions, and cubemaps
    from 1 to 2^21 texels a side (from 2^20 on everything must be ambiguous)."""
    rng = np.random.default_rng(77)
    for dim in (1, 2, 3, 64, 256, 1000, 2048, 8192, 1 << 20, 1 << 21):
        v = rng.normal(size=(400_000, 3))
        # points of the cube's surface that sit exactly on texel edges (and a few ulps off them), on every face
        face = rng.integers(0, 6, size=200_000)
        a = (rng.integers(0, dim + 1, size=200_000) / dim) * 2.0 - 1.0
        b = rng.uniform(-1, 1, size=200_000)
        a = a * (1.0 + rng.choice([0.0, 0.0, 1e-16, -1e-16, 1e-9, -1e-9, 1e-7, -1e-7, 3e-6, -3e-6], size=200_000))
        swap = rng.integers(0, 2, size=200_000).astype(bool)
        pa, pb = np.where(swap, b, a), np.where(swap, a, b)
        major = np.where(face & 1, -1.0, 1.0)
        cube = np.zeros((200_000, 3))
        ax = face >> 1
        for k in range(3):
            m = ax == k
            cube[m, k] = major[m]
            cube[m, (k + 1) % 3] = pa[m]
            cube[m, (k + 2) % 3] = pb[m]
        # edges and corners of the cube: two or three components of equal magnitude
        edge = rng.choice([-1.0, 1.0], size=(50_000, 3)) * np.where(rng.random((50_000, 3)) < 0.6, 1.0, rng.random((50_000, 3)))
        axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 1, 1], [-1, -1, -1], [1, -1, 0], [0, 1, -1]], dtype=np.float64)
        dirs = np.concatenate([v, cube, edge, axes])
        dirs = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
        # directions that were never normalised (TRT.c:444 leaves vectors shorter than 1e-4 alone) down among FP32's denormals,
        # some with one component a few thousand times smaller than the others: the cube instructions flush what the FP64 form keeps
        tiny = v[:60_000] * 10.0 ** rng.uniform(-46, -28, size=(60_000, 1))
        tiny[::3, rng.integers(0, 3)] *= 10.0 ** rng.uniform(-4, -1)
        dirs = np.concatenate([dirs, tiny])
        exact, est, amb = ctx.selftest_sky(dirs, dim)
        sure = amb == 0
        assert np.array_equal(exact[sure], est[sure]), (dim, int((exact[sure] != est[sure]).sum()))
        assert exact.min() >= 0 and exact.max() < 6 * dim * dim
        if dim >= 1 << 20:
            assert not sure.any()
        elif dim <= 2048:
            assert sure[: len(v)].mean() > 0.98, (dim, sure[: len(v)].mean())  # the estimate is what usually runs


@pytest.mark.gpu
def test_rays_beyond_the_tables_range(ctx):
    """A camera high above a small scene sees ground out to the horizon: the shadow rays and the reflections that start there lie
    beyond the tables' range (256 x the scene's reach) and their waves sweep the culling table instead -- a few of them per wave
    in the view towards the horizon, all of them in the view from 1e7 up.  The frames are the oracle's either way, and the sweeps
    must have happened."""
    for n, spp in ((6, 3), (64, 1)):
        base = (S.synth_scene(n, T.sky("synth"), T.bench_camera(192, 108)) if n > 6 else
                T.golden_scene(next(c for c in T.golden_cases() if c["name"] == "demo_160x48_b4")))
        ground = np.array(base.ground, dtype=np.float64).copy()
        ground[9] = 0.6  # a reflecting ground: path rays start out there too
        for height, tilt in ((40.0, -0.06), (1e7, -0.7)):
            cam = np.array(base.camera, dtype=np.float64).copy()
            cam[9:12] = [0.0, height, 0.0]
            fwd = np.array([0.0, tilt, -1.0]) / np.linalg.norm([0.0, tilt, -1.0])
            right = np.cross(fwd, [0.0, 1.0, 0.0]); right /= np.linalg.norm(right)
            up = np.cross(right, fwd)
            cam[0:3], cam[3:6], cam[6:9] = right, up, -fwd
            scene = S.SceneData(base.spheres, ground, base.dir_lights, base.point_lights, cam, base.sky)
            ctx.set_scene(scene)
            ctx.enable_counters(True)
            try:
                got = render(ctx, scene, 192, 108, 4, spp, hip.Context.PRODUCTION)
                ctx.read_counters()
                diag = ctx.read_diagnostics()
            finally:
                ctx.enable_counters(False)
            want = T.oracle_render(scene, 192, 108, 4, spp)[0]
            assert np.array_equal(bits(got), bits(want)), (n, height)
            assert diag["swept_traces"] > 0 or height < 1e6, (n, height, diag)
