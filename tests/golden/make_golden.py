#!/usr/bin/env python3
"""Generate tests/golden/* from the GENUINE reference (oracle/_ref/libtrtref_*.so, built by
oracle/Makefile from /root/reference/TerminalRayTracer.c where it lies).

Runs only in the build container (the reference does not exist on the GPU box); what it writes
is data: scene/camera inputs, expected framebuffers / hashes / call-independent outputs, and the
reference's skybox image files zlib-compressed (assets, not source).  Re-run after `make -C oracle`:

    python tests/golden/make_golden.py

Every frame is rendered by the reference's own project_scene (TerminalRayTracer.c:966); the
known-answer table of SURVEY.md section 8c is re-checked on the way and the script aborts on a
mismatch, which proves this harness equals the survey's.
"""
import ctypes as C
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from terminalraytracer_amd import layout as L  # noqa: E402
from terminalraytracer_amd import scenes as S  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
REFROOT = "/root/reference"


def fnv1a64(buf):
    data = np.frombuffer(memoryview(buf).cast("B"), dtype=np.uint8)
    h = 1469598103934665603
    # chunked pure-python would be slow for 50 MB; use the oracle's C helper
    return _oracle().trt_oracle_fnv1a64(data.ctypes.data, data.size)


_ORACLE = None


def _oracle():
    global _ORACLE
    if _ORACLE is None:
        lib = C.CDLL(os.path.join(ROOT, "oracle", "libtrt_oracle.so"))
        lib.trt_oracle_fnv1a64.restype = C.c_ulonglong
        lib.trt_oracle_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        lib.trt_oracle_rgb8.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        _ORACLE = lib
    return _ORACLE


class RefLib:
    """One compiled variant of the reference (bounce limit / rays per pixel / emitter size are macros there)."""
    _cache = {}

    def __init__(self, b, s, w=480, h=280):
        path = os.path.join(REFDIR, f"libtrtref_b{b}_s{s}_w{w}_h{h}.so")
        self.lib = C.CDLL(path)
        self.b, self.s, self.w, self.h = b, s, w, h
        lib = self.lib
        lib.project_scene.argtypes = [C.POINTER(L.Scene), C.POINTER(L.Screen)]
        lib.project_scene.restype = None
        lib.trace_ray.argtypes = [C.POINTER(L.Scene), C.POINTER(L.Ray), C.POINTER(L.Vector), C.POINTER(L.Vector),
                                  C.POINTER(L.Material)]
        lib.trace_ray.restype = C.c_int
        lib.get_skybox_color.argtypes = [C.POINTER(L.Scene), C.POINTER(L.Vector), C.POINTER(L.Color)]
        lib.get_skybox_color.restype = None
        lib.apply_lighting.argtypes = [C.POINTER(L.Scene), C.POINTER(L.Vector), C.POINTER(L.Vector), C.POINTER(L.Vector),
                                       C.POINTER(L.Material)]
        lib.apply_lighting.restype = None
        lib.init_camera.argtypes = [C.POINTER(L.Camera)]
        lib.init_frame.argtypes = [C.POINTER(L.Frame)]
        lib.rotate_basis_x.argtypes = [C.POINTER(L.Basis), C.c_double]
        lib.rotate_basis_y.argtypes = [C.POINTER(L.Basis), C.c_double]
        lib.transform_frame.argtypes = [C.POINTER(L.Frame), C.POINTER(L.Frame)]
        lib.triangle_wave.argtypes = [C.c_double]
        lib.triangle_wave.restype = C.c_double
        lib.read_ppm.argtypes = [C.c_char_p, C.POINTER(C.POINTER(L.Color)), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.initialize_screenbuffer.restype = None
        lib.buffered_draw_screen.argtypes = [C.POINTER(L.Screen)]

    @classmethod
    def get(cls, b, s, w=480, h=280):
        key = (b, s, w, h)
        if key not in cls._cache:
            cls._cache[key] = cls(*key)
        return cls._cache[key]

    def orbit_camera(self, t):
        """The reference's own camera construction, TerminalRayTracer.c:1292-1293 and :1327-1336."""
        cam = L.Camera()
        self.lib.init_camera(C.byref(cam))
        tf0, tf1 = L.Frame(), L.Frame()
        self.lib.init_frame(C.byref(tf0))
        self.lib.init_frame(C.byref(tf1))
        self.lib.init_frame(C.byref(cam.frame))
        self.lib.rotate_basis_x(C.byref(tf0.basis), 2.0 * S.PI * t * -0.03)
        self.lib.rotate_basis_y(C.byref(tf0.basis), 2.0 * S.PI * t * 0.05)
        tf1.origin.z = tf1.origin.z + 1.99
        self.lib.transform_frame(C.byref(cam.frame), C.byref(tf1))
        self.lib.transform_frame(C.byref(cam.frame), C.byref(tf0))
        return np.frombuffer(bytes(cam), dtype=np.float64).copy()

    def render(self, scene_data, width, height):
        scene = scene_data.as_scene()
        screen, pixels = S.new_screen(width, height)
        self.lib.project_scene(C.byref(scene), C.byref(screen))
        return pixels

    def read_ppm(self, path):
        colors = C.POINTER(L.Color)()
        w, h = C.c_int(), C.c_int()
        self.lib.read_ppm(path.encode(), C.byref(colors), C.byref(w), C.byref(h))
        n = w.value * h.value * 3
        arr = np.frombuffer((C.c_ubyte * n).from_address(C.addressof(colors.contents)), dtype=np.uint8).copy()
        return arr.reshape(h.value, w.value, 3)


FACES = ["+X", "-X", "+Y", "-Y", "+Z", "-Z"]  # TerminalRayTracer.c:390


def load_ref_sky(ref, name):
    return np.stack([ref.read_ppm(f"{REFROOT}/skybox/{name}/{f}.ppm") for f in FACES])


def rgb8(pixels):
    out = np.empty(pixels.shape, dtype=np.uint8)
    _oracle().trt_oracle_rgb8(pixels.ctypes.data, pixels.shape[0] * pixels.shape[1], out.ctypes.data)
    return out


def main():
    os.makedirs(os.path.join(HERE, "skybox"), exist_ok=True)
    ref10 = RefLib.get(10, 10)
    meta = {"cases": [], "note": "generated by tests/golden/make_golden.py from oracle/_ref (genuine reference)"}

    # ---- skybox assets: reference image files, zlib-compressed; decoded faces hashed --------------
    skies = {}
    sky_meta = {}
    for name in ("colors", "uv_checker"):
        sky = load_ref_sky(ref10, name)
        skies[name] = sky
        sky_meta[name] = {"dim": int(sky.shape[1]), "decoded_fnv": f"{fnv1a64(sky):016x}", "files": {}}
        for f in FACES:
            raw = open(f"{REFROOT}/skybox/{name}/{f}.ppm", "rb").read()
            dst = os.path.join(HERE, "skybox", name)
            os.makedirs(dst, exist_ok=True)
            with open(os.path.join(dst, f + ".ppm.z"), "wb") as fh:
                fh.write(zlib.compress(raw, 9))
            sky_meta[name]["files"][f] = {"bytes": len(raw), "fnv": f"{fnv1a64(raw):016x}"}
    skies["synth"] = S.synth_sky(64, seed=7)  # small, procedural, texel-level sensitivity
    meta["skybox"] = sky_meta

    # ---- cameras: the reference's own orbit at fixed t --------------------------------------------
    cam_t = [0.0, 0.5, 1.0, 2.5, 10.0, 33.3, 100.0 / 60.0]
    cams = np.stack([ref10.orbit_camera(t) for t in cam_t])
    np.savez(os.path.join(HERE, "cameras.npz"), t=np.array(cam_t), camera=cams)
    cam1 = ref10.orbit_camera(1.0)  # init_camera aspect: 5*480/280 x 5

    def bench_cam(w, h, t=1.0):
        c = ref10.orbit_camera(t)
        c[13] = 5 * float(w) / float(h)  # screen_width = 5*W/H
        return c

    # ---- frames --------------------------------------------------------------------------------
    arrays = {}
    known = {  # SURVEY.md 8c known-answer table: (fb hash, rgb8 hash)
        "demo_480x280_b10": ("453219f388ade6f2", "c915d01209577150"),
        "demo_160x48_b4": ("934969b6b4794876", "0e1baf81abb2bab5"),
        "demo_160x48_b4_s1": ("64b1239a098529bb", "6306ca17dbd48b70"),
        "demo_160x48_b10": ("fd5f93871b153060", "496c227a505ee65a"),
        "demo_1080p_b4": ("b46559293dcc89dd", "274af1814ee7cac9"),
        "demo_1080p_b8_uv": ("669a6a53737f9f09", "656c354182db925b"),
        "synth64_480x270_b8": ("ba85cef6ea36886a", None),
    }

    def case(name, scene, w, h, b, s, sky_name, keep_fb=False, size_class="small"):
        ref = RefLib.get(b, s)
        px = ref.render(scene, w, h)
        fb_h = f"{fnv1a64(px):016x}"
        rgb = rgb8(px)
        rgb_h = f"{fnv1a64(rgb):016x}"
        if name in known:
            want_fb, want_rgb = known[name]
            assert fb_h == want_fb, (name, fb_h, want_fb)
            assert want_rgb is None or rgb_h == want_rgb, (name, rgb_h, want_rgb)
            print(f"  known-answer OK  {name}: fb {fb_h} rgb8 {rgb_h}")
        for k, v in scene.to_arrays(name + "/").items():
            arrays[k] = v
        if keep_fb:
            arrays[name + "/fb"] = px
        meta["cases"].append({"name": name, "width": w, "height": h, "bounce_limit": b, "rays_per_pixel": s,
                              "sky": sky_name, "fb_fnv": fb_h, "rgb8_fnv": rgb_h, "has_fb": bool(keep_fb),
                              "size_class": size_class, "known_answer": name in known})
        print(f"case {name}: {w}x{h} B{b} S{s} sky={sky_name} fb={fb_h}")

    demo = lambda sky, cam, n=6: S.demo_scene(skies[sky], cam, n)  # noqa: E731
    synth = lambda n, sky, cam, **kw: S.synth_scene(n, skies[sky], cam, **kw)  # noqa: E731

    # SURVEY known answers
    case("demo_480x280_b10", demo("colors", cam1), 480, 280, 10, 10, "colors", size_class="medium")
    case("demo_160x48_b4", demo("colors", cam1), 160, 48, 4, 10, "colors", keep_fb=True)
    case("demo_160x48_b4_s1", demo("colors", cam1), 160, 48, 4, 1, "colors")
    case("demo_160x48_b10", demo("colors", cam1), 160, 48, 10, 10, "colors")
    case("demo_1080p_b4", demo("colors", cam1), 1920, 1080, 4, 10, "colors", size_class="large")
    case("demo_1080p_b8_uv", demo("uv_checker", cam1), 1920, 1080, 8, 10, "uv_checker", size_class="large")
    case("synth64_480x270_b8", synth(64, "colors", bench_cam(480, 270)), 480, 270, 8, 10, "colors", size_class="medium")

    # BASELINE config 1 wording: 160x48, spheres[0..2], 4 bounces
    case("demo3_160x48_b4", demo("uv_checker", cam1, 3), 160, 48, 4, 10, "uv_checker", keep_fb=True)
    # texel-sensitive procedural sky, other orbit times
    case("demo_96x54_b10_synthsky_t0", demo("synth", bench_cam(96, 54, 0.0)), 96, 54, 10, 10, "synth", keep_fb=True)
    case("demo_96x54_b4_synthsky_t33", demo("synth", bench_cam(96, 54, 33.3)), 96, 54, 4, 10, "synth")
    # synthetic sphere counts of the BASELINE configs at small resolution
    case("synth8_128x72_b4", synth(8, "synth", bench_cam(128, 72)), 128, 72, 4, 10, "synth")
    case("synth64_128x72_b8", synth(64, "synth", bench_cam(128, 72)), 128, 72, 8, 10, "synth")
    case("synth65_64x36_b8", synth(65, "synth", bench_cam(64, 36, 2.5)), 64, 36, 8, 10, "synth")
    case("synth256_64x36_b12", synth(256, "synth", bench_cam(64, 36)), 64, 36, 12, 10, "synth")
    case("synth64_mirror_96x54_b8", synth(64, "synth", bench_cam(96, 54, 10.0), mirror_fraction=0.25), 96, 54, 8, 10,
         "synth")
    # edge cases: no spheres, no lights, several lights, non-default rays-per-pixel, ragged sizes
    case("nospheres_64x36_b4", demo("synth", bench_cam(64, 36), 0), 64, 36, 4, 10, "synth")
    nolight = demo("synth", bench_cam(64, 36))
    nolight = S.SceneData(nolight.spheres, nolight.ground, np.zeros((0, 6)), np.zeros((0, 7)), nolight.camera, nolight.sky)
    case("nolights_64x36_b4", nolight, 64, 36, 4, 10, "synth")
    many = demo("synth", bench_cam(64, 36))
    many = S.SceneData(many.spheres, many.ground,
                       np.array([[-1.0, -1.0, -1.0, 0.6, 0.5, 0.4], [0.5, -1.0, 0.25, 0.2, 0.3, 0.5]]),
                       np.array([[0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 10.0], [2.0, 3.0, -1.5, 0.9, 0.7, 0.3, 25.0],
                                 [-3.0, 0.5, 2.0, 0.2, 0.8, 0.9, 4.0]]), many.camera, many.sky)
    case("manylights_64x36_b4", many, 64, 36, 4, 10, "synth")
    case("demo_67x13_b2_s3", demo("synth", bench_cam(67, 13)), 67, 13, 2, 3, "synth", keep_fb=True)
    case("demo_1x1_b4", demo("synth", bench_cam(1, 1)), 1, 1, 4, 10, "synth", keep_fb=True)
    case("demo_1x9_b4", demo("synth", bench_cam(1, 9)), 1, 9, 4, 10, "synth")
    # camera inside a sphere (near root only: the enclosing sphere is invisible from inside)
    inside = bench_cam(64, 36)
    inside[9:12] = [1.0, 0.1, 0.05]
    case("inside_sphere_64x36_b4", demo("synth", inside), 64, 36, 4, 10, "synth")
    # tilted, offset ground plane and a non-unit ground normal
    tilt = demo("synth", bench_cam(64, 36))
    g = tilt.ground.copy()
    g[0:6] = [0.3, -1.25, 0.2, 0.1, 2.0, -0.2]
    tilt = S.SceneData(tilt.spheres, g, tilt.dir_lights, tilt.point_lights, tilt.camera, tilt.sky)
    case("tilted_ground_64x36_b4", tilt, 64, 36, 4, 10, "synth")

    np.savez_compressed(os.path.join(HERE, "frames.npz"), **arrays)
    np.savez_compressed(os.path.join(HERE, "synth_sky64.npz"), sky=skies["synth"])

    # ---- single-ray vectors from the reference's trace_ray, lighting and skybox ---------------------
    rng = np.random.default_rng(20261004)
    sc = synth(64, "uv_checker", bench_cam(128, 72))
    scene = sc.as_scene()
    n = 600
    org = rng.uniform(-5, 5, (n, 3))
    org[:, 1] = rng.uniform(-1.9, 4, n)
    dirs = rng.normal(size=(n, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs[::7] *= rng.uniform(0.5, 2.0, (len(dirs[::7]), 1))  # non-unit directions (a != 1)
    # aim a third of the rays at sphere centres so that hits are well represented
    tgt = sc.spheres[rng.integers(0, 64, n // 3), :3] + rng.normal(scale=0.2, size=(n // 3, 3))
    dirs[: n // 3] = tgt - org[: n // 3]
    rays = np.ascontiguousarray(np.concatenate([org, dirs], axis=1))
    out_obj = np.zeros(n, dtype=np.int32)
    out_pt = np.zeros((n, 3))
    out_n = np.zeros((n, 3))
    out_mat = np.zeros((n, 5))
    out_lit = np.zeros((n, 3))
    for i in range(n):
        ray = L.Ray.from_buffer(rays[i])
        pt, nr, mt = L.Vector(), L.Vector(), L.Material()
        out_obj[i] = ref10.lib.trace_ray(C.byref(scene), C.byref(ray), C.byref(pt), C.byref(nr), C.byref(mt))
        out_pt[i] = [pt.x, pt.y, pt.z]
        out_n[i] = [nr.x, nr.y, nr.z]
        out_mat[i] = [mt.color.x, mt.color.y, mt.color.z, mt.reflectivity, mt.specularity]
        if out_obj[i] != L.NONE:
            view = L.Vector(-rays[i, 3], -rays[i, 4], -rays[i, 5])
            ref10.lib.apply_lighting(C.byref(scene), C.byref(pt), C.byref(view), C.byref(nr), C.byref(mt))
            out_lit[i] = [mt.color.x, mt.color.y, mt.color.z]
    sky_dirs = rng.normal(size=(3000, 3))
    sky_dirs[:6] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=float)
    sky_dirs[6] = [1.0, 1.0, 1.0]  # every unblocked directional-light shadow ray: u == 0.5 (index == dim, in bounds)
    sky_dirs[7:200] *= rng.uniform(1e-3, 1e3, (193, 1))
    sky_dirs = np.ascontiguousarray(sky_dirs)
    sky_rgb = np.zeros((len(sky_dirs), 3), dtype=np.uint8)
    for i in range(len(sky_dirs)):
        d = L.Vector.from_buffer(sky_dirs[i])
        col = L.Color()
        ref10.lib.get_skybox_color(C.byref(scene), C.byref(d), C.byref(col))
        sky_rgb[i] = [col.r, col.g, col.b]
    tw_t = np.concatenate([np.linspace(-1.0, 20.0, 211), 2 * S.PI * np.arange(10) / 10, S.PI * np.arange(10) / 10])
    tw = np.array([ref10.lib.triangle_wave(float(t)) for t in tw_t])
    np.savez_compressed(os.path.join(HERE, "rays.npz"), rays=rays, obj=out_obj, point=out_pt, normal=out_n,
                        material=out_mat, lit=out_lit, sky_dirs=sky_dirs, sky_rgb=sky_rgb, tw_t=tw_t, tw=tw,
                        **sc.to_arrays("scene/"))
    meta["rays"] = {"scene_sky": "uv_checker", "count": n, "sky_dirs": len(sky_dirs)}

    # ---- emitter: the reference's screenbuffer after buffered_draw_screen (TerminalRayTracer.c:1142) -----
    emit = {}
    for (w, h, b, cname) in ((160, 48, 4, "demo_160x48_b4"), (480, 280, 10, "demo_480x280_b10")):
        ref = RefLib.get(b, 10, w, h)
        px = ref.render(demo("colors", cam1), w, h)
        ref.lib.initialize_screenbuffer()
        size = (7 + 1) + (25 * w + 1) * h + 1  # sizeof(screenbuffer), TerminalRayTracer.c:1104: (sizeof(reset_str)+1) + ...
        buf = (C.c_char * size).in_dll(ref.lib, "screenbuffer")
        # patch digits exactly as buffered_draw_screen does, without its fwrite to stdout:
        # call it with stdout redirected to /dev/null
        sys.stdout.flush()
        saved = os.dup(1)
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
        screen = L.Screen()
        screen.pixels = px.ctypes.data_as(C.POINTER(L.Vector))
        screen.width, screen.height = w, h
        ref.lib.buffered_draw_screen(C.byref(screen))
        C.CDLL(None).fflush(None)
        os.dup2(saved, 1)
        os.close(devnull)
        os.close(saved)
        raw = bytes(buf)
        emit[cname] = {"width": w, "height": h, "bytes": size, "fnv": f"{fnv1a64(raw):016x}"}
        if w == 160:
            with open(os.path.join(HERE, "emit_demo_160x48_b4.bin.z"), "wb") as fh:
                fh.write(zlib.compress(raw, 9))
    meta["emitter"] = emit

    with open(os.path.join(HERE, "golden.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    print("wrote", HERE)


if __name__ == "__main__":
    main()
