"""Shared test plumbing: the oracle binding (checker only), golden loaders, hashes."""
import ctypes as C
import functools
import json
import os
import subprocess
import zlib

import numpy as np

from terminalraytracer_amd import layout as L
from terminalraytracer_amd import scenes as S

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")
FACES = ["+X", "-X", "+Y", "-Y", "+Z", "-Z"]


# TRT_TEST_SANITIZE=1 (set by tests/test_sanitized.py for a child run): the host builds of the table headers (filter / light-table /
# ray-table checkers) are compiled with the address and undefined-behaviour sanitizers, under another name
SANITIZE = os.environ.get("TRT_TEST_SANITIZE") == "1"
CHECKER_FLAGS = ["-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"] if SANITIZE else []


def checker_so(name):
    build = os.path.join(ROOT, "tests", "_build")
    os.makedirs(build, exist_ok=True)
    return os.path.join(build, name + ("_san" if SANITIZE else "") + ".so")


class OracleStats(C.Structure):
    _fields_ = [("path_rays", C.c_ulonglong), ("shadow_rays", C.c_ulonglong), ("sky_lookups", C.c_ulonglong),
                ("samples", C.c_ulonglong)]


@functools.lru_cache(maxsize=None)
def oracle():
    """oracle/libtrt_oracle.so -- the CPU restatement.  Checker only; never on a product path."""
    path = os.environ.get("TRT_ORACLE_LIB") or os.path.join(ROOT, "oracle", "libtrt_oracle.so") # the variable: a sanitized build (test_oracle_golden.py)
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    lib = C.CDLL(path)
    lib.trt_oracle_project_scene.argtypes = [C.POINTER(L.Scene), C.POINTER(L.Screen), C.c_int, C.c_int, C.c_int,
                                             C.POINTER(OracleStats)]
    lib.trt_oracle_project_scene.restype = None
    lib.trt_oracle_render_rows.argtypes = [C.POINTER(L.Scene), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.POINTER(OracleStats)]
    lib.trt_oracle_render_rows.restype = None
    lib.trt_oracle_trace_ray.argtypes = [C.POINTER(L.Scene), C.POINTER(L.Ray), C.POINTER(L.Vector), C.POINTER(L.Vector),
                                         C.POINTER(L.Material)]
    lib.trt_oracle_trace_ray.restype = C.c_int
    lib.trt_oracle_skybox_lookup.argtypes = [C.POINTER(L.Scene), C.POINTER(L.Vector), C.POINTER(C.c_int),
                                             C.POINTER(C.c_long)]
    lib.trt_oracle_skybox_lookup.restype = C.c_int
    lib.trt_oracle_apply_lighting.argtypes = [C.POINTER(L.Scene), C.POINTER(L.Vector), C.POINTER(L.Vector),
                                              C.POINTER(L.Material), C.POINTER(OracleStats)]
    lib.trt_oracle_apply_lighting.restype = None
    lib.trt_oracle_triangle_wave.argtypes = [C.c_double]
    lib.trt_oracle_triangle_wave.restype = C.c_double
    lib.trt_oracle_rgb8.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.trt_oracle_rgb8.restype = None
    lib.trt_oracle_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
    lib.trt_oracle_fnv1a64.restype = C.c_ulonglong
    lib.trt_oracle_div_sqrt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.trt_oracle_div_sqrt.restype = None
    lib.trt_oracle_project_scene_refractive.argtypes = [C.POINTER(L.Scene), C.c_void_p, C.POINTER(L.Screen), C.c_int, C.c_int, C.c_int,
                                                        C.POINTER(OracleStats)]
    lib.trt_oracle_project_scene_refractive.restype = None
    return lib


def fnv(buf):
    a = np.ascontiguousarray(buf)
    return f"{oracle().trt_oracle_fnv1a64(a.ctypes.data, a.nbytes):016x}"


def oracle_render(scene_data, width, height, bounce_limit, rays_per_pixel, threads=None, rows=None):
    """(pixels[H,W,3] float64, stats) from the CPU restatement."""
    threads = threads or min(8, os.cpu_count() or 1)
    scene = scene_data.as_scene()
    st = OracleStats()
    if rows is None:
        screen, px = S.new_screen(width, height)
        oracle().trt_oracle_project_scene(C.byref(scene), C.byref(screen), bounce_limit, rays_per_pixel, threads, C.byref(st))
        return px, st
    r0, r1 = rows
    px = np.zeros((r1 - r0, width, 3), dtype=np.float64)
    oracle().trt_oracle_render_rows(C.byref(scene), px.ctypes.data, width, height, r0, r1, bounce_limit, rays_per_pixel,
                                    threads, C.byref(st))
    return px, st


def oracle_render_refractive(scene_data, ior, width, height, bounce_limit, rays_per_pixel, threads=None):
    """EXTENSION, parity unpinned: the oracle's restatement of the refraction variant (not the reference, which has none)."""
    threads = threads or min(8, os.cpu_count() or 1)
    scene = scene_data.as_scene()
    ior = np.ascontiguousarray(ior, dtype=np.float64)
    assert ior.size == scene_data.num_spheres
    st = OracleStats()
    screen, px = S.new_screen(width, height)
    oracle().trt_oracle_project_scene_refractive(C.byref(scene), ior.ctypes.data, C.byref(screen), bounce_limit, rays_per_pixel, threads, C.byref(st))
    return px, st


def oracle_rgb8(pixels):
    px = np.ascontiguousarray(pixels, dtype=np.float64)
    out = np.empty(px.shape, dtype=np.uint8)
    oracle().trt_oracle_rgb8(px.ctypes.data, px.size // 3, out.ctypes.data)
    return out


@functools.lru_cache(maxsize=None)
def golden_meta():
    with open(os.path.join(GOLDEN, "golden.json")) as fh:
        return json.load(fh)


@functools.lru_cache(maxsize=None)
def _frames():
    return dict(np.load(os.path.join(GOLDEN, "frames.npz")))


def read_ppm_bytes(raw):
    """Minimal P6 decode for test fixtures (the product loader is the C one in csrc/host)."""
    assert raw[:2] == b"P6"
    pos = 3
    while raw[pos:pos + 1] == b"#":
        pos = raw.index(b"\n", pos) + 1
    toks = []
    while len(toks) < 3:
        end = pos
        while raw[end:end + 1] not in (b" ", b"\n", b"\t", b"\r"):
            end += 1
        toks.append(int(raw[pos:end]))
        pos = end + 1
    w, h, mx = toks
    assert mx == 255
    return np.frombuffer(raw, dtype=np.uint8, count=w * h * 3, offset=pos).reshape(h, w, 3).copy()


def golden_ppm_raw(name, face):
    with open(os.path.join(GOLDEN, "skybox", name, face + ".ppm.z"), "rb") as fh:
        return zlib.decompress(fh.read())


@functools.lru_cache(maxsize=None)
def sky(name):
    if name == "synth":
        return np.load(os.path.join(GOLDEN, "synth_sky64.npz"))["sky"]
    return np.stack([read_ppm_bytes(golden_ppm_raw(name, f)) for f in FACES])


def golden_cases(size_classes=("small",)):
    return [c for c in golden_meta()["cases"] if c["size_class"] in size_classes]


def golden_scene(case):
    return S.SceneData.from_arrays(_frames(), sky(case["sky"]), prefix=case["name"] + "/")


def golden_fb(case):
    return _frames()[case["name"] + "/fb"] if case["has_fb"] else None


def bench_camera(width, height, t=1.0):
    """Stored reference camera at orbit time t (tests/golden/cameras.npz) with screen_width = 5*W/H."""
    d = np.load(os.path.join(GOLDEN, "cameras.npz"))
    i = int(np.argmin(np.abs(d["t"] - t)))
    assert abs(d["t"][i] - t) < 1e-12
    cam = d["camera"][i].copy()
    cam[13] = 5 * float(width) / float(height)
    return cam


@functools.lru_cache(maxsize=None)
def golden_full():
    """Full-size frames hashed by the genuine reference (tests/golden/make_golden_full.py): BASELINE configs 2-5 at the
    sizes the bench and the GPU tests render them, and frames with 1024^2 / 2048^2 cubemaps."""
    with open(os.path.join(GOLDEN, "golden_full.json")) as fh:
        return {c["name"]: c for c in json.load(fh)["cases"]}


def full_scene(case):
    """The scene of a golden_full case: SYNTH-v0 spheres, procedural cubemap, the reference's stored camera."""
    return S.synth_scene(case["spheres"], S.synth_sky(case["sky_dim"], seed=case["sky_seed"]),
                         np.array(case["camera"], dtype=np.float64), seed=case["scene_seed"])
