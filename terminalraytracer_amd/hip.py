"""ctypes binding of libtrt_hip.so (C-ABI of include/trt_hip.h).

This is the only compute path of the package: there is no CPU or PyTorch fallback.  If the
HIP library is missing the import of `lib()` raises; if no GPU is present every compute call
fails with TRT_ERR_HIP.  PyTorch is used by callers only for device memory / streams /
torch.distributed; raw device pointers and the stream handle cross this boundary as integers.
"""
import ctypes as C
import functools
import os

import numpy as np

from . import layout as L

_HERE = os.path.dirname(os.path.abspath(__file__))
# TRT_HIP_LIB selects another build of the same library (kernel-tuning A/B runs); default is the in-tree one
LIB_PATH = os.environ.get("TRT_HIP_LIB") or os.path.join(_HERE, "libtrt_hip.so")

TRT_OK = 0
ERRORS = {-1: "TRT_ERR_HIP", -2: "TRT_ERR_ARGUMENT", -3: "TRT_ERR_NO_SCENE", -4: "TRT_ERR_CAPACITY",
          -5: "TRT_ERR_NOT_INITIALISED"}


class TrtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"{ERRORS.get(code, code)}: {message}")
        self.code = code


class RowSet(C.Structure):
    """trt_rowset: interleaved row tiles owned by one renderer (include/trt_hip.h)."""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("tile_rows", C.c_int), ("tile_first", C.c_int),
                ("tile_step", C.c_int)]

    @staticmethod
    def whole(width, height):
        return RowSet(width, height, height, 0, 1)

    @staticmethod
    def shard(width, height, rank, world, tile_rows=8):
        return RowSet(width, height, tile_rows, rank, world)


# every symbol include/trt_hip.h declares: (restype, argtypes)
_VP, _I, _SZ = C.c_void_p, C.c_int, C.c_size_t
SYMBOLS = {
    "project_scene": (None, [C.POINTER(L.Scene), C.POINTER(L.Screen)]),
    "trt_render_frame": (_I, [C.POINTER(L.Scene), C.POINTER(L.Screen), _I, _I]),
    "render_frame": (_I, [C.POINTER(L.Scene), C.POINTER(L.Screen), _I, _I]),
    "trt_render_frame_rgb8": (_I, [C.POINTER(L.Scene), _I, _I, _I, _I, _VP]),
    "trt_init": (_I, [_I]),
    "trt_shutdown": (_I, []),
    "trt_upload_skybox": (_I, [C.POINTER(L.Skybox)]),
    "trt_invalidate_skybox": (_I, []),
    "trt_rowset_rows": (_I, [C.POINTER(RowSet)]),
    "trt_rowset_frame_row": (_I, [C.POINTER(RowSet), _I]),
    "trt_create": (_I, [_I, C.POINTER(_VP)]),
    "trt_destroy": (_I, [_VP]),
    "trt_set_stream": (_I, [_VP, _VP]),
    "trt_set_scene": (_I, [_VP, C.POINTER(L.Scene)]),
    "trt_render_device": (_I, [_VP, C.POINTER(L.Camera), C.POINTER(RowSet), _I, _I, _VP, _SZ]),
    "trt_quantize_device": (_I, [_VP, _VP, _SZ, _VP]),
    "trt_render_host": (_I, [_VP, C.POINTER(L.Camera), C.POINTER(RowSet), _I, _I, _VP]),
    "trt_render_host_rgb8": (_I, [_VP, C.POINTER(L.Camera), C.POINTER(RowSet), _I, _I, _VP]),
    "trt_synchronize": (_I, [_VP]),
    "trt_kernel_times": (_I, [_VP, C.POINTER(C.c_float), _I]),
    "trt_render_kernel_times": (_I, [_VP, C.POINTER(C.c_float), C.POINTER(C.c_float), _I]),
    "trt_enable_counters": (_I, [_VP, _I]),
    "trt_read_counters": (_I, [_VP, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "trt_read_diagnostics": (_I, [_VP, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "trt_set_kernel": (_I, [_VP, _I]),
    "trt_set_light_grids": (_I, [_VP, _I, _I]),
    "trt_set_light_slabs": (_I, [_VP, _I, _I]),
    "trt_share_scene": (_I, [_VP, _VP]),
    "trt_set_scene_policy": (_I, [_I, _I]),
    "trt_scene_is_moving": (_I, []),
    "trt_scene_info": (_I, [_VP, C.POINTER(C.c_ulonglong), C.POINTER(_I), C.POINTER(C.c_double)]),
    "trt_set_list_pool_words": (_I, [_VP, C.c_size_t]),
    "trt_set_path_grids": (_I, [_VP, _I, _I]),
    "trt_set_path_grids_min_spheres": (_I, [_VP, _I]),
    "trt_set_path_patches": (_I, [_VP, _I]),
    "trt_get_path_patches": (_I, [_VP, C.POINTER(_I), C.POINTER(_I)]),
    "trt_path_family_code": (_I, [_VP, _I, _I, _VP]),
    "trt_set_compaction": (_I, [_VP, _I]),
    "trt_render_variant": (_I, [_VP, C.POINTER(_I), C.POINTER(_I)]),
    "trt_read_path_tables": (C.c_long, [_VP, C.POINTER(L.Camera), _VP, C.c_size_t, _VP, C.c_size_t, C.POINTER(C.c_long)]),
    "trt_read_sweep_fallbacks": (_I, [_VP, C.POINTER(C.c_ulonglong)]),
    "trt_read_shading_passes": (_I, [_VP, C.POINTER(C.c_ulonglong)]),
    "trt_read_loop_diagnostics": (_I, [_VP, C.POINTER(C.c_ulonglong)]),
    "trt_set_refraction": (_I, [_VP, _VP, _I]),
    "trt_reserve_cus": (_I, [_VP, _I]),
    "trt_get_stream": (_I, [_VP, C.POINTER(C.c_void_p)]),
    "trt_selftest_cube": (_I, [_VP, _VP, C.c_size_t, _VP, _VP]),
    "trt_selftest_unit": (_I, [_VP, _VP, C.c_size_t, _VP, _VP]),
    "trt_selftest_sky": (_I, [_VP, _VP, C.c_size_t, _I, _VP, _VP, _VP]),
    "trt_read_light_grid": (C.c_long, [_VP, _I, _I, _VP, C.c_size_t]),
    "trt_kernel_info": (_I, [_VP] + [C.POINTER(_I)] * 5),
    "trt_selftest_div_sqrt": (_I, [_VP, _VP, _VP, _SZ, _VP, _VP]),
    "trt_probe_rays": (_I, [_VP, _VP, _SZ, _VP, _VP, _VP, _VP, _VP]),
    "trt_probe_rays_production": (_I, [_VP, C.POINTER(L.Camera), _VP, _VP, _SZ, _VP, _VP, _VP, _VP, _VP]),
    "trt_launch_count": (C.c_long, [_VP]),
    "trt_launch_span_ms": (_I, [_VP, C.c_long, _VP, C.c_long, C.POINTER(C.c_float)]),
    "trt_dist_unique_id": (_I, [_VP]),
    "trt_dist_frame_times": (_I, [_VP, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "trt_dist_allow_rccl_override": (_I, [_I]),
    "trt_dist_rccl_library": (C.c_char_p, []),
    "trt_dist_comm_ranks": (_I, [_VP]),
    "trt_dist_create": (_I, [_I, C.POINTER(L.Scene), _VP, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_VP)]),
    "trt_dist_set_scene": (_I, [_VP, C.POINTER(L.Scene)]),
    "trt_dist_render": (_I, [_VP, C.POINTER(L.Camera), _I, _I, C.POINTER(_VP)]),
    "trt_dist_synchronize": (_I, [_VP]),
    "trt_dist_enable_rgb8": (_I, [_VP]),
    "trt_dist_render_rgb8": (_I, [_VP, C.POINTER(L.Camera), _I, _I, C.POINTER(_VP)]),
    "trt_dist_fetch_rgb8": (_I, [_VP, _VP, _VP]),
    "trt_dist_fetch": (_I, [_VP, _VP, _VP]),
    "trt_dist_source_rows": (_I, [_I, _I, _I, _I, C.POINTER(_I)]),
    "trt_dist_info": (_I, [_VP] + [C.POINTER(_I)] * 5),
    "trt_dist_context": (_VP, [_VP, _I]),
    "trt_dist_destroy": (_I, [_VP]),
    "trt_dist_last_error": (C.c_char_p, []),
    "trt_last_error": (C.c_char_p, []),
    "trt_version": (C.c_char_p, []),
}


@functools.lru_cache(maxsize=None)
def lib():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make lib` (hipcc --offload-arch=gfx950). "
                          "There is no fallback path.")
    # Load order matters when PyTorch shares the process: torch bundles its own libamdhip64/libhsa-runtime64,
    # and a process must initialise only ONE HIP runtime.  Importing torch first (when it is installed) makes
    # libtrt_hip.so's libamdhip64.so dependency resolve to the copy torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    dll = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(dll, name)  # AttributeError here = the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    return dll


def _check(code):
    if code != TRT_OK:
        raise TrtError(code, lib().trt_last_error().decode())


def camera_struct(camera_array):
    cam = L.Camera()
    a = np.ascontiguousarray(camera_array, dtype=np.float64).reshape(15)
    C.memmove(C.byref(cam), a.ctypes.data, 120)
    return cam


class Context:
    """One renderer on one GPU (trt_context)."""

    PRODUCTION, REFERENCE_ORDER = 0, 1

    def __init__(self, device=0, _borrowed=None):
        self._owned = _borrowed is None
        if self._owned:
            self._h = _VP()
            _check(lib().trt_create(device, C.byref(self._h)))
        else:
            self._h = _VP(_borrowed)  # a context owned by someone else (trt_dist_context)
        self.device = device

    def close(self):
        if self._h and self._owned:
            lib().trt_destroy(self._h)
        self._h = _VP()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle):
        _check(lib().trt_set_stream(self._h, _VP(stream_handle or 0)))

    def set_scene(self, scene_data):
        scene = scene_data.as_scene()
        _check(lib().trt_set_scene(self._h, C.byref(scene)))

    def set_kernel(self, which):
        _check(lib().trt_set_kernel(self._h, which))

    def reserve_cus(self, reserved):
        """keep `reserved` compute units free of this context's kernels (trt_reserve_cus)"""
        _check(lib().trt_reserve_cus(self._h, reserved))

    def stream_ptr(self):
        """the HIP stream the context launches on, as an integer (trt_get_stream)"""
        p = C.c_void_p()
        _check(lib().trt_get_stream(self._h, C.byref(p)))
        return p.value or 0

    def set_light_grids(self, directional_cells, point_cells):
        """cells per side of the light-space candidate tables; 0, 0 = off (trt_set_light_grids)"""
        _check(lib().trt_set_light_grids(self._h, directional_cells, point_cells))

    def share_scene(self, source):
        """render `source`'s scene from its tables: one copy of every read-only table per device (trt_share_scene)"""
        _check(lib().trt_share_scene(self._h, source._h))

    def scene_info(self):
        b, n, t = C.c_ulonglong(), _I(), C.c_double()
        _check(lib().trt_scene_info(self._h, C.byref(b), C.byref(n), C.byref(t)))
        return {"table_bytes": b.value, "sharers": n.value, "build_seconds": t.value}

    def set_list_pool_words(self, words):
        """tests: cap the pool of long candidate lists (trt_set_list_pool_words)"""
        _check(lib().trt_set_list_pool_words(self._h, words))

    def set_light_slabs(self, directional_slabs, point_shells):
        """the light tables' depth coordinate: slabs along a directional light, shells about a point light (trt_set_light_slabs)"""
        _check(lib().trt_set_light_slabs(self._h, directional_slabs, point_shells))

    def set_path_grids(self, eye_cells, sphere_cells):
        """cells per cube-map face side of the path rays' family tables; 0, 0 = off (trt_set_path_grids)"""
        _check(lib().trt_set_path_grids(self._h, eye_cells, sphere_cells))

    def set_path_patches(self, m):
        """sub-families of the spheres: 6 m^2 patches per sphere; 0 = one family per sphere, -1 = by the number of spheres (trt_set_path_patches)"""
        _check(lib().trt_set_path_patches(self._h, m))

    def path_patches(self):
        """(m, patches per sphere) of the current scene's tables (trt_get_path_patches)"""
        m, p = _I(), _I()
        _check(lib().trt_get_path_patches(self._h, C.byref(m), C.byref(p)))
        return m.value, p.value

    def family_codes(self, families, rays, num_spheres):
        """The kernel's family codes (trt_path_family_code) for rays stored along chains: families[j] in the numbering 0 eye,
        1 mirror eye, 2 + s starts on sphere s, 2 + N + s reflected by the ground with a parent that started on sphere s --
        that parent is ray j - 1 of the chain, whose origin names the patch."""
        families = np.asarray(families, dtype=np.int32)
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = families.copy()
        n = num_spheres
        for j in np.nonzero(families >= 2 + n)[0]:
            s = int(families[j]) - 2 - n
            assert j > 0 and families[j - 1] == 2 + s, "a mirror ray's parent must precede it"
            parent = np.ascontiguousarray(rays[j - 1, :3])
            out[j] = lib().trt_path_family_code(self._h, 3, s, parent.ctypes.data)
        return out

    def set_path_grids_min_spheres(self, min_spheres):
        """scenes with fewer spheres keep the sweep for their path rays (trt_set_path_grids_min_spheres)"""
        _check(lib().trt_set_path_grids_min_spheres(self._h, min_spheres))

    def set_compaction(self, mode):
        """shading decoupled from the owning lane: -1 when it costs no occupancy (default), 0 never, 1 whenever it fits (trt_set_compaction)"""
        _check(lib().trt_set_compaction(self._h, mode))

    def render_variant(self):
        """{"decoupled": bool, "workgroup_threads": int} of the kernel the next frame of the current scene runs (trt_render_variant)"""
        d, t = _I(), _I()
        _check(lib().trt_render_variant(self._h, C.byref(d), C.byref(t)))
        return {"decoupled": bool(d.value), "workgroup_threads": t.value}

    def read_path_tables(self, camera_array):
        """(info dict, list cells uint64[], pool uint64[]) of the path rays' tables as built for this camera's eye"""
        cam = camera_struct(camera_array)
        info = (C.c_long * 8)()
        probe = np.zeros(1, dtype=np.uint64)
        lib().trt_read_path_tables(self._h, C.byref(cam), probe.ctypes.data, 0, probe.ctypes.data, 0, info)  # sizes only
        cells = np.zeros(max(1, info[4]), dtype=np.uint64)
        pool = np.zeros(max(1, info[7]), dtype=np.uint64)
        got = lib().trt_read_path_tables(self._h, C.byref(cam), cells.ctypes.data, cells.size, pool.ctypes.data, pool.size, info)
        if got < 0:
            _check(int(got))
        keys = ("enabled", "eye_cells", "sphere_cells", "spheres", "cells", "pool_used_scene", "pool_used_eye", "pool_capacity")
        return dict(zip(keys, [int(x) for x in info])), cells[:got], pool

    def set_refraction(self, ior=None):
        """EXTENSION, parity unpinned: per-sphere indices of refraction (0 = opaque); None turns it off (trt_set_refraction)"""
        if ior is None:
            _check(lib().trt_set_refraction(self._h, None, 0))
        else:
            a = np.ascontiguousarray(ior, dtype=np.float64)
            _check(lib().trt_set_refraction(self._h, a.ctypes.data, a.size))

    def read_sweep_fallbacks(self):
        v = C.c_ulonglong()
        _check(lib().trt_read_sweep_fallbacks(self._h, C.byref(v)))
        return v.value

    def selftest_unit(self, xyzw):
        """(fast, reference): unit(x,y,z) and sqrt(w) by the kernels' lean code and by the compiler's plain expansions"""
        v = np.ascontiguousarray(xyzw, dtype=np.float64).reshape(-1, 4)
        fast, ref = np.zeros_like(v), np.zeros_like(v)
        _check(lib().trt_selftest_unit(self._h, v.ctypes.data, v.shape[0], fast.ctypes.data, ref.ctypes.data))
        return fast, ref

    def selftest_cube(self, xyz):
        """(device, host): {face, sc, tc, 2 * major} of directions by v_cubeid / sc / tc / ma and by their C restatement"""
        v = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        dev, host = np.zeros((v.shape[0], 4), dtype=np.float32), np.zeros((v.shape[0], 4), dtype=np.float32)
        _check(lib().trt_selftest_cube(self._h, v.ctypes.data, v.shape[0], dev.ctypes.data, host.ctypes.data))
        return dev, host

    def selftest_sky(self, dirs, dim):
        """(exact, estimate, ambiguous): texel index of unit directions by the FP64 form of the skybox look-up, by the FP32 estimate,
        and where the estimate does not vouch for itself (trt_selftest_sky)"""
        v = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        exact, est = np.zeros(v.shape[0], dtype=np.int64), np.zeros(v.shape[0], dtype=np.int64)
        amb = np.zeros(v.shape[0], dtype=np.int32)
        _check(lib().trt_selftest_sky(self._h, v.ctypes.data, v.shape[0], int(dim), exact.ctypes.data, est.ctypes.data, amb.ctypes.data))
        return exact, est, amb

    def read_light_grid(self, point_light, index, words):
        """one light's device-built candidate table as uint64 words (trt_read_light_grid)"""
        out = np.zeros(words, dtype=np.uint64)
        got = lib().trt_read_light_grid(self._h, int(point_light), index, out.ctypes.data, words)
        if got < 0:
            _check(int(got))
        return out[:got]

    def render_device(self, camera_array, rows, bounce_limit, rays_per_pixel, device_ptr, capacity_bytes):
        cam = camera_struct(camera_array)
        _check(lib().trt_render_device(self._h, C.byref(cam), C.byref(rows), bounce_limit, rays_per_pixel,
                                       _VP(device_ptr), capacity_bytes))

    def quantize_device(self, device_ptr, num_pixels, rgb_ptr):
        _check(lib().trt_quantize_device(self._h, _VP(device_ptr), num_pixels, _VP(rgb_ptr)))

    def render_host(self, camera_array, rows, bounce_limit, rays_per_pixel):
        n = lib().trt_rowset_rows(C.byref(rows))
        out = np.zeros((n, rows.width, 3), dtype=np.float64)
        cam = camera_struct(camera_array)
        _check(lib().trt_render_host(self._h, C.byref(cam), C.byref(rows), bounce_limit, rays_per_pixel,
                                     out.ctypes.data))
        return out

    def render_host_rgb8(self, camera_array, rows, bounce_limit, rays_per_pixel):
        """the frame as the emitter's bytes, (int)(c*255) done on the device: uint8 [rows, width, 3] (trt_render_host_rgb8)"""
        n = lib().trt_rowset_rows(C.byref(rows))
        out = np.zeros((n, rows.width, 3), dtype=np.uint8)
        cam = camera_struct(camera_array)
        _check(lib().trt_render_host_rgb8(self._h, C.byref(cam), C.byref(rows), bounce_limit, rays_per_pixel, out.ctypes.data))
        return out

    def synchronize(self):
        _check(lib().trt_synchronize(self._h))

    def kernel_times(self, max_count=256):
        buf = (C.c_float * max_count)()
        n = lib().trt_kernel_times(self._h, buf, max_count)
        if n < 0:
            _check(n)
        return [buf[i] for i in range(n)]

    def render_kernel_times(self, max_count=256):
        """(render_ms[], reduce_ms[]) of the most recent launches: the render kernel alone and the ordered mean"""
        a, b = (C.c_float * max_count)(), (C.c_float * max_count)()
        n = lib().trt_render_kernel_times(self._h, a, b, max_count)
        if n < 0:
            _check(n)
        return [a[i] for i in range(n)], [b[i] for i in range(n)]

    def launch_count(self):
        """frames this context has launched so far (trt_launch_count)"""
        return int(lib().trt_launch_count(self._h))

    def launch_span_ms(self, first_launch, last, last_launch):
        """device ms from the start of this context's launch `first_launch` to the end of context `last`'s launch `last_launch` (trt_launch_span_ms)"""
        ms = C.c_float()
        _check(lib().trt_launch_span_ms(self._h, first_launch, last._h, last_launch, C.byref(ms)))
        return float(ms.value)

    def enable_counters(self, on=True):
        _check(lib().trt_enable_counters(self._h, 1 if on else 0))

    def read_counters(self):
        p, s = C.c_ulonglong(), C.c_ulonglong()
        _check(lib().trt_read_counters(self._h, C.byref(p), C.byref(s)))
        return p.value, s.value

    def read_diagnostics(self):
        t, p = C.c_ulonglong(), C.c_ulonglong()
        _check(lib().trt_read_diagnostics(self._h, C.byref(t), C.byref(p)))
        v = C.c_ulonglong()
        _check(lib().trt_read_shading_passes(self._h, C.byref(v)))
        loops = (C.c_ulonglong * 8)()
        _check(lib().trt_read_loop_diagnostics(self._h, loops))
        kinds = ("path", "directional", "point")
        return {"wave_loop_trips": t.value, "phase2_rounds": p.value, "swept_traces": self.read_sweep_fallbacks(), "shading_passes": v.value,
                "exact_loop_iterations": {k: int(loops[i]) for i, k in enumerate(kinds)},
                "exact_loop_lane_activity": {k: round(loops[3 + i] / max(1, 64 * loops[i]), 4) for i, k in enumerate(kinds)},
                "point_light_closest_hit_fallbacks": int(loops[6])}

    def kernel_info(self):
        v = [_I() for _ in range(5)]
        _check(lib().trt_kernel_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("vgprs", "sgprs", "static_lds_bytes", "max_blocks_per_cu", "compute_units"),
                        [x.value for x in v]))

    def selftest_div_sqrt(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        q, r = np.empty_like(a), np.empty_like(a)
        _check(lib().trt_selftest_div_sqrt(self._h, a.ctypes.data, b.ctypes.data, a.size, q.ctypes.data, r.ctypes.data))
        return q, r

    def probe_rays(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = rays.shape[0]
        obj = np.zeros(n, dtype=np.int32)
        point, normal, material, lit = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 5)), np.zeros((n, 3))
        _check(lib().trt_probe_rays(self._h, rays.ctypes.data, n, obj.ctypes.data, point.ctypes.data,
                                    normal.ctypes.data, material.ctypes.data, lit.ctypes.data))
        return obj, point, normal, material, lit


    def probe_rays_production(self, camera_array, rays, families=None):
        """trt_probe_rays_production: the probe through the production kernel's stages (tables, sweep, exact tests, lighting)"""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = rays.shape[0]
        fam = None if families is None else np.ascontiguousarray(families, dtype=np.int32)
        obj = np.zeros(n, dtype=np.int32)
        point, normal, material, lit = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 5)), np.zeros((n, 3))
        cam = camera_struct(camera_array)
        _check(lib().trt_probe_rays_production(self._h, C.byref(cam), rays.ctypes.data, None if fam is None else fam.ctypes.data, n,
                                               obj.ctypes.data, point.ctypes.data, normal.ctypes.data, material.ctypes.data, lit.ctypes.data))
        return obj, point, normal, material, lit


def render_frame(scene_data, width, height, bounce_limit=10, rays_per_pixel=10, symbol="trt_render_frame"):
    """Host-in/host-out frame through the extended entry trt_render_frame, or its alias render_frame (default context)."""
    from .scenes import new_screen
    scene = scene_data.as_scene()
    screen, pixels = new_screen(width, height)
    _check(getattr(lib(), symbol)(C.byref(scene), C.byref(screen), bounce_limit, rays_per_pixel))
    return pixels


def render_frame_rgb8(scene_data, width, height, bounce_limit=10, rays_per_pixel=10):
    """Host-in, emitter-bytes-out frame: uint8 [height, width, 3] = (int)(c*255) formed on the device (trt_render_frame_rgb8)."""
    scene = scene_data.as_scene()
    out = np.zeros((height, width, 3), dtype=np.uint8)
    _check(lib().trt_render_frame_rgb8(C.byref(scene), width, height, bounce_limit, rays_per_pixel, out.ctypes.data))
    return out


def project_scene(scene_data, width, height):
    """The drop-in symbol itself: void project_scene(Scene*, Screen*) at B=10, spp=10 (TRT.c:966)."""
    from .scenes import new_screen
    scene = scene_data.as_scene()
    screen, pixels = new_screen(width, height)
    lib().project_scene(C.byref(scene), C.byref(screen))
    return pixels


def _dist_check(code):
    if code != TRT_OK:
        raise TrtError(code, lib().trt_dist_last_error().decode())


def dist_allow_rccl_override(allow=True):
    """TEST HOOK: let TRT_RCCL_LIB name the library bound in RCCL's place; before the first use of RCCL (trt_dist_allow_rccl_override)"""
    _dist_check(lib().trt_dist_allow_rccl_override(1 if allow else 0))


def dist_rccl_library():
    """which library this process bound for RCCL's entry points ("" before the first use)"""
    return lib().trt_dist_rccl_library().decode()


def dist_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0; trt_dist_unique_id)"""
    buf = C.create_string_buffer(128)
    _dist_check(lib().trt_dist_unique_id(buf))
    return buf.raw


class Dist:
    """trt_dist: this rank's part of one frame sharded over the GPUs of a node (include/trt_hip.h, section 3).
    Row tiles, the RCCL communicator, the gather and the frame pipeline all live in the library."""

    def __init__(self, device, scene_data, unique_id, rank, world, width, height, tile_rows=8, frames_in_flight=2, reserved_cus=0):
        self._h = _VP()
        self._scene = scene_data.as_scene()
        idbuf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        _dist_check(lib().trt_dist_create(device, C.byref(self._scene), idbuf, rank, world, width, height, tile_rows, frames_in_flight,
                                          reserved_cus, C.byref(self._h)))
        self.device, self.rank, self.world, self.width, self.height = device, rank, world, width, height
        self.frames_in_flight = frames_in_flight
        v = [_I() for _ in range(5)]
        _dist_check(lib().trt_dist_info(self._h, *[C.byref(x) for x in v]))
        self.local_rows, self.max_rows = v[2].value, v[3].value

    def context(self, slot=0):
        h = lib().trt_dist_context(self._h, slot)
        if not h:
            raise IndexError(slot)
        return Context(self.device, _borrowed=h)

    def render(self, camera_array, bounce_limit, rays_per_pixel):
        """enqueue one frame; returns the device address of the assembled frame on rank 0 (None elsewhere)"""
        cam = camera_struct(camera_array)
        out = _VP()
        _dist_check(lib().trt_dist_render(self._h, C.byref(cam), bounce_limit, rays_per_pixel, C.byref(out)))
        return out.value

    def synchronize(self):
        _dist_check(lib().trt_dist_synchronize(self._h))

    def fetch(self, device_frame):
        out = np.zeros((self.height, self.width, 3), dtype=np.float64)
        _dist_check(lib().trt_dist_fetch(self._h, _VP(device_frame), out.ctypes.data))
        return out

    def set_scene(self, scene_data):
        """another scene for every frame slot: the first slot builds its tables, the others share them (trt_dist_set_scene)"""
        scene = scene_data.as_scene()
        _dist_check(lib().trt_dist_set_scene(self._h, C.byref(scene)))

    def comm_ranks(self):
        """ranks of the communicator as RCCL reports them (ncclCommCount); 0 without a communicator (trt_dist_comm_ranks)"""
        n = lib().trt_dist_comm_ranks(self._h)
        if n < 0:
            _dist_check(n)
        return n

    def frame_times(self):
        """(render_ms, gather_ms) of this rank, averaged over the slots' most recent frames (trt_dist_frame_times)"""
        r, g = C.c_float(), C.c_float()
        _dist_check(lib().trt_dist_frame_times(self._h, C.byref(r), C.byref(g)))
        return float(r.value), float(g.value)

    def enable_rgb8(self):
        """byte buffers for frames gathered as the emitter's (int)(c*255) triplets (trt_dist_enable_rgb8)"""
        _dist_check(lib().trt_dist_enable_rgb8(self._h))

    def render_rgb8(self, camera_array, bounce_limit, rays_per_pixel):
        cam = camera_struct(camera_array)
        out = _VP()
        _dist_check(lib().trt_dist_render_rgb8(self._h, C.byref(cam), bounce_limit, rays_per_pixel, C.byref(out)))
        return out.value

    def fetch_rgb8(self, device_frame):
        out = np.zeros((self.height, self.width, 3), dtype=np.uint8)
        _dist_check(lib().trt_dist_fetch_rgb8(self._h, _VP(device_frame), out.ctypes.data))
        return out

    def close(self):
        if self._h:
            lib().trt_dist_destroy(self._h)
            self._h = _VP()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
