"""ctypes binding of the host-side C companions in libtrt_hip.so (include/trt_host.h):
camera orbit, PPM/cubemap loader, ANSI emitter -- the reference's code either side of
project_scene (TerminalRayTracer.c:290-436, 558-624, 1084-1172, 1327-1336)."""
import ctypes as C

import numpy as np

from . import layout as L
from .hip import lib as _lib

_VP, _I = C.c_void_p, C.c_int
HOST_SYMBOLS = {
    "trt_init_frame": (None, [C.POINTER(L.Frame)]),
    "trt_init_camera": (None, [C.POINTER(L.Camera), _I, _I]),
    "trt_rotate_basis": (None, [C.POINTER(L.Basis), C.POINTER(L.Basis)]),
    "trt_rotate_basis_x": (None, [C.POINTER(L.Basis), C.c_double]),
    "trt_rotate_basis_y": (None, [C.POINTER(L.Basis), C.c_double]),
    "trt_rotate_basis_z": (None, [C.POINTER(L.Basis), C.c_double]),
    "trt_transform_frame": (None, [C.POINTER(L.Frame), C.POINTER(L.Frame)]),
    "trt_orbit_camera": (None, [C.POINTER(L.Camera), C.c_double]),
    "trt_read_ppm": (_I, [C.c_char_p, C.POINTER(C.POINTER(L.Color)), C.POINTER(_I), C.POINTER(_I)]),
    "trt_load_skybox": (_I, [C.POINTER(L.Skybox), C.c_char_p]),
    "trt_free_skybox": (None, [C.POINTER(L.Skybox)]),
    "trt_emitter_create": (_I, [_I, _I, C.POINTER(_VP)]),
    "trt_emitter_destroy": (None, [_VP]),
    "trt_emitter_buffer": (_VP, [_VP]),
    "trt_emitter_size": (C.c_size_t, [_VP]),
    "trt_emitter_patch": (C.c_int, [_VP, C.POINTER(L.Screen)]),
    "trt_emitter_patch_rgb8": (C.c_int, [_VP, _VP]),
    "trt_emitter_write": (_I, [_VP, _VP]),
    "trt_draw_screen": (_I, [C.POINTER(L.Screen), _VP]),
    "trt_fnv1a64": (C.c_ulonglong, [_VP, C.c_size_t]),
}
def lib():
    dll = _lib()
    if not getattr(dll, "_trt_host_bound", False):  # per library object: hip.lib() may have been reloaded
        for name, (res, args) in HOST_SYMBOLS.items():
            fn = getattr(dll, name)
            fn.restype = res
            fn.argtypes = args
        dll._trt_host_bound = True
    return dll


def orbit_camera(t, aspect_w=480, aspect_h=280):
    """Camera (15 doubles) of the reference's frame loop at second t: init_camera + orbit."""
    cam = L.Camera()
    lib().trt_init_camera(C.byref(cam), aspect_w, aspect_h)
    lib().trt_orbit_camera(C.byref(cam), t)
    return np.frombuffer(bytes(cam), dtype=np.float64).copy()


def load_skybox(directory):
    """(6, dim, dim, 3) uint8 from <directory>/{+X,-X,+Y,-Y,+Z,-Z}.ppm; raises OSError(code) on failure."""
    sky = L.Skybox()
    rc = lib().trt_load_skybox(C.byref(sky), directory.encode())
    if rc != 0:
        raise OSError(rc, f"trt_load_skybox({directory}) failed with {rc}")
    try:
        dim = sky.dim
        faces = [np.frombuffer((C.c_ubyte * (dim * dim * 3)).from_address(C.addressof(sky.colors[f].contents)),
                               dtype=np.uint8).reshape(dim, dim, 3).copy() for f in range(6)]
    finally:
        lib().trt_free_skybox(C.byref(sky))
    return np.stack(faces)


class Emitter:
    def __init__(self, width, height):
        self._h = _VP()
        rc = lib().trt_emitter_create(width, height, C.byref(self._h))
        if rc != 0:
            raise ValueError(f"trt_emitter_create({width}, {height}) failed with {rc}")
        self.width, self.height = width, height

    def patch(self, pixels):
        px = np.ascontiguousarray(pixels, dtype=np.float64).reshape(self.height, self.width, 3)
        scr = L.Screen()
        scr.pixels = px.ctypes.data_as(C.POINTER(L.Vector))
        scr.width, scr.height = self.width, self.height
        rc = lib().trt_emitter_patch(self._h, C.byref(scr))
        if rc != 0:
            raise ValueError(f"trt_emitter_patch failed with {rc}")

    def patch_rgb8(self, rgb):
        a = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(self.height, self.width, 3)
        rc = lib().trt_emitter_patch_rgb8(self._h, a.ctypes.data)
        if rc != 0:
            raise ValueError(f"trt_emitter_patch_rgb8 failed with {rc}")

    def bytes(self):
        n = lib().trt_emitter_size(self._h)
        return C.string_at(lib().trt_emitter_buffer(self._h), n)

    def close(self):
        if self._h:
            lib().trt_emitter_destroy(self._h)
            self._h = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fnv1a64(buf):
    """FNV-1a-64 of an array's bytes as 16 hex digits: the fingerprint the golden frames are recorded with (trt_fnv1a64)."""
    a = np.ascontiguousarray(buf)
    return f"{lib().trt_fnv1a64(a.ctypes.data, a.nbytes):016x}"
