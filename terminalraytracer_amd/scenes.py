"""Scene containers and the scene/camera definitions the tests and the bench use.

A SceneData owns plain float64/uint8 numpy arrays whose memory layout IS the reference's
struct layout (Sphere = 9 doubles, Plane = 16, DirectionalLight = 6, PointLight = 7,
Camera = 15; TerminalRayTracer.c:146-208), so `as_scene()` only wires pointers.

Scenes:
  demo_scene()        the literals of the reference's main(), TerminalRayTracer.c:1256-1288
  synth_scene(n,seed) SYNTH-v0 of SURVEY.md section 8(d): splitmix64-driven random spheres
Cameras:
  orbit_camera(t,...) the per-frame orbit of TerminalRayTracer.c:1327-1336 (host-side, once per frame)
"""
import ctypes as C
import math

import numpy as np

from . import layout as L

PI = 3.14159265358979323846  # TerminalRayTracer.c:43


class SceneData:
    def __init__(self, spheres, ground, dir_lights, point_lights, camera, sky):
        self.spheres = np.ascontiguousarray(spheres, dtype=np.float64).reshape(-1, 9)
        self.ground = np.ascontiguousarray(ground, dtype=np.float64).reshape(16)
        self.dir_lights = np.ascontiguousarray(dir_lights, dtype=np.float64).reshape(-1, 6)
        self.point_lights = np.ascontiguousarray(point_lights, dtype=np.float64).reshape(-1, 7)
        self.camera = np.ascontiguousarray(camera, dtype=np.float64).reshape(15)
        self.sky = np.ascontiguousarray(sky, dtype=np.uint8)
        assert self.sky.ndim == 4 and self.sky.shape[0] == 6 and self.sky.shape[1] == self.sky.shape[2] \
            and self.sky.shape[3] == 3, self.sky.shape

    @property
    def num_spheres(self):
        return self.spheres.shape[0]

    @property
    def sky_dim(self):
        return self.sky.shape[1]

    def with_camera(self, camera):
        return SceneData(self.spheres, self.ground, self.dir_lights, self.point_lights, camera, self.sky)

    def with_spheres(self, spheres):
        return SceneData(spheres, self.ground, self.dir_lights, self.point_lights, self.camera, self.sky)

    def as_scene(self):
        """Build the reference's `Scene` (TerminalRayTracer.c:196-208) over this object's arrays.
        The returned struct borrows the arrays: keep `self` alive while it is in use."""
        s = L.Scene()
        s.spheres = self.spheres.ctypes.data_as(C.POINTER(L.Sphere))
        s.num_spheres = self.spheres.shape[0]
        C.memmove(C.byref(s.ground), self.ground.ctypes.data, 128)
        s.directional_lights = self.dir_lights.ctypes.data_as(C.POINTER(L.DirectionalLight))
        s.num_directional_lights = self.dir_lights.shape[0]
        s.point_lights = self.point_lights.ctypes.data_as(C.POINTER(L.PointLight))
        s.num_point_lights = self.point_lights.shape[0]
        C.memmove(C.byref(s.camera), self.camera.ctypes.data, 120)
        for f in range(6):
            s.skybox.colors[f] = self.sky[f].ctypes.data_as(C.POINTER(L.Color))
        s.skybox.dim = self.sky.shape[1]
        s._owner = self
        return s

    # (de)serialisation used by tests/golden
    def to_arrays(self, prefix=""):
        return {prefix + "spheres": self.spheres, prefix + "ground": self.ground, prefix + "dir_lights": self.dir_lights,
                prefix + "point_lights": self.point_lights, prefix + "camera": self.camera}

    @staticmethod
    def from_arrays(d, sky, prefix=""):
        return SceneData(d[prefix + "spheres"], d[prefix + "ground"], d[prefix + "dir_lights"],
                         d[prefix + "point_lights"], d[prefix + "camera"], sky)


def new_screen(width, height):
    """A `Screen` (TerminalRayTracer.c:188-193) over a fresh (H, W, 3) float64 array."""
    pixels = np.zeros((height, width, 3), dtype=np.float64)
    scr = L.Screen()
    scr.pixels = pixels.ctypes.data_as(C.POINTER(L.Vector))
    scr.width = width
    scr.height = height
    scr._owner = pixels
    return scr, pixels


# ---------------------------------------------------------------------------------------------
# cameras
# ---------------------------------------------------------------------------------------------
def _dot3(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def _rotate_basis(basis, rot):
    """TerminalRayTracer.c:558-573: every axis of `basis` expressed against the rows of `rot`."""
    return [[_dot3(axis, rot[0]), _dot3(axis, rot[1]), _dot3(axis, rot[2])] for axis in basis]


def _transform_frame(frame, tf):
    """TerminalRayTracer.c:607-624: (basis, origin) of `frame` pushed through `tf` as a 4x4 transform."""
    (fb, fo), (tb, to) = frame, tf
    basis = [[a[0] * tb[0][j] + a[1] * tb[1][j] + a[2] * tb[2][j] for j in range(3)] for a in fb]
    origin = [fo[0] * tb[0][j] + fo[1] * tb[1][j] + fo[2] * tb[2][j] + to[j] for j in range(3)]
    return basis, origin


def orbit_frame(t, radius=1.99):
    """Camera frame of TerminalRayTracer.c:1327-1336 at wall-clock second `t`.
    sin/cos come from the host libm, so bit-level parity tests feed stored cameras instead."""
    ident = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
    ax = 2.0 * PI * t * -0.03
    ay = 2.0 * PI * t * 0.05
    rx = [[1.0, 0.0, 0.0], [0.0, math.cos(ax), -math.sin(ax)], [0.0, math.sin(ax), math.cos(ax)]]
    ry = [[math.cos(ay), 0.0, math.sin(ay)], [0.0, 1.0, 0.0], [-math.sin(ay), 0.0, math.cos(ay)]]
    tf0_basis = _rotate_basis(_rotate_basis(ident, rx), ry)
    tf1 = (ident, [0.0 + 0.0, 0.0 + 0.0, 0.0 + radius])
    frame = (ident, [0.0, 0.0, 0.0])
    frame = _transform_frame(frame, tf1)
    frame = _transform_frame(frame, (tf0_basis, [0.0, 0.0, 0.0]))
    return frame


def camera_array(frame, screen_width, screen_height=5.0, screen_distance=1.0):
    basis, origin = frame
    return np.array([*basis[0], *basis[1], *basis[2], *origin, screen_distance, screen_width, screen_height],
                    dtype=np.float64)


def orbit_camera(t, width, height, reference_aspect=False):
    """Camera at orbit time t.  reference_aspect=True keeps init_camera's baked 480/280 aspect
    (TerminalRayTracer.c:303); otherwise screen_width = 5*W/H as the bench configs need."""
    sw = 5 * float(480) / float(280) if reference_aspect else 5 * float(width) / float(height)
    return camera_array(orbit_frame(t), sw)


# ---------------------------------------------------------------------------------------------
# scenes
# ---------------------------------------------------------------------------------------------
def demo_ground():
    # TerminalRayTracer.c:1269-1274 with GROUND_EVEN/ODD_COLOR of :88-89
    return np.array([0.0, -2.0, 0.0, 0.0, 1.0, 0.0,
                     1.0, 1.0, 1.0, 0.2, 100.0,
                     1.0, 0.0, 0.0, 0.2, 100.0], dtype=np.float64)


def demo_lights():
    # TerminalRayTracer.c:1278-1288: one directional light, one point light at the origin
    d = np.array([[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]], dtype=np.float64)
    p = np.array([[0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 10.0]], dtype=np.float64)
    return d, p


def demo_spheres():
    # TerminalRayTracer.c:1256-1263: six unit-axis spheres, radius 0.5
    rows = [
        (1.0, 0.0, 0.0, 0.5, 1.0, 0.0, 0.0, 1.0, 100.0),
        (0.0, 1.0, 0.0, 0.5, 0.0, 1.0, 0.0, 0.8, 100.0),
        (0.0, 0.0, 1.0, 0.5, 0.0, 0.0, 1.0, 0.8, 100.0),
        (-1.0, 0.0, 0.0, 0.5, 0.0, 1.0, 1.0, 0.8, 100.0),
        (0.0, -1.0, 0.0, 0.5, 1.0, 0.0, 1.0, 0.8, 100.0),
        (0.0, 0.0, -1.0, 0.5, 1.0, 1.0, 0.0, 0.8, 100.0),
    ]
    return np.array(rows, dtype=np.float64)


def demo_scene(sky, camera, num_spheres=6):
    d, p = demo_lights()
    return SceneData(demo_spheres()[:num_spheres], demo_ground(), d, p, camera, sky)


class SplitMix64:
    """splitmix64 exactly as SURVEY.md section 8(d) states it."""
    M = (1 << 64) - 1

    def __init__(self, seed):
        self.s = seed & self.M

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & self.M
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.M
        return z ^ (z >> 31)

    def u01(self):
        return (self.next() >> 11) * (2.0 ** -53)

    def ur(self, a, b):
        return a + self.u01() * (b - a)


def synth_spheres(n, seed=1234, mirror_fraction=0.0):
    """SYNTH-v0 spheres.  mirror_fraction>0 forces reflectivity 1.0 on that share of the spheres
    (every k-th), the harsher divergence variant mentioned in SURVEY 8(d)."""
    g = SplitMix64(seed)
    rows = []
    for i in range(n):
        cx = g.ur(-4.0, 4.0)
        cy = g.ur(-1.5, 2.5)
        cz = g.ur(-4.0, 4.0)
        radius = g.ur(0.1, 0.5)
        col = (g.u01(), g.u01(), g.u01())
        refl = g.u01()
        rows.append((cx, cy, cz, radius, col[0], col[1], col[2], refl, 100.0))
    out = np.array(rows, dtype=np.float64).reshape(n, 9)
    if mirror_fraction > 0.0 and n:
        step = max(1, int(round(1.0 / mirror_fraction)))
        out[::step, 7] = 1.0
    return out


def synth_scene(n, sky, camera, seed=1234, mirror_fraction=0.0):
    d, p = demo_lights()
    return SceneData(synth_spheres(n, seed, mirror_fraction), demo_ground(), d, p, camera, sky)


def synth_sky(dim=256, seed=7):
    """Procedural cubemap for bench / parity runs that must not depend on image files:
    a per-texel hash so that any wrong face, mirror, rotation or index shows up."""
    f, v, u = np.meshgrid(np.arange(6, dtype=np.uint64), np.arange(dim, dtype=np.uint64),
                          np.arange(dim, dtype=np.uint64), indexing="ij")
    h = (f * np.uint64(0x9E3779B97F4A7C15) + v * np.uint64(0xBF58476D1CE4E5B9) + u * np.uint64(0x94D049BB133111EB)
         + np.uint64(seed))
    h ^= h >> np.uint64(29)
    h *= np.uint64(0xD6E8FEB86659FD93)
    h ^= h >> np.uint64(32)
    sky = np.empty((6, dim, dim, 3), dtype=np.uint8)
    sky[..., 0] = (h & np.uint64(0xFF)).astype(np.uint8)
    sky[..., 1] = ((h >> np.uint64(8)) & np.uint64(0xFF)).astype(np.uint8)
    sky[..., 2] = ((h >> np.uint64(16)) & np.uint64(0xFF)).astype(np.uint8)
    return sky
