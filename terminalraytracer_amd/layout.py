"""ctypes mirrors of include/trt.h (the reference's structs, TerminalRayTracer.c:61-208).

Field names, order and types are the reference's; sizes/offsets are asserted at import so a
drift from include/trt.h fails immediately.
"""
import ctypes as C

NONE, SPHERE, GROUND = 0, 1, 2  # ObjectType, TerminalRayTracer.c:61-67


class Vector(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


Point = Vector  # same layout (TerminalRayTracer.c:70-83)


class Basis(C.Structure):
    _fields_ = [("x", Vector), ("y", Vector), ("z", Vector)]


class Frame(C.Structure):
    _fields_ = [("basis", Basis), ("origin", Point)]


class Ray(C.Structure):
    _fields_ = [("origin", Point), ("direction", Vector)]


class Material(C.Structure):
    _fields_ = [("color", Vector), ("reflectivity", C.c_double), ("specularity", C.c_double)]


class Color(C.Structure):
    _fields_ = [("r", C.c_ubyte), ("g", C.c_ubyte), ("b", C.c_ubyte)]


class Skybox(C.Structure):
    _fields_ = [("colors", C.POINTER(Color) * 6), ("dim", C.c_int)]


class DirectionalLight(C.Structure):
    _fields_ = [("direction", Vector), ("color", Vector)]


class PointLight(C.Structure):
    _fields_ = [("position", Point), ("color", Vector), ("intensity", C.c_double)]


class Sphere(C.Structure):
    _fields_ = [("center", Point), ("radius", C.c_double), ("material", Material)]


class Plane(C.Structure):
    _fields_ = [("point", Point), ("normal", Vector), ("even_material", Material), ("odd_material", Material)]


class Camera(C.Structure):
    _fields_ = [("frame", Frame), ("screen_distance", C.c_double), ("screen_width", C.c_double),
                ("screen_height", C.c_double)]


class Screen(C.Structure):
    _fields_ = [("pixels", C.POINTER(Vector)), ("width", C.c_int), ("height", C.c_int)]


class Scene(C.Structure):
    _fields_ = [("spheres", C.POINTER(Sphere)), ("num_spheres", C.c_int), ("ground", Plane),
                ("directional_lights", C.POINTER(DirectionalLight)), ("num_directional_lights", C.c_int),
                ("point_lights", C.POINTER(PointLight)), ("num_point_lights", C.c_int),
                ("camera", Camera), ("skybox", Skybox)]


_EXPECT = {Vector: 24, Basis: 72, Frame: 96, Ray: 48, Material: 40, Color: 3, Skybox: 56, DirectionalLight: 48,
           PointLight: 56, Sphere: 72, Plane: 128, Camera: 120, Screen: 16, Scene: 352}
for _t, _n in _EXPECT.items():
    assert C.sizeof(_t) == _n, (_t.__name__, C.sizeof(_t), _n)
assert Scene.ground.offset == 16 and Scene.camera.offset == 176 and Scene.skybox.offset == 296
assert Sphere.radius.offset == 24 and Sphere.material.offset == 32
