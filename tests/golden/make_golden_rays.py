#!/usr/bin/env python3
"""Single-ray vectors for the PRODUCTION kernel's probe (trt_probe_rays_production), from the GENUINE reference
(oracle/_ref, built by oracle/Makefile from /root/reference/TerminalRayTracer.c where it lies).

    python tests/golden/make_golden_rays.py        # build container only

Writes tests/golden/rays_families.npz: chains of path rays as project_scene produces them (TerminalRayTracer.c:1018-1057) --
a primary ray from the eye, its reflection off whatever it hits, and so on -- each ray with the FAMILY of the production
kernel's path-ray tables it belongs to (csrc/trt_raygrid.h: 0 eye, 1 mirror eye, 2 + i sphere i, 2 + N + i mirror sphere
i) and the outputs of the reference's own trace_ray (:793) and apply_lighting (:894) for it.  The reflections are formed
by the reference's reflect_vector (:627) and normalize_vector (:439)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from terminalraytracer_amd import layout as L  # noqa: E402
from terminalraytracer_amd import scenes as S  # noqa: E402
import make_golden as G  # noqa: E402


def main():
    ref = G.RefLib.get(10, 10)
    lib = ref.lib
    lib.reflect_vector.argtypes = [C.POINTER(L.Vector), C.POINTER(L.Vector)]
    lib.normalize_vector.argtypes = [C.POINTER(L.Vector)]
    sky = G.load_ref_sky(ref, "uv_checker")
    cam = ref.orbit_camera(2.5)
    cam[13] = 5 * 128.0 / 72.0
    out = {}
    for tag, n, mirror in (("a", 64, 0.25), ("b", 256, 0.0)):
        sc = S.synth_scene(n, sky, cam, seed=1234, mirror_fraction=mirror)
        ground = sc.ground.copy()
        ground[9] = ground[14] = 0.9  # a reflective floor: chains continue off the ground
        sc = S.SceneData(sc.spheres, ground, sc.dir_lights, sc.point_lights, sc.camera, sc.sky)
        scene = sc.as_scene()
        rng = np.random.default_rng(20261004 + n)
        eye = cam[9:12]
        rays, fams, obj, pts, nrm, mat, lit = [], [], [], [], [], [], []
        for chain in range(260):
            if chain % 2:
                d = sc.spheres[rng.integers(0, n), :3] + rng.normal(scale=0.3, size=3) - eye  # towards the spheres
            else:
                d = rng.normal(size=3) + np.array([0.0, -0.6, 0.0])                            # anywhere, floor-heavy
            d = d / np.linalg.norm(d)
            o, fam = eye.copy(), 0
            for depth in range(7):
                ray = L.Ray(L.Vector(*o), L.Vector(*d))
                pt, nr, mt = L.Vector(), L.Vector(), L.Material()
                what = lib.trace_ray(C.byref(scene), C.byref(ray), C.byref(pt), C.byref(nr), C.byref(mt))
                rays.append([*o, *d])
                fams.append(fam)
                obj.append(what)
                pts.append([pt.x, pt.y, pt.z])
                nrm.append([nr.x, nr.y, nr.z])
                mat.append([mt.color.x, mt.color.y, mt.color.z, mt.reflectivity, mt.specularity])
                if what == L.NONE:
                    lit.append([0.0, 0.0, 0.0])
                    break
                view = L.Vector(-d[0], -d[1], -d[2])
                shaded = L.Material(mt.color, mt.reflectivity, mt.specularity)
                lib.apply_lighting(C.byref(scene), C.byref(pt), C.byref(view), C.byref(nr), C.byref(shaded))
                lit.append([shaded.color.x, shaded.color.y, shaded.color.z])
                # the next path ray, TerminalRayTracer.c:1054-1056
                v = L.Vector(*d)
                lib.reflect_vector(C.byref(v), C.byref(nr))
                lib.normalize_vector(C.byref(v))
                if what == L.SPHERE:
                    p = np.array([pt.x, pt.y, pt.z])
                    fam = 2 + int(np.argmin(np.abs(np.linalg.norm(sc.spheres[:, :3] - p, axis=1) - sc.spheres[:, 3])))
                else:  # the ground: the mirror image of the parent's family
                    fam = 1 if fam == 0 else (fam + n if 2 <= fam < 2 + n else -1)
                o, d = np.array([pt.x, pt.y, pt.z]), np.array([v.x, v.y, v.z])
        fams = np.array(fams, dtype=np.int32)
        print(f"scene {tag}: {len(rays)} rays; eye {int((fams == 0).sum())}, mirror eye {int((fams == 1).sum())}, "
              f"sphere {int(((fams >= 2) & (fams < 2 + n)).sum())}, mirror sphere {int((fams >= 2 + n).sum())}, hits {int((np.array(obj) != 0).sum())}")
        out.update({f"{tag}/rays": np.array(rays), f"{tag}/families": fams, f"{tag}/obj": np.array(obj, dtype=np.int32),
                    f"{tag}/point": np.array(pts), f"{tag}/normal": np.array(nrm), f"{tag}/material": np.array(mat), f"{tag}/lit": np.array(lit)})
        out.update(sc.to_arrays(f"{tag}/scene/"))
    np.savez_compressed(os.path.join(HERE, "rays_families.npz"), **out)
    print("wrote rays_families.npz")


if __name__ == "__main__":
    main()
