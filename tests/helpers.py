"""Shared scene set-up for the parity tests: the same meshes, camera and settings go to the HIP path
(through the C ABI) and to the oracle."""
import numpy as np

import oracle
import ray_tracer_amd as rt

LIGHT = dict(origin=(0.0, 100.0, 100.0), emission=(0.3, 0.3, 0.3), ambient=(0.5, 0.5, 0.9))   # RayTracer.java:106


def to_orc_material(ci):
    return oracle.material(kind=ci.kind, diffuse=ci.diffuse, shininess=ci.shininess, brdf_id=ci.brdf_id,
                           specular=ci.specular, ambient=ci.ambient, emission=ci.emission)


def to_orc_ext(ci):
    e = ci._abi_material_ext()
    return oracle.material_ext(e.refraction_index, tuple(e.checker_color1), tuple(e.checker_color2), tuple(e.checker_cell))


def _both_sides(entry, use_kdtree):
    """(host object, oracle entry) of one Scene.objects entry: a (v, i, n, ColorInformation) mesh, an rt.Sphere or an rt.Plane."""
    if isinstance(entry, rt.Sphere):
        return entry, oracle.sphere(entry.origin, entry.radius, to_orc_material(entry.colorInfo), to_orc_ext(entry.colorInfo))
    if isinstance(entry, rt.Plane):
        return entry, oracle.plane(entry.corners.tolist(), to_orc_material(entry.colorInfo), to_orc_ext(entry.colorInfo))
    v, i, n, ci = entry[:4]
    kd = bool(entry[4]) if len(entry) > 4 else use_kdtree   # this mesh's own computeKDTree (TriangleMesh.java:23,89)
    return (rt.TriangleMesh((v, i) if n is None else (v, i, n), ci, kd),
            (v, i, n, to_orc_material(ci), to_orc_ext(ci), kd))


class Pair:
    """One scene built on both sides."""

    def __init__(self, meshes, orientation=(0, 0, 0), fov=15.0, phong=True, light=None, brdf_tables=None,
                 use_kdtree=True):
        # meshes: list of (vertices, indices, normals-or-None, ColorInformation[, computeKDTree]) and rt.Sphere / rt.Plane objects
        light = light or LIGHT
        self.light = rt.Light(light["origin"], light["emission"], light["ambient"])
        if brdf_tables is not None:
            rt.BRDFMaterial.setTables(*brdf_tables)
        sides = [_both_sides(e, use_kdtree) for e in meshes]
        tms = [h for h, _ in sides]
        lm = rt.PhongLightingModel(self.light, tms) if phong else rt.LightingModel()
        self.scene = rt.Scene(tms, self.light, orientation, lm, fov)
        self.oscene = oracle.OracleScene([o for _, o in sides],
                                         max_depth=rt.RTStatics.MAX_DEPTH, max_children=rt.RTStatics.MAX_CHILDREN,
                                         use_kdtree=use_kdtree)
        if brdf_tables is not None:
            self.oscene.set_brdf(*brdf_tables)
        self.phong = phong
        self.orientation = orientation
        self.fov = fov
        self.eps = self.oscene.epsilon()
        assert np.float32(self.eps) == np.float32(rt.RTStatics.EPSILON)

    def cameras(self, W, H, ms=1, brdf_samples=1, seed=0, origin=None):
        cam = rt.Camera(self.scene, ms, brdf_samples, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=seed)
        ocam = self.oscene.camera(self.orientation, self.fov, rt.RTStatics.NEAR_PLANE, W, H)
        if origin is not None:   # a hand-placed camera (the constructor's own placement is compared by the caller first)
            cam._cam.origin[:] = origin
            ocam.origin[:] = origin
        ost = oracle.settings(epsilon=self.eps, ms=ms, brdf_samples=brdf_samples, lighting=1 if self.phong else 0, seed=seed,
                              light_origin=self.light.origin, light_emission=self.light.emission,
                              light_ambient=self.light.ambient, bounces=rt.RTStatics.MAX_MIRROR_BOUNCES,
                              max_recursion_depth=rt.RTStatics.MAX_RECURSION_DEPTH)
        return cam, ocam, ost

    def close(self):
        self.scene.close()
        self.oscene.close()


def synthetic_brdf_tables(n_dirs=205, n_materials=8, seed=7):
    """Tables shaped like the CUReT ones (table.txt: theta_v phi_v theta_i phi_i; abrdf.dat rows)."""
    rs = np.random.RandomState(seed)
    ang = np.stack([rs.uniform(0.0, 1.45, n_dirs), rs.uniform(-np.pi, np.pi, n_dirs),
                    rs.uniform(0.0, 1.45, n_dirs), rs.uniform(-np.pi, np.pi, n_dirs)], axis=1).astype(np.float32)
    ang[1, 1] = 0.0   # a phi = 0 row, like row 2 of the real table
    w = rs.uniform(0.02, 0.35, size=(n_materials, n_dirs)).astype(np.float32)
    return ang, w


def curet_tables():
    """The reference's own CUReT tables (resources/table.txt, resources/abrdf.dat) as committed, parsed, in
    tests/golden/curet_tables.npz (tests/golden/make_curet_golden.py): angles[205, 4], weights[61, 205]."""
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "curet_tables.npz"))
    return f["angles"], f["weights"]


def compare_frames(g_rgb, g_id, g_w, o, rtol=1e-3, atol=1e-6, g_obj=None):
    """ids, hit objects and w bit-exact; radiance within 1e-3 relative (north_star), NaN-scrubbed pixels equal.
    g_obj: the frame's hit-object indices (a face index alone does not tell two objects' faces apart)."""
    same = g_id == o["hit_id"]
    if g_obj is not None:
        same = same & (g_obj == o["hit_obj"])
    ids_equal = bool(same.all())
    w_equal = np.array_equal(g_w.view(np.uint32), o["hit_w"].view(np.uint32))
    denom = np.maximum(np.abs(o["rgb"]), atol / rtol)
    rel = np.abs(g_rgb - o["rgb"]) / denom
    return ids_equal, w_equal, float(rel.max()), int((~same).sum())


# ---------------------------------------------------------------- the reference's analytic scenes (shared with tests/golden/make_golden.py)

YELLOW, RED, BLUE, GRAY, GREEN, CYAN, WHITE = (1, 1, 0), (1, 0, 0), (0, 0, 1), (128 / 255.0,) * 3, (0, 1, 0), (0, 1, 1), (1, 1, 1)


def whitted_objects(rt, use_brdfs):
    """RayTracer.getWhittedObjects (RayTracer.java:155-167)."""
    plane = rt.Plane([(-50, 0, -100), (-50, -40, 25), (50, -40, 25), (50, 0, -100)], rt.CheckerboardMaterial(YELLOW, RED, 10.0, 10.0, 10.0))
    blue = rt.ReflectionMaterial(BLUE)
    glass = rt.RefractionMaterial(GRAY, rt.RefractionMaterial.INDEX_OF_GLASS)
    s1 = rt.Sphere(5, (0, -12, 0), rt.BRDFMaterial(15, GREEN) if use_brdfs else glass)
    s2 = rt.Sphere(3, (5, -15, -10), rt.BRDFMaterial(16, CYAN) if use_brdfs else blue)
    return [plane, s1, s2]


def cornell_box(rt, use_spheres, use_whitted_materials):
    """RayTracer.getCornellBox (RayTracer.java:176-235)."""
    white, white2 = rt.ColorInformation((0.75, 0.75, 0.75)), rt.ColorInformation((0.65, 0.65, 0.65))
    red, green = rt.ColorInformation((0.63, 0.06, 0.04)), rt.ColorInformation((0.15, 0.48, 0.09))
    blue = rt.ColorInformation((0.392, 0.584, 0.93))
    mirror, glass = rt.ReflectionMaterial(WHITE), rt.RefractionMaterial(BLUE, rt.RefractionMaterial.INDEX_OF_GLASS)
    (x0, y0, z0), (x1, y1, z1) = (-278, -275, -800), (278, 275, 280)
    objs = [rt.Plane([(x1, y0, z0), (x0, y0, z0), (x0, y0, z1), (x1, y0, z1)], white),    # floor
            rt.Plane([(x1, y1, z0), (x1, y1, z1), (x0, y1, z1), (x0, y1, z0)], white),    # ceiling
            rt.Plane([(x1, y0, z1), (x0, y0, z1), (x0, y1, z1), (x1, y1, z1)], white),    # back
            rt.Plane([(x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0)], white),    # front
            rt.Plane([(x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0)], red),      # left
            rt.Plane([(x0, y0, z1), (x0, y0, z0), (x0, y1, z0), (x0, y1, z1)], green)]    # right
    if use_spheres:
        objs += [rt.Sphere(82.5, (-92, -192.5, -111.5), glass if use_whitted_materials else blue),
                 rt.Sphere(82.5, (116.5, -192.5, 71.5), mirror if use_whitted_materials else blue)]
    else:
        small = [[(-148, -110, -215), (-196, -110, -55), (-32, -110, -8), (12, -110, -166)],
                 [(12, -275, -166), (12, -110, -166), (-32, -110, -8), (-32, -275, -8)],
                 [(-148, -275, -215), (-148, -110, -215), (12, -110, -166), (12, -275, -166)],
                 [(-196, -275, -55), (-196, -110, -55), (-148, -110, -215), (-148, -275, -215)],
                 [(-32, -275, -8), (-32, -110, -8), (-196, -110, -55), (-196, -275, -55)]]
        large = [[(145, 55, -33), (-13, 55, 16), (36, 55, 176), (194, 55, 126)],
                 [(145, -275, -33), (145, 55, -33), (194, 55, 126), (194, -275, 126)],
                 [(194, -275, 126), (194, 55, 126), (36, 55, 176), (36, -275, 176)],
                 [(36, -275, 176), (36, 55, 176), (-13, 55, 16), (-13, -275, 16)],
                 [(-13, -275, 16), (-13, 55, 16), (145, 55, -33), (145, -275, -33)]]
        objs += [rt.Plane(c, white2) for c in small + large]
    return objs


CORNELL_LIGHT = dict(origin=(0.0, 270.0, 0.0), emission=(0.75, 0.75, 0.75), ambient=(0.75, 0.75, 0.75))   # RayTracer.java:109
