"""Randomised GPU-vs-oracle frames: random meshes, materials, lights inside and outside the scene, both shadow
pipelines.  Used by test_gpu_fuzz.py (a seeded subset) and tools/fuzz_parity.py (as many as asked for)."""
import os

import numpy as np

from helpers import Pair, compare_frames, synthetic_brdf_tables


def random_case(rt, seed):
    """Builds, renders on both sides and compares one random scene.  Returns (ok, one-line description)."""
    rs = np.random.RandomState(1000 + seed)
    scale = float(rs.choice([0.2, 3.0, 400.0]))   # 400: EPSILON = 1e-3 (Scene.java:36)
    objs = []
    for k in range(int(rs.randint(1, 4))):
        if rs.rand() < 0.7:
            v, i = rt.meshgen.blob(int(rs.randint(300, 5000)), span=scale * rs.uniform(0.3, 1.0), seed=int(rs.randint(1 << 20)),
                                   stretch=tuple(rs.uniform(0.5, 1.0, 3)))
        else:
            v, i = rt.meshgen.soup(int(rs.randint(1000, 6000)), span=scale * rs.uniform(0.5, 1.0), tri_size=rs.uniform(0.02, 0.08),
                                   seed=int(rs.randint(1 << 20)))
        v = (v + (rs.uniform(-0.4, 0.4, 3) * scale).astype(np.float32)).astype(np.float32)
        m = rs.rand()
        color = tuple(rs.uniform(0.2, 1.0, 3))
        mat = rt.ReflectionMaterial(color) if m < 0.25 else rt.BRDFMaterial(int(rs.randint(0, 8)), color) if m < 0.45 else rt.ColorInformation(color)
        objs.append((v, i, None, mat))
    # ties: equal distances between faces of one object and between objects are decided by the reference's fold order
    twist = rs.rand()
    if twist < 0.15:     # every face of object 0 twice (the second copy in shuffled order)
        v, i, n, mat = objs[0]
        objs[0] = (v, np.ascontiguousarray(np.concatenate([i, i[rs.permutation(i.shape[0])]]), dtype=np.int32), n, mat)
    elif twist < 0.3:    # object 0 again as another object, with its own material
        v, i, n, mat = objs[0]
        objs.append((v.copy(), i.copy(), None, rt.ColorInformation(tuple(rs.uniform(0.2, 1.0, 3)))))
    elif twist < 0.4:    # some degenerate faces (two equal corners) and given, non-unit normals
        v, i, n, mat = objs[0]
        i = i.copy()
        sel = rs.choice(i.shape[0], size=max(1, i.shape[0] // 50), replace=False)
        i[sel, 2] = i[sel, 1]
        objs[0] = (v, i, (rs.standard_normal(v.shape) * rs.uniform(0.5, 3.0)).astype(np.float32), mat)
    kind = int(rs.randint(0, 3))   # the reference's far light / a light inside the scene / one around it
    lo = (0.0, 100.0, 100.0) if kind == 0 else tuple(rs.uniform(-0.5, 0.5, 3) * scale) if kind == 1 else tuple(rs.uniform(-1.5, 1.5, 3) * scale)
    light = dict(origin=lo, emission=(0.3, 0.3, 0.3), ambient=(0.5, 0.5, 0.9))
    fuse = str(rs.choice(["0", "1", ""]))
    rt.debug_switch("FUSE_SHADOW", fuse or None)   # (the library reads the environment only once: mi355rt_debug_switch)
    try:
        pair = Pair(objs, orientation=tuple(rs.uniform(-40, 40, 3)), fov=float(rs.uniform(10, 30)), phong=bool(rs.rand() < 0.85),
                    light=light, brdf_tables=synthetic_brdf_tables(), use_kdtree=bool(rs.rand() < 0.9))
        W, H, ms = int(rs.randint(40, 120)), int(rs.randint(40, 120)), int(rs.randint(1, 4))
        cam, ocam, ost = pair.cameras(W, H, ms=ms, brdf_samples=int(rs.randint(1, 3)), seed=seed)
        cam.createImage()
        o = dict(pair.oscene.render(ocam, ost, n_threads=8))
        g = np.nan_to_num(cam._rgb_rows, nan=-1.0)   # a NaN-scrubbed pixel must be one on both sides
        o["rgb"] = np.nan_to_num(o["rgb"], nan=-1.0)
        ids_equal, w_equal, max_rel, n_diff = compare_frames(g, cam.hit_id.T, cam.hit_w.T, o, g_obj=cam.hit_obj.T)
        ok = ids_equal and w_equal and max_rel <= 1e-3
        desc = "seed %d scale %g objs %d light %d fuse %r %dx%dx%d hits %.2f: %s (ids %s w %s rel %.2e diff %d)" % (
            seed, scale, len(objs), kind, fuse, W, H, ms, float((o["hit_id"] >= 0).mean()), "ok" if ok else "MISMATCH", ids_equal,
            w_equal, max_rel, n_diff)
        pair.close()
    finally:
        rt.debug_switch("FUSE_SHADOW", None)
    return ok, desc


def random_grazing_case(rt, seed):
    """Exact shadows (the light inside the scene's box, so RTStatics.shadowIntersects' own rule runs) over large flat faces with
    the light IN their plane: the shadow rays of a floor point lie in the floor, every floor face is parallel to them, the
    triangle test's divisor is rounding noise, and a face that passes reports a point anywhere inside it -- off the ray.
    The walker's two exact-shadow shortcuts must not assume otherwise (device_util.h)."""
    rs = np.random.RandomState(9000 + seed)
    scale = float(rs.choice([0.2, 3.0, 400.0]))
    y0 = float(rs.uniform(-0.3, 0.0) * scale)
    n = int(rs.randint(1, 5))                       # floor of n x n quads: few, large faces
    ext = scale * float(rs.uniform(0.8, 1.5))
    xs = np.linspace(-ext, ext, n + 1)
    gx, gz = np.meshgrid(xs, xs, indexing="ij")
    tilt = float(rs.choice([0.0, 0.0, rs.uniform(-0.2, 0.2)]))     # the floor's plane: y = y0 + tilt * x
    fv = np.stack([gx.ravel(), y0 + tilt * gx.ravel(), gz.ravel()], axis=1).astype(np.float32)
    quads = [(a * (n + 1) + b, a * (n + 1) + b + 1, (a + 1) * (n + 1) + b + 1, (a + 1) * (n + 1) + b) for a in range(n) for b in range(n)]
    fi = np.array([t for q in quads for t in ((q[0], q[1], q[2]), (q[0], q[2], q[3]))], dtype=np.int32)
    if rs.rand() < 0.5:                             # and one face spanning the whole floor on top of the others
        fi = np.concatenate([fi, [[0, n, (n + 1) * (n + 1) - 1]]]).astype(np.int32)
    objs = [(fv, fi, None, rt.ColorInformation(tuple(rs.uniform(0.3, 1.0, 3))))]
    for k in range(int(rs.randint(1, 4))):          # things standing on it
        v, i = rt.meshgen.blob(int(rs.randint(200, 3000)), span=scale * rs.uniform(0.2, 0.6), seed=int(rs.randint(1 << 20)))
        off = np.array([rs.uniform(-0.6, 0.6) * ext, 0.0, rs.uniform(-0.6, 0.6) * ext])
        off[1] = y0 + tilt * off[0] - float(v[:, 1].min()) - rs.uniform(0.0, 0.05) * scale    # sunk a little into the floor
        objs.append(((v + off.astype(np.float32)).astype(np.float32), i, None, rt.ColorInformation(tuple(rs.uniform(0.3, 1.0, 3)))))
    lx, lz = (rs.uniform(-1.2, 1.2, 2) * ext).astype(np.float32)
    ly = np.float32(y0 + tilt * float(lx))          # in the floor's plane (as nearly as f32 has it) ...
    if rs.rand() < 0.3:
        ly = np.float32(ly + rs.uniform(-1e-4, 1e-4) * scale)   # ... or a hair off it
    light = dict(origin=(float(lx), float(ly), float(lz)), emission=(0.3, 0.3, 0.3), ambient=(0.5, 0.5, 0.9))
    pair = Pair(objs, orientation=(float(rs.uniform(-40, 40)), float(rs.uniform(15, 60)), 0.0), fov=float(rs.uniform(15, 35)),
                light=light, use_kdtree=bool(rs.rand() < 0.9))
    W, H, ms = int(rs.randint(60, 140)), int(rs.randint(60, 140)), int(rs.randint(1, 3))
    cam, ocam, ost = pair.cameras(W, H, ms=ms, seed=seed)
    cam.createImage()
    o = dict(pair.oscene.render(ocam, ost, n_threads=8))
    g = np.nan_to_num(cam._rgb_rows, nan=-1.0)
    o["rgb"] = np.nan_to_num(o["rgb"], nan=-1.0)
    ids_equal, w_equal, max_rel, n_diff = compare_frames(g, cam.hit_id.T, cam.hit_w.T, o, g_obj=cam.hit_obj.T)
    ok = ids_equal and w_equal and max_rel <= 1e-3
    floor_frac = float((o["hit_obj"] == 0).mean())
    desc = "grazing seed %d scale %g floor %dx%d tilt %.2f objs %d %dx%dx%d floor pixels %.2f: %s (ids %s w %s rel %.2e diff %d)" % (
        seed, scale, n, n, tilt, len(objs), W, H, ms, floor_frac, "ok" if ok else "MISMATCH", ids_equal, w_equal, max_rel, n_diff)
    pair.close()
    return ok, desc


def random_analytic_case(rt, seed):
    """The same with the analytic objects of the reference's Whitted / Cornell scenes: planes, spheres, an optional mesh;
    plain / mirror / refraction / checkerboard / BRDF materials (bounce chains through glass and mirrors)."""
    rs = np.random.RandomState(5000 + seed)

    def material():
        m = rs.rand()
        color = tuple(rs.uniform(0.1, 1.0, 3))
        if m < 0.2:
            return rt.ReflectionMaterial(color)
        if m < 0.4:
            return rt.RefractionMaterial(color, float(rs.choice([rt.RefractionMaterial.INDEX_OF_GLASS, 1.1, 1.33])))
        if m < 0.55:
            return rt.CheckerboardMaterial(color, tuple(rs.uniform(0.1, 1.0, 3)), *[float(x) for x in rs.uniform(2.0, 12.0, 3)])
        if m < 0.7:
            return rt.BRDFMaterial(int(rs.randint(0, 8)), color)
        return rt.ColorInformation(color)

    objs = []
    h = float(rs.uniform(8.0, 20.0))
    e = float(rs.uniform(30.0, 60.0))
    if rs.rand() < 0.8:   # floor
        objs.append(rt.Plane([(-e, -h, -e), (-e, -h, e), (e, -h, e), (e, -h, -e)], material()))
    if rs.rand() < 0.5:   # back wall
        objs.append(rt.Plane([(-e, -h, -e), (e, -h, -e), (e, e, -e), (-e, e, -e)], material()))
    if rs.rand() < 0.3:   # tilted plane, as the Whitted scene's
        objs.append(rt.Plane([(-e, 0.0, -2 * e), (-e, -h * 2, e / 2), (e, -h * 2, e / 2), (e, 0.0, -2 * e)], material()))
    for k in range(int(rs.randint(1, 5))):
        objs.append(rt.Sphere(float(rs.uniform(1.5, 6.0)), tuple(float(x) for x in rs.uniform(-12.0, 12.0, 3)), material()))
    if rs.rand() < 0.4:
        v, i = rt.meshgen.blob(int(rs.randint(300, 3000)), span=float(rs.uniform(6.0, 14.0)), seed=int(rs.randint(1 << 20)))
        v = (v + rs.uniform(-10.0, 10.0, 3).astype(np.float32)).astype(np.float32)
        objs.append((v, i, None, material()))
    lo = (0.0, 100.0, 100.0) if rs.rand() < 0.5 else tuple(float(x) for x in rs.uniform(-40.0, 40.0, 3))
    light = dict(origin=lo, emission=(0.3, 0.3, 0.3), ambient=(0.5, 0.5, 0.9))
    pair = Pair(objs, orientation=tuple(rs.uniform(-30, 30, 3)), fov=float(rs.uniform(15, 40)), phong=bool(rs.rand() < 0.8),
                light=light, brdf_tables=synthetic_brdf_tables())
    W, H, ms = int(rs.randint(40, 110)), int(rs.randint(40, 110)), int(rs.randint(1, 3))
    cam, ocam, ost = pair.cameras(W, H, ms=ms, brdf_samples=int(rs.randint(1, 3)), seed=seed)
    cam.createImage()
    o = dict(pair.oscene.render(ocam, ost, n_threads=8))
    g = np.nan_to_num(cam._rgb_rows, nan=-1.0)
    o["rgb"] = np.nan_to_num(o["rgb"], nan=-1.0)
    # hit ids of analytic objects are -1 on both sides (no faces): the object index is compared through w and radiance
    ids_equal, w_equal, max_rel, n_diff = compare_frames(g, cam.hit_id.T, cam.hit_w.T, o, g_obj=cam.hit_obj.T)
    ok = ids_equal and w_equal and max_rel <= 1e-3
    desc = "analytic seed %d objs %d %dx%dx%d hits(w>0) %.2f: %s (ids %s w %s rel %.2e diff %d)" % (
        seed, len(objs), W, H, ms, float((o["hit_w"] > 0).mean()), "ok" if ok else "MISMATCH", ids_equal, w_equal, max_rel, n_diff)
    pair.close()
    return ok, desc


def random_rays_case(rt, seed):
    """mi355rt_trace_rays against the oracle on arbitrary rays: origins inside and outside the mesh, zero direction
    components (1/d infinite), rays aimed exactly at vertices and edge midpoints (hits in several faces at one distance:
    the tie rule), origins on the surface."""
    rs = np.random.RandomState(7000 + seed)
    scale = float(rs.choice([0.2, 3.0, 400.0]))
    objs = []
    for k in range(int(rs.randint(1, 3))):
        v, i = rt.meshgen.blob(int(rs.randint(100, 4000)), span=scale * rs.uniform(0.4, 1.0), seed=int(rs.randint(1 << 20)))
        v = (v + (rs.uniform(-0.3, 0.3, 3) * scale).astype(np.float32)).astype(np.float32)
        objs.append((v, i, None, rt.ColorInformation((1, 1, 1))))
    pair = Pair(objs, use_kdtree=bool(rs.rand() < 0.9))
    cam, ocam, ost = pair.cameras(8, 8)
    n = 3000
    o = (rs.uniform(-0.8, 0.8, size=(n, 3)) * scale).astype(np.float32)
    d = rs.standard_normal((n, 3)).astype(np.float32)
    v0, i0 = objs[0][0], objs[0][1]
    vt = v0[rs.randint(0, v0.shape[0], 600)]                    # aimed at vertices
    d[:600] = vt - o[:600]
    f = i0[rs.randint(0, i0.shape[0], 600)]
    mid = ((v0[f[:, 0]].astype(np.float64) + v0[f[:, 1]]) / 2).astype(np.float32)   # aimed at edge midpoints
    d[600:1200] = mid - o[600:1200]
    o[1200:1400] = v0[rs.randint(0, v0.shape[0], 200)]         # origins on vertices
    for c in range(3):                                          # zero components, of both signs
        d[1400 + 100 * c:1500 + 100 * c, c] = 0.0 if c != 1 else -0.0
    d[1700:1750, 0] = 0.0
    d[1700:1750, 2] = 0.0
    g = cam.getClosestIntersection(o, d)
    r = pair.oscene.trace(ost, o, d)
    hit = r["face"] >= 0
    ok = (np.array_equal(g["face"], r["face"]) and np.array_equal(g["object"], r["object"])
          and np.array_equal(g["w"].view(np.uint32), r["w"].view(np.uint32))
          and np.array_equal(g["point"][hit].view(np.uint32), r["point"][hit].view(np.uint32))
          and np.array_equal(g["normal"][hit].view(np.uint32), r["normal"][hit].view(np.uint32)))
    desc = "rays seed %d scale %g objs %d hits %d of %d: %s (faces differing: %d)" % (
        seed, scale, len(objs), int(hit.sum()), n, "ok" if ok else "MISMATCH", int((g["face"] != r["face"]).sum()))
    pair.close()
    return ok, desc
