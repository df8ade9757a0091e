#!/usr/bin/env python3
"""Generates the committed golden fixtures from the ORACLE (oracle/liborc.so).

The reference ships no golden vectors and cannot run here (Java, no JDK), so these fixtures pin the
oracle against silent drift and give the HIP path a fixed target that does not depend on the oracle
being rebuilt: inputs (seeded meshes are regenerated from meshgen by name + seed; rays and cameras
are stored) and expected outputs (hit ids, distances, points, normals, radiance).

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
import ray_tracer_amd as rt  # noqa: E402
from helpers import CORNELL_LIGHT, LIGHT, _both_sides, cornell_box, synthetic_brdf_tables, whitted_objects  # noqa: E402

CASES = {
    # name: (mesh spec, material kind, frame WxH, spp, lighting, orientation, fov)
    "blob2k_phong": dict(faces=2000, span=0.156, seed=1, stretch=(1.0, 0.99, 0.78), W=48, H=40, ms=2, phong=1, orient=(0, 0, 0), fov=15.0),
    "blob3k_plain_large": dict(faces=3000, span=400.0, seed=4, stretch=(0.59, 1.0, 0.52), W=40, H=48, ms=1, phong=0, orient=(0, 0, 0), fov=10.0),
    "blob2k_brdf": dict(faces=2500, span=0.2, seed=3, stretch=(0.41, 1.0, 0.41), W=32, H=48, ms=1, phong=1, orient=(180, 0, 0), fov=10.0, brdf=3),
}


# the reference's own analytic scenes (RayTracer.java:111-113,147-149,155-235): spheres, planes, glass, mirror, checkerboard
ANALYTIC_CASES = {
    "whitted_phong": dict(objects=lambda: whitted_objects(rt, False), W=64, H=48, ms=2, orient=(0, 0, 0), fov=35.0, light=LIGHT),
    "cornell_phong": dict(objects=lambda: cornell_box(rt, True, True), W=48, H=48, ms=1, orient=(180, 0, 0), fov=15.0, light=CORNELL_LIGHT),
}


def build_analytic(case):
    """(host objects, oracle scene) of an analytic case."""
    sides = [_both_sides(o, True) for o in case["objects"]()]
    return [h for h, _ in sides], oracle.OracleScene([o for _, o in sides])


def analytic_settings(case, eps):
    lt = case["light"]
    return oracle.settings(epsilon=eps, ms=case["ms"], lighting=1, seed=99, light_origin=lt["origin"], light_emission=lt["emission"],
                           light_ambient=lt["ambient"])


def analytic_rays(case, sc):
    """Rays aimed into the scene's bounds from around it."""
    rs = np.random.RandomState(4321)
    mm = np.array([sc.object_info(k)["minmax"] for k in range(sc.n_objs)])
    lo, hi = mm[:, :3].min(axis=0), mm[:, 3:].max(axis=0)
    n = 400
    o = (lo + rs.uniform(-0.3, 1.3, size=(n, 3)) * (hi - lo)).astype(np.float32)
    tgt = (lo + rs.uniform(0.1, 0.9, size=(n, 3)) * (hi - lo)).astype(np.float32)
    return o, (tgt - o).astype(np.float32)


def build(case):
    v, i = rt.meshgen.blob(case["faces"], span=case["span"], stretch=case["stretch"], seed=case["seed"])
    kind = 1 if "brdf" in case else 0
    mat = oracle.material(kind=kind, diffuse=(0.2, 0.8, 0.3) if kind else (0.9, 0.9, 0.9), brdf_id=case.get("brdf", 0))
    sc = oracle.OracleScene([(v, i, None, mat)])
    if kind:
        sc.set_brdf(*synthetic_brdf_tables())
    return v, i, sc


def main():
    for name, case in CASES.items():
        v, i, sc = build(case)
        eps = sc.epsilon()
        cam = sc.camera(case["orient"], case["fov"], 0.01, case["W"], case["H"])
        st = oracle.settings(epsilon=eps, ms=case["ms"], brdf_samples=2, lighting=case["phong"], seed=99)
        fr = sc.render(cam, st, n_threads=1)
        rs = np.random.RandomState(1234)
        n = 300
        ext = float(case["span"])
        o = (rs.uniform(-0.45, 0.45, size=(n, 3)) * ext).astype(np.float32)
        d = rs.standard_normal((n, 3)).astype(np.float32)
        d[:10, 0] = 0.0
        d[10:20, 2] = -0.0
        tr = sc.trace(st, o, d)
        np.savez_compressed(os.path.join(HERE, name + ".npz"),
                            cam_origin=np.array(cam.origin[:], dtype=np.float32), cam_rotation=np.array(cam.rotation[:], dtype=np.float32),
                            epsilon=np.float32(eps), rgb=fr["rgb"], hit_id=fr["hit_id"], hit_w=fr["hit_w"],
                            ray_o=o, ray_d=d, ray_face=tr["face"], ray_w=tr["w"], ray_point=tr["point"], ray_normal=tr["normal"],
                            mesh_checksum=np.array([np.float64(v.astype(np.float64).sum()), np.float64(i.astype(np.int64).sum())]))
        print(name, "hits", int((fr["hit_id"] >= 0).sum()), "ray hits", int((tr["face"] >= 0).sum()))
        sc.close()


def main_analytic():
    for name, case in ANALYTIC_CASES.items():
        _, sc = build_analytic(case)
        eps = sc.epsilon()
        cam = sc.camera(case["orient"], case["fov"], 0.01, case["W"], case["H"])
        st = analytic_settings(case, eps)
        fr = sc.render(cam, st, n_threads=1)
        o, d = analytic_rays(case, sc)
        tr = sc.trace(st, o, d)
        np.savez_compressed(os.path.join(HERE, name + ".npz"),
                            cam_origin=np.array(cam.origin[:], dtype=np.float32), cam_rotation=np.array(cam.rotation[:], dtype=np.float32),
                            epsilon=np.float32(eps), rgb=fr["rgb"], hit_obj=fr["hit_obj"], hit_w=fr["hit_w"],
                            ray_o=o, ray_d=d, ray_object=tr["object"], ray_w=tr["w"], ray_point=tr["point"], ray_normal=tr["normal"])
        print(name, "hits", int((fr["hit_obj"] >= 0).sum()), "objects seen", sorted(set(fr["hit_obj"].ravel().tolist())),
              "ray hits", int((tr["object"] >= 0).sum()))
        sc.close()


if __name__ == "__main__":
    main_analytic()
    main()
