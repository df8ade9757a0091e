#!/usr/bin/env python3
"""Golden fixtures for the measured-BRDF path on the reference's OWN data (build container only).

    python tests/golden/make_curet_golden.py      # needs /root/reference/resources; rewrites tests/golden/curet_*.npz

The reference loads resources/table.txt (205 sampled directions: theta_v phi_v theta_i phi_i) and resources/abrdf.dat
(61 materials x 205 weights) as statics of BRDFMaterial (BRDFMaterial.java:33-43) with the token rule of
parseBRDFFile (:99-138).  /root/reference does not travel to the GPU box, so what is committed here is DATA:
  curet_tables.npz        the two files parsed with that rule (f32 arrays) + a checksum of each source file
  curet_whitted_brdf.npz  the reference's WHITTED_SCENE_BRDF (RayTracer.java:114-116,155-167: materials 15 and 16 on the
                          two spheres, plain LightingModel, fov 35) rendered by the ORACLE with those tables
  curet_blob_brdf.npz     a Buddha-class blob carrying BRDFMaterial(15) under Phong (configs[2]'s combination), oracle-rendered
The frames pin the oracle and give the HIP path a fixed target on the real tables' clustered directions (the
radius-doubling search of BRDFMaterial.java:68-83 behaves differently on them than on uniform random angles).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

RES = "/root/reference/resources"

import oracle  # noqa: E402
import ray_tracer_amd as rt  # noqa: E402
from helpers import LIGHT, _both_sides, whitted_objects  # noqa: E402

WHITTED = dict(W=96, H=72, ms=2, brdf_samples=2, orient=(0, 0, 0), fov=35.0, seed=15)
BLOB = dict(faces=30000, W=80, H=120, ms=2, brdf_samples=3, brdf_id=15, seed=16)


def load_tables():
    f = np.load(os.path.join(HERE, "curet_tables.npz"))
    return f["angles"], f["weights"]


def whitted_scene(tables):
    sides = [_both_sides(o, True) for o in whitted_objects(rt, True)]
    sc = oracle.OracleScene([o for _, o in sides])
    sc.set_brdf(*tables)
    return [h for h, _ in sides], sc


def whitted_settings(eps):
    return oracle.settings(epsilon=eps, ms=WHITTED["ms"], brdf_samples=WHITTED["brdf_samples"], lighting=0, seed=WHITTED["seed"],
                           light_origin=LIGHT["origin"], light_emission=LIGHT["emission"], light_ambient=LIGHT["ambient"])


def blob_scene(tables):
    cfg = rt.meshgen.STANFORD["buddha"]
    v, i = rt.meshgen.stanford_standin("buddha", BLOB["faces"])
    sc = oracle.OracleScene([(v, i, None, oracle.material(kind=1, diffuse=(0.0, 1.0, 0.0), brdf_id=BLOB["brdf_id"]))])
    sc.set_brdf(*tables)
    return v, i, cfg, sc


def blob_settings(eps):
    return oracle.settings(epsilon=eps, ms=BLOB["ms"], brdf_samples=BLOB["brdf_samples"], lighting=1, seed=BLOB["seed"],
                           light_origin=LIGHT["origin"], light_emission=LIGHT["emission"], light_ambient=LIGHT["ambient"])


def main():
    rows_a = rt.BRDFMaterial.parseBRDFFile(os.path.join(RES, "table.txt"))
    rows_w = rt.BRDFMaterial.parseBRDFFile(os.path.join(RES, "abrdf.dat"))
    assert len(rows_a) == 205 and all(len(r) == 4 for r in rows_a)
    assert len(rows_w) == 61 and all(len(r) == 205 for r in rows_w)
    angles = np.array(rows_a, dtype=np.float32)
    weights = np.array(rows_w, dtype=np.float32)
    sums = [hashlib.sha256(open(os.path.join(RES, n), "rb").read()).hexdigest() for n in ("table.txt", "abrdf.dat")]
    np.savez_compressed(os.path.join(HERE, "curet_tables.npz"), angles=angles, weights=weights, sha256=np.array(sums))
    tables = (angles, weights)

    _, sc = whitted_scene(tables)
    cam = sc.camera(WHITTED["orient"], WHITTED["fov"], 0.01, WHITTED["W"], WHITTED["H"])
    fr = sc.render(cam, whitted_settings(sc.epsilon()), n_threads=1)
    np.savez_compressed(os.path.join(HERE, "curet_whitted_brdf.npz"), epsilon=np.float32(sc.epsilon()),
                        cam_origin=np.array(cam.origin[:], dtype=np.float32), rgb=fr["rgb"], hit_obj=fr["hit_obj"], hit_w=fr["hit_w"])
    print("whitted brdf: objects seen", sorted(set(fr["hit_obj"].ravel().tolist())), "rgb range", float(fr["rgb"].min()), float(fr["rgb"].max()))
    sc.close()

    v, i, cfg, sc = blob_scene(tables)
    cam = sc.camera(cfg["orientation"], cfg["fov"], 0.01, BLOB["W"], BLOB["H"])
    fr = sc.render(cam, blob_settings(sc.epsilon()), n_threads=4)
    # a few luminance values straight from getBRDFLuminocity, for several materials (known answers on the real tables)
    L = oracle.lib()
    probes = []
    rs = np.random.RandomState(3)
    for m in (0, 15, 16, 60):
        for _ in range(8):
            d = rs.standard_normal(3)
            d[2] = -abs(d[2]) - 0.2
            n = rs.standard_normal(3) * 0.3 + np.array([0, 0, 1.0])
            args = (oracle.vec(*d), oracle.vec(*(rs.uniform(-0.1, 0.1, 3))), oracle.vec(*n), oracle.vec(0, 100, 100))
            probes.append([m] + list(d) + list(args[1]) + list(n) + [float(L.orc_brdf_luminocity(sc.h, m, 2, *args))])
    np.savez_compressed(os.path.join(HERE, "curet_blob_brdf.npz"), epsilon=np.float32(sc.epsilon()),
                        cam_origin=np.array(cam.origin[:], dtype=np.float32), rgb=fr["rgb"], hit_id=fr["hit_id"], hit_w=fr["hit_w"],
                        mesh_checksum=np.array([np.float64(v.astype(np.float64).sum()), np.float64(i.astype(np.int64).sum())]),
                        luminocity_probes=np.array(probes, dtype=np.float64))
    print("blob brdf: hits", int((fr["hit_id"] >= 0).sum()), "rgb range", float(fr["rgb"].min()), float(fr["rgb"].max()))
    sc.close()


if __name__ == "__main__":
    main()
