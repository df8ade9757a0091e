"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle).
CPU: the oracle still reproduces them bit for bit.  GPU: the HIP path reproduces them through the C ABI."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402

NAMES = sorted(make_golden.CASES)


def _load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(orc, name):
    g = _load(name)
    case = make_golden.CASES[name]
    v, i, sc = make_golden.build(case)
    np.testing.assert_array_equal(g["mesh_checksum"], [np.float64(v.astype(np.float64).sum()), np.float64(i.astype(np.int64).sum())])
    cam = sc.camera(case["orient"], case["fov"], 0.01, case["W"], case["H"])
    np.testing.assert_array_equal(np.array(cam.origin[:], dtype=np.float32), g["cam_origin"])
    np.testing.assert_array_equal(np.array(cam.rotation[:], dtype=np.float32), g["cam_rotation"])
    st = orc.settings(epsilon=float(g["epsilon"]), ms=case["ms"], brdf_samples=2, lighting=case["phong"], seed=99)
    fr = sc.render(cam, st, n_threads=2)
    np.testing.assert_array_equal(fr["hit_id"], g["hit_id"])
    np.testing.assert_array_equal(fr["hit_w"].view(np.uint32), g["hit_w"].view(np.uint32))
    np.testing.assert_array_equal(fr["rgb"].view(np.uint32), g["rgb"].view(np.uint32))
    tr = sc.trace(st, g["ray_o"], g["ray_d"])
    np.testing.assert_array_equal(tr["face"], g["ray_face"])
    np.testing.assert_array_equal(tr["w"].view(np.uint32), g["ray_w"].view(np.uint32))
    sc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_reproduces_golden(rt, name):
    from helpers import synthetic_brdf_tables
    g = _load(name)
    case = make_golden.CASES[name]
    v, i = rt.meshgen.blob(case["faces"], span=case["span"], stretch=case["stretch"], seed=case["seed"])
    if "brdf" in case:
        rt.BRDFMaterial.setTables(*synthetic_brdf_tables())
        ci = rt.BRDFMaterial(case["brdf"], (0.2, 0.8, 0.3))
    else:
        ci = rt.ColorInformation((0.9, 0.9, 0.9))
    light = rt.Light((0, 100, 100), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
    mesh = rt.TriangleMesh((v, i), ci, True)
    lm = rt.PhongLightingModel(light, [mesh]) if case["phong"] else rt.LightingModel()
    scene = rt.Scene([mesh], light, case["orient"], lm, case["fov"])
    assert np.float32(rt.RTStatics.EPSILON) == g["epsilon"]
    cam = rt.Camera(scene, case["ms"], 2, rt.RTStatics.NEAR_PLANE, case["W"], case["H"], rng_seed=99)
    np.testing.assert_array_equal(cam.origin, g["cam_origin"])
    np.testing.assert_array_equal(cam.rotation.ravel(), g["cam_rotation"])
    cam.createImage()
    np.testing.assert_array_equal(cam.hit_id.T, g["hit_id"])
    np.testing.assert_array_equal(cam.hit_w.T.view(np.uint32), g["hit_w"].view(np.uint32))
    denom = np.maximum(np.abs(g["rgb"]), 1e-3)
    assert (np.abs(cam._rgb_rows - g["rgb"]) / denom).max() <= 1e-3   # north star: radiance within 1e-3 relative
    tr = cam.getClosestIntersection(g["ray_o"], g["ray_d"])
    np.testing.assert_array_equal(tr["face"], g["ray_face"])
    np.testing.assert_array_equal(tr["w"].view(np.uint32), g["ray_w"].view(np.uint32))
    hit = g["ray_face"] >= 0
    np.testing.assert_array_equal(tr["point"][hit].view(np.uint32), g["ray_point"][hit].view(np.uint32))
    np.testing.assert_array_equal(tr["normal"][hit].view(np.uint32), g["ray_normal"][hit].view(np.uint32))
    scene.close()


ANALYTIC_NAMES = sorted(make_golden.ANALYTIC_CASES)


@pytest.mark.parametrize("name", ANALYTIC_NAMES)
def test_oracle_reproduces_analytic_golden(orc, name):
    g = _load(name)
    case = make_golden.ANALYTIC_CASES[name]
    _, sc = make_golden.build_analytic(case)
    assert np.float32(sc.epsilon()) == g["epsilon"]
    cam = sc.camera(case["orient"], case["fov"], 0.01, case["W"], case["H"])
    np.testing.assert_array_equal(np.array(cam.origin[:], dtype=np.float32), g["cam_origin"])
    fr = sc.render(cam, make_golden.analytic_settings(case, float(g["epsilon"])), n_threads=2)
    np.testing.assert_array_equal(fr["hit_obj"], g["hit_obj"])
    np.testing.assert_array_equal(fr["hit_w"].view(np.uint32), g["hit_w"].view(np.uint32))
    np.testing.assert_array_equal(fr["rgb"].view(np.uint32), g["rgb"].view(np.uint32))
    tr = sc.trace(make_golden.analytic_settings(case, float(g["epsilon"])), g["ray_o"], g["ray_d"])
    np.testing.assert_array_equal(tr["object"], g["ray_object"])
    np.testing.assert_array_equal(tr["w"].view(np.uint32), g["ray_w"].view(np.uint32))
    sc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ANALYTIC_NAMES)
def test_hip_reproduces_analytic_golden(rt, name):
    g = _load(name)
    case = make_golden.ANALYTIC_CASES[name]
    objs, osc = make_golden.build_analytic(case)
    osc.close()
    lt = case["light"]
    light = rt.Light(lt["origin"], lt["emission"], lt["ambient"])
    scene = rt.Scene(objs, light, case["orient"], rt.PhongLightingModel(light, objs), case["fov"])
    assert np.float32(rt.RTStatics.EPSILON) == g["epsilon"]
    cam = rt.Camera(scene, case["ms"], 1, rt.RTStatics.NEAR_PLANE, case["W"], case["H"], rng_seed=99)
    np.testing.assert_array_equal(cam.origin, g["cam_origin"])
    np.testing.assert_array_equal(cam.rotation.ravel(), g["cam_rotation"])
    cam.createImage()
    np.testing.assert_array_equal(cam.hit_id.T >= 0, g["hit_obj"] >= 0)
    np.testing.assert_array_equal(cam.hit_w.T.view(np.uint32), g["hit_w"].view(np.uint32))
    denom = np.maximum(np.abs(g["rgb"]), 1e-3)
    assert float((np.abs(cam._rgb_rows - g["rgb"]) / denom).max()) <= 1e-3      # north star: radiance within 1e-3 relative
    tr = cam.getClosestIntersection(g["ray_o"], g["ray_d"])
    np.testing.assert_array_equal(tr["object"], g["ray_object"])
    hit = g["ray_object"] >= 0
    for key in ("w", "point", "normal"):
        np.testing.assert_array_equal(tr[key][hit].view(np.uint32), g["ray_" + key][hit].view(np.uint32))
    scene.close()
