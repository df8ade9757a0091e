"""The measured-BRDF path on the reference's own CUReT tables (resources/table.txt, resources/abrdf.dat), which do not
travel to the GPU box: tests/golden/curet_tables.npz holds them parsed with BRDFMaterial.parseBRDFFile's rule, and
curet_*.npz hold oracle-rendered frames on them (tests/golden/make_curet_golden.py, build container only).
CPU: the fixture equals a fresh parse of the files (where they exist) and the oracle reproduces the frames bit for bit.
GPU: the HIP path reproduces them through the C ABI -- hit objects / faces and w bit-exact, radiance within 1e-3
relative (north star) -- with materials 15 and 16 as RayTracer.java:162-163 uses them."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_curet_golden as mc  # noqa: E402

RES = "/root/reference/resources"


def _g(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


def test_tables_fixture_shape_and_content():
    ang, wts = mc.load_tables()
    assert ang.shape == (205, 4) and wts.shape == (61, 205) and ang.dtype == np.float32 and wts.dtype == np.float32
    assert np.isfinite(ang).all() and np.isfinite(wts).all()
    assert ang[1, 1] == 0.0                      # row 2 of table.txt has phi_v = 0
    assert 0.0 <= ang[:, 0].min() and ang[:, 0].max() < np.pi / 2 and ang[:, 2].max() < np.pi / 2
    # the illumination directions sit on a few rings (20 distinct theta_i among 205 rows), unlike uniform random angles
    assert len(np.unique(np.round(ang[:, 2], 3))) == 20
    assert wts.min() < 0.0 < wts.max()           # measured weights: some are slightly negative, one is 2.59


@pytest.mark.skipif(not os.path.isdir(RES), reason="reference data files not present (GPU box)")
def test_tables_fixture_equals_a_fresh_parse_of_the_reference_files(rt):
    ang, wts = mc.load_tables()
    a = np.array(rt.BRDFMaterial.parseBRDFFile(os.path.join(RES, "table.txt")), dtype=np.float32)
    w = np.array(rt.BRDFMaterial.parseBRDFFile(os.path.join(RES, "abrdf.dat")), dtype=np.float32)
    np.testing.assert_array_equal(a.view(np.uint32), ang.view(np.uint32))
    np.testing.assert_array_equal(w.view(np.uint32), wts.view(np.uint32))


def test_oracle_reproduces_the_curet_frames(orc):
    tables = mc.load_tables()
    g = _g("curet_whitted_brdf")
    _, sc = mc.whitted_scene(tables)
    cam = sc.camera(mc.WHITTED["orient"], mc.WHITTED["fov"], 0.01, mc.WHITTED["W"], mc.WHITTED["H"])
    np.testing.assert_array_equal(np.array(cam.origin[:], dtype=np.float32), g["cam_origin"])
    fr = sc.render(cam, mc.whitted_settings(float(g["epsilon"])), n_threads=4)
    np.testing.assert_array_equal(fr["hit_obj"], g["hit_obj"])
    np.testing.assert_array_equal(fr["hit_w"].view(np.uint32), g["hit_w"].view(np.uint32))
    np.testing.assert_array_equal(fr["rgb"].view(np.uint32), g["rgb"].view(np.uint32))
    sc.close()
    g = _g("curet_blob_brdf")
    v, i, cfg, sc = mc.blob_scene(tables)
    np.testing.assert_array_equal(g["mesh_checksum"], [np.float64(v.astype(np.float64).sum()), np.float64(i.astype(np.int64).sum())])
    cam = sc.camera(cfg["orientation"], cfg["fov"], 0.01, mc.BLOB["W"], mc.BLOB["H"])
    fr = sc.render(cam, mc.blob_settings(float(g["epsilon"])), n_threads=4)
    np.testing.assert_array_equal(fr["hit_id"], g["hit_id"])
    np.testing.assert_array_equal(fr["rgb"].view(np.uint32), g["rgb"].view(np.uint32))
    for row in g["luminocity_probes"]:
        m, d, p, n, want = int(row[0]), row[1:4], row[4:7], row[7:10], row[10]
        got = orc.lib().orc_brdf_luminocity(sc.h, m, 2, orc.vec(*d), orc.vec(*p), orc.vec(*n), orc.vec(0, 100, 100))
        assert np.float32(got) == np.float32(want)
    lum = g["luminocity_probes"][:, 10]
    assert lum.min() > -0.1 and lum.max() < 20.0 and (lum > 0).mean() > 0.5   # measured weights go slightly negative, and so does the result
    sc.close()


@pytest.mark.gpu
def test_hip_reproduces_the_whitted_brdf_scene_on_the_real_tables(rt):
    """Scenes.WHITTED_SCENE_BRDF: BRDFMaterial(15, green) and BRDFMaterial(16, cyan) on the spheres over the checkerboard
    plane, plain LightingModel (RayTracer.java:114-116,155-167)."""
    from helpers import LIGHT, whitted_objects
    g = _g("curet_whitted_brdf")
    rt.BRDFMaterial.setTables(*mc.load_tables())
    objs = whitted_objects(rt, True)
    light = rt.Light(LIGHT["origin"], LIGHT["emission"], LIGHT["ambient"])
    scene = rt.Scene(objs, light, mc.WHITTED["orient"], rt.LightingModel(), mc.WHITTED["fov"])
    assert np.float32(rt.RTStatics.EPSILON) == g["epsilon"]
    cam = rt.Camera(scene, mc.WHITTED["ms"], mc.WHITTED["brdf_samples"], rt.RTStatics.NEAR_PLANE, mc.WHITTED["W"], mc.WHITTED["H"],
                    rng_seed=mc.WHITTED["seed"])
    np.testing.assert_array_equal(cam.origin, g["cam_origin"])
    cam.createImage()
    np.testing.assert_array_equal(cam.hit_obj.T, g["hit_obj"])
    np.testing.assert_array_equal(cam.hit_w.T.view(np.uint32), g["hit_w"].view(np.uint32))
    denom = np.maximum(np.abs(g["rgb"]), 1e-3)
    assert float((np.abs(cam._rgb_rows - g["rgb"]) / denom).max()) <= 1e-3
    assert set(np.unique(g["hit_obj"])) == {-1, 0, 1, 2}
    scene.close()


@pytest.mark.gpu
def test_hip_reproduces_the_brdf_blob_on_the_real_tables(rt):
    from helpers import LIGHT
    g = _g("curet_blob_brdf")
    rt.BRDFMaterial.setTables(*mc.load_tables())
    cfg = rt.meshgen.STANFORD["buddha"]
    v, i = rt.meshgen.stanford_standin("buddha", mc.BLOB["faces"])
    light = rt.Light(LIGHT["origin"], LIGHT["emission"], LIGHT["ambient"])
    mesh = rt.TriangleMesh((v, i), rt.BRDFMaterial(mc.BLOB["brdf_id"], (0.0, 1.0, 0.0)), True)
    scene = rt.Scene([mesh], light, cfg["orientation"], rt.PhongLightingModel(light, [mesh]), cfg["fov"])
    cam = rt.Camera(scene, mc.BLOB["ms"], mc.BLOB["brdf_samples"], rt.RTStatics.NEAR_PLANE, mc.BLOB["W"], mc.BLOB["H"], rng_seed=mc.BLOB["seed"])
    np.testing.assert_array_equal(cam.origin, g["cam_origin"])
    cam.createImage()
    np.testing.assert_array_equal(cam.hit_id.T, g["hit_id"])
    np.testing.assert_array_equal(cam.hit_w.T.view(np.uint32), g["hit_w"].view(np.uint32))
    denom = np.maximum(np.abs(g["rgb"]), 1e-3)
    assert float((np.abs(cam._rgb_rows - g["rgb"]) / denom).max()) <= 1e-3
    scene.close()
