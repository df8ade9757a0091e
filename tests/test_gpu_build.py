"""The GPU hierarchy builder (gpu_build.hip) against the host builder (scene_build.cpp), which the
parity tests pin to the oracle's pointer tree: the same node records, leaf slots and normals, byte for byte."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(rt, meshes, builder, max_depth=None, max_children=None):
    sd, sc = rt.RTStatics.MAX_DEPTH, rt.RTStatics.MAX_CHILDREN
    try:
        rt.debug_switch("BUILDER", "host" if builder == "host" else None)   # (the library reads the environment only once)
        if max_depth is not None:
            rt.RTStatics.MAX_DEPTH = max_depth
        if max_children is not None:
            rt.RTStatics.MAX_CHILDREN = max_children
        light = rt.Light()
        tms = [rt.TriangleMesh((v, i), rt.ColorInformation((1, 1, 1)), True) for v, i in meshes]
        return rt.Scene(tms, light, (0, 0, 0), rt.LightingModel(), 15.0)
    finally:
        rt.RTStatics.MAX_DEPTH, rt.RTStatics.MAX_CHILDREN = sd, sc
        rt.debug_switch("BUILDER", None)


def _same(rt, meshes, **kw):
    host = _scene(rt, meshes, "host", **kw)
    gpu = _scene(rt, meshes, "gpu", **kw)
    dh, on_gpu_h = host.digest()
    dg, on_gpu_g = gpu.digest()
    assert not on_gpu_h and on_gpu_g
    for k in range(len(meshes)):
        ih, ig = host.object_info(k), gpu.object_info(k)
        for key in ("n_nodes", "n_leaves", "n_leaf_slots", "max_depth"):
            assert ih[key] == ig[key], (k, key, ih[key], ig[key])
    assert dh == dg
    info = gpu.object_info(0)
    host.close()
    gpu.close()
    return info


def test_blob_sizes(rt):
    for faces, seed in ((12, 1), (300, 2), (5000, 3), (60000, 4)):
        v, i = rt.meshgen.blob(faces, span=0.3, seed=seed)
        info = _same(rt, [(v, i)])
        assert info["n_nodes"] >= 1


def test_soup_with_heavy_duplication(rt):
    v, i = rt.meshgen.soup(20000, span=0.3, tri_size=0.05, seed=9)
    info = _same(rt, [(v, i)])
    assert info["n_leaf_slots"] > 3 * 20000          # straddlers are duplicated as KDNode.java:58-69 duplicates them


def test_limits_and_several_objects(rt):
    v1, i1 = rt.meshgen.blob(4000, span=0.2, seed=5)
    v2, i2 = rt.meshgen.blob(900, span=0.1, seed=6)
    fv, fi = rt.meshgen.quad(-0.12, 0.6)               # two faces: the root does not split, host path inside a mixed scene
    _same(rt, [(v1, i1), (fv, fi), (v2 + np.float32([0.3, 0, 0]), i2)])
    assert _same(rt, [(v1, i1)], max_depth=3)["max_depth"] == 3
    _same(rt, [(v1, i1)], max_depth=1)
    _same(rt, [(v1, i1)], max_children=40)
    _same(rt, [(v1, i1)], max_children=0, max_depth=12)


def test_degenerate_and_unreferenced_vertices(rt):
    v, i = rt.meshgen.blob(2000, span=0.2, seed=7)
    v = np.concatenate([v, np.float32([[5, 5, 5], [-3, 0, 0]])])      # unreferenced vertices widen the root box (KDTree.java:35-44)
    i = np.concatenate([i, i[:50], np.int32([[0, 0, 0], [1, 1, 2]])])   # repeated and degenerate faces
    _same(rt, [(v, i)])


def test_dragon_class_full_size(rt):
    """BASELINE configs[1]: 871,414 faces; the GPU build gives the hierarchy the bench and the full-size parity test use."""
    v, i = rt.meshgen.stanford_standin("dragon")
    info = _same(rt, [(v, i)])
    assert info["n_leaf_slots"] == 6023242 and info["n_nodes"] == 798804 and info["max_depth"] == 20


def test_thai_class_ten_million_faces(rt):
    """BASELINE configs[3]'s mesh size: 10,000,000 faces (leaves of tens of faces under the reference's depth limit) -- the
    device's hierarchy, leaf groups, records and normals against the host builder's, byte for byte."""
    v, i = rt.meshgen.blob(10000000, span=400.0, stretch=(0.59, 1.0, 0.52), seed=4)
    info = _same(rt, [(v, i)])
    assert info["max_depth"] == 20 and info["n_leaf_slots"] > 10000000


def test_normals_on_the_device_are_the_oracles(rt, orc):
    """TriangleMesh.computeNormals on the device (face-order sums per vertex) against the oracle's loop, bit for bit:
    a blob, with repeated faces, faces with two equal corners (NaN normals that spread, TriangleMesh.java:96-126),
    an unreferenced vertex (0 / 0) and a vertex of very high valence."""
    v, i = rt.meshgen.blob(300000, span=0.3, seed=21)
    fan = np.stack([np.zeros(3000, dtype=np.int32), np.arange(1, 3001, dtype=np.int32), np.arange(2, 3002, dtype=np.int32)], axis=1)   # vertex 0: +3000 faces
    i = np.ascontiguousarray(np.concatenate([i, i[:500], fan, np.int32([[7, 7, 9], [11, 12, 12]])]))
    v = np.ascontiguousarray(np.concatenate([v, np.float32([[9, 9, 9]])]))
    got = rt.compute_normals(v, i)                      # more than 50,000 faces: the device path
    exp = np.empty_like(v)
    orc.lib().orc_compute_normals(orc.P(v), v.shape[0], orc.PI(i), i.shape[0], orc.P(exp))
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    assert np.isnan(got[-1]).all() and np.isnan(got[7]).all() and np.isfinite(got[100]).all()
