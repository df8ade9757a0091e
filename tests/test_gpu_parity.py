"""HIP path vs oracle, through the C ABI, on the same seeded inputs (bit-exact ids and w; radiance
within the north star's 1e-3 relative)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _blob_pair(rt, faces, name="bunny", **kw):
    from helpers import Pair
    cfg = rt.meshgen.STANFORD[name]
    v, i = rt.meshgen.stanford_standin(name, faces)
    ci = kw.pop("colorInfo", None) or rt.ColorInformation((1.0, 1.0, 1.0))
    return Pair([(v, i, None, ci)], orientation=cfg["orientation"], fov=cfg["fov"], **kw)


def _check(pair, W, H, ms=1, brdf_samples=1, seed=0, rtol=1e-3):
    from helpers import compare_frames
    cam, ocam, ost = pair.cameras(W, H, ms=ms, brdf_samples=brdf_samples, seed=seed)
    np.testing.assert_array_equal(np.array(ocam.origin[:], dtype=np.float32), cam.origin)
    np.testing.assert_array_equal(np.array(ocam.rotation[:], dtype=np.float32).reshape(3, 3), cam.rotation)
    cam.createImage()
    o = pair.oscene.render(ocam, ost, n_threads=8)
    ids_equal, w_equal, max_rel, n_diff = compare_frames(cam._rgb_rows, cam.hit_id.T, cam.hit_w.T, o, rtol=rtol, g_obj=cam.hit_obj.T)
    assert ids_equal, "%d pixels differ in hit id" % n_diff
    assert w_equal
    assert max_rel <= rtol, max_rel
    hit_frac = float((o["hit_id"] >= 0).mean())
    return hit_frac


def test_device_arithmetic_is_correctly_rounded(rt):
    """f32 divide / sqrt and the f64 normalise scale must equal IEEE (= Java) results bit for bit."""
    import ctypes as C
    rs = np.random.RandomState(3)
    n = 1 << 18
    a = np.abs(rs.standard_normal(n).astype(np.float32)) * np.float32(10.0) ** rs.randint(-20, 20, n).astype(np.float32)
    a[:4] = [1e-40, 3e-39, 1.0, 2.0]   # subnormals too
    b = rs.standard_normal(n).astype(np.float32)
    b[b == 0] = 1
    outs = [np.empty(n, dtype=np.float32) for _ in range(4)]
    lib = rt._abi.load()
    P = lambda x: x.ctypes.data_as(rt._abi.c_float_p)
    rt._abi.check(lib.mi355rt_debug_arith(n, P(a), P(b), *[P(x) for x in outs]))
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(outs[0].view(np.uint32), (a / b).view(np.uint32))
        np.testing.assert_array_equal(outs[1].view(np.uint32), np.sqrt(a).view(np.uint32))
        ref = (1.0 / np.sqrt(a.astype(np.float64))).astype(np.float32)
        np.testing.assert_array_equal(outs[2].view(np.uint32), ref.view(np.uint32))
        refp = np.power(b.astype(np.float64), 20.0).astype(np.float32)
        ok = np.isclose(outs[3], refp, rtol=1e-6, atol=0) | (outs[3] == refp)
        assert ok.all()


def test_bunny_small_phong(rt):
    pair = _blob_pair(rt, 5000)
    frac = _check(pair, 96, 96)
    assert 0.05 < frac < 0.95
    pair.close()


def test_bunny_config1_phong(rt):
    """BASELINE config 1: ~69k faces, 512x512, 1 spp, primary + shadow."""
    pair = _blob_pair(rt, None)
    _check(pair, 512, 512)
    pair.close()


def test_multisample_jitter(rt):
    pair = _blob_pair(rt, 20000)
    _check(pair, 160, 90, ms=4, seed=1234)
    pair.close()


def test_plain_lighting_model(rt):
    pair = _blob_pair(rt, 3000, phong=False)
    _check(pair, 64, 64)
    pair.close()


def test_no_kdtree(rt):
    pair = _blob_pair(rt, 600, use_kdtree=False)
    _check(pair, 48, 48)
    pair.close()


def test_kdtree_flag_is_per_mesh(rt):
    """computeKDTree belongs to each TriangleMesh (TriangleMesh.java:23,89): a mesh built without the hierarchy next
    to one built with it.  Both hold every face twice, so equal distances are everywhere, and the two forms break
    the ties differently (one leaf: the first face; a tree: the later leaf) -- each object must keep its own rule."""
    from helpers import Pair
    v, i = rt.meshgen.blob(1500, span=0.15, seed=3)
    i2 = np.concatenate([i, i[::-1]], axis=0)
    shift = np.array([0.09, 0.02, -0.03], dtype=np.float32)
    for flags in ((True, False), (False, True)):
        pair = Pair([(v, i2, None, rt.ColorInformation((1.0, 0.8, 0.6)), flags[0]),
                     ((v + shift).astype(np.float32), i2, None, rt.ColorInformation((0.6, 0.8, 1.0)), flags[1])], fov=20.0)
        info = [pair.scene.object_info(k) for k in range(2)]
        assert [bool(x["n_nodes"]) for x in info] == list(flags)
        assert [pair.oscene.object_info(k)["n_leaves"] > 1 for k in range(2)] == list(flags)
        frac = _check(pair, 128, 96, ms=2, seed=5)
        assert frac > 0.1
        pair.close()


def test_brdf_material(rt):
    from helpers import synthetic_brdf_tables
    pair = _blob_pair(rt, 8000, name="buddha", colorInfo=rt.BRDFMaterial(3, (0.2, 0.8, 0.3)),
                      brdf_tables=synthetic_brdf_tables())
    _check(pair, 96, 128, brdf_samples=2)
    pair.close()


def test_reflection_two_objects(rt):
    from helpers import Pair
    v, i = rt.meshgen.blob(6000, span=0.2, seed=11)
    fv, fi = rt.meshgen.quad(-0.12, 0.6)
    pair = Pair([(v, i, None, rt.ReflectionMaterial((0.9, 0.9, 1.0))), (fv, fi, None, rt.ColorInformation((0.8, 0.3, 0.2)))],
                orientation=(0, 20, 0), fov=20.0)
    _check(pair, 128, 96, ms=2, seed=5)
    pair.close()


def test_large_scale_epsilon_branch(rt):
    """span >= 10 selects EPSILON = 1e-3 (Scene.java:36)."""
    from helpers import Pair
    v, i = rt.meshgen.blob(4000, span=400.0, seed=4)
    pair = Pair([(v, i, None, rt.ColorInformation((1, 1, 1)))], fov=10.0)
    assert np.float32(pair.eps) == np.float32(1e-3)
    _check(pair, 80, 80)
    pair.close()


def test_triangle_soup(rt):
    from helpers import Pair
    v, i = rt.meshgen.soup(20000, span=0.3, tri_size=0.03, seed=9)
    pair = Pair([(v, i, None, rt.ColorInformation((0.9, 0.9, 0.9)))], fov=23.0)
    _check(pair, 128, 128)
    pair.close()


@pytest.mark.parametrize("light_origin", [(0.0, 0.0, 0.0), (0.1, 0.12, -0.05), (0.0, 0.4, 0.4), (0.02, 0.151, 0.0)])
def test_exact_shadows_with_the_light_inside_the_scene(rt, light_origin):
    """With the light inside (or just outside) the scene's box the any-hit shortcut cannot be proven and the shadow
    rays run the reference-exact form: per object the closest accepted hit, which occludes only if it is nearer to the
    ray's origin than to the light (RTStatics.java:628-633).  The walker looks no farther than half the way to the
    light plus the origin offset (shadow_reach2); a soup of triangles puts faces before and beyond that distance on
    most shadow rays, and every pixel's radiance must still match the oracle."""
    from helpers import Pair
    v, i = rt.meshgen.soup(20000, span=0.3, tri_size=0.03, seed=12)
    fv, fi = rt.meshgen.quad(-0.16, 0.5)
    light = dict(origin=light_origin, emission=(0.3, 0.3, 0.3), ambient=(0.5, 0.5, 0.9))
    pair = Pair([(v, i, None, rt.ColorInformation((0.9, 0.9, 0.9))), (fv, fi, None, rt.ColorInformation((0.8, 0.3, 0.2)))],
                fov=23.0, light=light)
    hit_frac = _check(pair, 112, 112, ms=2, seed=8)
    assert hit_frac > 0.1
    pair.close()


@pytest.mark.parametrize("light_origin", [(0.0, 260.0, 0.0), (120.0, 120.0, 150.0)])
def test_exact_shadows_large_epsilon(rt, light_origin):
    """The same at EPSILON = 1e-3 (span 400), where the shadow ray's origin offset N * EPSILON is not negligible
    (synthesised normals are not unit length): a light between two blobs, and one beside them."""
    from helpers import Pair
    v, i = rt.meshgen.blob(4000, span=400.0, seed=6)
    v2, i2 = rt.meshgen.blob(2000, span=150.0, seed=2)
    v2 = (v2 + np.float32([40.0, 330.0, 0.0])).astype(np.float32)
    light = dict(origin=light_origin, emission=(0.3, 0.3, 0.3), ambient=(0.5, 0.5, 0.9))
    pair = Pair([(v, i, None, rt.ColorInformation((1, 1, 1))), (v2, i2, None, rt.ColorInformation((0.8, 0.3, 0.2)))],
                fov=14.0, light=light)
    assert np.float32(pair.eps) == np.float32(1e-3)
    _check(pair, 96, 96, ms=2, seed=3)
    pair.close()


def test_trace_rays_arbitrary(rt):
    """Rays from inside and behind the mesh (hits behind the origin are legal, Triangle.java has no t>0 test),
    axis-aligned directions with +0.0 / -0.0 components."""
    from helpers import Pair
    v, i = rt.meshgen.blob(3000, span=0.2, seed=21)
    pair = Pair([(v, i, None, rt.ColorInformation((1, 1, 1)))])
    cam, ocam, ost = pair.cameras(8, 8)
    rs = np.random.RandomState(5)
    n = 4000
    o = rs.uniform(-0.25, 0.25, size=(n, 3)).astype(np.float32)
    d = rs.standard_normal((n, 3)).astype(np.float32)
    d[:50, 0] = 0.0
    d[50:100, 1] = -0.0
    d[100:150, 0] = 0.0
    d[100:150, 1] = 0.0
    o[150:200] = 0.0
    g = cam.getClosestIntersection(o, d)
    r = pair.oscene.trace(ost, o, d)
    np.testing.assert_array_equal(g["face"], r["face"])
    np.testing.assert_array_equal(g["object"], r["object"])
    np.testing.assert_array_equal(g["w"].view(np.uint32), r["w"].view(np.uint32))
    hit = r["face"] >= 0
    assert hit.sum() > 200
    np.testing.assert_array_equal(g["point"][hit].view(np.uint32), r["point"][hit].view(np.uint32))
    np.testing.assert_array_equal(g["normal"][hit].view(np.uint32), r["normal"][hit].view(np.uint32))
    pair.close()


def test_sixty_four_objects_and_one_too_many(rt):
    """MAX_OBJECTS meshes in one scene (the walker's object index fields are 7 bits wide: layout.h) -- an 8 x 8 grid of small
    blobs, every one its own hierarchy, colour and root box, one of them a mirror -- against the oracle; a 65th is refused
    with MI355RT_E_LIMIT before anything is built."""
    from helpers import Pair
    objs = []
    for k in range(64):
        v, i = rt.meshgen.blob(300 + 20 * (k % 5), span=0.02, seed=100 + k)
        v = v + np.array([0.03 * (k % 8 - 3.5), 0.03 * (k // 8 - 3.5), 0.005 * (k % 3)], dtype=np.float32)
        mat = rt.ReflectionMaterial((0.9, 0.9, 1.0)) if k == 27 else rt.ColorInformation((0.3 + 0.01 * k, 1.0 - 0.01 * k, 0.5))
        objs.append((v.astype(np.float32), i, None, mat))
    pair = Pair(objs, orientation=(0, 10, 0), fov=25.0)
    hit_frac = _check(pair, 160, 120, ms=2, seed=9)
    assert hit_frac > 0.2
    cam, ocam, ost = pair.cameras(160, 120, ms=1)
    cam.createImage()
    assert len(np.unique(cam.hit_obj[cam.hit_obj >= 0])) >= 60   # (nearly) every object is seen
    pair.close()
    light = rt.Light()
    meshes = [rt.TriangleMesh(rt.meshgen.blob(60, span=0.01, seed=k), rt.ColorInformation((1, 1, 1)), True) for k in range(65)]
    with pytest.raises(rt._abi.Mi355rtError) as e:
        rt.Scene(meshes, light, (0, 0, 0), rt.LightingModel(), 15.0)
    assert "64" in str(e.value)


@pytest.mark.parametrize("W,H,ms", [(1, 1, 1), (1, 9, 3), (9, 1, 2), (3, 2, 5), (17, 1, 1), (8, 8, 33)])
def test_smallest_and_thinnest_images(rt, W, H, ms):
    """Images of one pixel, one row, one column (every 8 x 8 pixel group ragged) and a sample count far above the usual."""
    pair = _blob_pair(rt, 2000)
    _check(pair, W, H, ms=ms, seed=4)
    pair.close()
