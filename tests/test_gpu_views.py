"""mi355rt_scene_view: frames in flight on one scene.  Views share the scene's arrays in
HBM and own their frame scratch; frames rendered through them on different streams must be the frames the scene
itself renders, bit for bit, in any interleaving."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


def _scene(rt, mirror=False):
    v, i = rt.meshgen.blob(60000, span=2.0, seed=5)
    light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
    objs = [rt.TriangleMesh((v, i), rt.ReflectionMaterial((0.9, 0.9, 1.0)) if mirror else rt.ColorInformation((0.9, 0.8, 0.7)), True)]
    if mirror:
        fv, fi = rt.meshgen.quad(-1.3, 3.0)
        objs.append(rt.TriangleMesh((fv, fi), rt.ColorInformation((0.8, 0.3, 0.2)), True))
    return rt.Scene(objs, light, (0, -25, 0), rt.PhongLightingModel(light, objs), 40.0)


@pytest.mark.parametrize("mirror", [False, True])
def test_views_render_the_scenes_frame_with_frames_in_flight(rt, mirror):
    import torch
    from ray_tracer_amd import tiles
    scene = _scene(rt, mirror)
    W, H, spp = 640, 400, 3
    make = lambda sc: rt.Camera(sc, spp, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=11)
    cam = make(scene)
    ref = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"), torch.zeros((H, W), dtype=torch.int32, device="cuda"),
           torch.zeros((H, W), dtype=torch.int32, device="cuda"), torch.zeros((H, W), dtype=torch.float32, device="cuda")]
    cam.render_device(rt._abi.Frame(ref[0].data_ptr(), None, ref[1].data_ptr(), ref[2].data_ptr(), ref[3].data_ptr()))
    torch.cuda.synchronize()
    assert int((ref[1] >= 0).sum()) > W * H // 20
    before = scene.device_bytes()
    pipe = tiles.FramePipeline(scene, make, W, H, depth=3)
    assert all(v.device_bytes() < before // 4 for v in pipe.views)      # a view holds scratch, not a second copy of the scene
    for _ in range(7):
        pipe.submit()
    pipe.drain()
    for o in pipe.outs:
        assert torch.equal(o[1], ref[1]) and torch.equal(o[2], ref[2]) and torch.equal(o[3], ref[3])
        assert torch.equal(o[0].view(torch.int32), ref[0].view(torch.int32))
    # the scene may go first: its arrays live until the last view is destroyed
    scene.close()
    for o in pipe.outs:
        o[1].fill_(-7)
    for _ in range(3):
        pipe.submit()
    pipe.drain()
    for o in pipe.outs:
        assert torch.equal(o[1], ref[1])
    pipe.close()


def test_view_errors(rt):
    scene = _scene(rt)
    lib = rt._abi.load()
    out = C.c_void_p()
    assert lib.mi355rt_scene_view(None, C.byref(out)) == rt._abi.E_ARG
    assert lib.mi355rt_scene_view(scene._handle, None) == rt._abi.E_ARG
    view = scene.view()
    view2 = view.view()     # a view of a view is a view of the scene
    import numpy as np
    ang = np.zeros((4, 4), dtype=np.float32)
    wts = np.zeros((1, 4), dtype=np.float32)
    for h in (view._handle, scene._handle):   # tables are set before views exist
        rc = lib.mi355rt_brdf_tables(h, ang.ctypes.data_as(C.POINTER(C.c_float)), 4, wts.ctypes.data_as(C.POINTER(C.c_float)), 1)
        assert rc == rt._abi.E_ARG
    view2.close()
    view.close()
    scene.close()
