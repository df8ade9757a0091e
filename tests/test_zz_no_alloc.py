"""mi355rt_render_device_noalloc (include/mi355rt.h): the frame rendered out of the library's static device scratch
equals mi355rt_render_device's frame bit for bit -- on two calls in a row, the second with a smaller input, so the
second call runs over whatever the first left in the shared array -- and frames beyond the array are refused.
Collected last (test_zz_*): the round's end runs pytest -x."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _buffers(W, H, poison):
    import torch
    dev = torch.device("cuda", 0)
    rgb = torch.full((H, W, 3), float(poison), dtype=torch.float32, device=dev)
    argb = torch.full((H, W), 7, dtype=torch.int32, device=dev)
    hid = torch.full((H, W), -5, dtype=torch.int32, device=dev)
    hobj = torch.full((H, W), -5, dtype=torch.int32, device=dev)
    hw = torch.full((H, W), float(poison), dtype=torch.float32, device=dev)
    return (rgb, argb, hid, hobj, hw)


def _frame(rt, bufs):
    return rt._abi.Frame(*[b.data_ptr() for b in bufs])


def _scenes(rt):
    """A mesh-only Phong scene (fused and unfused shadow pipelines by frame size) and a mesh over a mirror floor
    (bounce queues: the 204-byte-per-slot layout, and the bounce loop's read-back)."""
    light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
    v, i = rt.meshgen.stanford_standin("dragon", 30000)
    blob = rt.TriangleMesh((v, i), rt.ColorInformation((0.9, 0.9, 0.9)), True)
    plain = rt.Scene([blob], light, (0, 0, 0), rt.PhongLightingModel(light, [blob]), 23.0)
    lo, hi = v.min(axis=0), v.max(axis=0)
    y = lo[1] - 0.02 * (hi[1] - lo[1])
    ex = 1.5 * (hi - lo)
    fv = np.array([[lo[0] - ex[0], y, lo[2] - ex[2]], [hi[0] + ex[0], y, lo[2] - ex[2]], [hi[0] + ex[0], y, hi[2] + ex[2]],
                   [lo[0] - ex[0], y, hi[2] + ex[2]]], dtype=np.float32)
    fi = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)
    floor = rt.TriangleMesh((fv, fi), rt.ReflectionMaterial((0.8, 0.8, 0.8)), True)
    blob2 = rt.TriangleMesh((v, i), rt.ColorInformation((0.9, 0.6, 0.3)), True)
    mirror = rt.Scene([blob2, floor], light, (0, -20, 0), rt.PhongLightingModel(light, [blob2, floor]), 23.0)
    return [("plain", plain), ("mirror", mirror)]


def test_noalloc_frames_equal_render_device_frames(rt):
    import torch
    for name, scene in _scenes(rt):
        # two calls in a row on the shared array, the second smaller; then a larger one again
        for (W, H, spp) in ((640, 480, 2), (200, 120, 1), (320, 200, 3)):
            cam = rt.Camera(scene, spp, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=11)
            want = _buffers(W, H, -1.0)
            got = _buffers(W, H, -2.0)
            cam.render_device(_frame(rt, want))
            torch.cuda.synchronize()
            cam.render_device_noalloc(_frame(rt, got))
            torch.cuda.synchronize()
            # ... and the property the entry point exists for: a further call leaves the device's free memory as it was
            free_before = torch.cuda.mem_get_info()[0]
            cam.render_device_noalloc(_frame(rt, got))
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info()[0] == free_before, (name, W, H, spp)
            for k, (a, b) in enumerate(zip(want, got)):
                # bit-exact, NaN-safe: compare the words
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (name, W, H, spp, k)
            assert int((want[2] >= 0).sum()) > W * H // 20   # the frame shows the object
            if name == "mirror":
                assert int((want[3] == 1).sum()) > 0 and int((want[3] == 0).sum()) > 0   # the mirror floor and the mesh on it
        scene.close()


def test_noalloc_refuses_what_does_not_fit(rt):
    import torch
    light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
    v, i = rt.meshgen.stanford_standin("bunny", 2000)
    mesh = rt.TriangleMesh((v, i), rt.ColorInformation((1.0, 1.0, 1.0)), True)
    scene = rt.Scene([mesh], light, (0, 0, 0), rt.PhongLightingModel(light, [mesh]), 15.0)
    W, H = 1920, 1080
    cam = rt.Camera(scene, 4, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=1)
    bufs = _buffers(W, H, -3.0)
    with pytest.raises(rt._abi.Mi355rtError) as ei:
        cam.render_device_noalloc(_frame(rt, bufs))
    assert ei.value.code == rt._abi.E_LIMIT and "static scratch" in ei.value.message
    torch.cuda.synchronize()
    assert bool((bufs[0] == -3.0).all()) and bool((bufs[2] == -5).all())   # nothing was written
    # the largest frame the header promises does fit: 1280 x 960 at 1 spp
    cam = rt.Camera(scene, 1, 1, rt.RTStatics.NEAR_PLANE, 1280, 960, rng_seed=1)
    want, got = _buffers(1280, 960, -1.0), _buffers(1280, 960, -2.0)
    cam.render_device(_frame(rt, want))
    cam.render_device_noalloc(_frame(rt, got))
    torch.cuda.synchronize()
    for a, b in zip(want, got):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    # counting / timing would create events and read back: refused
    with pytest.raises(rt._abi.Mi355rtError) as ei:
        cam.render_device_noalloc(_frame(rt, got), collect_stats=True)
    assert ei.value.code == rt._abi.E_ARG
    scene.close()
