"""Several GPUs behind one handle (mi355rt_scene_set_devices; SURVEY.md 8b/8e, Camera.java:129-182): a scene given a device
list renders the frame the one-GPU scene renders, bit for bit, through mi355rt_render_device, mi355rt_render_host and
through views with frames in flight.

This box has ONE GPU, so the list names it several times: every entry is a rank with its own replica of the scene's
arrays (copied device to device), its own host thread, stream and tile-compact block -- the N-rank code path of the
library as it runs on N GPUs, except that the exchange step is peer copies (RCCL refuses a device named twice).  The
RCCL calls themselves are rehearsed with a one-device list (comm of one rank, the block sent to itself)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bufs(W, H, poison=-3.0):
    import torch
    dev = torch.device("cuda", 0)
    return [torch.full((H, W, 3), poison, dtype=torch.float32, device=dev), torch.full((H, W), 7, dtype=torch.int32, device=dev),
            torch.full((H, W), -5, dtype=torch.int32, device=dev), torch.full((H, W), -5, dtype=torch.int32, device=dev),
            torch.full((H, W), poison, dtype=torch.float32, device=dev)]


def _frame(rt, b):
    return rt._abi.Frame(*[t.data_ptr() for t in b])


def _same(a, b, what):
    import torch
    for k, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32)), (what, k)


def _dragon(rt):
    cfg = rt.meshgen.STANFORD["dragon"]
    v, i = rt.meshgen.stanford_standin("dragon")
    light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
    mesh = rt.TriangleMesh((v, i), rt.ColorInformation((0.9, 0.9, 0.9)), True)
    return rt.Scene([mesh], light, cfg["orientation"], rt.PhongLightingModel(light, [mesh]), cfg["fov"]), (1920, 1080, 4)


def _mirrors(rt):
    """a mirrored blob over a mirror floor beside a plain blob: bounce chains of several bounces (two mirror objects)"""
    light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
    v, i = rt.meshgen.blob(60000, span=2.0, seed=5)
    v2 = (v * 0.5 + np.array([1.4, 0.0, 0.3], dtype=np.float32)).astype(np.float32)
    fv, fi = rt.meshgen.quad(-1.3, 4.0)
    objs = [rt.TriangleMesh((v, i), rt.ReflectionMaterial((0.9, 0.9, 1.0)), True), rt.TriangleMesh((v2, i), rt.ColorInformation((0.9, 0.5, 0.2)), True),
            rt.TriangleMesh((fv, fi), rt.ReflectionMaterial((0.8, 0.8, 0.8)), True)]
    return rt.Scene(objs, light, (0, -25, 0), rt.PhongLightingModel(light, objs), 40.0), (640, 400, 3)


@pytest.mark.parametrize("which", ["dragon", "mirrors"])
def test_a_device_list_renders_the_one_gpu_frame(rt, which):
    import torch
    scene, (W, H, spp) = _dragon(rt) if which == "dragon" else _mirrors(rt)
    cam = rt.Camera(scene, spp, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=2024)
    ref = _bufs(W, H)
    cam.render_device(_frame(rt, ref), collect_stats=True)
    torch.cuda.synchronize()
    st1 = cam.frame_stats()
    assert int((ref[2] >= 0).sum()) > W * H // 20
    assert scene.multi_info()["transport"] == "none"

    scene.setDevices([0, 0, 0])
    info = scene.multi_info()
    assert info["n_ranks"] == 3 and info["devices"] == [0, 0, 0] and info["transport"] == "peer"
    for tile in (0, 8, 32):
        got = _bufs(W, H, -4.0)
        cam.render_device(_frame(rt, got), tile_size=tile)
        torch.cuda.synchronize()
        _same(ref, got, (which, "render_device", tile))
    # counted over the ranks: the same rays, tests and hits as the one-GPU frame (box tests of the root boxes included)
    got = _bufs(W, H, -4.0)
    cam.render_device(_frame(rt, got), collect_stats=True)
    torch.cuda.synchronize()
    _same(ref, got, (which, "counted"))
    stn = cam.frame_stats()
    for k in ("primary_rays", "shadow_rays", "secondary_rays", "hits"):
        assert stn[k] == st1[k], (k, stn[k], st1[k])
    ms = scene.multi_info()["render_ms"]
    assert len(ms) == 3 and all(x > 0 for x in ms)
    # the blocking call (Camera.createImage) and its 8-bit image
    cam.createImage(outputs=("pixels", "image", "hit_id", "hit_obj", "hit_w"))
    assert np.array_equal(cam._rgb_rows.view(np.uint32), ref[0].cpu().numpy().view(np.uint32))
    assert np.array_equal(cam.hit_id.T, ref[2].cpu().numpy()) and np.array_equal(cam.hit_obj.T, ref[3].cpu().numpy())
    assert np.array_equal(cam.hit_w.T.view(np.uint32), ref[4].cpu().numpy().view(np.uint32))
    assert np.array_equal(cam.image.view(np.int32), ref[1].cpu().numpy())
    # a caller-set tile_world is refused: the library shards the frame itself
    with pytest.raises(rt._abi.Mi355rtError):
        cam.render_device(_frame(rt, got), tile_rank=0, tile_world=2)
    # frames in flight: views of the scene span the same ranks
    from ray_tracer_amd import tiles
    pipe = tiles.FramePipeline(scene, lambda sc: rt.Camera(sc, spp, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=2024), W, H, depth=2)
    assert pipe.views[0].multi_info()["n_ranks"] == 3
    for _ in range(5):
        pipe.submit()
    pipe.drain()
    for o in pipe.outs:
        _same([ref[0], ref[2], ref[3], ref[4]], o, (which, "views"))
    scene.close()   # the scene first: the views keep every rank's arrays alive
    pipe.submit()
    pipe.drain()
    _same([ref[0], ref[2], ref[3], ref[4]], pipe.outs[1], (which, "views after close"))
    pipe.close()


def test_the_rccl_calls_on_a_one_rank_list(rt, switch):
    """TRANSPORT=rccl with a one-device list: librccl.so.1 is loaded, ncclCommInitAll makes the communicator, and the frame's
    block goes through ncclSend / ncclRecv (to itself, one group) before the unpack kernel -- the library's RCCL calls,
    as far as one GPU can run them."""
    import torch
    light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
    v, i = rt.meshgen.stanford_standin("dragon", 30000)
    mesh = rt.TriangleMesh((v, i), rt.ColorInformation((0.9, 0.9, 0.9)), True)
    scene = rt.Scene([mesh], light, (0, 0, 0), rt.PhongLightingModel(light, [mesh]), 23.0)
    W, H, spp = 640, 400, 2
    cam = rt.Camera(scene, spp, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=3)
    ref = _bufs(W, H)
    cam.render_device(_frame(rt, ref))
    torch.cuda.synchronize()
    switch("TRANSPORT", "rccl")
    scene.setDevices([0])
    assert scene.multi_info()["transport"] == "rccl"
    for _ in range(3):
        got = _bufs(W, H, -4.0)
        cam.render_device(_frame(rt, got))
        torch.cuda.synchronize()
        _same(ref, got, "rccl self send")
    scene.close()


def test_device_list_errors(rt):
    light = rt.Light()
    v, i = rt.meshgen.blob(500)
    mesh = rt.TriangleMesh((v, i), rt.ColorInformation((1, 1, 1)), True)
    scene = rt.Scene([mesh], light, (0, 0, 0), rt.LightingModel(), 15.0)
    for bad in ([], [0, 99], [99, 0], list(range(17))):
        with pytest.raises(rt._abi.Mi355rtError):
            scene.setDevices(bad)
    view = scene.view()
    with pytest.raises(rt._abi.Mi355rtError):
        view.setDevices([0, 0])
    with pytest.raises(rt._abi.Mi355rtError):
        scene.setDevices([0, 0])   # views exist already
    view.close()
    scene.setDevices([-1, 0])
    with pytest.raises(rt._abi.Mi355rtError):
        scene.setDevices([0, 0])   # once per scene
    scene.close()


_ONE_RANK_RCCL = r'''
import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
import ray_tracer_amd as rt
from ray_tracer_amd import tiles

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
v, i = rt.meshgen.stanford_standin("dragon", 30000)
mesh = rt.TriangleMesh((v, i), rt.ColorInformation((0.9, 0.9, 0.9)), True)
scene = rt.Scene([mesh], light, (0, 0, 0), rt.PhongLightingModel(light, [mesh]), 23.0)
W, H, spp, tile = 648, 400, 2, 16
cam = rt.Camera(scene, spp, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=3)
ref = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev), torch.zeros((H, W), dtype=torch.int32, device=dev),
       torch.zeros((H, W), dtype=torch.int32, device=dev), torch.zeros((H, W), dtype=torch.float32, device=dev)]
cam.render_device(rt._abi.Frame(ref[0].data_ptr(), None, ref[1].data_ptr(), ref[2].data_ptr(), ref[3].data_ptr()))
torch.cuda.synchronize()
sh = tiles.TileSharder(W, H, 0, 1, tile, dev)
side = torch.cuda.Stream(device=dev)
for rep in range(3):
    got = [torch.full_like(t, 7) for t in ref]
    with torch.cuda.stream(side):
        cam.render_device(sh.blocks[0].frame(), stream=side.cuda_stream, tile_rank=0, tile_world=1, tile_size=tile, tile_compact=True)
        # what TileSharder.gather does on rank 0 of N ranks, with N = 1: the block through dist.gather on the RCCL backend
        # (asynchronous, on a side stream, into slices of one receive area), then the library's unpack kernel
        recv = torch.empty(tiles.PLANES * sh.n, dtype=torch.float32, device=dev)
        work = dist.gather(sh.blocks[0].flat, gather_list=list(recv.view(1, -1).unbind(0)), dst=0, async_op=True)
        work.wait()
        sh.unpack(recv, *got)
    side.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), rep
dist.destroy_process_group()
scene.close()
print("one-rank RCCL gather ok")
'''


def test_the_process_per_gpu_gather_over_rccl_with_one_rank():
    """tiles.TileSharder's exchange as rank 0 runs it -- dist.gather on torch.distributed's "nccl" backend (RCCL), asynchronous,
    on a side stream, into slices of the receive area, then mi355rt_tiles_unpack_device -- with a process group of ONE rank
    (all that one GPU allows: RCCL refuses two ranks on a device).  In a process of its own: the group is process-wide."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _ONE_RANK_RCCL], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    assert "one-rank RCCL gather ok" in r.stdout
