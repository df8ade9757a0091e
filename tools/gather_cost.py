"""Cost of TileSharder.gather's packing and unpacking kernels without the RCCL exchange (profiling helper, one GPU):
rank 0 of an N-rank run of the bench frame, dist.gather replaced by a device copy of the same size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from ray_tracer_amd import tiles

W, H = 1920, 1080
dev = torch.device("cuda", 0)
rgb = torch.rand((H, W, 3), device=dev)
hw = torch.rand((H, W), device=dev)
hid = torch.randint(0, 1 << 20, (H, W), dtype=torch.int32, device=dev)


def fake_gather(send, gather_list=None, dst=0):
    for r in gather_list:
        r.copy_(send)


for world in (2, 4, 8):
    sh = tiles.TileSharder(W, H, 0, world, tiles.DEFAULT_TILE, dev)
    real = dist.gather
    tiles.dist.gather = fake_gather
    try:
        for _ in range(3):
            sh.gather(rgb, hw, hid)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            sh.gather(rgb, hw, hid)
        torch.cuda.synchronize()
        print("world %d: pack + unpack (no exchange) %.3f ms per frame" % (world, (time.perf_counter() - t0) / n * 1e3))
    finally:
        tiles.dist.gather = real
