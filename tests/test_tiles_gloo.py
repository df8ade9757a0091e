"""The N > 1 path on CPU: processes (gloo), each holding the tiles it owns of one frame in the library's tile-compact
block layout; after TileSharder.gather rank 0 must hold the whole frame, bit for bit."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from ray_tracer_amd import tiles
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H, T = 150, 97, 16
g = torch.Generator().manual_seed(5)
full_rgb = torch.rand(H, W, 3, generator=g)
full_w = torch.rand(H, W, generator=g)
full_id = torch.randint(-1, 871414, (H, W), generator=g, dtype=torch.int32)
full_obj = torch.randint(-1, 3, (H, W), generator=g, dtype=torch.int32)
sh = tiles.TileSharder(W, H, rank, world, T, "cpu", n_buffers=2)
# this rank "rendered" only its own tiles, in the library's compact order, into block 1 (frames in flight use several)
tiles.tiles_pack_reference(full_rgb, full_id, full_obj, full_w, rank, world, T, sh.blocks[1])
rgb, w = torch.full_like(full_rgb, -7.0), torch.full_like(full_w, -7.0)
ids, objs = torch.full_like(full_id, -7), torch.full_like(full_obj, -7)
finish = sh.gather(rgb, ids, objs, w, buffer=1, async_op=True)
finish()
if rank == 0:
    assert torch.equal(rgb, full_rgb) and torch.equal(w, full_w) and torch.equal(ids, full_id) and torch.equal(objs, full_obj)
    print("GATHER_OK")
dist.barrier()
dist.destroy_process_group()
'''


import pytest


@pytest.mark.parametrize("world", [2, 3])
def test_gather_gloo(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]
