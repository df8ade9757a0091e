"""Image-tile sharding of one frame over the GPUs of a node.

The reference scatters pixels over `availableProcessors()` threads (Camera.java:129-143); samples
are independent given the read-only scene.  Here the frame is cut into square tiles; tile t belongs
to rank t % world (interleaved, because a statue sits mid-frame and bands would be imbalanced); the
scene is replicated on every GPU.

Each rank renders its tiles straight into ONE compact block (settings.tile_compact: the library's
resolve / shade / generate kernels write pixel (x, y) of the rank's k-th owned tile at element
k*T*T + (y%T)*T + (x%T) of four planes [rgb: 3n][hit_id: n][hit_obj: n][hit_w: n], n = slots*T*T),
so there is nothing to pack; the ONE exchange step is a gather of those blocks to rank 0, and one
library kernel there (mi355rt_tiles_unpack_device) scatters all of them back into image order.
torch.distributed is plumbing: backend "nccl" (RCCL over xGMI) on GPUs, "gloo" in the CPU tests,
where the same layout is unpacked with index tensors (tiles_unpack_reference).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _abi

DEFAULT_TILE = 16
PLANES = 6   # words per pixel of a block: r, g, b, hit_id, hit_obj, hit_w


def tile_grid(width, height, tile=DEFAULT_TILE):
    return (width + tile - 1) // tile, (height + tile - 1) // tile


def owned_tiles(width, height, rank, world, tile=DEFAULT_TILE):
    """Tile ids of `rank`, in the order libmi355rt enumerates them (t = rank, rank + world, ...)."""
    tx, ty = tile_grid(width, height, tile)
    return torch.arange(rank, tx * ty, world, dtype=torch.int64)


def max_owned(width, height, world, tile=DEFAULT_TILE):
    tx, ty = tile_grid(width, height, tile)
    return (tx * ty + world - 1) // world


def tile_pixels(width, height, tiles, tile=DEFAULT_TILE, device="cpu"):
    """[len(tiles), tile*tile] linear pixel indices (y*W + x) of each tile; -1 where the tile hangs off the image."""
    tx, _ = tile_grid(width, height, tile)
    tiles = tiles.to(device)
    ox = (tiles % tx) * tile
    oy = (tiles // tx) * tile
    d = torch.arange(tile, device=device)
    x = ox[:, None, None] + d[None, None, :]
    y = oy[:, None, None] + d[None, :, None]
    idx = y * width + x
    idx = torch.where((x < width) & (y < height), idx, torch.full_like(idx, -1))
    return idx.reshape(tiles.numel(), tile * tile)


class Block:
    """One rank's compact block: a flat f32 tensor of PLANES*n words and views of its four planes."""

    def __init__(self, flat, n):
        self.flat, self.n = flat, n
        self.rgb = flat[: 3 * n].view(n, 3)
        self.hit_id = flat[3 * n: 4 * n].view(torch.int32)
        self.hit_obj = flat[4 * n: 5 * n].view(torch.int32)
        self.hit_w = flat[5 * n: 6 * n]

    def frame(self):
        """_abi.Frame of device pointers into the block (what Camera.render_device(..., tile_compact=True) writes)."""
        return _abi.Frame(self.rgb.data_ptr(), None, self.hit_id.data_ptr(), self.hit_obj.data_ptr(), self.hit_w.data_ptr())


def tiles_pack_reference(rgb, hit_id, hit_obj, hit_w, rank, world, tile, block):
    """What the library's kernels write with tile_compact, restated with index tensors (CPU tests): this rank's tiles
    of full-size frame tensors -> its block."""
    H, W = hit_id.shape
    pix = tile_pixels(W, H, owned_tiles(W, H, rank, world, tile), tile, rgb.device)
    n_own, tt = pix.shape
    ok = (pix >= 0).reshape(-1)
    src = pix.reshape(-1)[ok]
    dst = torch.arange(n_own * tt, device=rgb.device)[ok]
    block.rgb[dst] = rgb.reshape(-1, 3)[src]
    block.hit_id[dst] = hit_id.reshape(-1)[src]
    block.hit_obj[dst] = hit_obj.reshape(-1)[src]
    block.hit_w[dst] = hit_w.reshape(-1)[src]


def tiles_unpack_reference(blocks, world, slots, tile, rgb, hit_id, hit_obj, hit_w):
    """mi355rt_tiles_unpack_device restated with index tensors (CPU tests): `world` consecutive blocks -> image order."""
    H, W = hit_id.shape
    n = slots * tile * tile
    for r in range(world):
        b = Block(blocks[r * PLANES * n: (r + 1) * PLANES * n], n)
        pix = tile_pixels(W, H, owned_tiles(W, H, r, world, tile), tile, rgb.device)
        ok = (pix >= 0).reshape(-1)
        dst = pix.reshape(-1)[ok]
        src = torch.arange(pix.numel(), device=rgb.device)[ok]
        rgb.reshape(-1, 3)[dst] = b.rgb[src]
        hit_id.reshape(-1)[dst] = b.hit_id[src]
        hit_obj.reshape(-1)[dst] = b.hit_obj[src]
        hit_w.reshape(-1)[dst] = b.hit_w[src]


class TileSharder:
    """The per-frame exchange of one rank.  `n_buffers` independent blocks let frames be in flight: frame k's block
    travels while frame k + 1 renders into another."""

    def __init__(self, width, height, rank, world, tile=DEFAULT_TILE, device="cpu", n_buffers=1, via_host=False):
        self.width, self.height, self.rank, self.world, self.tile = width, height, rank, world, tile
        self.via_host = via_host   # rehearsals only (several ranks on ONE GPU over gloo): the blocks travel through host memory
        self.device = torch.device(device)
        self.slots = max_owned(width, height, world, tile)
        self.n = self.slots * tile * tile
        self.blocks = [Block(torch.zeros(PLANES * self.n, dtype=torch.float32, device=self.device), self.n) for _ in range(n_buffers)]
        self.recv = None
        if rank == 0 and world > 1:
            self.recv = [torch.empty(world * PLANES * self.n, dtype=torch.float32, device=self.device) for _ in range(n_buffers)]
        self._lib = _abi.load() if self.device.type == "cuda" else None

    def gather(self, rgb, hit_id, hit_obj, hit_w, buffer=0, async_op=False):
        """Sends this rank's block `buffer` to rank 0 and, there, scatters all ranks' blocks into the frame tensors
        (rgb [H,W,3] f32, hit_id / hit_obj [H,W] i32, hit_w [H,W] f32; any may be None on the GPU).  world == 1: the
        frame was rendered in image order already, nothing to do.  async_op: returns a function that finishes the
        exchange (waits for the gather, then unpacks on rank 0) -- call it after queueing the next frame."""
        if self.world == 1:
            return (lambda: None) if async_op else None
        send = self.blocks[buffer].flat
        staged = None
        if self.via_host:
            send = send.cpu()   # (waits for the stream that rendered into the block)
            staged = [torch.empty_like(send) for _ in range(self.world)] if self.rank == 0 else None
            work = dist.gather(send, gather_list=staged, dst=0, async_op=True)
        elif self.rank == 0:
            recv = self.recv[buffer]
            work = dist.gather(send, gather_list=list(recv.view(self.world, -1).unbind(0)), dst=0, async_op=True)
        else:
            work = dist.gather(send, gather_list=None, dst=0, async_op=True)

        def finish():
            work.wait()   # on the GPU: the current stream waits for the collective; the host does not
            if self.rank == 0:
                if staged is not None:
                    self.recv[buffer].copy_(torch.cat(staged))
                self.unpack(self.recv[buffer], rgb, hit_id, hit_obj, hit_w)

        if async_op:
            return finish
        finish()
        return None

    def unpack(self, blocks, rgb, hit_id, hit_obj, hit_w):
        if self.device.type == "cuda":
            fr = _abi.Frame(rgb.data_ptr() if rgb is not None else None, None, hit_id.data_ptr() if hit_id is not None else None,
                            hit_obj.data_ptr() if hit_obj is not None else None, hit_w.data_ptr() if hit_w is not None else None)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            with torch.cuda.device(self.device):
                _abi.check(self._lib.mi355rt_tiles_unpack_device(C.c_void_p(blocks.data_ptr()), self.world, self.slots, self.tile, self.width,
                                                                 self.height, C.byref(fr), C.c_void_p(stream) if stream else None))
        else:
            tiles_unpack_reference(blocks, self.world, self.slots, self.tile, rgb, hit_id, hit_obj, hit_w)


class FramePipeline:
    """Frames in flight on one rank (mi355rt_scene_view).  Every launch of a frame ends on its longest rays
    with the chip draining, and with N ranks each frame ends in an exchange; a sequence of frames hides both: `depth`
    views of ONE scene (shared arrays in HBM, own scratch) on `depth` streams, round robin and without host
    synchronisation, so frame k + 1 renders while frame k drains, and with N > 1 the gather of frame k's compact block
    runs under it (rank 0 unpacks it when its slot comes round again, or at drain()).

    make_camera(scene_or_view) -> host.Camera; slot b's outputs are `outs[b]` = [rgb, hit_id, hit_obj, hit_w] device
    tensors in image order (on rank 0, or with world == 1)."""

    def __init__(self, scene, make_camera, width, height, rank=0, world=1, depth=2, tile=DEFAULT_TILE, device="cuda", tile_world=None,
                 via_host=False, streams=None):
        self.rank, self.world, self.depth, self.tile = rank, world, depth, tile
        self.tile_world = world if tile_world is None else tile_world   # (tile_world != world: one rank's share of a larger run, for measurements)
        self.device = torch.device(device)
        self.views = [scene.view() for _ in range(depth)]
        self.cams = [make_camera(v) for v in self.views]
        # HIP deals streams out over a few hardware queues (4 per process by default); two streams that share a queue run
        # their frames one after the other, so a caller that builds many pipelines hands the same `depth` streams to each
        self.streams = list(streams) if streams is not None else [torch.cuda.Stream(device=self.device) for _ in range(depth)]
        if len(self.streams) != depth:
            raise ValueError("FramePipeline: %d streams for depth %d" % (len(self.streams), depth))
        self.sharder = TileSharder(width, height, rank, world, tile, self.device, n_buffers=depth, via_host=via_host)
        full = rank == 0 or world == 1
        self.outs = [[torch.zeros((height, width, 3), dtype=torch.float32, device=self.device),
                      torch.zeros((height, width), dtype=torch.int32, device=self.device),
                      torch.zeros((height, width), dtype=torch.int32, device=self.device),
                      torch.zeros((height, width), dtype=torch.float32, device=self.device)] if full else [None] * 4 for _ in range(depth)]
        self.frames = [_abi.Frame(o[0].data_ptr(), None, o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr()) if world == 1
                       else self.sharder.blocks[b].frame() for b, o in enumerate(self.outs)]
        self._pending = [None] * depth
        self._k = 0

    def submit(self):
        """Queues the next frame; returns its slot.  The slot's previous frame is finished first (stream-side)."""
        b = self._k % self.depth
        self._k += 1
        with torch.cuda.stream(self.streams[b]):
            if self._pending[b]:
                self._pending[b]()
            self.cams[b].render_device(self.frames[b], stream=self.streams[b].cuda_stream, tile_rank=self.rank, tile_world=self.tile_world,
                                       tile_size=self.tile, tile_compact=self.world > 1)
            self._pending[b] = self.sharder.gather(*self.outs[b], buffer=b, async_op=True) if self.world > 1 else None
        return b

    def drain(self):
        """Finishes every frame in flight (their exchanges included) and waits for the device."""
        for b in range(self.depth):
            if self._pending[b]:
                with torch.cuda.stream(self.streams[b]):
                    self._pending[b]()
                self._pending[b] = None
        torch.cuda.synchronize(self.device)

    def close(self):
        self.drain()
        for v in self.views:
            v.close()
        self.views = []
