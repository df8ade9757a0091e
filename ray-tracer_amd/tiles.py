"""Image-tile sharding of one frame over the GPUs of a node.

The reference scatters pixels over `availableProcessors()` threads (Camera.java:129-143); samples
are independent given the read-only scene.  Here the frame is cut into square tiles; tile t belongs
to rank t % world (interleaved, because a statue sits mid-frame and bands would be imbalanced); the
scene is replicated on every GPU.  Each rank renders its tiles into a full-size device frame
(libmi355rt writes only owned tiles), packs them into one compact [tiles, T*T, 5] f32 tensor
(r, g, b, w, id-bits) and the ONE exchange step is a gather of those tensors to rank 0, which
scatters them back into frame order.  torch.distributed is plumbing: backend "nccl" (RCCL over xGMI)
on GPUs, "gloo" in the CPU tests.
"""
import torch
import torch.distributed as dist

DEFAULT_TILE = 16
CHANNELS = 5   # r, g, b, hit_w, hit_id (bit pattern)


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


def pack_tiles(rgb, hit_w, hit_id, pix, n_slots):
    """Gathers this rank's tiles out of its full-size frame into [n_slots, T*T, CHANNELS] (zero padded)."""
    n, tt = pix.shape
    out = torch.zeros((n_slots, tt, CHANNELS), dtype=torch.float32, device=rgb.device)
    safe = pix.clamp(min=0).reshape(-1)
    out[:n, :, 0:3] = rgb.reshape(-1, 3)[safe].reshape(n, tt, 3)
    out[:n, :, 3] = hit_w.reshape(-1)[safe].reshape(n, tt)
    out[:n, :, 4] = hit_id.reshape(-1)[safe].view(torch.float32).reshape(n, tt)
    return out


def unpack_tiles(packed, pix, rgb, hit_w, hit_id):
    """Scatters one rank's packed tiles into the full frame tensors (in place)."""
    n, _ = pix.shape
    valid = pix.reshape(-1) >= 0
    dst = pix.reshape(-1)[valid]
    src = packed[:n].reshape(-1, CHANNELS)[valid]
    rgb.reshape(-1, 3)[dst] = src[:, 0:3]
    hit_w.reshape(-1)[dst] = src[:, 3]
    hit_id.reshape(-1)[dst] = src[:, 4].contiguous().view(torch.int32)


class TileSharder:
    """Pre-computes every index tensor once; gather() is the per-frame exchange step and launches a fixed, small
    number of kernels (no boolean masks, so no device-to-host synchronisation inside the timed step)."""

    def __init__(self, width, height, rank, world, tile=DEFAULT_TILE, device="cpu"):
        self.width, self.height, self.rank, self.world, self.tile = width, height, rank, world, tile
        self.device = torch.device(device)
        self.n_slots = max_owned(width, height, world, tile)
        tt = tile * tile
        self.my_pix = tile_pixels(width, height, owned_tiles(width, height, rank, world, tile), tile, self.device)
        self.n_mine = self.my_pix.shape[0]
        self.my_safe = self.my_pix.clamp(min=0).reshape(-1)
        self.send = torch.zeros((self.n_slots, tt, CHANNELS), dtype=torch.float32, device=self.device)
        if rank == 0 and world > 1:
            self.recv_all = torch.empty((world, self.n_slots, tt, CHANNELS), dtype=torch.float32, device=self.device)
            self.recv = [self.recv_all[r] for r in range(world)]
            src, dst = [], []
            for r in range(1, world):   # rank 0's own tiles are already in place
                pix = tile_pixels(width, height, owned_tiles(width, height, r, world, tile), tile, self.device).reshape(-1)
                keep = torch.nonzero(pix >= 0).squeeze(1)
                src.append(keep + r * self.n_slots * tt)
                dst.append(pix[keep])
            self.src = torch.cat(src)
            self.dst = torch.cat(dst)

    def gather(self, rgb, hit_w, hit_id):
        """rgb [H,W,3] f32, hit_w [H,W] f32, hit_id [H,W] i32: this rank's frame (owned tiles valid).
        On return rank 0's tensors hold the whole frame."""
        if self.world == 1:
            return
        n, tt = self.n_mine, self.tile * self.tile
        self.send[:n, :, 0:3] = rgb.reshape(-1, 3).index_select(0, self.my_safe).reshape(n, tt, 3)
        self.send[:n, :, 3] = hit_w.reshape(-1).index_select(0, self.my_safe).reshape(n, tt)
        self.send[:n, :, 4] = hit_id.reshape(-1).index_select(0, self.my_safe).view(torch.float32).reshape(n, tt)
        if self.rank == 0:
            dist.gather(self.send, gather_list=self.recv, dst=0)
            got = self.recv_all.reshape(-1, CHANNELS).index_select(0, self.src)
            rgb.reshape(-1, 3).index_copy_(0, self.dst, got[:, 0:3])
            hit_w.reshape(-1).index_copy_(0, self.dst, got[:, 3])
            hit_id.reshape(-1).index_copy_(0, self.dst, got[:, 4].contiguous().view(torch.int32))
        else:
            dist.gather(self.send, gather_list=None, dst=0)
