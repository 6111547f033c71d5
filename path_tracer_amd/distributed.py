"""Multi-GPU orchestration: one process per GPU, interleaved destination tiles, one RCCL exchange.

The reference splits a render over N threads, each adding full frames into a private buffer that is
summed under a mutex at the end (main.cpp:436-458, 642-654).  Here rank r of N renders only the
32x8-pixel destination tiles whose number is congruent to r modulo N (pt_set_partition) and leaves
every other pixel of its FP64 accumulator untouched.  Tile numbers run along the tile rows, each
row starting one column further right than the row above (number ty*tiles_x + c sits at column
(c + ty) % tiles_x), so that a rank's tiles do not line up in columns of the picture.  Tiles are disjoint, so the one
exchange step is an all-gather of the owned tiles (each rank sends W*H*3/N doubles; `Exchange`);
the older form, an all-reduce (sum) of the zero-padded W*H*3 accumulators, is kept
(`all_reduce_accumulator`) and gives the same bits, since x + 0.0 == x.  Either way the frame is
bit-identical to a single-GPU render whatever N is.
"""
import numpy as np

TILE_W, TILE_H = 32, 8
TILE_PIXELS = TILE_W * TILE_H


def tile_owner(width, height, world, tile_w=TILE_W, tile_h=TILE_H):
    """(H, W) int array: the rank that renders each destination pixel."""
    tiles_x = (width + tile_w - 1) // tile_w
    ty = np.arange(height)[:, None] // tile_h
    tx = np.arange(width)[None, :] // tile_w
    return ((ty * tiles_x + (tx - ty) % tiles_x) % world).astype(np.int32)


def owned_mask(width, height, rank, world, tile_w=TILE_W, tile_h=TILE_H):
    return tile_owner(width, height, world, tile_w, tile_h) == rank


def tiles_per_rank(width, height, world, tile_w=TILE_W, tile_h=TILE_H):
    total = -(-width // tile_w) * -(-height // tile_h)
    return -(-total // world)


def _tile_slots(width, height, tile, tile_w, tile_h):
    """Destination pixel index of each of the 256 pixels of tile `tile` (-1 outside the image)."""
    tiles_x = -(-width // tile_w)
    ty = tile // tiles_x
    tx = (tile % tiles_x + ty) % tiles_x
    px = tx * tile_w + np.arange(tile_w * tile_h) % tile_w
    py = ty * tile_h + np.arange(tile_w * tile_h) // tile_w
    return np.where((px < width) & (py < height), py * width + px, -1)


def pack_tiles_host(frame, rank, world, tile_w=TILE_W, tile_h=TILE_H):
    """Host mirror of pt_tiles_pack: (H, W, 3) frame -> (tiles_per_rank, 256, 3) owned tiles of
    `rank`, zero outside the image and beyond the owned count."""
    H, W = frame.shape[0], frame.shape[1]
    T = tiles_per_rank(W, H, world, tile_w, tile_h)
    total = -(-W // tile_w) * -(-H // tile_h)
    flat = frame.reshape(-1, 3)
    out = np.zeros((T, tile_w * tile_h, 3), dtype=frame.dtype)
    for q in range(T):
        tile = q * world + rank
        if tile >= total:
            break
        slots = _tile_slots(W, H, tile, tile_w, tile_h)
        ok = slots >= 0
        out[q, ok] = flat[slots[ok]]
    return out


def unpack_tiles_host(gathered, width, height, world, tile_w=TILE_W, tile_h=TILE_H):
    """Host mirror of pt_tiles_unpack: (world, tiles_per_rank, 256, 3) -> (H, W, 3)."""
    total = -(-width // tile_w) * -(-height // tile_h)
    flat = np.zeros((width * height, 3), dtype=gathered.dtype)
    for tile in range(total):
        slots = _tile_slots(width, height, tile, tile_w, tile_h)
        ok = slots >= 0
        flat[slots[ok]] = gathered[tile % world, tile // world][ok]
    return flat.reshape(height, width, 3)


def all_reduce_accumulator(accum, group=None):
    """Sum the per-rank accumulators in place (torch tensor on the rank's device; RCCL over xGMI
    with backend "nccl", gloo on CPU).  No-op without an initialised process group."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if accum.is_cuda and dist.get_backend(group) == "gloo":
            # rehearsal on one GPU shared by several ranks: stage through host memory
            host = accum.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            accum.copy_(host)
        else:
            dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
    return accum


class Exchange:
    """All-gather of the owned tiles.  Holds the packed and gathered device buffers of one renderer
    (W*H*3/N and W*H*3 doubles, rounded up to whole tiles).  `stream_ptr` passed to run() must be
    torch's current stream on the device: the collective is ordered after and before the work queued
    there.  `always=True` runs the collective even for a single rank (tests)."""

    def __init__(self, renderer, world, device, always=False):
        import torch

        self.world = world
        self.always = always
        self.tiles = renderer.tiles_per_rank(world)
        n = self.tiles * TILE_PIXELS * 3
        self.packed = torch.zeros(n, dtype=torch.float64, device=device)
        self.gathered = torch.zeros(world * n, dtype=torch.float64, device=device)

    def run(self, renderer, accum, stream_ptr=0, group=None):
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return accum
        if dist.get_world_size(group) != self.world:
            raise ValueError("Exchange was sized for another world size")
        if self.world == 1 and not self.always:
            return accum
        renderer.tiles_pack(accum.data_ptr(), self.packed.data_ptr(), self.tiles, stream_ptr)
        if accum.is_cuda and dist.get_backend(group) == "gloo":
            # rehearsal on one GPU shared by several ranks: stage through host memory
            src = self.packed.cpu()
            dst = self.gathered.cpu()
            dist.all_gather_into_tensor(dst, src, group=group)
            self.gathered.copy_(dst)
        else:
            dist.all_gather_into_tensor(self.gathered, self.packed, group=group)
        renderer.tiles_unpack(self.world, self.tiles, self.gathered.data_ptr(), accum.data_ptr(), stream_ptr)
        return accum


def render_distributed(renderer, first_iteration, n_iterations, accum, rank, world, stream_ptr=0, group=None,
                       exchange=None):
    """Render this rank's tiles into the zero-initialised device tensor `accum` (float64, W*H*3)
    and complete the frame on every rank: through `exchange` (all-gather of owned tiles) when given,
    else by summing the accumulators."""
    renderer.set_partition(rank, world, TILE_W, TILE_H)
    renderer.render_device(first_iteration, n_iterations, accum.data_ptr(), stream_ptr)
    if exchange is not None:
        return exchange.run(renderer, accum, stream_ptr, group)
    return all_reduce_accumulator(accum, group)
