"""Multi-GPU work split for one render: one process per GPU, `torch.distributed` (RCCL on ROCm, gloo on CPU).

Every (pixel, sample index) pair is independent and the random stream is keyed by exactly that pair, so a
render can be cut either way and recombined with a single collective on the additive film:

  * spp-slice  : rank r renders sample indices [begin, begin+count) of the whole frame into its own film;
                 the accumulation buffers are summed (one reduce of W*H*3 floats, 24.9 MB at 1080p);
  * tile-slice : rank r renders the 32x32 tiles t with t % world == r (interleaved, so expensive regions are
                 spread over ranks); the films are disjoint, and the same sum gathers them.

The reference has no counterpart (it is a single-process thread pool, main.cpp:30-47); the row-band split of
RayTraceParallel is what tile-slice generalises.
"""
import torch
import torch.distributed as dist


def spp_slice(rank, world, spp_total):
    """Contiguous sample-index range of `rank`: sizes differ by at most one, the union is [0, spp_total)."""
    base, extra = divmod(spp_total, world)
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


def tile_rects(rank, world, width, height, tile=32):
    """Rectangles (x0, y0, x1, y1) of the interleaved tiles owned by `rank` (row-major tile index % world)."""
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    out = []
    for t in range(rank, tx * ty, world):
        x0, y0 = (t % tx) * tile, (t // tx) * tile
        out.append((x0, y0, min(x0 + tile, width), min(y0 + tile, height)))
    return out


def reduce_film(accum, samples=None, dst=0, all_ranks=False):
    """Sums the per-rank accumulation buffers (and optionally the per-pixel sample counts) onto `dst`
    (or onto every rank). One collective per tensor; tensors are reduced in place."""
    if not dist.is_initialized():  # a single process without a process group; a group of one still runs the collective
        return
    for t in (accum, samples):
        if t is None:
            continue
        if t.is_cuda and dist.get_backend() == "gloo":
            # gloo has no reduce on device tensors (two ranks sharing one GPU in tests): sum on the host, copy back
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            if all_ranks or dist.get_rank() == dst:
                t.copy_(h)
            continue
        if all_ranks:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:
            dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)


def render_sharded(render_rect, width, height, spp_total, mode="spp", rank=None, world=None, tile=32, render_tiles=None):
    """Drives one rank's share. render_rect(spp_begin, spp_count, rect_or_None) must add the samples to this
    rank's film. Returns the number of (pixel, sample) pairs this rank rendered. render_tiles(spp_begin,
    spp_count, tile, phase, stride), when given, renders the rank's interleaved tiles in one launch
    (ptamd.Render(..., tiles=...)) instead of one launch per tile."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    if mode == "spp":
        begin, count = spp_slice(rank, world, spp_total)
        if count:
            render_rect(begin, count, None)
        return width * height * count
    if mode == "tile":
        rects = tile_rects(rank, world, width, height, tile)
        if render_tiles is not None:
            if rects:
                render_tiles(0, spp_total, tile, rank, world)
        else:
            for r in rects:
                render_rect(0, spp_total, r)
        return sum((r[2] - r[0]) * (r[3] - r[1]) for r in rects) * spp_total
    raise ValueError("mode must be 'spp' or 'tile'")
