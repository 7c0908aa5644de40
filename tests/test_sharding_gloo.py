"""CPU tests of the N>1 path: two processes over gloo, each rendering its share (the CPU oracle stands in for
the GPU renderer: same counter-based random stream, same film semantics), one reduce, compared with a
single-process render."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import shard
from oracle import pyoracle as po
from scenes import presets

W, H, SPP, DEPTH = 48, 40, 6, 4


def _worker(rank, world, mode, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = po.Scene(presets.sphere_cornell(W, H))
    accum = np.zeros((H, W, 3), np.float32)
    samples = np.zeros((H, W), np.uint32)

    def render_rect(begin, count, rect):
        S.render(count, DEPTH, spp_begin=begin, rect=rect, threads=2, accum=accum, samples=samples)

    n = shard.render_sharded(render_rect, W, H, SPP, mode=mode, tile=16)
    ta, ts = torch.from_numpy(accum), torch.from_numpy(samples.astype(np.int32))
    shard.reduce_film(ta, ts, dst=0)
    tn = torch.tensor([n], dtype=torch.int64)
    dist.all_reduce(tn)
    if rank == 0:
        np.save(os.path.join(out_dir, "accum.npy"), ta.numpy())
        np.save(os.path.join(out_dir, "samples.npy"), ts.numpy())
        np.save(os.path.join(out_dir, "n.npy"), tn.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,port", [("spp", 29531), ("tile", 29532)])
def test_two_rank_render_equals_single_process(mode, port, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, mode, port, str(tmp_path)), nprocs=world, join=True)
    accum = np.load(tmp_path / "accum.npy")
    samples = np.load(tmp_path / "samples.npy")
    assert int(np.load(tmp_path / "n.npy")[0]) == W * H * SPP  # every (pixel, sample) rendered exactly once
    ref_a, ref_s = po.Scene(presets.sphere_cornell(W, H)).render(SPP, DEPTH, threads=2)
    assert np.array_equal(samples, ref_s.astype(np.int32))
    if mode == "tile":
        assert np.array_equal(accum, ref_a)  # disjoint pixels: the sum only adds zeros
    else:
        # the per-sample values are identical; only the order of the float additions differs
        assert np.allclose(accum, ref_a, rtol=1e-5, atol=1e-6)


def test_slices_partition_the_work():
    for world in (1, 2, 3, 4, 8):
        for total in (0, 1, 7, 1024, 4097):
            parts = [shard.spp_slice(r, world, total) for r in range(world)]
            assert sum(c for _, c in parts) == total
            pos = 0
            for b, c in parts:
                assert b == pos
                pos += c
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
        seen = np.zeros((1080 // 8, 1920 // 8), np.int32)
        for r in range(world):
            for x0, y0, x1, y1 in shard.tile_rects(r, world, 1920 // 8, 1080 // 8, tile=32):
                seen[y0:y1, x0:x1] += 1
        assert seen.min() == 1 and seen.max() == 1


def test_tile_mode_with_a_single_launch_callback():
    """render_tiles replaces the per-tile loop by one call per rank carrying (tile size, rank, world); the work
    accounting is the same, and a rank that owns no tile makes no call."""
    calls = []
    total = 0
    for rank in range(3):
        total += shard.render_sharded(None, 100, 40, 5, mode="tile", rank=rank, world=3, tile=32,
                                      render_tiles=lambda b, c, t, ph, st: calls.append((b, c, t, ph, st)))
    assert total == 100 * 40 * 5
    assert calls == [(0, 5, 32, 0, 3), (0, 5, 32, 1, 3), (0, 5, 32, 2, 3)]
    calls.clear()
    assert shard.render_sharded(None, 30, 30, 2, mode="tile", rank=1, world=2, tile=32, render_tiles=lambda *a: calls.append(a)) == 0
    assert calls == []
