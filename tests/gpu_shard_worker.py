"""One rank of tests/test_gpu_shard_two_ranks.py: renders its share of a frame with the GPU library and joins the film sum
over gloo.  python gpu_shard_worker.py <rank> <world> <mode: spp|tile|tiles1> <port> <out_dir>"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "path-tracer_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

W, H, SPP, DEPTH, SEED = 200, 120, 12, 6, 77


def scene():
    from scenes import presets
    return presets.mixed("tests/golden/blob.glb", W, H, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0))


def main():
    rank, world, mode, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.chdir(ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ptamd
    import shard
    ptamd.init(0)  # every rank on the one GPU of the box
    pt, cam = ptamd.build_scene(scene())
    film = ptamd.Film(W, H)

    def render_rect(begin, count, rect):
        ptamd.Render(cam, pt, film, count, DEPTH, spp_begin=begin, rect=rect, seed=SEED)

    def render_tiles(begin, count, tile, phase, stride):
        ptamd.Render(cam, pt, film, count, DEPTH, spp_begin=begin, tiles=(tile, phase, stride), seed=SEED)

    n = shard.render_sharded(render_rect, W, H, SPP, mode="spp" if mode == "spp" else "tile", tile=32,
                             render_tiles=render_tiles if mode == "tiles1" else None)
    ta = torch.from_numpy(film.Accum().copy())
    ts = torch.from_numpy(film.Samples().astype(np.int32))
    shard.reduce_film(ta, ts, dst=0)
    tn = torch.tensor([n], dtype=torch.int64)
    dist.all_reduce(tn)
    if rank == 0:
        np.save(os.path.join(out_dir, "accum.npy"), ta.numpy())
        np.save(os.path.join(out_dir, "samples.npy"), ts.numpy())
        np.save(os.path.join(out_dir, "n.npy"), tn.numpy())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
