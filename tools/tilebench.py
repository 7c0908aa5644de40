"""One rank's share of an 8-way interleaved 32x32 tile split of a 4K film (config 4): a launch per tile against
ptamd_render_tiles. MI355X, 16 spp: 2277 ms vs 11.5 ms (the whole film in one launch: 82 ms)."""
import sys, time
sys.path.insert(0, 'path-tracer_amd'); sys.path.insert(0, '.')
import torch, ptamd, shard
from scenes import presets
ptamd.init(0); ptamd.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 3840, 2160
scene = presets.mixed("tests/golden/blob.glb", w, h, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0))
pt, cam = ptamd.build_scene(scene)
film = ptamd.Film(w, h)
spp = 16
def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
def per_tile():
    for r in shard.tile_rects(0, 8, w, h, 32): ptamd.Render(cam, pt, film, spp, 8, rect=r)
def one():
    ptamd.Render(cam, pt, film, spp, 8, tiles=(32, 0, 8))
def full():
    ptamd.Render(cam, pt, film, spp, 8)
for name, fn in (("one launch per 32x32 tile", per_tile), ("one launch for the rank's tiles", one), ("whole film (8 ranks' work)", full)):
    fn(); print(f"{name:34s} {min(timed(fn) for _ in range(2)):9.2f} ms")
