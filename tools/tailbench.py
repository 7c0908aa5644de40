"""How much of a render launch is tail? One wave renders every sample of an 8x8 tile and a workgroup holds four of them,
so a launch is (W/8)(H/8)/4 workgroups that the GPU runs 1280 at a time (256 CUs x 5). This renders the mesh Cornell
at film heights that put the launch just below and just above a whole number of such rounds: if the throughput saw-tooths
with the height, the last partial round (and not the work per sample) is costing time.
    python tools/tailbench.py [spp]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "path-tracer_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import ptamd  # noqa: E402
from scenes import presets  # noqa: E402

os.chdir(ROOT)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ptamd.init(0)
stream = torch.cuda.current_stream()
ptamd.set_stream(stream.cuda_stream)
W = 1920
for H in (840, 856, 1016, 1024, 1032, 1080, 1184, 1192, 1200, 1360, 1368):
    scene = presets.mesh_cornell("tests/golden/cornell_mesh.glb", W, H)
    pt, cam = ptamd.build_scene(scene)
    film = ptamd.Film(W, H)
    ptamd.Render(cam, pt, film, 16, 8)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    ptamd.Render(cam, pt, film, spp, 8, spp_begin=16)
    b.record(stream)
    torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    blocks = (W // 8) * ((H + 7) // 8) / 4
    print("H %4d  blocks %6.0f  rounds of 1280: %.2f   %.1f ms   %.0f Msamples/s" % (H, blocks, blocks / 1280, ms, W * H * spp / ms * 1e-3), flush=True)
