"""Frame times of an interactive session (GPU box, repo root): passes of 1, 8 and 64 samples per pixel at 1080p on the mesh
Cornell and on the room with the reference's dragon model.  python3 tools/frame_rates.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "path-tracer_amd")):
    sys.path.insert(0, p)
import ptamd
from scenes import presets
os.chdir(ROOT)
ptamd.init(0)
W, H = 1920, 1080
import tempfile
from scenes import dragon
field = os.path.join(tempfile.mkdtemp(prefix="frame_rates_"), "dragon_field.glb")
dragon.write_dragon_field(field)
for label, scene in (("mesh Cornell", presets.mesh_cornell("tests/golden/cornell_mesh.glb", W, H)),
                     ("room + spheres + 100 000-triangle dragon", presets.mixed("tests/golden/chinese_dragon.glb", W, H)),
                     ("glass and mirror spheres in the room", presets.glass_mirror(W, H)),
                     ("room + 7680-triangle blob", presets.mixed("tests/golden/blob.glb", W, H, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0))),
                     ("1 M-triangle dragon field in the room", dragon.config3_scene(field, W, H))):
    pt, cam = ptamd.build_scene(scene)
    film = ptamd.Film(W, H)
    for spp in [int(v) for v in os.environ.get('FRAME_SPP', '1 8 64').split()]:
        ptamd.Render(cam, pt, film, spp, 8)
        film.Samples()
        n = 20
        t0 = time.perf_counter()
        for i in range(n):
            ptamd.Render(cam, pt, film, spp, 8, spp_begin=(i + 1) * spp)
        film.Samples()
        dt = (time.perf_counter() - t0) / n
        print("%-45s %3d spp per pass: %7.2f ms per pass, %7.1f Msamples/s" % (label, spp, dt * 1e3, W * H * spp / dt * 1e-6), flush=True)
