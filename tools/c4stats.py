"""Raw counters of one counting render of the config-4 scene (4K mixed scene): python tools/c4stats.py [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "path-tracer_amd")):
    sys.path.insert(0, p)
import ptamd
from scenes import presets
os.chdir(ROOT)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scene = presets.mixed("tests/golden/blob.glb", 3840, 2160, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0))
pt, cam = ptamd.build_scene(scene)
film = ptamd.Film(3840, 2160)
ptamd.Render(cam, pt, film, spp, 8, count=True)
s = ptamd.RenderStats()
print(s)
print("per 64 rays:", {k: round(64.0 * v / s["rays"], 2) for k, v in s.items()})
print(pt.Info())
