"""Raw counters of one counting render of the config-3 scene (dragon field in the closed room): python tools/c3stats.py [spp]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "path-tracer_amd")):
    sys.path.insert(0, p)
import ptamd
from scenes import dragon
os.chdir(ROOT)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
glb = os.path.join(tempfile.mkdtemp(), "f.glb")
dragon.write_dragon_field(glb)
pt, cam = ptamd.build_scene(dragon.config3_scene(glb, 1920, 1080))
film = ptamd.Film(1920, 1080)
ptamd.Render(cam, pt, film, spp, 8, count=True)
s = ptamd.RenderStats()
print(s)
print({k: v / s["rays"] for k, v in s.items()})
print(pt.Info())
