import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/path-tracer_amd")
import ptamd
from scenes import presets
os.chdir("/root/repo")
pt, cam = ptamd.build_scene(presets.mesh_cornell("tests/golden/cornell_mesh.glb", 1920, 1080))
film = ptamd.Film(1920, 1080)
ptamd.Render(cam, pt, film, 64, 8, count=True)
s = ptamd.RenderStats()
tot = s["steal_rounds"]
print(s)
print("closest_hit %.1f%%  shade %.1f%%  done-block %.1f%% of loop cycles (wave-leader view)" % (100.0 * s["node_trips"] / tot, 100.0 * s["tri_trips"] / tot, 100.0 * s["shade_trips"] / tot))
