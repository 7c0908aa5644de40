"""Development check on the GPU box: the deferred shared walk (PTAMD_WALK_DEFER=<lanes>) must leave the film of the mixed
scene bit for bit as it is, and its rate at several thresholds.  python3 tools/defer_check.py [width height spp]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
gpu = importlib.import_module("path-tracer_amd.ptamd")
from scenes import presets

w, h, spp = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1920, 1080, 64)
if os.environ.get("DEFER_SCENE") == "dragon":  # the room, two spheres and the reference's dragon model (100 000 triangles), as main.cpp:229-311 places its car
    scene = presets.mixed("tests/golden/chinese_dragon.glb", w, h)
else:
    scene = presets.mixed("tests/golden/blob.glb", w, h, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0))
pt, cam = gpu.build_scene(scene)
print("lds_resident", pt.Info()["lds_resident"], "triangles", pt.Info()["triangles"], "wide nodes", pt.Info()["wide_nodes"])
ref = None
for lanes in os.environ.get("DEFER_LANES", "0 1 16 24 32 40 48 64").split():
    os.environ["PTAMD_WALK_DEFER"] = lanes
    f = gpu.Film(w, h)
    gpu.Render(cam, pt, f, spp, 8, seed=5, count=True)
    st = gpu.RenderStats()
    a = f.Accum()
    if ref is None:
        ref, ref_st = a, st
    same = np.array_equal(ref.view(np.uint32), a.view(np.uint32))
    best = 1e9
    for rep in range(3):
        g = gpu.Film(w, h)
        t0 = time.perf_counter()
        gpu.Render(cam, pt, g, spp, 8, seed=5)
        g.Samples()
        best = min(best, time.perf_counter() - t0)
    print("defer %2s  same film %s  rays %d nodes %d tris %d  node_trips %d tri_trips %d ray_trips %d steal %d shade_lanes %d shade_trips %d   %.1f Msamples/s" % (
        lanes, same, st["rays"], st["nodes"], st["tris"], st["node_trips"], st["tri_trips"], st["ray_trips"], st.get("steal_rounds", 0), st["shade_lanes"], st["shade_trips"],
        w * h * spp / best / 1e6), flush=True)
os.environ.pop("PTAMD_WALK_DEFER")
pt2, cam2 = gpu.build_scene(scene)
gpu.Render(cam2, pt2, gpu.Film(w, h), spp, 8, seed=5)
print("the library's own measurement chooses", pt2.Info()["walk_cut"])
