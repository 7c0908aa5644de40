"""Walk cut on scenes where a room of boxes and two spheres stands in front of the LDS-resident mesh:
python tools/budget_mixed.py   (GPU box, repo root). Prints Msamples/s without the cut, with it, and with what the library measures."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "path-tracer_amd")):
    sys.path.insert(0, p)
import ptamd
from scenes import presets
os.chdir(ROOT)
ptamd.init(0)
for label, kw in (("small mesh among primitives", dict(mesh_pos=(2.0, 0.1, 1.5), mesh_rot=(0.0, 25.0, 0.0), mesh_scale=(0.3, 0.3, 0.3))),
                  ("mesh filling the room", dict(mesh_pos=(0.0, 0.0, 0.0), mesh_rot=(0.0, 0.0, 0.0), mesh_scale=(1.0, 1.0, 1.0)))):
    scene = presets.mixed("tests/golden/cornell_mesh.glb", 1920, 1080, **kw)
    pt, cam = ptamd.build_scene(scene)
    assert pt.Info()["lds_resident"] == 1
    film = ptamd.Film(1920, 1080)
    for budget in ("0", "24", None):
        if budget is None: os.environ.pop("PTAMD_WALK_BUDGET", None)
        else: os.environ["PTAMD_WALK_BUDGET"] = budget
        ptamd.Render(cam, pt, film, 64, 8)
        film.Accum()
        t0 = time.time()
        for _ in range(3): ptamd.Render(cam, pt, film, 128, 8)
        film.Accum()
        dt = (time.time() - t0) / 3
        ptamd.Render(cam, pt, film, 64, 8, count=True)
        s = ptamd.RenderStats()
        print("%-30s budget %-7s %8.1f Msamples/s  node lanes %.2f shading lanes %.2f" % (label, budget or "measured %d" % pt.Info()["walk_cut"], 1920 * 1080 * 128 / dt * 1e-6,
              s["nodes"] / max(1, 64 * s["node_trips"]), s["shade_lanes"] / max(1, 64 * s["shade_trips"])))
