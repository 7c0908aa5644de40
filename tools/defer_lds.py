"""Development check (GPU box, repo root): LDS-resident mesh among primitives rendered from LDS (plain, with the walk cut) against
the same scene read from global memory with deferred shared walks.  python3 tools/defer_lds.py"""
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
    for nolds in (False, True):
        if nolds: os.environ["PTAMD_NO_LDS_GEOM"] = "1"
        else: os.environ.pop("PTAMD_NO_LDS_GEOM", None)
        scene = presets.mixed("tests/golden/cornell_mesh.glb", 1920, 1080, **kw)
        pt, cam = ptamd.build_scene(scene)
        film = ptamd.Film(1920, 1080)
        for var, val in ((("PTAMD_WALK_DEFER", v) for v in ("0", "20", "32", "40")) if nolds else (("PTAMD_WALK_BUDGET", v) for v in ("0", "24"))):
            os.environ[var] = val
            ptamd.Render(cam, pt, film, 64, 8)
            film.Accum()
            t0 = time.time()
            for _ in range(3): ptamd.Render(cam, pt, film, 128, 8)
            film.Accum()
            dt = (time.time() - t0) / 3
            ptamd.Render(cam, pt, film, 64, 8, count=True)
            s = ptamd.RenderStats()
            print("%-30s lds %d %s=%-3s %8.1f Msamples/s  node lanes %.2f shading lanes %.2f" % (label, pt.Info()["lds_resident"], var, val, 1920 * 1080 * 128 / dt * 1e-6,
                  s["nodes"] / max(1, 64 * s["node_trips"]), s["shade_lanes"] / max(1, 64 * s["shade_trips"])), flush=True)
            os.environ.pop(var)
