"""Walk cut on other scenes that are one LDS-resident mesh (GPU box, repo root): python tools/budget_lone.py
Rates without the cut, with it, and with what the library measures for itself.
A bumpy sphere of ~200 triangles seen from outside under the constant background (most camera rays miss, paths are short),
and the same mesh turned inside out around the camera (every ray hits, eight bounces)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "path-tracer_amd")):
    sys.path.insert(0, p)
import numpy as np
import ptamd
from scenes import gltf_io, presets
os.chdir(ROOT)
ptamd.init(0)
tmp = tempfile.mkdtemp()
for label, nlat, nlon, scale, pos in (("sphere seen from outside", 8, 12, (1.5, 1.5, 1.5), (2.78, 2.7, 2.0)),
                                      ("camera inside the sphere", 8, 12, (14.0, 14.0, 14.0), (2.78, 2.73, -8.0)),
                                      ("camera inside, 96 triangles", 6, 8, (14.0, 14.0, 14.0), (2.78, 2.73, -8.0))):
    prim = gltf_io.bumpy_sphere_primitive(nlat, nlon, bump=0.25, freq=3)
    path = os.path.join(tmp, "s%d_%d.glb" % (nlat, nlon))
    gltf_io.write_glb(path, [prim], [gltf_io.gltf_material("m", presets.material(albedo=(0.8, 0.7, 0.6), emission_colour=(1.0, 1.0, 1.0), emission_strength=0.3))])
    scene = {"film": (1920, 1080), "camera": presets.default_camera(), "background": (0.2, 0.2, 0.2),
             "objects": [{"type": "mesh", "path": path, "pos": pos, "rot": (0.0, 0.0, 0.0), "scale": scale}]}
    pt, cam = ptamd.build_scene(scene)
    info = pt.Info()
    if not info["lds_resident"]:
        print(label, "not LDS-resident", info["triangles"]); continue
    film = ptamd.Film(1920, 1080)
    for budget in ("0", "24", None):
        os.environ.pop("PTAMD_WALK_BUDGET", None)
        if budget is not None: os.environ["PTAMD_WALK_BUDGET"] = budget
        ptamd.Render(cam, pt, film, 64, 8, count=True)
        s = ptamd.RenderStats()
        t0 = time.time()
        for _ in range(3): ptamd.Render(cam, pt, film, 128, 8)
        film.Accum()
        dt = (time.time() - t0) / 3
        print("%-28s %4d triangles cut %-8s %8.1f Msamples/s  rays/sample %.2f node lanes %.2f shading lanes %.2f" % (label, info["triangles"], budget or ("measured: %d" % pt.Info()["walk_cut"]),
              1920 * 1080 * 128 / dt * 1e-6, s["rays"] / s["samples"], s["nodes"] / max(1, 64 * s["node_trips"]), s["shade_lanes"] / max(1, 64 * s["shade_trips"])))
