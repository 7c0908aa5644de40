"""Wave-level trip counters of scenes that are one LDS-resident mesh, with and without the walk cut (GPU box, repo root):
python tools/cut_stats.py. Per 64 samples: searches, node trips, triangle trips, shading trips of a wave."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "path-tracer_amd")):
    sys.path.insert(0, p)
import ptamd
from scenes import gltf_io, presets
os.chdir(ROOT)
ptamd.init(0)
tmp = tempfile.mkdtemp()
scenes = [("mesh cornell", presets.mesh_cornell("tests/golden/cornell_mesh.glb", 1920, 1080))]
for label, nlat, nlon, scale, pos in (("sphere seen from outside", 8, 12, (1.5, 1.5, 1.5), (2.78, 2.7, 2.0)),
                                      ("camera inside the sphere", 8, 12, (14.0, 14.0, 14.0), (2.78, 2.73, -8.0)),
                                      ("camera inside, 96 triangles", 6, 8, (14.0, 14.0, 14.0), (2.78, 2.73, -8.0))):
    prim = gltf_io.bumpy_sphere_primitive(nlat, nlon, bump=0.25, freq=3)
    path = os.path.join(tmp, "s%d_%d.glb" % (nlat, nlon))
    gltf_io.write_glb(path, [prim], [gltf_io.gltf_material("m", presets.material(albedo=(0.8, 0.7, 0.6), emission_colour=(1.0, 1.0, 1.0), emission_strength=0.3))])
    scenes.append((label, {"film": (1920, 1080), "camera": presets.default_camera(), "background": (0.2, 0.2, 0.2),
                           "objects": [{"type": "mesh", "path": path, "pos": pos, "rot": (0.0, 0.0, 0.0), "scale": scale}]}))
for label, scene in scenes:
    pt, cam = ptamd.build_scene(scene)
    film = ptamd.Film(1920, 1080)
    for cut in ("0", "24"):
        os.environ["PTAMD_WALK_BUDGET"] = cut
        ptamd.Render(cam, pt, film, 64, 8, count=True)
        s = ptamd.RenderStats()
        ptamd.Render(cam, pt, film, 64, 8); film.Accum()
        t0 = time.time()
        for _ in range(3): ptamd.Render(cam, pt, film, 128, 8)
        film.Accum()
        dt = (time.time() - t0) / 3
        k = 64.0 / s["samples"]
        print("%-28s cut %-3s %9.1f Msamples/s | per 64 samples: searches %.2f node trips %.1f tri trips %.1f shade trips %.2f | lanes: node %.2f tri %.2f shade %.2f" % (
            label, cut, 1920 * 1080 * 128 / dt * 1e-6, s["ray_trips"] * k, s["node_trips"] * k, s["tri_trips"] * k, s["shade_trips"] * k,
            s["nodes"] / max(1, 64 * s["node_trips"]), s["tris"] / max(1, 64 * s["tri_trips"]), s["shade_lanes"] / max(1, 64 * s["shade_trips"])))
