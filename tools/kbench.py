"""Kernel micro-benchmark for development: times ptamd_render on the preset scenes (HIP events around each
launch, median of a few launches). Usage: python tools/kbench.py [--spp 128] [--reps 5] [--scenes a,b]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "path-tracer_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import ptamd
from scenes import presets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--scenes", default="mesh_cornell,sphere_cornell,glass_mirror")
    args = ap.parse_args()
    os.chdir(ROOT)
    torch.cuda.set_device(0)
    ptamd.init(0)
    ptamd.set_stream(torch.cuda.current_stream().cuda_stream)
    table = {
        "mesh_cornell": (presets.mesh_cornell("tests/golden/cornell_mesh.glb", 1920, 1080), 8),
        "sphere_cornell": (presets.sphere_cornell(1920, 1080), 5),
        "box_cornell": (presets.box_cornell(1920, 1080), 8),
        "glass_mirror": (presets.glass_mirror(1920, 1080), 32),
        "blob_mixed": (presets.mixed("tests/golden/blob.glb", 1920, 1080, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0)), 8),
    }
    if "dragon" in args.scenes.split(","):  # the reference's 100 000-triangle model alone under the sky colour
        table["dragon"] = ({"film": (1920, 1080), "camera": {"pos": (28., 12., -60.), "rot": (0., 180., 0.), "fov": 45., "near": .1, "far": 100.},
                            "objects": [{"type": "mesh", "path": "tests/golden/chinese_dragon.glb", "pos": (0, 0, 0), "rot": (-90., 0, 0), "scale": (1, 1.2, 0.8)}]}, 6)
    for name, (rings, segments) in {"bumpy_1m": (500, 1000), "bumpy_280": (10, 14)}.items():
        if name not in args.scenes.split(","):
            continue
        from scenes import gltf_io
        prim = gltf_io.bumpy_sphere_primitive(rings, segments, radius=1.0, bump=0.06, freq=9 if rings > 100 else 5)
        idx = prim["indices"].reshape(-1, 3)
        data = {"positions": np.ascontiguousarray(prim["positions"][idx].reshape(-1, 9)),
                "normals": np.ascontiguousarray(prim["normals"][idx].reshape(-1, 9)),
                "uvs": np.ascontiguousarray(prim["uvs"][idx].reshape(-1, 6)),
                "group": np.zeros(len(idx), np.int32),
                "groups": [dict(gltf_io.default_pbr(), base_color_factor=[0.7, 0.6, 0.5, 1.0], metallic_factor=0.0, roughness_factor=0.6, name="m")],
                "images": []}
        objs = presets.cornell_room_boxes()
        objs.append({"type": "mesh", "data": data, "pos": (2.775, 2.2, 2.2), "rot": (20.0, 30.0, 0.0), "scale": (2.3, 2.0, 2.1)})
        table[name] = ({"film": (1920, 1080), "camera": presets.default_camera(), "background": (0.2, 0.2, 0.2), "objects": objs}, 8)
    for name in args.scenes.split(","):
        scene, depth = table[name]
        w, h = scene["film"]
        pt, cam = ptamd.build_scene(scene)
        film = ptamd.Film(w, h)
        # executed traversal work: primary rays and random interior rays
        prim = ptamd.Camera((240, 135), cam.mPosition, cam.mRotation)
        prim.SetFov(cam.mFov)
        rays = prim.GetRays().reshape(-1, 6)
        rng = np.random.default_rng(1)
        o = rng.random((20000, 3)).astype(np.float32) * 5.0 + 0.25
        d = rng.normal(size=(20000, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        _, c1 = pt.TraversalCounts(rays)
        _, c2 = pt.TraversalCounts(np.concatenate([o, d], 1))
        print("   traversal: primary %.1f nodes %.1f tris | interior %.1f nodes %.1f tris (max %d / %d)" % (
            c1[:, 0].mean(), c1[:, 1].mean(), c2[:, 0].mean(), c2[:, 1].mean(), c2[:, 0].max(), c2[:, 1].max()))
        ptamd.Render(cam, pt, film, 8, depth, count=True)
        torch.cuda.synchronize()
        st = ptamd.RenderStats()
        rps = st["rays"] / st["samples"]
        fill = lambda a, b: st[a] / max(1, 64 * st[b])
        if st["ray_trips"]:
            print("   per ray: %.1f nodes %.1f tris %.2f shades | wave trips per ray-trip: %.1f node %.1f tri %.2f shade | lanes filled: bounce loop %.0f%% node loop %.0f%% tri loop %.0f%% shading %.0f%% | %.1f steal rounds per ray-trip" % (
                st["nodes"] / st["rays"], st["tris"] / st["rays"], st["shade_lanes"] / st["rays"],
                st["node_trips"] / st["ray_trips"], st["tri_trips"] / st["ray_trips"], st["shade_trips"] / st["ray_trips"],
                100 * fill("rays", "ray_trips"), 100 * fill("nodes", "node_trips"), 100 * fill("tris", "tri_trips"), 100 * fill("shade_lanes", "shade_trips"), st["steal_rounds"] / st["ray_trips"]))
        ms = []
        for _ in range(args.reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ptamd.Render(cam, pt, film, args.spp, depth)
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        t = float(np.median(ms))
        msps = w * h * args.spp / t * 1e-3
        print("%-16s depth %2d  %8.2f ms  %9.1f Msamples/s  %9.1f Mrays/s  (%.2f rays/sample, min %.2f ms)" % (name, depth, t, msps, msps * rps, rps, min(ms)), flush=True)


if __name__ == "__main__":
    main()
