"""The ~1 M-triangle input of BASELINE.json's config 3, derived from the reference's only real model.

`tests/golden/chinese_dragon.glb` is a byte copy of the reference's assets/models/chinese_dragon.glb (a data file:
100 000 triangles, 59 089 vertices with normals, one material, no textures). `write_dragon_field` instances it ten
times -- rotated about its up axis, scaled and laid out in two rows -- and bakes the copies into ONE indexed .glb
(the reference's loader ignores node transforms, ModelLoader.h:204-206, so instancing through nodes would collapse
the copies onto each other). Both the reference and this library load that same file.
"""
import os

import numpy as np

from . import gltf_io

HERE = os.path.dirname(os.path.abspath(__file__))
DRAGON = os.path.join(os.path.dirname(HERE), "tests", "golden", "chinese_dragon.glb")


def read_indexed(path):
    """(positions (V,3), normals (V,3), indices (T*3,), material dict) of the first primitive."""
    doc, buffers, _ = gltf_io._read_doc(path)
    prim = doc["meshes"][0]["primitives"][0]
    pos, _ = gltf_io._accessor(doc, buffers, prim["attributes"]["POSITION"])
    nrm, _ = gltf_io._accessor(doc, buffers, prim["attributes"]["NORMAL"])
    idx, _ = gltf_io._accessor(doc, buffers, prim["indices"])
    mat = doc["materials"][prim.get("material", 0)]
    return (np.ascontiguousarray(pos, np.float32), np.ascontiguousarray(nrm, np.float32),
            np.ascontiguousarray(idx.reshape(-1), np.uint32), mat)


def write_dragon_field(dst, copies=10, src=DRAGON):
    """Writes `copies` baked instances of the dragon (copies x 100 000 triangles) into dst; returns the face count.
    The model is Z-up as stored (bbox (0,0,0)-(56.4,25.2,39.8)); copy k is turned by 36 k degrees about Z around the
    bbox centre, scaled by 0.8 + 0.04 k, and placed on a grid of 2 columns x 5 rows with a pitch of 62 x 26 units
    (neighbours interpenetrate, which a path tracer does not mind). Deterministic: float32 arithmetic on fixed inputs."""
    pos, nrm, idx, mat = read_indexed(src)
    centre = ((pos.min(axis=0) + pos.max(axis=0)) * np.float32(0.5)).astype(np.float32)
    prims = []
    for k in range(copies):
        a = np.float32(np.radians(36.0 * k))
        c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)
        scale = np.float32(0.8 + 0.04 * k)
        off = np.array([62.0 * (k % 2), 26.0 * (k // 2), 0.0], np.float32)
        p = ((pos - centre) * scale) @ R.T + centre + off
        n = nrm @ R.T
        prims.append({"positions": p.astype(np.float32), "normals": n.astype(np.float32), "uvs": None, "indices": idx, "material": 0})
    gltf_io.write_glb(dst, prims, [mat])
    return copies * (len(idx) // 3)


def config3_scene(glb_path, w=1920, h=1080):
    """Closed Cornell room (main.cpp:229-305) with the dragon field standing on its floor. The field as stored spans
    x 4.3..118.4, y -4.5..145.2, z -3.2..42.9 (Z-up); Rx(-90) Rz(180) turns it upright (world = pos + s (-x, z, y)) and
    s = 0.034 makes it 3.9 room units wide, 5.1 deep and 1.6 high, between the side walls and in front of the back wall."""
    from . import presets
    objs = presets.cornell_room_boxes()
    s = 0.034
    objs.append({"type": "mesh", "path": glb_path, "pos": (4.861, 0.13, 0.4), "rot": (-90.0, 0.0, 180.0), "scale": (s, s, s)})
    # the reference's default camera stands 8 units in front of the room (main.cpp:60-63), where the field covers 4 % of
    # the frame; this one stands at the room's open front and looks down on the field
    cam = dict(presets.default_camera(), pos=(2.78, 2.4, -1.8), rot=(17.0, 180.0, 0.0), fov=52.0)
    return {"film": (w, h), "camera": cam, "background": (0.2, 0.2, 0.2), "objects": objs}
