"""Generates the converged-image fixtures tests/golden/converged_<scene>.npz from the reference renderer's own sources
(oracle/_ref/ref_oracle, built by `make -C oracle ref`; only possible where /root/reference exists).

For every scene the reference renders CHUNKS independent runs of SPP samples per pixel (its generator seeds itself from
the thread ids, PathTracer.cpp:8-16, so separate runs are separate sample sets). Stored per scene:
    mean      (H, W, 3) f32   average of the chunk averages = the reference's clamped-mean image at CHUNKS * SPP spp
    stderr    (H, W, 3) f32   standard error of that mean, from the spread of the chunks
    region    (H, W)    u8    which material the camera ray meets (0 = miss), from the reference's own hit records
    region_lum (R, CHUNKS) f32  mean luminance of each region in each chunk
    materials (R, 11)   f32   the material rows behind the region numbers
    spp, chunks, depth
Also writes tests/golden/emitter_panel.glb (a quad lit only by its emissive texture).

    python tests/golden/make_converged.py [scene ...]        (about 25 minutes on 8 cores for all of them)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po  # noqa: E402
from scenes import dragon, gltf_io, presets  # noqa: E402

CHUNKS, SPP = 8, 4096
LUM = np.array([0.2126, 0.7152, 0.0722], np.float32)


def write_emitter_panel(path):
    """A 2.4 x 2.4 quad facing -z (towards the camera and the sphere) whose material has no factor-only emission: the
    light comes from an 8 x 8 emissive texture with dark, coloured and bright texels."""
    n = 8
    yy, xx = np.mgrid[0:n, 0:n]
    em = np.zeros((n, n, 3), np.uint8)
    em[..., 0] = np.where((xx + yy) % 2 == 0, 255, 40)
    em[..., 1] = np.where(xx < 4, 200, 90)
    em[..., 2] = np.where(yy < 4, 60, 230)
    em[(xx % 3 == 0) & (yy % 3 == 0)] = 0
    h = 1.2
    pos = np.array([[-h, -h, 0], [h, -h, 0], [h, h, 0], [-h, h, 0]], np.float32)
    nrm = np.tile(np.array([[0, 0, -1]], np.float32), (4, 1))
    uvs = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    prim = {"positions": pos, "normals": nrm, "uvs": uvs, "indices": np.array([0, 2, 1, 0, 3, 2], np.uint32), "material": 0}
    mat = {"name": "panel", "emissiveFactor": [2.2, 2.2, 2.2], "emissiveTexture": {"index": 0},
           "pbrMetallicRoughness": {"baseColorFactor": [0.1, 0.1, 0.1, 1.0], "metallicFactor": 0.0, "roughnessFactor": 1.0}}
    gltf_io.write_glb(path, [prim], [mat], [em])


def scenes(tmp):
    """name -> (scene dict, depth). The golden scenes of conftest.py at about 2300 pixels, and the lobe scenes."""
    panel = "tests/golden/emitter_panel.glb"
    field = os.path.join(tmp, "dragon_field.glb")
    out = {
        "sphere_cornell": (presets.sphere_cornell(48, 48), 5),
        "glass_mirror": (presets.glass_mirror(48, 48), 12),
        "box_cornell": (presets.box_cornell(64, 36), 6),
        "mesh_cornell": (presets.mesh_cornell("tests/golden/cornell_mesh.glb", 64, 36), 8),
        "blob_mixed": (presets.mixed("tests/golden/blob.glb", 48, 48, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0)), 8),
        "textured": (presets.mixed("tests/golden/textured.glb", 48, 48, mesh_pos=(2.7, 1.6, 2.6), mesh_rot=(0.0, 25.0, 0.0), mesh_scale=(1.4, 1.4, 1.4)), 6),
        "dragon_field": (lambda: dragon.config3_scene(field, 64, 36), 8),
    }
    for k in presets.LOBE_KINDS:
        out["lobe_" + k] = (presets.lobe_scene(k, 64, 36, mesh_path=panel), 8)
    return out, field


def regions_of(hits):
    """Region numbers from closest-hit records (H, W, 20): 0 = miss, else 1 + index of the material row."""
    h, w, _ = hits.shape
    mats = hits[..., 9:20].reshape(-1, 11)
    hit = hits[..., 0].reshape(-1) > 0
    rows, inv = np.unique(np.round(mats[hit], 4), axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    # materials seen by fewer than 24 pixels (the texels of a textured surface, slivers) share one last region
    big = [i for i in range(len(rows)) if (inv == i).sum() >= 24]
    remap = np.full(len(rows), len(big), np.int64)
    remap[big] = np.arange(len(big))
    reg = np.zeros(h * w, np.uint8)
    reg[hit] = 1 + remap[inv]
    mats = np.concatenate([rows[big], np.zeros((1, 11))]) if len(big) < len(rows) else rows
    return reg.reshape(h, w), mats.astype(np.float32)


def main():
    import tempfile
    assert po.have_ref(), "oracle/_ref/ref_oracle is missing: make -C oracle ref"
    os.chdir(ROOT)
    write_emitter_panel(os.path.join(HERE, "emitter_panel.glb"))
    tmp = tempfile.mkdtemp(prefix="ptconv_")
    table, field = scenes(tmp)
    want = sys.argv[1:] or list(table)
    threads = len(os.sched_getaffinity(0))
    for name in want:
        scene, depth = table[name]
        if callable(scene):
            dragon.write_dragon_field(field)
            scene = scene()
        w, h = scene["film"]
        t0 = time.time()
        chunks = []
        for c in range(CHUNKS):
            a, s, _, _ = po.ref_render(scene, SPP, depth, threads=threads, tasks=128)
            assert (s == SPP).all()
            chunks.append(a / np.float32(SPP))
        chunks = np.stack(chunks).astype(np.float64)
        mean = chunks.mean(axis=0)
        stderr = chunks.std(axis=0, ddof=1) / np.sqrt(CHUNKS)
        region, mats = regions_of(po.ref_hits(scene))
        nreg = int(region.max()) + 1
        lum = (chunks * LUM).sum(-1)
        region_lum = np.stack([[lum[c][region == r].mean() if (region == r).any() else 0.0 for c in range(CHUNKS)] for r in range(nreg)])
        np.savez_compressed(os.path.join(HERE, "converged_%s.npz" % name), mean=mean.astype(np.float32), stderr=stderr.astype(np.float32),
                            region=region, region_lum=region_lum.astype(np.float32), materials=mats, spp=SPP, chunks=CHUNKS, depth=depth)
        print("%-22s %dx%d depth %d: %d regions, mean lum %.5f, median stderr %.5f, %.0f s" %
              (name, w, h, depth, nreg, float((mean * LUM).sum(-1).mean()), float(np.median(stderr)), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
