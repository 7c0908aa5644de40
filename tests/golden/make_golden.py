"""Generates the golden fixtures in this directory from the reference renderer's own sources
(oracle/_ref/ref_oracle, built by `make -C oracle ref`; only possible where /root/reference exists).

Run:  python tests/golden/make_golden.py            (every scene + the film vectors)
      python tests/golden/make_golden.py dragon dragon_field   (only the named scenes)
Writes small .npz files (inputs + expected outputs) and the .glb inputs the reference loaded.
Nothing of the reference's source text is stored: only scene parameters and output buffers.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from scenes import dragon, gltf_io, presets  # noqa: E402


def lum(a):
    return (a * np.array([0.2126, 0.7152, 0.0722], np.float32)).sum(-1)


def write_textured_glb(path):
    """A .glb that exercises every texture path of the reference (Mesh.cpp:15-37,106-125,170-227,447-486): base
    colour + alpha MASK cutout, metallic-roughness, emissive, normal map, transmission texture."""
    rng = np.random.default_rng(11)
    n = 16
    yy, xx = np.mgrid[0:n, 0:n]
    base = np.zeros((n, n, 4), np.uint8)
    base[..., 0] = 60 + 12 * xx
    base[..., 1] = 40 + 13 * yy
    base[..., 2] = rng.integers(30, 255, (n, n))
    base[..., 3] = np.where(((xx // 2 + yy // 2) % 3) == 0, 40, 255)      # holes for the MASK cutout
    mr = np.zeros((n, n, 3), np.uint8)
    mr[..., 1] = rng.integers(20, 255, (n, n))                             # G = roughness
    mr[..., 2] = np.where((xx + yy) % 4 < 2, 255, 0)                       # B = metallic
    em = np.zeros((n, n, 3), np.uint8)
    em[(xx % 5 == 0) & (yy % 5 == 0)] = (255, 180, 90)
    nm = np.zeros((n, n, 3), np.uint8)
    nm[..., 0] = 128 + (40 * np.sin(xx * 0.9)).astype(int)
    nm[..., 1] = 128 + (40 * np.cos(yy * 0.7)).astype(int)
    nm[..., 2] = 230
    tr = np.zeros((n, n, 1), np.uint8)
    tr[..., 0] = np.where(xx < n // 2, 255, 64)
    images = [base, mr, em, nm, tr]
    p0 = gltf_io.bumpy_sphere_primitive(24, 32, radius=0.8, bump=0.05, freq=3, material_index=0)
    p0["uvs"] = p0["uvs"] * np.array([3.0, 2.0], np.float32) - np.float32(0.25)   # wraps and negative uvs
    p1 = gltf_io.box_primitive((1.2, -0.3, 0.2), (10.0, 20.0, 0.0), (0.7, 0.7, 0.7), 1)
    p2 = gltf_io.box_primitive((-1.1, 0.5, 0.3), (0.0, 45.0, 30.0), (0.5, 0.9, 0.5), 2)
    mats = [
        {"name": "masked_pbr", "alphaMode": "MASK", "alphaCutoff": 0.5, "emissiveFactor": [1.5, 1.5, 1.5],
         "pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.8, 1.0, 0.9], "baseColorTexture": {"index": 0},
                                  "metallicFactor": 0.8, "roughnessFactor": 0.9, "metallicRoughnessTexture": {"index": 1}},
         "emissiveTexture": {"index": 2}, "normalTexture": {"index": 3, "scale": 1.5}},
        {"name": "glassy", "pbrMetallicRoughness": {"baseColorFactor": [1, 1, 1, 1], "metallicFactor": 0.0, "roughnessFactor": 0.1},
         "extensions": {"KHR_materials_transmission": {"transmissionFactor": 0.9, "transmissionTexture": {"index": 4}},
                        "KHR_materials_ior": {"ior": 1.4}}},
        {"name": "bumpy_plain", "pbrMetallicRoughness": {"baseColorFactor": [0.3, 0.7, 0.4, 1], "metallicFactor": 0.0},
         "normalTexture": {"index": 3}},
    ]
    gltf_io.write_glb(path, [p0, p1, p2], mats, images)


def jpeg_pictures():
    """Small JPEG files (PIL / libjpeg encodes; only the decoder is under test) that between them use every path of the
    decoder: baseline and progressive, 4:4:4 / 4:2:2 / 4:2:0, sizes that are not multiples of the MCU, one-pixel rows and
    columns, grey, CMYK, restart intervals, optimised Huffman tables."""
    import io
    from PIL import Image
    rng = np.random.default_rng(3)

    def picture(w, h):
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.zeros((h, w, 3), np.float64)
        a[..., 0] = 128 + 100 * np.sin(xx * 0.37) * np.cos(yy * 0.21)
        a[..., 1] = (xx * 7 + yy * 3) % 256
        a[..., 2] = 255 * ((xx // 5 + yy // 3) % 2)
        return np.clip(a + rng.normal(0, 12, a.shape), 0, 255).astype(np.uint8)
    cases = [(40, 24, "RGB", dict(quality=90, subsampling=0)), (17, 13, "RGB", dict(quality=35, subsampling=1)),
             (33, 19, "RGB", dict(quality=75, subsampling=2)), (1, 1, "RGB", dict(quality=90, subsampling=2)),
             (9, 40, "RGB", dict(quality=60, subsampling=2, progressive=True)), (33, 1, "RGB", dict(quality=90, subsampling=1, progressive=True)),
             (48, 32, "RGB", dict(quality=85, subsampling=0, progressive=True, optimize=True)), (50, 31, "L", dict(quality=80)),
             (50, 31, "L", dict(quality=60, progressive=True)), (48, 32, "RGB", dict(quality=50, subsampling=1, restart_marker_blocks=3)),
             (48, 32, "RGB", dict(quality=50, subsampling=2, progressive=True, restart_marker_rows=1)), (31, 29, "RGB", dict(quality=100, subsampling=0)),
             (40, 24, "CMYK", dict(quality=80))]
    out = []
    for w, h, mode, kw in cases:
        im = Image.fromarray(picture(w, h))
        im = im.convert(mode) if mode != "RGB" else im
        buf = io.BytesIO()
        im.save(buf, "JPEG", **kw)
        out.append(buf.getvalue())
    return out


ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def png_encode_any(samples, depth, ctype, interlace=False, palette=None, trns=None):
    """A PNG file of any flavour the format allows. samples: (H, W, C) integers below 2^depth, C = channels of the colour
    type (1 for grey and for palette indices). Rows are filtered with the five filter types in turn."""
    import struct
    import zlib
    h, w, c = samples.shape
    assert c == {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, c * depth // 8)

    def pack(rows):  # (h', w', c) -> list of bytes per row
        out = []
        for r in rows:
            flat = r.reshape(-1).astype(np.uint32)
            if depth == 16:
                out.append(flat.astype(">u2").tobytes())
            elif depth == 8:
                out.append(flat.astype(np.uint8).tobytes())
            else:
                per = 8 // depth
                pad = (-len(flat)) % per
                f = np.concatenate([flat, np.zeros(pad, np.uint32)]).reshape(-1, per)
                byte = np.zeros(len(f), np.uint32)
                for k in range(per):
                    byte |= f[:, k] << (8 - depth * (k + 1))
                out.append(byte.astype(np.uint8).tobytes())
        return out

    def filtered(rows, start):
        out = bytearray()
        prev = bytes(len(rows[0])) if rows else b""
        for i, row in enumerate(rows):
            ft = (start + i) % 5
            line = bytearray(len(row))
            for k in range(len(row)):
                a = row[k - bpp] if k >= bpp else 0
                b = prev[k]
                cc = prev[k - bpp] if k >= bpp else 0
                if ft == 0:
                    pred = 0
                elif ft == 1:
                    pred = a
                elif ft == 2:
                    pred = b
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    pp = a + b - cc
                    pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - cc)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
                line[k] = (row[k] - pred) & 255
            out.append(ft)
            out += line
            prev = row
        return bytes(out)

    raw = b""
    if interlace:
        for n, (x0, y0, dx, dy) in enumerate(ADAM7):
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += filtered(pack(sub), n)
    else:
        raw = filtered(pack(samples), 1)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    comp = zlib.compress(raw, 6)
    half = len(comp) // 2  # two IDAT chunks: decoders must join them
    return out + chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b"")


def png_pictures():
    """PNG files covering what stb_image (the reference's decoder, through tinygltf) accepts beyond 8-bit non-interlaced
    colour: bit depths 1, 2, 4 and 16, palettes with and without transparency, colour-key transparency, grey + alpha,
    Adam7 interlacing (sizes that leave some passes empty included)."""
    import struct
    rng = np.random.default_rng(17)

    def img(w, h, c, depth):
        return rng.integers(0, 1 << depth, (h, w, c), dtype=np.uint32)
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    pal256 = rng.integers(0, 256, (256, 3), dtype=np.uint8)
    files = [
        png_encode_any(img(13, 9, 3, 8), 8, 2, interlace=True),
        png_encode_any(img(8, 8, 4, 8), 8, 6, interlace=True),
        png_encode_any(img(3, 2, 4, 8), 8, 6, interlace=True),          # passes 2..6 partly empty
        png_encode_any(img(1, 1, 3, 8), 8, 2, interlace=True),
        png_encode_any(img(17, 5, 1, 1), 1, 0), png_encode_any(img(10, 7, 1, 2), 2, 0), png_encode_any(img(9, 6, 1, 4), 4, 0),
        png_encode_any(img(11, 7, 1, 4), 4, 0, interlace=True),
        png_encode_any(img(12, 5, 1, 4), 4, 3, palette=pal), png_encode_any(img(7, 9, 1, 2), 2, 3, palette=pal[:4]),
        png_encode_any(img(15, 4, 1, 1), 1, 3, palette=pal[:2], trns=[40]),
        png_encode_any(img(9, 9, 1, 8), 8, 3, palette=pal256, trns=list(rng.integers(0, 256, 100)), interlace=True),
        png_encode_any(img(10, 6, 2, 8), 8, 4), png_encode_any(img(10, 6, 1, 8), 8, 0, trns=struct.pack(">H", 77)),
        png_encode_any(np.minimum(img(10, 6, 3, 8), 3), 8, 2, trns=struct.pack(">HHH", 1, 2, 3)),
        png_encode_any(np.minimum(img(12, 5, 1, 2), 3), 2, 0, trns=struct.pack(">H", 2)),
        png_encode_any(img(7, 5, 3, 16), 16, 2), png_encode_any(img(6, 4, 4, 16), 16, 6, interlace=True),
        png_encode_any(img(9, 3, 1, 16), 16, 0), png_encode_any(img(5, 5, 2, 16), 16, 4),
        png_encode_any(img(6, 6, 1, 16) & 0x0303, 16, 0, trns=struct.pack(">H", 0x0101)),
    ]
    return files


def write_png_fixtures():
    """png_cases.npz: the PNG files above + the bytes ModelLoader holds for each after loading them through the reference
    (ref_oracle images). A 16-bit file arrives as 16-bit samples (tiny_gltf.h:2645-2661) that ModelLoader copies as they are
    (ModelLoader.h:190-197), so its entry is twice as long as width * height * channels."""
    import subprocess
    import tempfile
    files = png_pictures()
    prim = gltf_io.box_primitive((0, 0, 0), (0, 0, 0), (1, 1, 1), 0)
    mat = gltf_io.gltf_material("m", presets.material())
    tmp = os.path.join(tempfile.mkdtemp(prefix="ptpng_"), "cases.glb")
    gltf_io.write_glb(tmp, [prim], [mat], files)
    pre = os.path.join(os.path.dirname(tmp), "im")
    subprocess.run([po.REF_BIN, "images", tmp, pre], check=True, capture_output=True)
    out = {}
    for i, line in enumerate(open(pre + ".txt")):
        w, h, c = map(int, line.split())
        out["file_%d" % i] = np.frombuffer(files[i], np.uint8)
        out["size_%d" % i] = np.array([w, h, c], np.int32)
        out["bytes_%d" % i] = np.fromfile("%s.%d.u8" % (pre, i), np.uint8)
    assert len(out) == 3 * len(files)
    np.savez_compressed(os.path.join(HERE, "png_cases.npz"), **out)
    return out


def reference_images(glb):
    """The texels ModelLoader holds after loading glb (ref_oracle images): list of (H, W, C) uint8."""
    import subprocess
    import tempfile
    pre = os.path.join(tempfile.mkdtemp(prefix="ptimg_"), "im")
    subprocess.run([po.REF_BIN, "images", glb, pre], check=True, capture_output=True)
    out = []
    for i, line in enumerate(open(pre + ".txt")):
        w, h, c = map(int, line.split())
        out.append(np.fromfile("%s.%d.u8" % (pre, i), np.uint8).reshape(h, w, c))
    return out


def write_jpeg_fixtures():
    """jpeg_cases.npz: JPEG files + the reference's decode of each. textured_jpeg.glb: the textured test mesh with its
    base-colour and emissive maps as JPEG (4:2:0 baseline, 4:4:4 progressive), + textured_jpeg.images.npz: the
    reference's decode of its images, which the Python glTF reader (CPU oracle side) uses in place of a JPEG decoder."""
    import io
    import tempfile
    from PIL import Image
    files = jpeg_pictures()
    prim = gltf_io.box_primitive((0, 0, 0), (0, 0, 0), (1, 1, 1), 0)
    mat = gltf_io.gltf_material("m", presets.material())
    tmp = os.path.join(tempfile.mkdtemp(prefix="ptjpg_"), "cases.glb")
    gltf_io.write_glb(tmp, [prim], [mat], files)
    dec = reference_images(tmp)
    assert len(dec) == len(files)
    np.savez_compressed(os.path.join(HERE, "jpeg_cases.npz"), **{"file_%d" % i: np.frombuffer(f, np.uint8) for i, f in enumerate(files)},
                        **{"texels_%d" % i: d for i, d in enumerate(dec)})
    # the textured scene with JPEG maps
    n = 32
    yy, xx = np.mgrid[0:n, 0:n]
    base = np.zeros((n, n, 3), np.uint8)
    base[..., 0] = 60 + 6 * xx; base[..., 1] = 40 + 6 * yy; base[..., 2] = 128 + (100 * np.sin(xx * 0.5) * np.cos(yy * 0.4)).astype(int)
    em = np.zeros((n, n, 3), np.uint8)
    em[(xx % 8 < 2) & (yy % 8 < 2)] = (255, 180, 90)

    def jpg(a, **kw):
        b = io.BytesIO()
        Image.fromarray(a).save(b, "JPEG", **kw)
        return b.getvalue()
    rng = np.random.default_rng(5)
    mr = np.zeros((16, 16, 3), np.uint8)
    mr[..., 1] = rng.integers(20, 255, (16, 16)); mr[..., 2] = 128
    images = [jpg(base, quality=88, subsampling=2), mr, jpg(em, quality=92, subsampling=0, progressive=True)]
    p0 = gltf_io.bumpy_sphere_primitive(24, 32, radius=0.8, bump=0.05, freq=3, material_index=0)
    p0["uvs"] = p0["uvs"] * np.array([2.0, 1.5], np.float32)
    p1 = gltf_io.box_primitive((1.2, -0.3, 0.2), (10.0, 20.0, 0.0), (0.7, 0.7, 0.7), 1)
    mats = [{"name": "jpeg_pbr", "emissiveFactor": [1.2, 1.2, 1.2],
             "pbrMetallicRoughness": {"baseColorFactor": [1.0, 0.9, 1.0, 1.0], "baseColorTexture": {"index": 0}, "metallicFactor": 0.6,
                                      "roughnessFactor": 0.8, "metallicRoughnessTexture": {"index": 1}},
             "emissiveTexture": {"index": 2}},
            {"name": "jpeg_plain", "pbrMetallicRoughness": {"baseColorFactor": [1, 1, 1, 1], "baseColorTexture": {"index": 0}, "metallicFactor": 0.0}}]
    glb = os.path.join(HERE, "textured_jpeg.glb")
    gltf_io.write_glb(glb, [p0, p1], mats, images)
    np.savez_compressed(os.path.join(HERE, "textured_jpeg.images.npz"), **{"image_%d" % i: d for i, d in enumerate(reference_images(glb))})


def dragon_scenes(field_glb):
    """The reference's own 100 000-triangle model (tests/golden/chinese_dragon.glb is a byte copy of its
    assets/models/chinese_dragon.glb) alone under the constant background, and the ten-instance field derived from
    it (scenes/dragon.py, ~1 M triangles, regenerated wherever it is needed) in the closed room."""
    single = {"film": (96, 72), "camera": {"pos": (28., 12., -60.), "rot": (0., 180., 0.), "fov": 45., "near": .1, "far": 100.},
              "objects": [{"type": "mesh", "path": "tests/golden/chinese_dragon.glb", "pos": (0, 0, 0), "rot": (-90., 0, 0), "scale": (1, 1.2, 0.8)}]}
    return {"dragon": (single, 6), "dragon_field": (dragon.config3_scene(field_glb, 96, 54), 8)}


def golden_scenes():
    """name -> (scene dict, depth for the stochastic render)."""
    cm = os.path.join(HERE, "cornell_mesh.glb")
    blob = os.path.join(HERE, "blob.glb")
    gltf_io.write_box_scene_glb(cm, presets.box_cornell()["objects"])
    # a smooth procedural mesh with two materials (one glass, one rough metal) and vertex normals
    p0 = gltf_io.bumpy_sphere_primitive(48, 64, radius=1.0, material_index=0)
    p1 = gltf_io.bumpy_sphere_primitive(24, 32, radius=0.45, bump=0.15, freq=5, material_index=1)
    p1["positions"] = p1["positions"] + np.array([1.3, 0.9, -0.2], np.float32)
    mats = [gltf_io.gltf_material("copper", presets.material(albedo=(0.9, 0.5, 0.3), roughness=0.3, metallic=1.0)),
            gltf_io.gltf_material("glass", presets.material(albedo=(1, 1, 1), roughness=0.0, ior=1.45, transmission=1.0))]
    gltf_io.write_glb(blob, [p0, p1], mats)
    write_textured_glb(os.path.join(HERE, "textured.glb"))
    tex_scene = presets.mixed("tests/golden/textured.glb", 96, 96, mesh_pos=(2.7, 1.6, 2.6), mesh_rot=(0.0, 25.0, 0.0), mesh_scale=(1.4, 1.4, 1.4))
    blob_scene = presets.mixed("tests/golden/blob.glb", 96, 96, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0))
    return {
        "sphere_cornell": (presets.sphere_cornell(96, 96), 5),
        "glass_mirror": (presets.glass_mirror(96, 96), 12),
        "box_cornell": (presets.box_cornell(96, 64), 6),
        "mesh_cornell": (presets.mesh_cornell("tests/golden/cornell_mesh.glb", 96, 54), 8),
        "blob_mixed": (blob_scene, 8),
        "textured": (tex_scene, 6),
        # 1920/128 = 15 is odd, so pixel (x, y) here has the same centre ray as pixel (15x+7, 15y+7) of the 1080p frame
        "mesh_cornell_128x72": (presets.mesh_cornell("tests/golden/cornell_mesh.glb", 128, 72), 8),
    }


def camera_4k():
    """Camera::GetRay known answers at 3840x2160 for three cameras (the default one, the raised and turned one of the dragon
    scene, a rolled wide-angle one): the reference's rays on a lattice of rows and columns that includes the frame's edges."""
    ys = np.r_[0:2160:83, 2159].astype(np.int32)
    xs = np.r_[0:3840:97, 3839].astype(np.int32)
    cams = [presets.default_camera(),
            dict(presets.default_camera(), pos=(2.78, 2.4, -1.8), rot=(17.0, 180.0, 0.0), fov=52.0),
            dict(presets.default_camera(), pos=(-3.5, 7.25, 11.0), rot=(-31.0, 23.0, 14.0), fov=97.0)]
    out = {"ys": ys, "xs": xs}
    for i, cam in enumerate(cams):
        scene = {"film": (3840, 2160), "camera": cam, "objects": presets.sphere_cornell(8, 8)["objects"]}
        rays = po.ref_rays(scene).reshape(2160, 3840, 6)
        out["rays%d" % i] = np.ascontiguousarray(rays[ys][:, xs])
        out["cam%d" % i] = np.array(list(cam["pos"]) + list(cam["rot"]) + [cam["fov"], cam["near"], cam["far"]], np.float32)
    np.savez_compressed(os.path.join(HERE, "camera_4k.npz"), **out)
    print("camera_4k", {k: v.shape for k, v in out.items()})


def main():
    assert po.have_ref(), "build oracle/_ref first: make -C oracle ref"
    os.chdir(ROOT)  # mesh paths in the scenes are relative to the repo root
    only = sys.argv[1:]
    if not only or "camera_4k" in only:
        camera_4k()
        if only == ["camera_4k"]:
            return
    import tempfile
    field = os.path.join(tempfile.mkdtemp(prefix="ptgold_"), "dragon_field.glb")
    dragon.write_dragon_field(field)
    if not only or "textured_jpeg" in only:
        write_jpeg_fixtures()
        write_png_fixtures()
    late = {"textured_jpeg": (presets.mixed("tests/golden/textured_jpeg.glb", 96, 96, mesh_pos=(2.7, 1.6, 2.6), mesh_rot=(0.0, 25.0, 0.0),
                                            mesh_scale=(1.4, 1.4, 1.4)), 6)}
    late.update(dragon_scenes(field))
    scenes = dict(golden_scenes() if not only or any(n not in late for n in only) else {}, **late)
    for name, (scene, depth) in scenes.items():
        if only and name not in only:
            continue
        rays = po.ref_rays(scene)
        hits = po.ref_hits(scene)
        alb, alb_s, alb_rgba, _ = po.ref_render(scene, 1, 10, True, threads=4, tasks=16)
        d1, d1_s, _, _ = po.ref_render(scene, 2, 1, False, threads=4, tasks=16)
        spp = 256
        a1, s1, rgba1, _ = po.ref_render(scene, spp, depth, False)
        a2, s2, _, _ = po.ref_render(scene, spp, depth, False)
        A1, A2 = po.average(a1, s1), po.average(a2, s2)
        self_rmse = float(np.sqrt(((A1 - A2) ** 2).mean()))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rays=rays, hits=hits, albedo_accum=alb,
                            albedo_rgba=alb_rgba, depth1_accum=d1, depth1_samples=d1_s,
                            avg_a=A1.astype(np.float16), avg_b=A2.astype(np.float16), spp=spp, depth=depth,
                            self_rmse=self_rmse, mean_lum_a=float(lum(A1).mean()), mean_lum_b=float(lum(A2).mean()),
                            rgba_a=rgba1)
        print(name, "self_rmse", self_rmse, "lum", float(lum(A1).mean()), float(lum(A2).mean()))
    if only:
        return
    # Film known-answer vectors: AddSample clamp + the four resolve variants
    rng = np.random.default_rng(7)
    vals = np.concatenate([rng.random((200, 3)) * np.array([0.02, 1.0, 30.0])[rng.integers(0, 3, (200, 1))],
                           np.array([[0, 0, 0], [12, 12, 12], [15, 12.6, 8.85], [0.0031308, 0.0031307, 0.0031309],
                                     [1e-8, 1e3, 0.5], [100, 0, 0], [0, 100, 0], [0, 0, 100]])]).astype(np.float32)
    import tempfile
    d = tempfile.mkdtemp()
    vals.tofile(os.path.join(d, "in.f32"))
    import subprocess
    subprocess.run([po.REF_BIN, "film", os.path.join(d, "in.f32"), str(len(vals)), os.path.join(d, "out")], check=True)
    out = {"values": vals, "accum": np.fromfile(os.path.join(d, "out.accum.f32"), np.float32).reshape(-1, 3),
           "samples": np.fromfile(os.path.join(d, "out.samples.u32"), np.uint32),
           "avg": np.fromfile(os.path.join(d, "out.avg.f32"), np.float32).reshape(-1, 3)}
    for cs in (0, 1):
        for tm in (0, 1):
            out["rgba_cs%d_tm%d" % (cs, tm)] = np.fromfile(os.path.join(d, "out.cs%d.tm%d.rgba8" % (cs, tm)), np.uint8).reshape(-1, 4)
    np.savez_compressed(os.path.join(HERE, "film_kat.npz"), **out)
    print("film_kat", len(vals))


if __name__ == "__main__":
    main()
