"""Minimal glTF 2.0 / GLB writer and reader for fixtures (test-side; numpy + json only).

`load_gltf` restates what the reference's ModelLoader::LoadGLTF (ModelLoader.h:171-420) extracts
from a file: de-indexed faces in file order of meshes/primitives, node transforms ignored,
material groups merged by material *name*, PBR factors + KHR_materials_transmission / _ior.
It is the test-side twin of the product's C++ loader (libptamd: ptamd_model_load) and feeds the
CPU oracle, so that the oracle does not depend on product code.
"""
import base64
import json
import os
import struct
import zlib

import numpy as np

_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


# ----------------------------------------------------------------------------- PNG (8-bit, non-interlaced)
def png_encode(img):
    """img: (H, W, C) uint8 with C in {1,3,4}."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, c = img.shape
    ctype = {1: 0, 3: 2, 4: 6}[c]
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def png_decode(data):
    """Returns (H, W, C) uint8. Supports 8-bit grey/RGB/RGBA/grey+alpha, no interlace, no palette."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n", "not a PNG"
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 4, 6):
        raise ValueError("unsupported PNG flavour")
    c = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8)
    stride = w * c
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        ft = raw[y * (stride + 1)]
        line = raw[y * (stride + 1) + 1:(y + 1) * (stride + 1)].astype(np.int32)
        cur = np.zeros(stride, dtype=np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:
            for i in range(stride):
                a = cur[i - c] if i >= c else 0
                b = prev[i]
                cc = prev[i - c] if i >= c else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - cc
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - cc)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
                cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, c)


# ----------------------------------------------------------------------------- writer
def write_glb(path, primitives, materials, images=None, nodes=None, scene_nodes=None):
    """primitives: list of dicts {positions (V,3) f32, normals (V,3) f32|None, uvs (V,2) f32|None,
    indices (T*3,) u32, material: index|None}; each primitive becomes its own glTF mesh.
    materials: list of glTF material dicts (already in glTF JSON form, textures by image index are
    wired here: any {"index": i} refers to textures[i] == images[i]).  images: list of (H,W,C) uint8 arrays (stored as
    PNG) or of encoded PNG / JPEG files as bytes (stored as they are).
    nodes / scene_nodes: an explicit scene graph in glTF JSON form (node dicts with mesh / children / matrix or
    translation, rotation, scale; root indices); by default one untransformed node per mesh."""
    blob = bytearray()
    views, accessors, meshes = [], [], []

    def add_view(arr, target=None):
        while len(blob) % 4:
            blob.append(0)
        off = len(blob)
        blob.extend(arr.tobytes())
        v = {"buffer": 0, "byteOffset": off, "byteLength": arr.nbytes}
        if target:
            v["target"] = target
        views.append(v)
        return len(views) - 1

    def add_acc(arr, ctype, typ, target=None, minmax=False):
        a = {"bufferView": add_view(arr, target), "componentType": ctype, "count": int(arr.shape[0]), "type": typ}
        if minmax:
            a["min"] = [float(x) for x in arr.min(axis=0)]
            a["max"] = [float(x) for x in arr.max(axis=0)]
        accessors.append(a)
        return len(accessors) - 1

    for p in primitives:
        attrs = {"POSITION": add_acc(np.ascontiguousarray(p["positions"], np.float32), 5126, "VEC3", 34962, True)}
        if p.get("normals") is not None:
            attrs["NORMAL"] = add_acc(np.ascontiguousarray(p["normals"], np.float32), 5126, "VEC3", 34962)
        if p.get("uvs") is not None:
            attrs["TEXCOORD_0"] = add_acc(np.ascontiguousarray(p["uvs"], np.float32), 5126, "VEC2", 34962)
        idx = add_acc(np.ascontiguousarray(p["indices"], np.uint32).reshape(-1), 5125, "SCALAR", 34963)
        prim = {"attributes": attrs, "indices": idx, "mode": 4}
        if p.get("material") is not None:
            prim["material"] = int(p["material"])
        meshes.append({"primitives": [prim]})

    doc = {"asset": {"version": "2.0", "generator": "ptamd scenes/gltf_io.py"},
           "scene": 0, "scenes": [{"nodes": list(range(len(meshes))) if nodes is None else list(scene_nodes)}],
           "nodes": [{"mesh": i} for i in range(len(meshes))] if nodes is None else nodes,
           "meshes": meshes, "accessors": accessors, "bufferViews": views}
    if materials:
        doc["materials"] = materials
        ext = set()
        for m in materials:
            ext.update(m.get("extensions", {}).keys())
        if ext:
            doc["extensionsUsed"] = sorted(ext)
    if images:
        doc["images"], doc["textures"], doc["samplers"] = [], [], [{}]
        for img in images:
            if isinstance(img, (bytes, bytearray)):  # an encoded file as it is (JPEG by its signature, else PNG)
                raw, mime = bytes(img), "image/jpeg" if bytes(img[:2]) == b"\xff\xd8" else "image/png"
            else:
                raw, mime = png_encode(img), "image/png"
            doc["images"].append({"bufferView": add_view(np.frombuffer(raw, dtype=np.uint8)), "mimeType": mime})
            doc["textures"].append({"source": len(doc["images"]) - 1, "sampler": 0})
    while len(blob) % 4:
        blob.append(0)
    doc["buffers"] = [{"byteLength": len(blob)}]
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * (-len(js) % 4)
    total = 12 + 8 + len(js) + 8 + len(blob)
    with open(path, "wb") as f:
        f.write(struct.pack("<III", 0x46546C67, 2, total))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(blob), 0x004E4942) + bytes(blob))


# ----------------------------------------------------------------------------- reader
def _read_doc(path):
    data = open(path, "rb").read()
    base = os.path.dirname(os.path.abspath(path))
    if data[:4] == b"glTF":
        _, _, total = struct.unpack("<III", data[:12])
        pos, doc, bin_chunk = 12, None, None
        while pos < total:
            n, tag = struct.unpack("<II", data[pos:pos + 8])
            body = data[pos + 8:pos + 8 + n]
            pos += 8 + n
            if tag == 0x4E4F534A:
                doc = json.loads(body.decode())
            elif tag == 0x004E4942 and bin_chunk is None:
                bin_chunk = body
    else:
        doc, bin_chunk = json.loads(data.decode()), None
    buffers = []
    for i, b in enumerate(doc.get("buffers", [])):
        uri = b.get("uri")
        if uri is None:
            buffers.append(bin_chunk)
        elif uri.startswith("data:"):
            buffers.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            buffers.append(open(os.path.join(base, uri), "rb").read())
    return doc, buffers, base


def _accessor(doc, buffers, idx):
    acc = doc["accessors"][idx]
    view = doc["bufferViews"][acc["bufferView"]]
    dt = np.dtype(_COMP[acc["componentType"]])
    nc = _NCOMP[acc["type"]]
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = view.get("byteStride", 0) or dt.itemsize * nc
    buf = buffers[view["buffer"]]
    count = acc["count"]
    if stride == dt.itemsize * nc:
        arr = np.frombuffer(buf, dtype=dt, count=count * nc, offset=start).reshape(count, nc)
    else:
        arr = np.lib.stride_tricks.as_strided(
            np.frombuffer(buf, dtype=np.uint8, offset=start), shape=(count, nc * dt.itemsize), strides=(stride, 1))
        arr = np.ascontiguousarray(arr).view(dt).reshape(count, nc)
    return arr, acc


def default_pbr():
    """ModelLoader::PBRMaterial defaults (ModelLoader.h:56-85)."""
    return {"base_color_tex": -1, "normal_tex": -1, "metallic_roughness_tex": -1, "emissive_tex": -1,
            "transmission_tex": -1, "base_color_factor": [1.0, 1.0, 1.0, 1.0], "metallic_factor": 0.0,
            "roughness_factor": 1.0, "alpha_mode": 0, "alpha_cutoff": 0.5, "normal_scale": 1.0,
            "emissive_factor": [0.0, 0.0, 0.0], "transmission_factor": 0.0, "ior": 1.5}


def load_gltf(path):
    """Returns dict(positions (F,9) f32, normals (F,9) f32, uvs (F,6) f32, group (F,) i32,
    groups [pbr dict + name], images [(H,W,C) uint8])."""
    ext = path.rsplit(".", 1)[-1]
    if ext not in ("glb", "gltf"):
        raise RuntimeError("Model only supports .glb/.gltf: " + path)  # ModelLoader.h:142
    doc, buffers, base = _read_doc(path)
    images = []
    for img_index, img in enumerate(doc.get("images", [])):
        if "bufferView" in img:
            v = doc["bufferViews"][img["bufferView"]]
            raw = bytes(buffers[v["buffer"]][v.get("byteOffset", 0):v.get("byteOffset", 0) + v["byteLength"]])
        elif img["uri"].startswith("data:"):
            raw = base64.b64decode(img["uri"].split(",", 1)[1])
        else:
            raw = open(os.path.join(base, img["uri"]), "rb").read()
        if raw[:2] == b"\xff\xd8":
            # JPEG: this Python reader has no decoder for it (and an independent one would not reproduce the reference's
            # texels bit for bit); the texels that the reference's own loader decoded are kept next to the model
            side = os.path.splitext(path)[0] + ".images.npz"
            if not os.path.exists(side):
                raise RuntimeError("JPEG image in %s: decoded texels expected in %s" % (path, side))
            images.append(np.load(side)["image_%d" % img_index])
            continue
        px = png_decode(raw)
        # tinygltf asks stb_image for 4 channels whatever the file holds (tiny_gltf.h:2653)
        h, w, c = px.shape
        rgba = np.full((h, w, 4), 255, np.uint8)
        if c == 1:
            rgba[..., :3] = px
        elif c == 2:
            rgba[..., :3] = px[..., :1]
            rgba[..., 3] = px[..., 1]
        else:
            rgba[..., :c] = px
        images.append(rgba)
    textures = doc.get("textures", [])
    mats = doc.get("materials", [])
    P, N, T, G, groups = [], [], [], [], []
    for mesh in doc.get("meshes", []):
        for prim in mesh["primitives"]:
            if prim.get("mode", 4) != 4:
                raise RuntimeError("Only TRIANGLES are supported")
            if "POSITION" not in prim["attributes"]:
                raise RuntimeError("Missing POSITION")
            pos, pacc = _accessor(doc, buffers, prim["attributes"]["POSITION"])
            if pacc["type"] != "VEC3" or pacc["componentType"] != 5126:
                raise RuntimeError("POSITION must be float vec3")
            nrm = uv = None
            if "NORMAL" in prim["attributes"]:
                a, acc = _accessor(doc, buffers, prim["attributes"]["NORMAL"])
                if acc["type"] == "VEC3" and acc["componentType"] == 5126:
                    nrm = a
            if "TEXCOORD_0" in prim["attributes"]:
                a, acc = _accessor(doc, buffers, prim["attributes"]["TEXCOORD_0"])
                if acc["type"] == "VEC2" and acc["componentType"] == 5126:
                    uv = a
            if prim.get("indices") is None:
                raise RuntimeError("Indexed geometry required")
            idx, iacc = _accessor(doc, buffers, prim["indices"])
            if iacc["type"] != "SCALAR":
                raise RuntimeError("indices accessor must be SCALAR")
            if iacc["componentType"] not in (5121, 5123, 5125):
                raise RuntimeError("Unsupported index type")
            idx = idx.reshape(-1).astype(np.int64)
            nf = len(idx) // 3
            idx = idx[:nf * 3]
            gi = -1
            mi = prim.get("material", -1)
            if mi is not None and mi >= 0:
                m = mats[mi]
                name = m.get("name", "")
                gi = next((i for i, g in enumerate(groups) if g["name"] == name), -1)
                if gi < 0:
                    g = default_pbr()
                    g["name"] = name
                    groups.append(g)
                    gi = len(groups) - 1
                g = groups[gi]
                pbr = m.get("pbrMetallicRoughness", {})

                def tex(t):
                    return textures[t["index"]].get("source", -1) if (t and t.get("index", -1) >= 0) else None
                s = tex(pbr.get("baseColorTexture"))
                if s is not None:
                    g["base_color_tex"] = s
                g["base_color_factor"] = [float(x) for x in pbr.get("baseColorFactor", [1, 1, 1, 1])]
                s = tex(pbr.get("metallicRoughnessTexture"))
                if s is not None:
                    g["metallic_roughness_tex"] = s
                # tinygltf's parsed defaults (metallic 1, roughness 1) overwrite the loader's own
                g["metallic_factor"] = float(pbr.get("metallicFactor", 1.0))
                g["roughness_factor"] = float(pbr.get("roughnessFactor", 1.0))
                s = tex(m.get("normalTexture"))
                if s is not None:
                    g["normal_tex"] = s
                    g["normal_scale"] = float(m["normalTexture"].get("scale", 1.0))
                s = tex(m.get("emissiveTexture"))
                if s is not None:
                    g["emissive_tex"] = s
                am = m.get("alphaMode", "OPAQUE")
                if am == "BLEND":
                    g["alpha_mode"] = 2
                elif am == "MASK":
                    g["alpha_mode"] = 1
                    g["alpha_cutoff"] = float(m.get("alphaCutoff", 0.5))
                else:
                    g["alpha_mode"] = 0
                g["emissive_factor"] = [float(x) for x in m.get("emissiveFactor", [0, 0, 0])]
                exts = m.get("extensions", {})
                tr = exts.get("KHR_materials_transmission")
                if tr is not None:
                    if isinstance(tr.get("transmissionFactor"), (int, float)):
                        g["transmission_factor"] = float(tr["transmissionFactor"])
                    tt = tr.get("transmissionTexture")
                    if isinstance(tt, dict) and isinstance(tt.get("index"), int) and 0 <= tt["index"] < len(textures):
                        src = textures[tt["index"]].get("source", -1)
                        if 0 <= src < len(images):
                            g["transmission_tex"] = src
                io = exts.get("KHR_materials_ior")
                if io is not None and isinstance(io.get("ior"), (int, float)):
                    g["ior"] = float(io["ior"])
            P.append(pos[idx].reshape(nf, 9))
            N.append(nrm[idx].reshape(nf, 9) if nrm is not None else np.zeros((nf, 9), np.float32))
            T.append(uv[idx].reshape(nf, 6) if uv is not None else np.zeros((nf, 6), np.float32))
            G.append(np.full(nf, gi, np.int32))
    return {"positions": np.ascontiguousarray(np.concatenate(P), np.float32),
            "normals": np.ascontiguousarray(np.concatenate(N), np.float32),
            "uvs": np.ascontiguousarray(np.concatenate(T), np.float32),
            "group": np.ascontiguousarray(np.concatenate(G), np.int32),
            "groups": groups, "images": images}


# ----------------------------------------------------------------------------- generators
def _euler_xyz(rot_deg):
    """R = Rx * Ry * Rz (glm post-multiplication order used by Box.cpp:16-19 / Mesh.cpp:49-51)."""
    rx, ry, rz = np.radians(np.asarray(rot_deg, np.float64))
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def box_primitive(pos, rot_deg, size, material_index):
    """12 triangles (24 vertices, flat normals, outward winding) of an oriented box."""
    h = np.asarray(size, np.float64) * 0.5
    R = _euler_xyz(rot_deg)
    V, Nn, UV, I = [], [], [], []
    for axis in range(3):
        for sign in (+1.0, -1.0):
            n = np.zeros(3)
            n[axis] = sign
            u_ax, v_ax = (axis + 1) % 3, (axis + 2) % 3
            if sign < 0:
                u_ax, v_ax = v_ax, u_ax
            corners = []
            for su, sv in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
                c = np.zeros(3)
                c[axis] = sign * h[axis]
                c[u_ax] = su * h[u_ax]
                c[v_ax] = sv * h[v_ax]
                corners.append(c)
            base = len(V)
            for c, uvc in zip(corners, ((0, 0), (1, 0), (1, 1), (0, 1))):
                V.append(R @ c + np.asarray(pos, np.float64))
                Nn.append(R @ n)
                UV.append(uvc)
            I += [base, base + 1, base + 2, base, base + 2, base + 3]
    return {"positions": np.asarray(V, np.float32), "normals": np.asarray(Nn, np.float32),
            "uvs": np.asarray(UV, np.float32), "indices": np.asarray(I, np.uint32), "material": material_index}


def gltf_material(name, mat):
    """glTF material that the reference's loader + FillMaterialAt turn back into the 11-float `mat`.
    Mesh emission is emissive * |emissive| (Mesh.cpp:476-477 with PathTracer.cpp:76), so the factor
    written is target / sqrt(|target|) with target = emissionColour * emissionStrength."""
    target = np.asarray(mat[5:8], np.float64) * mat[8]
    ln = np.linalg.norm(target)
    em = (target / np.sqrt(ln)) if ln > 0 else np.zeros(3)
    m = {"name": name,
         "pbrMetallicRoughness": {"baseColorFactor": [mat[0], mat[1], mat[2], 1.0],
                                  "metallicFactor": mat[4], "roughnessFactor": mat[3]},
         "emissiveFactor": [float(x) for x in em]}
    ext = {}
    if mat[10] > 0:
        ext["KHR_materials_transmission"] = {"transmissionFactor": mat[10]}
    if mat[9] != 1.5:
        ext["KHR_materials_ior"] = {"ior": mat[9]}
    if ext:
        m["extensions"] = ext
    return m


def write_box_scene_glb(path, boxes):
    """Writes the given Box objects (scenes/presets.py dicts) as one .glb, one material per distinct `mat`."""
    mats, keys, prims = [], {}, []
    for b in boxes:
        key = tuple(b["mat"])
        if key not in keys:
            keys[key] = len(mats)
            mats.append(gltf_material("mat%d" % len(mats), b["mat"]))
        prims.append(box_primitive(b["pos"], b["rot"], b["size"], keys[key]))
    write_glb(path, prims, mats)


def bumpy_sphere_primitive(n_lat, n_lon, radius=1.0, bump=0.08, freq=7, material_index=0):
    """Procedural closed mesh with ~2*n_lat*n_lon triangles and smooth per-vertex normals."""
    th = np.linspace(0.0, np.pi, n_lat + 1)
    ph = np.linspace(0.0, 2 * np.pi, n_lon, endpoint=False)
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    r = radius * (1.0 + bump * np.sin(freq * TH) * np.cos(freq * PH) + 0.5 * bump * np.sin(3 * freq * PH + TH))
    x, y, z = r * np.sin(TH) * np.cos(PH), r * np.cos(TH), r * np.sin(TH) * np.sin(PH)
    V = np.stack([x, y, z], -1).reshape(-1, 3)
    idx = []
    i0 = (np.arange(n_lat)[:, None] * n_lon + np.arange(n_lon)[None, :])
    i1 = (np.arange(n_lat)[:, None] * n_lon + (np.arange(n_lon)[None, :] + 1) % n_lon)
    a, b, c, d = i0, i1, i1 + n_lon, i0 + n_lon
    tri = np.concatenate([np.stack([a, c, b], -1).reshape(-1, 3), np.stack([a, d, c], -1).reshape(-1, 3)])
    # drop degenerate pole triangles
    p = V[tri]
    area = np.linalg.norm(np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), axis=1)
    tri = tri[area > 1e-12]
    # area-weighted vertex normals
    p = V[tri]
    fn = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    Nn = np.zeros_like(V)
    for k in range(3):
        np.add.at(Nn, tri[:, k], fn)
    Nn /= np.maximum(np.linalg.norm(Nn, axis=1, keepdims=True), 1e-20)
    uv = np.stack([PH / (2 * np.pi), TH / np.pi], -1).reshape(-1, 2)
    return {"positions": V.astype(np.float32), "normals": Nn.astype(np.float32), "uvs": uv.astype(np.float32),
            "indices": tri.reshape(-1).astype(np.uint32), "material": material_index}
