"""Loader options that go beyond the reference (it ignores glTF nodes, ModelLoader.h:204-206, and rejects
anything but .glb/.gltf, ModelLoader.h:139-142). No reference output exists for them: the expected values are
computed independently here with numpy. Host code only, no GPU needed."""
import numpy as np
import pytest

import ptamd
from scenes import gltf_io


def quat_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def trs(t=(0, 0, 0), q=(0, 0, 0, 1), s=(1, 1, 1)):
    M = np.eye(4)
    M[:3, :3] = quat_matrix(q) @ np.diag(s)
    M[:3, 3] = t
    return M


def soup(prim):
    idx = prim["indices"].reshape(-1, 3)
    return prim["positions"][idx].astype(np.float64), prim["normals"][idx].astype(np.float64)


def apply(M, P, N):
    P2 = P @ M[:3, :3].T + M[:3, 3]
    N2 = N @ np.linalg.inv(M[:3, :3])  # rows times inverse = inverse-transpose applied to column vectors
    N2 /= np.linalg.norm(N2, axis=-1, keepdims=True)
    return P2, N2


def test_node_transforms_hierarchy_trs_matrix_and_instancing(tmp_path):
    a = gltf_io.box_primitive((0, 0, 0), (0, 0, 0), (1, 2, 3), 0)
    b = gltf_io.bumpy_sphere_primitive(6, 8, radius=0.5, material_index=1)
    mats = [gltf_io.gltf_material("a", [0.8, 0.2, 0.2, 0.5, 0.0, 1, 1, 1, 0.0, 1.5, 0.0]),
            gltf_io.gltf_material("b", [0.2, 0.8, 0.2, 0.5, 0.0, 1, 1, 1, 0.0, 1.5, 0.0])]
    h = np.sqrt(0.5)
    root = trs(t=(1, 2, 3), q=(0, h, 0, h), s=(2, 2, 2))           # 90 degrees about y, uniform scale
    child = trs(t=(0, 1, 0), q=(h, 0, 0, h), s=(1, 0.5, 2))        # non-uniform scale under a rotation
    other = np.array([[0, -1, 0, 4], [1, 0, 0, 0], [0, 0, 3, -2], [0, 0, 0, 1]], float)
    nodes = [{"translation": [1, 2, 3], "rotation": [0, h, 0, h], "scale": [2, 2, 2], "mesh": 0, "children": [1]},
             {"translation": [0, 1, 0], "rotation": [h, 0, 0, h], "scale": [1, 0.5, 2], "mesh": 1},
             {"matrix": [float(v) for v in other.T.reshape(-1)], "mesh": 1},   # column-major, second instance of mesh 1
             {"mesh": 0}]                                                       # not in the scene: must not appear
    path = str(tmp_path / "nodes.glb")
    gltf_io.write_glb(path, [a, b], mats, nodes=nodes, scene_nodes=[0, 2])
    flat = ptamd.load_model(path)                       # the reference's behaviour: meshes as stored, once each
    assert flat["positions"].shape[0] == 12 + len(b["indices"]) // 3
    Pa, Na = soup(a)
    Pb, Nb = soup(b)
    assert np.array_equal(flat["positions"][:12].reshape(-1, 3, 3), Pa.astype(np.float32))
    got = ptamd.load_model(path, node_transforms=True)
    nb = len(b["indices"]) // 3
    assert got["positions"].shape[0] == 12 + 2 * nb
    want = [apply(root, Pa, Na), apply(root @ child, Pb, Nb), apply(other, Pb, Nb)]
    P = np.concatenate([w[0] for w in want]).reshape(-1, 9)
    N = np.concatenate([w[1] for w in want]).reshape(-1, 9)
    assert np.abs(got["positions"] - P).max() <= 1e-5
    assert np.abs(got["normals"] - N).max() <= 1e-5
    assert list(got["group"]) == [0] * 12 + [1] * (2 * nb)
    assert [g["name"] for g in got["groups"]] == ["a", "b"]


def test_obj_reader(tmp_path):
    (tmp_path / "m.mtl").write_text(
        "newmtl red\nKd 0.8 0.1 0.1\nPr 0.3\nPm 0.0\n"
        "newmtl lamp\nKd 0 0 0\nKe 2 2 1\nNi 1.33\nd 0.25\nNs 98\n")
    (tmp_path / "m.obj").write_text(
        "# quad + triangle\nmtllib m.mtl\n"
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\n"
        "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
        "vn 0 0 1\n"
        "usemtl red\nf 1/1/1 2/2/1 3/3/1 4/4/1\n"
        "usemtl lamp\nf -1 -5 -4\n"
        "usemtl red\nf 1//1 3//1 5//1\n")
    with pytest.raises(ptamd.PtamdError, match="only supports .glb/.gltf"):
        ptamd.load_model(str(tmp_path / "m.obj"))          # the reference's behaviour stays the default
    m = ptamd.load_model(str(tmp_path / "m.obj"), allow_obj=True)
    assert m["positions"].shape == (4, 9)
    V = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    assert np.array_equal(m["positions"].reshape(-1, 3, 3), V[[[0, 1, 2], [0, 2, 3], [4, 0, 1], [0, 2, 4]]])
    assert np.array_equal(m["uvs"][0], np.array([0, 0, 1, 0, 1, 1], np.float32)) and np.all(m["uvs"][2:] == 0)
    assert np.all(m["normals"][:2].reshape(-1, 3) == (0, 0, 1))
    n = np.cross(V[0] - V[4], V[1] - V[4]); n /= np.linalg.norm(n)  # face without normals: geometric normal
    assert np.allclose(m["normals"][2].reshape(3, 3), n, atol=1e-6)
    assert list(m["group"]) == [0, 0, 1, 0] and [g["name"] for g in m["groups"]] == ["red", "lamp"]
    red, lamp = m["groups"]
    assert np.allclose(red["base_color_factor"][:3], (0.8, 0.1, 0.1)) and abs(red["roughness_factor"] - 0.3) < 1e-6 and red["metallic_factor"] == 0
    assert np.allclose(lamp["emissive_factor"], (2, 2, 1)) and abs(lamp["ior"] - 1.33) < 1e-6
    assert abs(lamp["transmission_factor"] - 0.75) < 1e-6 and abs(lamp["roughness_factor"] - np.sqrt(2 / 100)) < 1e-6


def test_load_flags_are_validated():
    L = ptamd.lib()
    import ctypes as C
    h = C.c_void_p()
    assert L.ptamd_model_load_ex(b"x.glb", 8, C.byref(h)) == -1 and b"unknown load flag" in L.ptamd_last_error()


def test_material_pointing_at_a_missing_image_is_refused(tmp_path):
    prim = gltf_io.box_primitive((0, 0, 0), (0, 0, 0), (1, 1, 1), 0)
    mat = gltf_io.gltf_material("m", [0.8, 0.2, 0.2, 0.5, 0.0, 1, 1, 1, 0.0, 1.5, 0.0])
    mat["pbrMetallicRoughness"]["baseColorTexture"] = {"index": 0}
    path = str(tmp_path / "bad.gltf")
    # a texture whose source image index is past the (empty) image list
    import json, struct
    gltf_io.write_glb(str(tmp_path / "ok.glb"), [prim], [mat])
    raw = open(tmp_path / "ok.glb", "rb").read()
    n = struct.unpack_from("<I", raw, 12)[0]
    doc = json.loads(raw[20:20 + n])
    blob = raw[20 + n + 8:]
    doc["textures"] = [{"source": 3}]
    import base64
    doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(blob).decode()
    open(path, "w").write(json.dumps(doc))
    with pytest.raises(ptamd.PtamdError, match="image that the file does not hold"):
        ptamd.load_model(path)


def _doc_and_blob(tmp_path):
    import base64
    import json
    import struct
    prim = gltf_io.box_primitive((0, 0, 0), (0, 0, 0), (1, 1, 1), 0)
    mat = gltf_io.gltf_material("m", [0.8, 0.2, 0.2, 0.5, 0.0, 1, 1, 1, 0.0, 1.5, 0.0])
    gltf_io.write_glb(str(tmp_path / "ok.glb"), [prim], [mat])
    raw = open(tmp_path / "ok.glb", "rb").read()
    n = struct.unpack_from("<I", raw, 12)[0]
    doc = json.loads(raw[20:20 + n])
    blob = raw[20 + n + 8:]
    doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(blob).decode()
    return doc


@pytest.mark.parametrize("edit", [
    lambda d: d["accessors"][0].__setitem__("count", 2 ** 62),                       # count * stride wraps around 2^64
    lambda d: (d["bufferViews"][0].__setitem__("byteStride", 2 ** 63), d["accessors"][0].__setitem__("count", 3)),
    lambda d: d["bufferViews"][0].__setitem__("byteOffset", -8),
    lambda d: d["accessors"][0].__setitem__("byteOffset", 1e300),
    lambda d: d["accessors"][0].__setitem__("count", 2.5),
    lambda d: d["accessors"][0].__setitem__("bufferView", -1),
    lambda d: d["accessors"][0].__setitem__("count", 1 << 40),                        # valid number, far past the buffer
    lambda d: d["meshes"][0]["primitives"][0].__setitem__("indices", 1e20),
    lambda d: d["meshes"][0]["primitives"][0]["attributes"].__setitem__("POSITION", "zero"),
])
def test_hostile_numbers_in_a_model_file_are_refused(tmp_path, edit):
    """Counts, offsets, strides and indices of a file are validated before any pointer arithmetic: negative, huge,
    fractional or non-numeric values raise a clean error (the reference gets the same protection from tinygltf)."""
    import json
    doc = _doc_and_blob(tmp_path)
    edit(doc)
    path = str(tmp_path / "bad.gltf")
    open(path, "w").write(json.dumps(doc))
    with pytest.raises(ptamd.PtamdError):
        ptamd.load_model(path)


def test_deeply_nested_json_and_short_png_header_are_refused(tmp_path):
    import base64
    import json
    import struct
    import zlib
    path = str(tmp_path / "deep.gltf")
    open(path, "w").write("[" * 200000)
    with pytest.raises(ptamd.PtamdError, match="nesting"):
        ptamd.load_model(path)
    # a PNG whose IHDR chunk is shorter than its 13 bytes
    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">II", 4, 4)) + chunk(b"IEND", b"")
    doc = _doc_and_blob(tmp_path)
    doc["images"] = [{"uri": "data:image/png;base64," + base64.b64encode(png).decode()}]
    path = str(tmp_path / "png.gltf")
    open(path, "w").write(json.dumps(doc))
    with pytest.raises(ptamd.PtamdError, match="PNG"):
        ptamd.load_model(path)


def test_jpeg_images_decode_to_the_reference_texels(tmp_path):
    """JPEG-encoded glTF images (the reference takes them through stb_image): thirteen files covering baseline and
    progressive coding, 4:4:4 / 4:2:2 / 4:2:0, ragged sizes, one-pixel rows and columns, grey, CMYK, restart intervals
    and optimised tables must decode to exactly the RGBA texels the reference's loader holds (tests/golden/
    jpeg_cases.npz, written by make_golden.py from the reference itself), as must the maps of textured_jpeg.glb."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "jpeg_cases.npz"))
    n = sum(1 for k in g.files if k.startswith("file_"))
    assert n == 13
    prim = gltf_io.box_primitive((0, 0, 0), (0, 0, 0), (1, 1, 1), 0)
    mat = gltf_io.gltf_material("m", [1, 1, 1, 1.0, 0.0, 1, 1, 1, 0.0, 1.5, 0.0])
    path = str(tmp_path / "cases.glb")
    gltf_io.write_glb(path, [prim], [mat], [g["file_%d" % i].tobytes() for i in range(n)])
    got = ptamd.load_model(path)["images"]
    assert len(got) == n
    for i in range(n):
        want = g["texels_%d" % i]
        assert want.shape[2] == 4 and got[i].shape == want.shape and np.array_equal(got[i], want), i
    side = np.load(os.path.join(GOLDEN, "textured_jpeg.images.npz"))
    got = ptamd.load_model(os.path.join(GOLDEN, "textured_jpeg.glb"))["images"]
    assert len(got) == 3
    for i in range(3):
        assert np.array_equal(got[i], side["image_%d" % i])
    # truncated and corrupt files end in a clean error or a partial picture, never a crash
    raw = g["file_4"].tobytes()
    for cut in (3, 20, len(raw) // 2, len(raw) - 2):
        gltf_io.write_glb(path, [prim], [mat], [raw[:cut]])
        try:
            ptamd.load_model(path)
        except ptamd.PtamdError as e:
            assert "JPEG" in str(e) or "image" in str(e)


def test_png_flavours_decode_to_the_reference_bytes():
    """Everything stb_image (the reference's decoder, through tinygltf) accepts as PNG: bit depths 1 / 2 / 4 / 16, palettes
    with and without transparency, colour-key transparency, grey + alpha, Adam7 interlacing. 21 files (written by
    tests/golden/make_golden.py; png_cases.npz keeps them with the bytes the reference's ModelLoader held after loading
    each). The texel bytes must be identical. A 16-bit file reaches the reference's sampler as 16-bit samples read as
    bytes (tiny_gltf.h:2645-2661, ModelLoader.h:190-197, Mesh.cpp:31-34): what it can index is the first width * height * 4
    bytes of that buffer, and that is what the loader must return."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "png_cases.npz"))
    n = sum(1 for k in g.files if k.startswith("file_"))
    assert n == 21
    sixteen = 0
    for i in range(n):
        w, h, c = (int(v) for v in g["size_%d" % i])
        held = g["bytes_%d" % i]
        assert c == 4 and len(held) in (w * h * 4, w * h * 8)
        sixteen += len(held) == w * h * 8
        got = ptamd.decode_image(g["file_%d" % i].tobytes())
        assert got.shape == (h, w, 4) and np.array_equal(got.reshape(-1), held[:w * h * 4]), i
    assert sixteen == 5
    # and through the model loader
    prim = gltf_io.box_primitive((0, 0, 0), (0, 0, 0), (1, 1, 1), 0)
    mat = gltf_io.gltf_material("m", [1, 1, 1, 1.0, 0.0, 1, 1, 1, 0.0, 1.5, 0.0])
    import tempfile
    path = os.path.join(tempfile.mkdtemp(prefix="ptpng_"), "cases.glb")
    gltf_io.write_glb(path, [prim], [mat], [g["file_%d" % i].tobytes() for i in (0, 8, 17)])
    imgs = ptamd.load_model(path)["images"]
    for k, i in enumerate((0, 8, 17)):
        assert np.array_equal(imgs[k].reshape(-1), g["bytes_%d" % i][:imgs[k].size])
    # truncated and corrupted files end in a clean error or an image, never a crash
    raw = g["file_11"].tobytes()
    rng = np.random.default_rng(5)
    for trial in range(200):
        m = bytearray(raw[:int(rng.integers(20, len(raw) + 1))] if trial % 2 else raw)
        for _ in range(int(rng.integers(1, 4))):
            m[int(rng.integers(8, len(m)))] = int(rng.integers(0, 256))
        try:
            ptamd.decode_image(bytes(m))
        except ptamd.PtamdError as e:
            assert "image" in str(e) or "PNG" in str(e)


def test_jpeg_header_that_claims_more_than_the_file_holds_is_refused():
    """A forged frame header: a small file whose SOF claims 23170 x 23170 pixels would make the decoder allocate
    gigabytes of component planes before reading a single coefficient. It is refused from the header alone."""
    import os
    from conftest import GOLDEN
    raw = bytearray(np.load(os.path.join(GOLDEN, "jpeg_cases.npz"))["file_0"].tobytes())
    sof = next(i for i in range(len(raw) - 1) if raw[i] == 0xFF and raw[i + 1] in (0xC0, 0xC1, 0xC2))
    import struct
    raw[sof + 5:sof + 9] = struct.pack(">HH", 23170, 23170)  # height, width
    with pytest.raises(ptamd.PtamdError, match="larger than the file"):
        ptamd.decode_image(bytes(raw))


def test_mutated_model_files_never_crash_the_loader(tmp_path):
    """Byte-level mutations of the committed .glb files (flipped bytes anywhere: header, JSON, accessors' data, PNG and
    JPEG streams; truncations): every load either succeeds or raises a clean error. Runs in a child process, so a crash
    of the native loader fails this test instead of ending the session."""
    import subprocess
    import sys
    script = r'''
import os, sys
import numpy as np
sys.path.insert(0, "path-tracer_amd"); sys.path.insert(0, ".")
import ptamd
tmp = sys.argv[1]
rng = np.random.default_rng(2024)
ok = bad = 0
for name in ("cornell_mesh", "textured", "textured_jpeg", "blob"):
    raw = np.frombuffer(open("tests/golden/%s.glb" % name, "rb").read(), np.uint8)
    for trial in range(120):
        m = raw.copy()
        kind = trial % 4
        if kind == 0:    # a few flipped bytes anywhere
            idx = rng.integers(0, len(m), int(rng.integers(1, 9)))
            m[idx] = rng.integers(0, 256, len(idx))
        elif kind == 1:  # flipped bytes in the JSON chunk
            n = int(np.frombuffer(m[12:16].tobytes(), "<u4")[0])
            idx = 20 + rng.integers(0, n, int(rng.integers(1, 5)))
            m[idx] = rng.integers(32, 127, len(idx))
        elif kind == 2:  # truncated
            m = m[:int(rng.integers(1, len(m)))]
        else:            # a run of bytes overwritten
            a = int(rng.integers(0, len(m) - 64)); m[a:a + int(rng.integers(4, 64))] = int(rng.integers(0, 256))
        path = os.path.join(tmp, "m.glb")
        m.tofile(path)
        try:
            ptamd.load_model(path); ok += 1
        except ptamd.PtamdError:
            bad += 1
print("loaded", ok, "refused", bad)
assert ok > 0 and bad > 0
'''
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", script, str(tmp_path)], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
