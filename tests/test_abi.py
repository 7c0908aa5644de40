"""CPU tests (no GPU): the C-ABI library loads, exports every symbol include/ptamd.h declares, refuses to
compute without a device, and its host-side logic (glTF ingest) agrees with the test-side loader."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import ptamd
from conftest import GOLDEN, ROOT
from scenes import gltf_io, presets


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ptamd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptamd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = ptamd.lib()
    syms = header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(L, s), "libptamd.so does not export " + s
    assert sorted(ptamd.EXPORTS) == syms  # the Python mirror binds exactly the declared surface


def test_image_decode_needs_no_gpu_and_matches_the_test_side_decoder():
    rng = np.random.default_rng(3)
    for shape in [(7, 5, 4), (16, 9, 3), (3, 31, 1)]:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        got = ptamd.decode_image(gltf_io.png_encode(img))
        want = np.full(shape[:2] + (4,), 255, np.uint8)
        want[..., :3] = img[..., :3] if shape[2] >= 3 else img[..., :1]
        if shape[2] == 4:
            want[..., 3] = img[..., 3]
        assert np.array_equal(got, want)
    with pytest.raises(ptamd.PtamdError, match="PNG"):
        ptamd.decode_image(b"not an image at all")
    # one face of the committed sky map (a byte copy of the reference's assets/skyboxes/sky/px.png): size and a few rows
    # against the slow test-side decoder
    data = open(os.path.join(ROOT, "scenes", "sky", "px.png"), "rb").read()
    face = ptamd.decode_image(data)
    assert face.shape == (1024, 1024, 4) and face[..., 3].min() == 255


def test_defaults_match_reference_structs():
    m = ptamd.CMaterial()
    ptamd.lib().ptamd_material_default(C.byref(m))  # RayObject.h:12-19
    assert list(m.albedo) == [1, 1, 1] and m.roughness == 1 and m.metallic == 0
    assert list(m.emission_colour) == [1, 1, 1] and m.emission_strength == 0 and m.ior == 1.5 and m.transmission == 0
    p = ptamd.CPbr()
    ptamd.lib().ptamd_pbr_default(C.byref(p))  # ModelLoader.h:60-84
    assert p.base_color_tex == -1 and p.metallic_factor == 0 and p.roughness_factor == 1 and p.alpha_cutoff == 0.5 and p.ior == 1.5


def test_host_only_calls_and_error_reporting():
    L = ptamd.lib()
    h = C.c_void_p()
    assert L.ptamd_scene_create(C.byref(h)) == 0
    m = ptamd.Material().to_c()
    f3 = lambda *v: (C.c_float * 3)(*v)
    assert L.ptamd_scene_add_sphere(h, f3(0, 0, 0), 1.0, C.byref(m)) == 0
    assert L.ptamd_scene_add_box(h, f3(0, 0, 0), f3(0, 0, 0), f3(1, 1, 1), C.byref(m)) == 1
    assert L.ptamd_scene_object_count(h) == 2
    assert L.ptamd_scene_set_radius(h, 1, 2.0) == -1 and b"not a sphere" in L.ptamd_last_error()
    assert L.ptamd_scene_set_radius(h, 7, 2.0) == -1 and b"out of range" in L.ptamd_last_error()
    rc = L.ptamd_scene_add_mesh_file(h, b"/nonexistent/model.obj", f3(0, 0, 0), f3(0, 0, 0), f3(1, 1, 1))
    assert rc == -2 and b"only supports .glb/.gltf" in L.ptamd_last_error()  # ModelLoader.h:142
    rc = L.ptamd_scene_add_mesh_file(h, b"/nonexistent/model.glb", f3(0, 0, 0), f3(0, 0, 0), f3(1, 1, 1))
    assert rc == -2 and b"Failed to load GLTF model" in L.ptamd_last_error()  # ModelLoader.h:137
    assert L.ptamd_scene_clear(h) == 0 and L.ptamd_scene_object_count(h) == 0
    assert L.ptamd_scene_destroy(h) == 0


def test_no_cpu_fallback_without_device():
    """Without a usable HIP device every compute entry point fails loudly (there is no CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = ptamd.lib()
    assert L.ptamd_init(0) == -4 and b"no CPU fallback" in L.ptamd_last_error()
    h = C.c_void_p()
    assert L.ptamd_film_create(8, 8, C.byref(h)) == -4
    s = C.c_void_p()
    L.ptamd_scene_create(C.byref(s))
    assert L.ptamd_scene_commit(s) == -4
    L.ptamd_scene_destroy(s)


@pytest.mark.parametrize("fname", ["cornell_mesh.glb", "blob.glb", "textured.glb"])
def test_gltf_ingest_matches_test_side_loader(fname):
    """ModelLoader::LoadGLTF replacement (C++ in libptamd) vs the independent numpy reader that feeds the oracle
    (the oracle + numpy reader pair is pinned to the reference bit for bit in test_oracle_golden.py)."""
    path = os.path.join(GOLDEN, fname)
    a = ptamd.load_model(path)
    b = gltf_io.load_gltf(path)
    for k in ("positions", "normals", "uvs", "group"):
        assert np.array_equal(a[k], b[k]), k
    assert len(a["groups"]) == len(b["groups"])
    for ga, gb in zip(a["groups"], b["groups"]):
        for k, vb in gb.items():
            va = ga[k]
            if isinstance(vb, list):
                assert np.allclose(va, np.asarray(vb, np.float32), rtol=0, atol=0), k
            elif isinstance(vb, float):
                assert va == np.float32(vb), k
            else:
                assert va == vb, k
    assert len(a["images"]) == len(b["images"])
    for ia, ib in zip(a["images"], b["images"]):
        assert np.array_equal(ia, ib)


def test_gltf_ingest_textures_indices_and_material_quirks(tmp_path):
    """u16 indices, strided vertex buffers are not produced by our writer, so this builds a .gltf by hand:
    external .bin, 16-bit indices, a material without metallicFactor (tinygltf's default 1.0 wins over the
    loader's 0.0, ModelLoader.h:294), MASK alpha mode, name-merged groups, embedded PNG as data URI."""
    import base64
    import json
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], np.float32)
    uv = np.array([[0, 0], [1, 0], [0, 1], [1, 1]], np.float32)
    idx = np.array([0, 1, 2, 2, 1, 3], np.uint16)
    blob = pos.tobytes() + uv.tobytes() + idx.tobytes()
    (tmp_path / "geo.bin").write_bytes(blob)
    img = (np.arange(4 * 2 * 4) % 256).astype(np.uint8).reshape(2, 4, 4)
    png = gltf_io.png_encode(img)
    doc = {"asset": {"version": "2.0"},
           "buffers": [{"uri": "geo.bin", "byteLength": len(blob)}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 48}, {"buffer": 0, "byteOffset": 48, "byteLength": 32},
                           {"buffer": 0, "byteOffset": 80, "byteLength": 12}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"},
                         {"bufferView": 1, "componentType": 5126, "count": 4, "type": "VEC2"},
                         {"bufferView": 2, "componentType": 5123, "count": 6, "type": "SCALAR"}],
           "images": [{"uri": "data:image/png;base64," + base64.b64encode(png).decode()}],
           "textures": [{"source": 0}],
           "materials": [{"name": "leaf", "alphaMode": "MASK", "alphaCutoff": 0.3,
                          "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}},
                         {"name": "leaf", "emissiveFactor": [0.5, 0.25, 0.0],
                          "extensions": {"KHR_materials_ior": {"ior": 1.33}, "KHR_materials_transmission": {"transmissionFactor": 0.5}}}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1}, "indices": 2, "material": 0}]},
                      {"primitives": [{"attributes": {"POSITION": 0}, "indices": 2, "material": 1}]}]}
    p = tmp_path / "quad.gltf"
    p.write_text(json.dumps(doc))
    a = ptamd.load_model(str(p))
    b = gltf_io.load_gltf(str(p))
    assert a["positions"].shape == (4, 9)
    for k in ("positions", "normals", "uvs", "group"):
        assert np.array_equal(a[k], b[k]), k
    assert list(a["group"]) == [0, 0, 0, 0] and len(a["groups"]) == 1  # merged by name (ModelLoader.h:268-283)
    g = a["groups"][0]
    assert g["metallic_factor"] == 1.0 and g["roughness_factor"] == 1.0
    # the second primitive's material overwrote the group (alpha mode back to opaque, cutoff kept)
    assert g["alpha_mode"] == 0 and abs(g["alpha_cutoff"] - 0.3) < 1e-7 and g["base_color_tex"] == 0
    assert abs(g["ior"] - 1.33) < 1e-6 and g["transmission_factor"] == 0.5 and g["emissive_factor"][:2] == [0.5, 0.25]
    assert np.all(a["normals"] == 0)
    assert len(a["images"]) == 1 and np.array_equal(a["images"][0], img) and np.array_equal(b["images"][0], img)


def test_gltf_rejects_what_the_reference_rejects(tmp_path):
    import json
    L = ptamd.lib()
    base = {"asset": {"version": "2.0"}, "buffers": [{"uri": "data:application/octet-stream;base64,AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAABAAIA", "byteLength": 42}],
            "bufferViews": [{"buffer": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 6}],
            "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"},
                          {"bufferView": 1, "componentType": 5123, "count": 3, "type": "SCALAR"}]}
    cases = [({"attributes": {"POSITION": 0}, "indices": 1, "mode": 1}, b"Only TRIANGLES are supported"),
             ({"attributes": {}, "indices": 1}, b"Missing POSITION"),
             ({"attributes": {"POSITION": 0}}, b"Indexed geometry required"),
             ({"attributes": {"POSITION": 1}, "indices": 1}, b"POSITION must be float vec3")]
    for i, (prim, msg) in enumerate(cases):
        doc = dict(base, meshes=[{"primitives": [prim]}])
        p = tmp_path / ("bad%d.gltf" % i)
        p.write_text(json.dumps(doc))
        h = C.c_void_p()
        assert L.ptamd_model_load(str(p).encode(), C.byref(h)) == -2
        assert msg in L.ptamd_last_error(), L.ptamd_last_error()


def test_mesh_description_is_validated():
    """Null arrays behind non-zero counts are refused instead of dereferenced."""
    L = ptamd.lib()
    h = C.c_void_p()
    assert L.ptamd_scene_create(C.byref(h)) == 0
    d = ptamd.CMeshDesc()
    P = (C.c_float * 9)(0, 0, 0, 1, 0, 0, 0, 1, 0)
    d.num_faces = 1
    d.positions = C.cast(P, C.c_void_p)
    f3 = lambda *v: (C.c_float * 3)(*v)
    d.num_groups = 2
    assert L.ptamd_scene_add_mesh(h, C.byref(d), f3(0, 0, 0), f3(0, 0, 0), f3(1, 1, 1)) == -1 and b"no group array" in L.ptamd_last_error()
    d.num_groups = 0
    d.num_images = 1
    assert L.ptamd_scene_add_mesh(h, C.byref(d), f3(0, 0, 0), f3(0, 0, 0), f3(1, 1, 1)) == -1 and b"no image array" in L.ptamd_last_error()
    d.num_images = 0
    assert L.ptamd_scene_add_mesh(h, C.byref(d), f3(0, 0, 0), f3(0, 0, 0), f3(1, 1, 1)) == 0
    assert L.ptamd_scene_destroy(h) == 0
