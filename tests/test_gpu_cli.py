"""GPU test of the headless C++ path: ptamd_cli (host classes of host/ptamd.hpp over the C ABI) renders the
same film as the Python mirror, and its PNG / PFM / raw outputs hold what Film::ResolveToRGBA8 / AverageAt /
Accum return (PNG rows flipped as Window::SaveImagePNG does, Window.cpp:173-184)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from scenes import gltf_io, presets

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "path-tracer_amd", "ptamd_cli")


def test_cli_matches_python_mirror(gpu, tmp_path):
    assert os.path.exists(CLI), "ptamd_cli is not built (make -C path-tracer_amd)"
    scene = presets.mixed("tests/golden/textured.glb", 72, 56, mesh_pos=(2.7, 1.6, 2.6), mesh_rot=(0.0, 25.0, 0.0), mesh_scale=(1.4, 1.4, 1.4))
    sp = str(tmp_path / "scene.txt")
    presets.write_scene_txt(scene, sp)
    w, h = scene["film"]
    pt, cam = gpu.build_scene(scene)
    film = gpu.Film(w, h)
    gpu.Render(cam, pt, film, 5, 4, seed=77)
    accum, rgba, avg = film.Accum(), film.ResolveToRGBA8(), film.Average()
    for extra in ([], ["--frames"]):
        out = subprocess.run([CLI, sp, "--out", str(tmp_path / "img.png"), "--pfm", str(tmp_path / "img.pfm"), "--raw", str(tmp_path / "raw"),
                              "--spp", "5", "--depth", "4", "--seed", "77"] + extra, check=True, capture_output=True, text=True, cwd=ROOT)
        info = json.loads(out.stdout.strip().splitlines()[-1])
        assert info["width"] == w and info["height"] == h and info["objects"] == len(scene["objects"])
        a = np.fromfile(str(tmp_path / "raw.accum.f32"), np.float32).reshape(h, w, 3)
        s = np.fromfile(str(tmp_path / "raw.samples.u32"), np.uint32).reshape(h, w)
        assert np.array_equal(a.view(np.uint32), accum.view(np.uint32)) and np.all(s == 5)
        png = gltf_io.png_decode(open(tmp_path / "img.png", "rb").read())
        assert png.shape == (h, w, 4) and np.array_equal(png[::-1], rgba)
        raw = open(tmp_path / "img.pfm", "rb").read()
        header = ("PF\n%d %d\n-1.0\n" % (w, h)).encode()
        assert raw.startswith(header)
        assert np.array_equal(np.frombuffer(raw[len(header):], np.float32).reshape(h, w, 3), avg)


def test_cli_reports_loader_errors(gpu, tmp_path):
    sp = tmp_path / "bad.txt"
    sp.write_text("film 8 8\ncamera 0 0 0 0 0 0 45\nmesh /nonexistent/model.glb 0 0 0 0 0 0 1 1 1\n")
    out = subprocess.run([CLI, str(sp), "--out", str(tmp_path / "x.png")], capture_output=True, text=True)
    assert out.returncode == 1 and "Failed to load GLTF model" in out.stderr


def test_cli_skybox_and_roulette(gpu, tmp_path):
    """The two options outside the reference reach the C++ host classes: `skybox` scene record and --rr."""
    from test_extensions import gradient_faces
    scene = presets.glass_mirror(64, 48)
    faces = gradient_faces(16, seed=3)
    sp = str(tmp_path / "scene.txt")
    presets.write_scene_txt(scene, sp)
    faces.tofile(str(tmp_path / "sky.rgba8"))
    with open(sp, "a") as f:
        f.write("skybox %s 16 0.75\n" % (tmp_path / "sky.rgba8"))
    pt, cam = gpu.build_scene(dict(scene, skybox={"faces": faces, "intensity": 0.75}))
    film = gpu.Film(64, 48)
    gpu.Render(cam, pt, film, 3, 9, seed=12, russian_roulette=True)
    subprocess.run([CLI, sp, "--raw", str(tmp_path / "raw"), "--spp", "3", "--depth", "9", "--seed", "12", "--rr"],
                   check=True, capture_output=True, text=True, cwd=ROOT)
    a = np.fromfile(str(tmp_path / "raw.accum.f32"), np.float32).reshape(48, 64, 3)
    assert np.array_equal(a.view(np.uint32), film.Accum().view(np.uint32))


def test_cli_obj_and_node_transforms(gpu, tmp_path):
    """--obj and --node-transforms reach the loader: an OBJ quad lit by an emissive OBJ triangle renders the same
    film as the Python mirror given the same options, and differs from the untransformed load of a glTF whose
    only node is translated."""
    (tmp_path / "m.mtl").write_text("newmtl wall\nKd 0.7 0.7 0.7\nnewmtl lamp\nKd 0 0 0\nKe 3 3 3\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 5\nv 5.5 0 5\nv 5.5 5.5 5\nv 0 5.5 5\nv 2 5 3\nv 3.5 5 3\nv 2.7 5 4.5\n"
                                    "usemtl wall\nf 1 2 3 4\nusemtl lamp\nf 5 6 7\n")
    cam = presets.default_camera()
    scene_txt = "film 48 40\ncamera %s %s %g\nmesh %s 0 0 0 0 0 0 1 1 1\n" % (
        " ".join("%.9g" % v for v in cam["pos"]), " ".join("%.9g" % v for v in cam["rot"]), cam["fov"], tmp_path / "m.obj")
    (tmp_path / "s.txt").write_text(scene_txt)
    bad = subprocess.run([CLI, str(tmp_path / "s.txt"), "--raw", str(tmp_path / "r0")], capture_output=True, text=True)
    assert bad.returncode == 1 and "only supports .glb/.gltf" in bad.stderr
    subprocess.run([CLI, str(tmp_path / "s.txt"), "--raw", str(tmp_path / "r1"), "--spp", "4", "--depth", "3", "--obj"], check=True, capture_output=True)
    a = np.fromfile(str(tmp_path / "r1.accum.f32"), np.float32).reshape(40, 48, 3)
    pt = gpu.PathTracer()
    m = gpu.Mesh(str(tmp_path / "m.obj"), allow_obj=True)
    pt.AddRayObject(m)
    pt.Commit()
    c = gpu.Camera((48, 40), cam["pos"], cam["rot"]); c.SetFov(cam["fov"])
    film = gpu.Film(48, 40)
    gpu.Render(c, pt, film, 4, 3)
    assert np.array_equal(film.Accum().view(np.uint32), a.view(np.uint32))
    assert a.max() > 0.5  # the lamp is seen

    prim = gltf_io.box_primitive((0, 0, 0), (0, 0, 0), (1, 1, 1), 0)
    glb = str(tmp_path / "n.glb")
    gltf_io.write_glb(glb, [prim], [gltf_io.gltf_material("m", presets.material(albedo=(0.9, 0.2, 0.2)))],
                      nodes=[{"mesh": 0, "translation": [2.78, 2.73, 2.0], "scale": [2, 2, 2]}], scene_nodes=[0])
    (tmp_path / "s2.txt").write_text(scene_txt.replace(str(tmp_path / "m.obj"), glb))
    outs = []
    for extra in ([], ["--node-transforms"]):
        subprocess.run([CLI, str(tmp_path / "s2.txt"), "--raw", str(tmp_path / "r2"), "--spp", "1", "--depth", "1", "--albedo"] + extra, check=True, capture_output=True)
        outs.append(np.fromfile(str(tmp_path / "r2.accum.f32"), np.float32).reshape(40, 48, 3))
    red = lambda img: int(((img[12:28, 16:32, 0] > 0.3) & (img[12:28, 16:32, 1] < 0.3)).sum())  # centre of the view
    assert red(outs[0]) == 0 and red(outs[1]) > 50  # stored at the origin the box is off-centre; its node moves it in front of the camera
