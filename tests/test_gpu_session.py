"""GPU test of the progressive-session path (the per-frame editors of Sphere.cpp:72-88, Box.cpp:94-110, Mesh.cpp:259-287
imply a scene that changes between frames): a commit after edits of transforms, sizes, materials or group factors must
rewrite table rows only -- the geometry buffers of a 1 M-triangle mesh keep their addresses -- and the next frame must
equal, bit for bit, the frame of a scene built from scratch with the edited values."""
import numpy as np
import pytest

from scenes import dragon, presets

pytestmark = pytest.mark.gpu


def build(gpu, glb, sphere_pos, sphere_mat, box_size, mesh_rot, rough):
    pt = gpu.PathTracer()
    scene = dragon.config3_scene(glb, 96, 54)
    objs = {}
    for i, o in enumerate(scene["objects"]):
        if o["type"] == "box":
            b = gpu.Box("Box")
            b.mPosition, b.mRotation, b.mSize = tuple(o["pos"]), tuple(o["rot"]), tuple(o["size"])
            if i == 6:  # the ceiling light
                b.mSize = box_size
                objs["light"] = b
            b.mMaterial = gpu.Material.from_list(o["mat"])
            pt.AddRayObject(b)
        else:
            m = gpu.Mesh(o["path"])
            m.mPosition, m.mRotation, m.mScale = tuple(o["pos"]), mesh_rot, tuple(o["scale"])
            pt.AddRayObject(m)
            objs["mesh"] = m
    s = gpu.Sphere("S", sphere_pos, 0.5)
    s.mMaterial = sphere_mat
    pt.AddRayObject(s)
    objs["sphere"] = s
    pt.Commit()
    if rough is not None:
        g = dict(gpu.load_model(glb)["groups"][0])
        g.pop("name")
        g["roughness_factor"] = rough
        objs["mesh"].SetGroupMaterial(0, g)
        pt.Commit()
    c = scene["camera"]
    cam = gpu.Camera(scene["film"], c["pos"], c["rot"])
    cam.SetFov(c["fov"])
    return pt, cam, objs


def frame(gpu, pt, cam):
    film = gpu.Film(96, 54)
    gpu.Render(cam, pt, film, 2, 5, seed=17)
    return film.Accum()


def test_edits_rewrite_rows_and_leave_geometry_in_place(gpu, dragon_field_glb):
    mat0 = gpu.Material(albedo=(0.8, 0.3, 0.2), roughness=0.4, metallic=0.0)
    light0 = (1.3, 0.006, 1.05)
    pt, cam, o = build(gpu, dragon_field_glb, (2.7, 2.2, 0.4), mat0, light0, (-90.0, 0.0, 180.0), None)
    before = pt.Info()
    assert before["triangles"] == 1_000_000 and before["geometry_uploads"] == 1 and before["wide_nodes"] > 0
    first = frame(gpu, pt, cam)
    # the edits an interactive session makes between two frames
    mat1 = gpu.Material(albedo=(0.2, 0.6, 0.9), roughness=0.1, metallic=1.0)
    o["sphere"].SetPosition((3.4, 2.6, 0.9)); o["sphere"].SetRadius(0.5); o["sphere"].SetMaterial(mat1)
    o["light"].SetSize((2.0, 0.006, 1.6))
    o["mesh"].SetRotation((-90.0, 0.0, 170.0))
    g = dict(gpu.load_model(dragon_field_glb)["groups"][0]); g.pop("name"); g["roughness_factor"] = 0.25
    o["mesh"].SetGroupMaterial(0, g)
    with pytest.raises(gpu.PtamdError, match="uncommitted"):
        frame(gpu, pt, cam)
    pt.Commit()
    after = pt.Info()
    assert after["geometry_uploads"] == 1 and after["row_updates"] >= 5
    for k in ("d_nodes", "d_tris", "d_shade"):
        assert after[k] == before[k] and after[k]
    assert after["last_bvh_seconds"] == before["last_bvh_seconds"] and after["last_commit_seconds"] < 0.05 < before["last_commit_seconds"]
    edited = frame(gpu, pt, cam)
    assert not np.array_equal(edited, first)
    pt2, cam2, _ = build(gpu, dragon_field_glb, (3.4, 2.6, 0.9), mat1, (2.0, 0.006, 1.6), (-90.0, 0.0, 170.0), 0.25)
    assert np.array_equal(edited.view(np.uint32), frame(gpu, pt2, cam2).view(np.uint32))
    print("full commit of the 1 M-triangle scene %.3f s (BVH build %.3f s), commit after five edits %.6f s"
          % (before["last_commit_seconds"], before["last_bvh_seconds"], after["last_commit_seconds"]))
    # a new object changes the table sizes: everything is uploaded again, and the trees are not rebuilt
    pt.AddRayObject(gpu.Sphere("T", (1.0, 3.0, 1.0), 0.3))
    pt.Commit()
    third = pt.Info()
    assert third["geometry_uploads"] == 2 and third["last_bvh_seconds"] == 0.0 and third["objects"] == before["objects"] + 1


def test_counting_renders_of_two_scenes_do_not_share_tallies(gpu):
    """Each scene counts into a buffer of its own; ptamd_render_stats reports the counting render that ran last, and
    says so when that scene is gone."""
    a, cam_a = gpu.build_scene(presets.sphere_cornell(48, 32))
    b, cam_b = gpu.build_scene(presets.mesh_cornell("tests/golden/cornell_mesh.glb", 40, 24))
    fa, fb = gpu.Film(48, 32), gpu.Film(40, 24)
    gpu.Render(cam_a, a, fa, 3, 4, count=True)
    gpu.Render(cam_b, b, fb, 2, 4, count=True)
    sb = gpu.RenderStats()
    assert sb["samples"] == 40 * 24 * 2 and sb["nodes"] > 0
    gpu.Render(cam_a, a, fa, 3, 4, spp_begin=3, count=True)
    sa = gpu.RenderStats()
    assert sa["samples"] == 48 * 32 * 3 and sa["nodes"] == 0  # spheres and boxes only: no tree was walked
    a.close()
    with pytest.raises(gpu.PtamdError, match="PTAMD_FLAG_COUNT"):
        gpu.RenderStats()


def test_leading_run_of_unrotated_boxes_changes_no_image_and_follows_edits(gpu):
    """The unrotated boxes at the head of the object list are tested with one shared set of direction reciprocals
    (closest_hit in pt_device.h): the same values in the same order as the per-box test (Box.cpp:16-54), so the film is
    bit-identical to that of the same scene behind a leading sphere that no ray meets (which ends the run before it
    starts), in both render kernels. A rotation edit shortens or restores the run at the next commit."""
    scene = presets.box_cornell(96, 64)
    pt, cam = gpu.build_scene(scene)
    assert pt.Info()["leading_boxes"] == 7  # six walls and the light; the two inner blocks are rotated
    behind = dict(scene, objects=[{"type": "sphere", "pos": (0.0, -500.0, 0.0), "radius": 0.25, "mat": scene["objects"][0]["mat"]}] + scene["objects"])
    pt0, cam0 = gpu.build_scene(behind)
    assert pt0.Info()["leading_boxes"] == 0
    for spp in (3, 64):  # pixel-major and sample-major launches
        a, b = gpu.Film(96, 64), gpu.Film(96, 64)
        gpu.Render(cam, pt, a, spp, 6, seed=5)
        gpu.Render(cam0, pt0, b, spp, 6, seed=5)
        assert np.array_equal(a.Accum().view(np.uint32), b.Accum().view(np.uint32))
    hits = pt.ClosestHits(np.load("tests/golden/box_cornell.npz")["rays"])
    hits0 = pt0.ClosestHits(np.load("tests/golden/box_cornell.npz")["rays"])
    assert np.array_equal(hits.view(np.uint32), hits0.view(np.uint32))
    # the spheres that follow the walls share d.d and its reciprocal (Sphere.cpp:36,46-48): same film as behind a leading sphere,
    # where neither run exists
    room = presets.sphere_cornell(80, 60)
    rs, cs = gpu.build_scene(room)
    r0, c0 = gpu.build_scene(dict(room, objects=behind["objects"][:1] + room["objects"]))
    assert rs.Info()["leading_boxes"] == 7 and r0.Info()["leading_boxes"] == 0
    for spp in (3, 64):
        a, b = gpu.Film(80, 60), gpu.Film(80, 60)
        gpu.Render(cs, rs, a, spp, 5, seed=9)
        gpu.Render(c0, r0, b, spp, 5, seed=9)
        assert np.array_equal(a.Accum().view(np.uint32), b.Accum().view(np.uint32))
    # an edit that rotates the second wall ends the run after the first box (one box is no run); undoing it restores it
    first = pt.GetRayObjects()[1]
    first.SetRotation((0.0, 0.0, 3.0)); pt.Commit()
    assert pt.Info()["leading_boxes"] == 0
    first.SetRotation((0.0, 0.0, 0.0)); pt.Commit()
    assert pt.Info()["leading_boxes"] == 7
    c = gpu.Film(96, 64)
    gpu.Render(cam, pt, c, 3, 6, seed=5)
    d = gpu.Film(96, 64)
    gpu.Render(cam0, pt0, d, 3, 6, seed=5)
    assert np.array_equal(c.Accum().view(np.uint32), d.Accum().view(np.uint32))
