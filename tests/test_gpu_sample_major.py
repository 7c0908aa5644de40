"""The sample-major render kernel (launches of 16 samples per pixel and more) against
the pixel-major kernel, which adds a pixel's samples in the reference's order (Film::AddSample, Film.cpp:34-46, called once
per RayTraceParallel pass, main.cpp:17-28). Both run the same paths -- the random stream is keyed by (pixel, sample) -- so
ray counts agree and the films differ by float rounding: the sample-major kernel adds the exactly rounded sum of a
launch's samples in one float addition, the pixel-major kernel rounds after every sample. Tolerance: 1e-5 relative (128
sequential additions of like-signed values drift by at most 128 * 2^-24 = 8e-6) on at least 99.5 % of the pixels; the two
kernels are separate compilations of the same source (multiply-adds may be fused differently), so on glass a last-bit
difference in a direction now and then flips a discrete choice (an edge, total internal reflection) and one sample of a
pixel takes another path: those pixels must still agree in the mean, and the ray totals to 1e-4."""
import numpy as np
import pytest

from scenes import presets
from test_extensions import empty_scene, gradient_faces

pytestmark = pytest.mark.gpu


def both(gpu, scene, spp, depth, spp_begin=0, rect=None, rr=False, seed=5, tiles=None):
    """(film of one sample-major launch, film of pixel-major launches of < 16 samples each, stats of both)."""
    pt, cam = gpu.build_scene(scene)
    w, h = scene["film"]
    fs, fp = gpu.Film(w, h), gpu.Film(w, h)
    gpu.Render(cam, pt, fs, spp, depth, spp_begin=spp_begin, rect=rect, russian_roulette=rr, seed=seed, count=True, tiles=tiles)
    ss = gpu.RenderStats()
    rays = 0
    b = spp_begin
    while b < spp_begin + spp:
        n = min(12, spp_begin + spp - b)
        gpu.Render(cam, pt, fp, n, depth, spp_begin=b, rect=rect, russian_roulette=rr, seed=seed, count=True, tiles=tiles)
        rays += gpu.RenderStats()["rays"]
        b += n
    return fs, fp, ss, rays


def mixed_blob(w, h):
    return presets.mixed("tests/golden/blob.glb", w, h, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0))


def agree(a, b):
    close = np.isclose(a, b, rtol=1e-5, atol=1e-6).all(axis=-1)
    assert close.mean() >= 0.995, (float(close.mean()), float(np.abs(a - b).max()))
    assert abs(float(a.mean()) - float(b.mean())) <= 2e-4 * float(b.mean()), (float(a.mean()), float(b.mean()))


@pytest.mark.parametrize("name", ["mesh_cornell", "sphere_cornell", "glass_mirror", "box_cornell"])
def test_sample_major_equals_pixel_major(gpu, name):
    scene = {"mesh_cornell": lambda: presets.mesh_cornell("tests/golden/cornell_mesh.glb", 100, 54),
             "sphere_cornell": lambda: presets.sphere_cornell(90, 70), "glass_mirror": lambda: presets.glass_mirror(64, 61),
             "box_cornell": lambda: presets.box_cornell(72, 40)}[name]()
    assert gpu.build_scene(scene)[0].Info()["lds_resident"] == 1
    depth = 12 if name == "glass_mirror" else 8
    fs, fp, ss, rays = both(gpu, scene, 130, depth, spp_begin=7)
    a, b = fs.Accum(), fp.Accum()
    assert np.array_equal(fs.Samples(), fp.Samples()) and int(fs.Samples().min()) == 130
    agree(a, b)
    assert float(b.mean()) > 1.0  # 130 samples of a lit room
    w, h = scene["film"]
    assert ss["samples"] == w * h * 130 and abs(ss["rays"] - rays) <= 1e-4 * rays, (ss["rays"], rays)
    # deterministic: a second launch gives the same bits
    pt, cam = gpu.build_scene(scene)
    f2 = gpu.Film(w, h)
    gpu.Render(cam, pt, f2, 130, depth, spp_begin=7, seed=5)
    assert np.array_equal(a.view(np.uint32), f2.Accum().view(np.uint32))


@pytest.mark.parametrize("spp", [16, 23])
def test_sample_major_at_its_smallest_launches(gpu, spp, monkeypatch):
    """The kernel takes launches from 16 samples on: four tickets per lane at 16, a ragged last round at 23. Against the
    pixel-major kernel as above, on an LDS-resident scene (with its walks cut) and on one read from global memory (with its
    walks deferred); a launch of 15 samples still adds its samples one by one (bit-identical to three launches of 5)."""
    monkeypatch.setenv("PTAMD_WALK_BUDGET", "24")
    monkeypatch.setenv("PTAMD_WALK_DEFER", "20")
    for scene in (presets.mesh_cornell("tests/golden/cornell_mesh.glb", 100, 54), mixed_blob(96, 60)):
        fs, fp, ss, rays = both(gpu, scene, spp, 8, spp_begin=3)
        assert np.array_equal(fs.Samples(), fp.Samples()) and int(fs.Samples().min()) == spp
        agree(fs.Accum(), fp.Accum())
        w, h = scene["film"]
        assert ss["samples"] == w * h * spp and abs(ss["rays"] - rays) <= 1e-4 * rays, (ss["rays"], rays)
        pt, cam = gpu.build_scene(scene)
        one, three = gpu.Film(w, h), gpu.Film(w, h)
        gpu.Render(cam, pt, one, 15, 8, seed=5)
        for k in range(3):
            gpu.Render(cam, pt, three, 5, 8, spp_begin=5 * k, seed=5)
        assert np.array_equal(one.Accum().view(np.uint32), three.Accum().view(np.uint32))


def test_sample_major_rect_roulette_and_sky(gpu):
    """A ragged rectangle, Russian roulette and a sky map behind the open front (camera rays that leave the scene take the
    whole-pixel shortcut of the kernel), sample ranges that add up over launches."""
    scene = dict(presets.glass_mirror(80, 60), skybox={"faces": gradient_faces(32, seed=3), "intensity": 1.3})
    scene["camera"] = dict(scene["camera"], pos=(2.78, 2.73, -14.0), fov=80.0)  # far outside: most camera rays miss the room
    fs, fp, ss, rays = both(gpu, scene, 96, 20, spp_begin=0, rect=(3, 5, 77, 58), rr=True)
    a, b = fs.Accum(), fp.Accum()
    assert np.array_equal(fs.Samples(), fp.Samples())
    assert (fs.Samples()[5:58, 3:77] == 96).all() and fs.Samples().sum() == 96 * 74 * 53
    agree(a, b)
    assert abs(ss["rays"] - rays) <= 1e-4 * rays, (ss["rays"], rays)
    # two launches of 64 + 64 samples against one of 128: float rounding of one addition apart
    pt, cam = gpu.build_scene(scene)
    f1, f2 = gpu.Film(80, 60), gpu.Film(80, 60)
    gpu.Render(cam, pt, f1, 128, 20, russian_roulette=True)
    gpu.Render(cam, pt, f2, 64, 20, russian_roulette=True)
    gpu.Render(cam, pt, f2, 64, 20, spp_begin=64, russian_roulette=True)
    assert np.allclose(f1.Accum(), f2.Accum(), rtol=3e-7, atol=1e-7)  # the same kernel: the same paths
    # an empty scene: every pixel is the sky texel, 70 times
    sky = dict(empty_scene(40, 24, (10, 30, 0)), skybox={"faces": gradient_faces(16, seed=1), "intensity": 1.0})
    pt, cam = gpu.build_scene(sky)
    f70, f1 = gpu.Film(40, 24), gpu.Film(40, 24)
    gpu.Render(cam, pt, f70, 70, 4)
    gpu.Render(cam, pt, f1, 1, 4)
    assert np.allclose(f70.Accum(), 70.0 * f1.Accum(), rtol=2e-7)


@pytest.mark.parametrize("wide", [False, True])
def test_sample_major_on_meshes_in_global_memory(gpu, monkeypatch, wide):
    """Meshes too large for LDS are walked from global memory by the whole wave together (binary nodes, or eight-wide
    quantised nodes): the sample-major kernel's lanes enter every walk with a ray, or as helpers once the tile's tickets are
    gone. A rectangle launch with ragged edges and one rank's interleaved tiles, with Russian roulette."""
    monkeypatch.setenv("PTAMD_WIDE_MIN_FACES", "1" if wide else "100000000")
    w, h = 150, 84
    scene = presets.mixed("tests/golden/blob.glb", w, h, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0))
    info = gpu.build_scene(scene)[0].Info()
    assert info["lds_resident"] == 0 and (info["wide_nodes"] > 0) == wide
    fs, fp, ss, rays = both(gpu, scene, 70, 6, spp_begin=3, rect=(2, 1, 147, 83))
    agree(fs.Accum(), fp.Accum())
    assert np.array_equal(fs.Samples(), fp.Samples()) and fs.Samples().sum() == 70 * 145 * 82
    assert ss["samples"] == 70 * 145 * 82 and abs(ss["rays"] - rays) <= 1e-4 * rays, (ss["rays"], rays)
    assert float(fp.Accum().mean()) > 0.5
    # one of three ranks' 32-pixel tiles (ragged at the right and bottom edges), with roulette
    fs, fp, ss, rays = both(gpu, scene, 64, 9, rr=True, tiles=(32, 1, 3))
    agree(fs.Accum(), fp.Accum())
    assert np.array_equal(fs.Samples(), fp.Samples()) and 0 < fs.Samples().sum() < 64 * w * h
    assert abs(ss["rays"] - rays) <= 1e-4 * rays, (ss["rays"], rays)
    # all three ranks together cover the film once, and equal the rectangle launch bit for bit (the same kernel, the same paths)
    pt, cam = gpu.build_scene(scene)
    ft, fr = gpu.Film(w, h), gpu.Film(w, h)
    for rank in range(3):
        gpu.Render(cam, pt, ft, 64, 9, russian_roulette=True, seed=5, tiles=(32, rank, 3))
    gpu.Render(cam, pt, fr, 64, 9, russian_roulette=True, seed=5)
    assert (ft.Samples() == 64).all()
    assert np.array_equal(ft.Accum().view(np.uint32), fr.Accum().view(np.uint32))


def test_sample_major_tiles_from_lds(gpu):
    """Tile launches of the LDS-resident build: ranks' tiles add up to the rectangle launch bit for bit."""
    scene = presets.mesh_cornell("tests/golden/cornell_mesh.glb", 100, 54)
    pt, cam = gpu.build_scene(scene)
    ft, fr = gpu.Film(100, 54), gpu.Film(100, 54)
    for rank in range(2):
        gpu.Render(cam, pt, ft, 80, 8, seed=2, tiles=(16, rank, 2))
    gpu.Render(cam, pt, fr, 80, 8, seed=2)
    assert (ft.Samples() == 80).all()
    assert np.array_equal(ft.Accum().view(np.uint32), fr.Accum().view(np.uint32))


@pytest.mark.parametrize("name", ["mesh_cornell", "two_meshes"])
def test_walk_cut_changes_the_schedule_not_the_film(gpu, name, monkeypatch):
    """In an LDS-resident scene whose last object is a mesh, the sample-major kernel can cut a wave's walk of that mesh when
    few of its lanes are still walking and let the unfinished lanes go on in the wave's next search. A lane's walk is the
    same sequence of steps wherever it is cut, so the film, the ray total and the shaded hits must be bit-identical without
    the cut (PTAMD_WALK_BUDGET=0), with the usual one (24 of 64 lanes) and with one so eager that most walks are cut many
    times; only the wave-level trip counts may move. `two_meshes`: a lane that resumes walks the mesh in front of the last
    one again (more lane-level work, the same hits)."""
    if name == "mesh_cornell":
        scene = presets.mesh_cornell("tests/golden/cornell_mesh.glb", 96, 56)
    else:
        scene = presets.mesh_cornell("tests/golden/cornell_mesh.glb", 64, 40)
        first = dict(scene["objects"][0], path="tests/golden/emitter_panel.glb", pos=(0.4, 0.2, -0.3), rot=(10.0, 30.0, 5.0), scale=(0.8, 0.8, 0.8))
        scene["objects"] = [first] + scene["objects"]
    pt, cam = gpu.build_scene(scene)
    assert pt.Info()["lds_resident"] == 1
    w, h = scene["film"]
    films, stats = [], []
    for cut in ("0", "24", "56"):
        monkeypatch.setenv("PTAMD_WALK_BUDGET", cut)
        f = gpu.Film(w, h)
        gpu.Render(cam, pt, f, 70, 8, seed=11, count=True)
        films.append(f.Accum())
        stats.append(gpu.RenderStats())
    for f, s in zip(films[1:], stats[1:]):
        assert np.array_equal(films[0].view(np.uint32), f.view(np.uint32))
        for k in ("samples", "rays", "shade_lanes"):
            assert s[k] == stats[0][k], (k, s[k], stats[0][k])
        for k in ("nodes", "tris"):
            assert s[k] == stats[0][k] if name == "mesh_cornell" else s[k] >= stats[0][k], (k, s[k], stats[0][k])
    assert float(films[0].mean()) > 0.5
    if name == "mesh_cornell":  # the cut packs the node loop: fewer wave-level node trips for the same lane-level visits
        assert stats[1]["node_trips"] < 0.8 * stats[0]["node_trips"], (stats[1]["node_trips"], stats[0]["node_trips"])
        assert stats[2]["ray_trips"] > stats[1]["ray_trips"] > stats[0]["ray_trips"]


def test_walk_cut_is_measured_once_per_commit_and_leaves_the_film_alone(gpu, monkeypatch):
    """Without the variable the first launch of 16 samples and more on a scene that is one LDS-resident mesh measures whether
    the cut pays (two bands of the frame, with and without) and keeps the answer until the next commit; launches that
    cannot use it, the entry point that allocates nothing and small films measure nothing."""
    monkeypatch.delenv("PTAMD_WALK_BUDGET", raising=False)
    scene = presets.mesh_cornell("tests/golden/cornell_mesh.glb", 1280, 720)
    pt, cam = gpu.build_scene(scene)
    unknown = 0xffffffff
    assert pt.Info()["walk_cut"] == unknown
    few = gpu.Film(1280, 720)
    gpu.Render(cam, pt, few, 8, 8, seed=3)          # under 16 samples: the pixel-major kernel, nothing to measure
    assert pt.Info()["walk_cut"] == unknown
    a = gpu.Film(1280, 720)
    gpu.Render(cam, pt, a, 64, 8, seed=3)
    chosen = pt.Info()["walk_cut"]
    assert chosen in (0, 28)
    print("walk cut chosen for the mesh Cornell at 1280x720:", chosen)
    monkeypatch.setenv("PTAMD_WALK_BUDGET", "28" if chosen == 0 else "0")
    b = gpu.Film(1280, 720)
    gpu.Render(cam, pt, b, 64, 8, seed=3)           # the other choice: same film
    assert np.array_equal(a.Accum().view(np.uint32), b.Accum().view(np.uint32))
    assert pt.Info()["walk_cut"] == chosen          # the variable overrides, it does not overwrite
    monkeypatch.delenv("PTAMD_WALK_BUDGET")
    mesh = pt.GetRayObjects()[0]
    mesh.SetRotation((0.0, 1.0, 0.0)); pt.Commit()  # a measurement that fewer than 32 launches have used survives a moved object
    assert pt.Info()["walk_cut"] == chosen          # (a session that moves something before every frame measures once in 32 frames)
    scratch = gpu.Film(1280, 720)
    for _ in range(32):
        gpu.Render(cam, pt, scratch, 64, 8, seed=3)
    assert pt.Info()["walk_cut"] == chosen
    mesh.SetRotation((0.0, 2.0, 0.0)); pt.Commit()  # 32 launches later a move asks for a new measurement at the next eligible launch
    assert pt.Info()["walk_cut"] == unknown
    gpu.Render(cam, pt, gpu.Film(1280, 720), 64, 8, seed=3)
    assert pt.Info()["walk_cut"] in (0, 28)
    small, cam_s = gpu.build_scene(presets.mesh_cornell("tests/golden/cornell_mesh.glb", 96, 56))
    gpu.Render(cam_s, small, gpu.Film(96, 56), 64, 8, seed=3)
    assert small.Info()["walk_cut"] == 0            # a small film: not worth the launches


@pytest.mark.parametrize("name", ["blob_among_primitives", "roulette_sky_tiles", "textured_alpha_mask", "dragon_field_wide_nodes"])
def test_deferred_walk_changes_the_schedule_not_the_film(gpu, name, monkeypatch, dragon_field_glb):
    """A scene read from global memory whose only mesh is its last object: the sample-major kernel can let the lanes whose
    rays meet the mesh's bounds wait until enough of them make the shared walk worth its trips (PTAMD_WALK_DEFER = the
    number of waiting lanes that starts a walk; 0 = every search walks the mesh, as before). A ray's closest hit does not
    depend on when its walk runs or on how many rays walk beside it (ties go to the lower triangle index), so the film, the
    ray total and the shaded hits must be bit-identical for every threshold: 1 = the bounds test alone decides who walks,
    20 and 40 = the values the host chooses between, 64 = a walk only when every lane of the wave waits or has nothing else
    to do. Node visits fall (rays that miss the bounds no longer enter, helpers speculate less); triangle tests depend on
    the order in which a wave finds its hits and may move by a little either way."""
    rr, tiles = False, None
    if name == "blob_among_primitives":
        scene, (w, h), spp = mixed_blob(160, 96), (160, 96), 70
    elif name == "roulette_sky_tiles":
        scene = dict(mixed_blob(128, 96), skybox={"faces": gradient_faces(16, seed=2), "intensity": 1.1})
        scene["objects"] = scene["objects"][:4] + scene["objects"][5:]  # a wall less: paths reach the sky
        (w, h), spp, rr, tiles = (128, 96), 64, True, 2
    elif name == "textured_alpha_mask":  # the walk's own alpha test (Mesh.cpp:196-233) and the textured hit fill behind a deferred walk
        scene = presets.mixed("tests/golden/textured.glb", 96, 96, mesh_pos=(2.7, 1.6, 2.6), mesh_rot=(0.0, 25.0, 0.0), mesh_scale=(1.6, 1.6, 1.6))
        (w, h), spp = (96, 96), 64
    else:
        from scenes import dragon
        scene, (w, h), spp = dragon.config3_scene(dragon_field_glb, 96, 54), (96, 54), 64
    pt, cam = gpu.build_scene(scene)
    info = pt.Info()
    assert info["lds_resident"] == 0 and (info["wide_nodes"] > 0) == (name == "dragon_field_wide_nodes")
    films, stats = [], []
    for lanes in ("0", "1", "20", "40", "64"):
        monkeypatch.setenv("PTAMD_WALK_DEFER", lanes)
        f = gpu.Film(w, h)
        tot = {}
        for rank in range(tiles or 1):
            gpu.Render(cam, pt, f, spp, 8, seed=11, count=True, russian_roulette=rr, tiles=(16, rank, tiles) if tiles else None)
            for k, v in gpu.RenderStats().items():
                tot[k] = tot.get(k, 0) + v
        assert (f.Samples() == spp).all()
        films.append(f.Accum())
        stats.append(tot)
    for f, s in zip(films[1:], stats[1:]):
        assert np.array_equal(films[0].view(np.uint32), f.view(np.uint32))
        for k in ("samples", "rays", "shade_lanes"):
            assert s[k] == stats[0][k], (k, s[k], stats[0][k])
        assert s["nodes"] <= stats[0]["nodes"] and s["tris"] <= 1.01 * stats[0]["tris"]
    assert float(films[0].mean()) > 0.05
    if name == "blob_among_primitives":  # fuller walks: fewer wave-level node trips and steal rounds for the mesh
        assert stats[2]["node_trips"] < 0.75 * stats[0]["node_trips"], (stats[2]["node_trips"], stats[0]["node_trips"])
        assert stats[2]["steal_rounds"] < stats[0]["steal_rounds"]
        assert stats[4]["ray_trips"] > stats[2]["ray_trips"] > stats[0]["ray_trips"]  # ... for more trips of the bounce loop


def test_deferred_walk_is_chosen_by_measurement(gpu, monkeypatch):
    """Without the variable the first launch of 16 samples and more on such a scene measures the plain schedule against
    walks deferred until 20 and until 40 lanes wait and keeps the fastest (ptamd_scene_info.walk_cut), once per commit; a
    scene with a second mesh has no deferred walk to choose."""
    monkeypatch.delenv("PTAMD_WALK_DEFER", raising=False)
    scene = mixed_blob(1280, 720)
    pt, cam = gpu.build_scene(scene)
    unknown = 0xffffffff
    assert pt.Info()["lds_resident"] == 0 and pt.Info()["walk_cut"] == unknown
    a = gpu.Film(1280, 720)
    gpu.Render(cam, pt, a, 64, 8, seed=3)
    chosen = pt.Info()["walk_cut"]
    assert chosen in (0, 20, 40)
    print("deferred walk chosen for the blob among primitives at 1280x720:", chosen)
    monkeypatch.setenv("PTAMD_WALK_DEFER", "0" if chosen else "20")
    b = gpu.Film(1280, 720)
    gpu.Render(cam, pt, b, 64, 8, seed=3)           # another choice: same film
    assert np.array_equal(a.Accum().view(np.uint32), b.Accum().view(np.uint32))
    assert pt.Info()["walk_cut"] == chosen
    monkeypatch.delenv("PTAMD_WALK_DEFER")
    sphere = pt.GetRayObjects()[7]
    m = sphere.GetMaterial(); m.roughness = 0.3; sphere.SetMaterial(m); pt.Commit()  # a material edit leaves the walks as they are
    assert pt.Info()["walk_cut"] == chosen
    sphere.SetPosition((1.1, 0.6, 1.4)); pt.Commit()                                # a moved object, few launches since: kept
    assert pt.Info()["walk_cut"] == chosen
    two = mixed_blob(1280, 720)
    two["objects"].append(dict(two["objects"][-1], pos=(1.2, 1.0, 3.4)))
    pt2, cam2 = gpu.build_scene(two)
    gpu.Render(cam2, pt2, gpu.Film(1280, 720), 64, 8, seed=3)
    assert pt2.Info()["walk_cut"] == unknown        # nothing to measure: every search walks both meshes


def test_deferred_walk_on_random_scenes(gpu, monkeypatch):
    """The bounds test that decides which rays wait for the shared walk must never turn away a ray that the walk would
    have given a hit: random rooms (rotated and unrotated boxes, spheres, some of glass or mirror) in front of a mesh with
    a random placement -- tiny, larger than the room (camera and every ray origin inside its bounds), flattened by an
    uneven scale, partly behind the camera -- render the same film bit for bit with every search walking the mesh and
    with walks deferred until 20 lanes wait or until every lane waits."""
    rng = np.random.default_rng(20261112)
    meshes = ["tests/golden/blob.glb", "tests/golden/textured.glb", "tests/golden/cornell_mesh.glb"]
    for case in range(12):
        objs = presets.room_without_light()[: int(rng.integers(0, 7))]
        objs.append({"type": "box", "pos": (2.78, 5.44, 2.8), "rot": (0.0, 0.0, 0.0), "size": (3.0, 0.006, 3.0),
                     "mat": presets.material(albedo=(0, 0, 0), emission_colour=(1.0, 0.9, 0.8), emission_strength=4.0)})
        for _ in range(int(rng.integers(0, 4))):
            kind = rng.integers(0, 3)
            mat = presets.material(albedo=tuple(rng.uniform(0.2, 1.0, 3)), roughness=float(rng.choice([0.0, 0.3, 1.0])),
                                   metallic=float(rng.choice([0.0, 1.0])), transmission=float(rng.choice([0.0, 0.0, 1.0])))
            pos = tuple(rng.uniform(0.5, 5.0, 3))
            if kind == 0:
                objs.append({"type": "sphere", "pos": pos, "radius": float(rng.uniform(0.2, 1.0)), "mat": mat})
            else:
                objs.append({"type": "box", "pos": pos, "rot": tuple(rng.uniform(-40, 40, 3)) if kind == 1 else (0.0, 0.0, 0.0),
                             "size": tuple(rng.uniform(0.3, 1.5, 3)), "mat": mat})
        size = float(rng.choice([0.02, 0.4, 1.0, 2.5, 8.0]))
        scale = tuple(size * rng.uniform(0.3, 1.0, 3)) if case % 3 else (size, size, size)
        path = meshes[case % 3]
        if path.endswith("cornell_mesh.glb"):
            scale = tuple(0.3 * v for v in scale)  # authored at room size
        objs.append({"type": "mesh", "path": path, "pos": tuple(rng.uniform(0.5, 5.0, 3)), "rot": tuple(rng.uniform(-180, 180, 3)), "scale": scale})
        scene = {"film": (64, 48), "camera": presets.default_camera(), "background": (0.3, 0.35, 0.4), "objects": objs}
        monkeypatch.setenv("PTAMD_NO_LDS_GEOM", "1")  # the small meshes too are read from global memory here
        pt, cam = gpu.build_scene(scene)
        assert pt.Info()["lds_resident"] == 0
        films = []
        for lanes in ("0", "20", "64"):
            monkeypatch.setenv("PTAMD_WALK_DEFER", lanes)
            f = gpu.Film(64, 48)
            gpu.Render(cam, pt, f, 64, 6, seed=case, russian_roulette=bool(case & 1))
            films.append(f.Accum())
        assert np.isfinite(films[0]).all() and float(films[0].mean()) > 0.0
        for f in films[1:]:
            assert np.array_equal(films[0].view(np.uint32), f.view(np.uint32)), case
