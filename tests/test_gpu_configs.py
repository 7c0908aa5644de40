"""GPU parity cases at the sizes of BASELINE.json's other configurations, checked through properties that do
not need a full CPU render: a ~1 M-triangle mesh (config 3: BVH in global memory, deep tree), the 4K film of
config 4 with tile sharding, the 32-bounce glass/mirror scene of config 5, and the full 1080p mesh Cornell of
config 2."""
import numpy as np
import pytest

from conftest import lum
from oracle import pyoracle as po
from scenes import gltf_io, presets

pytestmark = pytest.mark.gpu


def soup(prim, group=0):
    idx = prim["indices"].reshape(-1, 3)
    return {"positions": np.ascontiguousarray(prim["positions"][idx].reshape(-1, 9)),
            "normals": np.ascontiguousarray(prim["normals"][idx].reshape(-1, 9)),
            "uvs": np.ascontiguousarray(prim["uvs"][idx].reshape(-1, 6)),
            "group": np.full(len(idx), group, np.int32),
            "groups": [dict(gltf_io.default_pbr(), base_color_factor=[0.7, 0.6, 0.5, 1.0], metallic_factor=0.0, roughness_factor=0.6, name="m")],
            "images": []}


@pytest.fixture(scope="module")
def million_tri_scene():
    prim = gltf_io.bumpy_sphere_primitive(500, 1000, radius=1.0, bump=0.06, freq=9)
    data = soup(prim)
    assert data["positions"].shape[0] > 990_000
    objs = presets.cornell_room_boxes()
    objs.append({"type": "mesh", "data": data, "pos": (2.775, 2.2, 2.2), "rot": (20.0, 30.0, 0.0), "scale": (2.3, 2.0, 2.1)})
    return {"film": (160, 120), "camera": presets.default_camera(), "background": (0.2, 0.2, 0.2), "objects": objs}


def test_million_triangle_mesh_hits_and_samples(gpu, million_tri_scene):
    """Config 3 geometry scale: closest hits against the oracle's median-split BVH (a different tree over the
    same triangles), then same-seed radiance."""
    scene = million_tri_scene
    pt, cam = gpu.build_scene(scene)
    S = po.Scene(scene)
    rays = S.rays().reshape(-1, 6)
    ref = S.closest_hits(rays)
    got, cnt = pt.TraversalCounts(rays)
    on_mesh = (ref[:, 0] == 1) & (ref[:, 12] == np.float32(0.6))  # the mesh's roughness factor
    assert on_mesh.mean() > 0.1
    assert (got[:, 0] == ref[:, 0]).mean() >= 0.999
    same = (got[:, 0] == 1) & (ref[:, 0] == 1) & (np.abs(got[:, 1] - ref[:, 1]) <= 2e-5 * np.abs(ref[:, 1]) + 1e-6)
    assert same.sum() >= 0.997 * (ref[:, 0] == 1).sum()  # silhouettes of ~0.01-unit triangles may pick a neighbour
    sel = same & on_mesh
    assert np.abs(got[sel, 2:5] - ref[sel, 2:5]).max() <= 5e-4
    assert (np.abs(got[sel, 5:8] - ref[sel, 5:8]).max(axis=1) <= 1e-3).mean() >= 0.995
    assert np.array_equal(got[sel, 8], ref[sel, 8])
    assert cnt[on_mesh, 0].mean() < 120 and cnt[:, 0].max() < 1000  # the SAH tree is walked, not brute-forced
    w, h = scene["film"]
    film = gpu.Film(w, h)
    gpu.Render(cam, pt, film, 1, 8, seed=3)
    a = film.Accum()
    oa, _ = S.render(1, 8, seed=3)
    close = np.abs(a - oa).max(axis=-1) <= 1e-3 + 1e-3 * np.abs(oa).max(axis=-1)
    assert close.mean() >= 0.95, close.mean()


def test_config5_depth32_same_seed(gpu):
    """Glass / mirror / rough metal, 32 bounces, no Russian roulette (the reference has none)."""
    scene = presets.glass_mirror(96, 96)
    pt, cam = gpu.build_scene(scene)
    film = gpu.Film(96, 96)
    gpu.Render(cam, pt, film, 2, 32, seed=9, count=True)
    st = gpu.RenderStats()
    assert st["rays"] / st["samples"] > 8  # long specular chains really happen
    a = film.Accum()
    oa, _ = po.Scene(scene).render(2, 32, seed=9)
    close = np.abs(a - oa).max(axis=-1) <= 2e-3 + 2e-3 * np.abs(oa).max(axis=-1)
    assert close.mean() >= 0.93, close.mean()  # refraction chains amplify rounding: a larger share of paths decorrelate
    assert abs(lum(a).mean() - lum(oa).mean()) <= 0.03 * lum(oa).mean()


def test_config4_4k_film_tile_sharding(gpu):
    """3840x2160 film, interleaved 32x32 tiles of 8 'ranks' rendered into one film equal the single launch bit
    for bit; counts are exact."""
    import shard
    w, h = 3840, 2160
    scene = presets.mixed("tests/golden/blob.glb", w, h, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0))
    pt, cam = gpu.build_scene(scene)
    full = gpu.Film(w, h)
    gpu.Render(cam, pt, full, 2, 3, seed=5)
    fa = full.Accum()
    tiled = gpu.Film(w, h)
    total = 0
    for rank in range(8):
        rects = shard.tile_rects(rank, 8, w, h, tile=240)  # coarse tiles keep the launch count of this test low
        for r in rects:
            gpu.Render(cam, pt, tiled, 2, 3, seed=5, rect=r)
            total += (r[2] - r[0]) * (r[3] - r[1])
    assert total == w * h
    assert np.array_equal(tiled.Accum().view(np.uint32), fa.view(np.uint32))
    assert np.array_equal(tiled.Samples(), np.full((h, w), 2, np.uint32))
    # the same split with one launch per rank: 32-pixel tiles (the default of shard.render_sharded), ragged at the
    # film's bottom edge (2160 = 67.5 tiles)
    one = gpu.Film(w, h)
    total = 0
    for rank in range(8):
        total += shard.render_sharded(None, w, h, 2, mode="tile", rank=rank, world=8, tile=32,
                                      render_tiles=lambda b, c, t, ph, st: gpu.Render(cam, pt, one, c, 3, seed=5, spp_begin=b, tiles=(t, ph, st)))
    assert total == w * h * 2
    assert np.array_equal(one.Accum().view(np.uint32), fa.view(np.uint32))
    assert np.array_equal(one.Samples(), np.full((h, w), 2, np.uint32))
    # more ranks than tiles, tile larger than the film, bad arguments
    small = gpu.Film(40, 24)
    for rank in range(3):
        gpu.Render(cam, pt, small, 1, 2, tiles=(64, rank, 3))
    assert np.array_equal(small.Samples(), np.ones((24, 40), np.uint32))
    for bad in [(12, 0, 1), (32, 2, 2), (32, -1, 2), (32, 0, 0)]:
        with pytest.raises(Exception):
            gpu.Render(cam, pt, small, 1, 2, tiles=bad)


def test_config2_full_size_properties(gpu, golden):
    """1920x1080 mesh Cornell at 256 spp: exact sample counts, sample-range additivity, and agreement with the
    reference on the pixels whose centre rays coincide with the reference's 128x72 golden render of the same view
    (1920 / 128 = 15 is odd: golden pixel (x, y) and frame pixel (15x+7, 15y+7) share their primary ray)."""
    w, h = 1920, 1080
    scene = presets.mesh_cornell("tests/golden/cornell_mesh.glb", w, h)
    pt, cam = gpu.build_scene(scene)
    g = golden("mesh_cornell_128x72")
    spp = int(g["spp"])
    film = gpu.Film(w, h)
    gpu.Render(cam, pt, film, spp, 8, count=True)
    st = gpu.RenderStats()
    assert st["samples"] == w * h * spp and 5.5 < st["rays"] / st["samples"] < 7.5
    assert np.array_equal(film.Samples(), np.full((h, w), spp, np.uint32))
    A = film.Average()
    rays = cam.GetRays()[7::15, 7::15]
    assert np.abs(rays - g["rays"]).max() <= 2e-6
    sub = A[7::15, 7::15]
    ra, rb = g["avg_a"].astype(np.float32), g["avg_b"].astype(np.float32)
    self_rmse = float(g["self_rmse"])
    rm = lambda x, y: float(np.sqrt(((x - y) ** 2).mean()))
    assert rm(sub, ra) <= 1.25 * self_rmse and rm(sub, rb) <= 1.25 * self_rmse
    ref_lum = 0.5 * (float(g["mean_lum_a"]) + float(g["mean_lum_b"]))
    assert abs(lum(sub).mean() - ref_lum) <= 0.02 * ref_lum
    f2 = gpu.Film(w, h)
    gpu.Render(cam, pt, f2, 100, 8, spp_begin=0)
    gpu.Render(cam, pt, f2, spp - 100, 8, spp_begin=100)
    # launches of 16 samples and more on this scene sum their samples exactly and add the sum once: a split moves float rounding
    assert np.allclose(f2.Accum()[::7], film.Accum()[::7], rtol=3e-7, atol=1e-7)


def test_quad_primitives_match_separate_triangles(gpu, monkeypatch):
    """Two triangles that form a parallelogram are tested as one primitive. Switching the pairing off must give the
    same closest hits (same t, point, normal, material) and the same per-sample radiance."""
    scene = presets.mesh_cornell("tests/golden/cornell_mesh.glb", 96, 64)
    pt, cam = gpu.build_scene(scene)
    rays = cam.GetRays().reshape(-1, 6)
    rng = np.random.default_rng(5)
    o = (rng.random((20000, 3)) * 5.0 + 0.25).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([rays, np.concatenate([o, d], 1)])
    a, ca = pt.TraversalCounts(rays)
    film = gpu.Film(96, 64)
    gpu.Render(cam, pt, film, 4, 6, seed=8)
    fa = film.Accum()
    monkeypatch.setenv("PTAMD_NO_QUADS", "1")
    pt2, _ = gpu.build_scene(scene)
    b, cb = pt2.TraversalCounts(rays)
    film2 = gpu.Film(96, 64)
    gpu.Render(cam, pt2, film2, 4, 6, seed=8)
    fb = film2.Accum()
    assert ca[:, 1].sum() < 0.85 * cb[:, 1].sum()  # fewer primitive tests: the pairing really happened
    assert np.array_equal(a[:, 0], b[:, 0])
    hit = a[:, 0] == 1
    assert np.abs(a[hit, 1] - b[hit, 1]).max() <= 2e-6 * np.abs(b[hit, 1]).max()
    assert np.abs(a[hit, 2:5] - b[hit, 2:5]).max() <= 2e-5
    assert (np.abs(a[hit, 5:8] - b[hit, 5:8]).max(axis=1) <= 1e-5).mean() >= 0.999  # a ray through a shared edge may pick the neighbour
    assert np.array_equal(a[hit, 9:], b[hit, 9:])
    close = np.abs(fa - fb).max(axis=-1) <= 1e-4 + 1e-4 * np.abs(fb).max(axis=-1)
    assert close.mean() >= 0.99, close.mean()


def test_lds_and_global_traversal_agree(gpu, monkeypatch):
    """Small scenes are traversed from LDS through nodes stored in both ray orders, large ones from the 64-byte
    nodes in global memory with the work-sharing loop. Two different small meshes, one of them instanced twice with
    rotations and non-uniform scales (node and triangle bases other than 0, every sign combination of the ray
    direction), must give the same per-sample radiance on both paths, and the oracle's."""
    def solid(verts, faces, colour):
        v = np.asarray(verts, np.float32)
        tri = v[np.asarray(faces)].reshape(-1, 9)
        e1, e2 = tri[:, 3:6] - tri[:, 0:3], tri[:, 6:9] - tri[:, 0:3]
        n = np.cross(e1, e2)
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        return {"positions": tri, "normals": np.tile(n, 3).astype(np.float32), "uvs": np.zeros((len(tri), 6), np.float32),
                "group": np.zeros(len(tri), np.int32),
                "groups": [dict(gltf_io.default_pbr(), base_color_factor=list(colour) + [1.0], metallic_factor=0.0, name="m")], "images": []}
    octa = solid([(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)],
                 [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)], (0.2, 0.7, 0.3))
    c = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]
    cube = solid(c, [(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (2, 3, 7), (2, 7, 6), (1, 2, 6), (1, 6, 5), (3, 0, 4), (3, 4, 7)],
                 (0.7, 0.4, 0.2))
    objs = presets.cornell_room_boxes()
    objs.append({"type": "mesh", "data": cube, "pos": (1.6, 1.0, 2.6), "rot": (15.0, 40.0, -10.0), "scale": (0.8, 0.5, 0.6)})
    objs.append({"type": "mesh", "data": octa, "pos": (3.7, 1.6, 3.2), "rot": (0.0, 25.0, 65.0), "scale": (0.7, 1.2, 0.6)})
    objs.append({"type": "mesh", "data": cube, "pos": (2.8, 3.6, 3.4), "rot": (50.0, 0.0, 20.0), "scale": (0.4, 0.4, 0.9)})
    scene = {"film": (96, 72), "camera": presets.default_camera(), "objects": objs}
    pt, cam = gpu.build_scene(scene)
    film = gpu.Film(96, 72)
    gpu.Render(cam, pt, film, 4, 6, seed=21, count=True)
    lds = film.Accum()
    monkeypatch.setenv("PTAMD_NO_LDS_GEOM", "1")
    pt2, _ = gpu.build_scene(scene)
    film2 = gpu.Film(96, 72)
    gpu.Render(cam, pt2, film2, 4, 6, seed=21, count=True)
    glob = film2.Accum()
    assert gpu.RenderStats()["steal_rounds"] > 0  # the second render really took the work-sharing kernel
    close = np.abs(lds - glob).max(axis=-1) <= 1e-4 + 1e-4 * np.abs(glob).max(axis=-1)
    assert close.mean() >= 0.99, close.mean()
    oa, _ = po.Scene(scene).render(4, 6, seed=21)
    close = np.abs(lds - oa).max(axis=-1) <= 1e-3 + 1e-3 * np.abs(oa).max(axis=-1)
    assert close.mean() >= 0.95, close.mean()


def test_mid_size_mesh_stays_in_lds(gpu, monkeypatch):
    """A mesh of a few hundred triangles no longer fits the LDS budget of five workgroups per CU but still fits that
    of three, which is faster than reading it from global memory: it must take the LDS kernel (no steal rounds, the
    mark of the work-sharing kernel) and give the same radiance as the global-memory kernel."""
    prim = gltf_io.bumpy_sphere_primitive(10, 14, radius=1.0, bump=0.05, freq=5)
    data = soup(prim)
    assert 250 <= data["positions"].shape[0] <= 300
    objs = presets.cornell_room_boxes()
    objs.append({"type": "mesh", "data": data, "pos": (2.775, 2.0, 2.8), "rot": (10.0, 30.0, 0.0), "scale": (1.6, 1.6, 1.6)})
    scene = {"film": (96, 72), "camera": presets.default_camera(), "objects": objs}
    pt, cam = gpu.build_scene(scene)
    film = gpu.Film(96, 72)
    gpu.Render(cam, pt, film, 4, 6, seed=9, count=True)
    stats = gpu.RenderStats()
    assert stats["steal_rounds"] == 0 and stats["nodes"] > 0
    lds = film.Accum()
    monkeypatch.setenv("PTAMD_NO_LDS_GEOM", "1")
    pt2, _ = gpu.build_scene(scene)
    film2 = gpu.Film(96, 72)
    gpu.Render(cam, pt2, film2, 4, 6, seed=9, count=True)
    assert gpu.RenderStats()["steal_rounds"] > 0
    glob = film2.Accum()
    close = np.abs(lds - glob).max(axis=-1) <= 1e-4 + 1e-4 * np.abs(glob).max(axis=-1)
    assert close.mean() >= 0.99, close.mean()


def test_non_finite_triangles_are_inert(gpu):
    """Triangles with NaN or infinite coordinates can never be hit (Mesh.cpp:418-445: every comparison fails) and must
    not disturb the others: same closest hits and radiance as the mesh without them, on both traversal kernels."""
    for rings, segments in [(5, 6), (24, 40)]:  # LDS-resident / global memory
        good = soup(gltf_io.bumpy_sphere_primitive(rings, segments, radius=1.0, bump=0.05, freq=3))
        bad = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in good.items()}
        n = good["positions"].shape[0]
        extra = good["positions"][:4].copy()
        extra[0, 4] = np.nan; extra[1, 0] = np.inf; extra[2, :] = np.nan; extra[3, 8] = -np.inf
        where = [0, n // 3, n // 2, n]
        for key, rows in (("positions", extra), ("normals", good["normals"][:4]), ("uvs", good["uvs"][:4])):
            a = good[key]
            for i, w in enumerate(where):
                a = np.insert(a, w + i, rows[i], axis=0)
            bad[key] = np.ascontiguousarray(a)
        bad["group"] = np.zeros(n + 4, np.int32)
        out = []
        for data in (good, bad):
            objs = presets.cornell_room_boxes()
            objs.append({"type": "mesh", "data": data, "pos": (2.775, 2.0, 2.8), "rot": (10.0, 30.0, 0.0), "scale": (1.5, 1.5, 1.5)})
            scene = {"film": (64, 48), "camera": presets.default_camera(), "objects": objs}
            pt, cam = gpu.build_scene(scene)
            hits = pt.ClosestHits(cam.GetRays().reshape(-1, 6))
            film = gpu.Film(64, 48)
            gpu.Render(cam, pt, film, 2, 5, seed=4)
            out.append((hits, film.Accum()))
        (ha, fa), (hb, fb) = out
        assert np.array_equal(ha[:, 0], hb[:, 0])
        assert np.isfinite(hb).all() and np.isfinite(fb).all()
        assert np.abs(ha[:, 1:8] - hb[:, 1:8]).max() <= 1e-5
        close = np.abs(fa - fb).max(axis=-1) <= 1e-4 + 1e-4 * np.abs(fa).max(axis=-1)
        assert close.mean() >= 0.99, close.mean()


@pytest.mark.parametrize("faces", [1, 2, 3])
def test_tiny_meshes(gpu, faces, monkeypatch):
    """One triangle (the tree is a single leaf), a quad (one primitive) and three triangles (quad + triangle), with a
    zero-area triangle among them: closest hits as the oracle's, from LDS and from global memory."""
    tri = np.array([[-1, -1, 0, 1, -1, 0, 1, 1, 0], [-1, -1, 0, 1, 1, 0, -1, 1, 0], [1, -1, 0, 2, 0, 0.5, 1, 1, 0]], np.float32)[:faces]
    tri = np.concatenate([tri, np.array([[0.5, 0.5, 1, 0.5, 0.5, 1, 0.5, 0.5, 1]], np.float32)])  # degenerate: never hit
    nrm = np.tile(np.array([0, 0, -1], np.float32), (len(tri), 3))
    data = {"positions": tri, "normals": nrm, "uvs": np.zeros((len(tri), 6), np.float32), "group": np.zeros(len(tri), np.int32),
            "groups": [dict(gltf_io.default_pbr(), base_color_factor=[0.9, 0.5, 0.1, 1.0], metallic_factor=0.0, name="m")], "images": []}
    objs = presets.cornell_room_boxes()
    objs.append({"type": "mesh", "data": data, "pos": (2.775, 2.5, 3.0), "rot": (0.0, 15.0, 0.0), "scale": (1.2, 1.2, 1.2)})
    scene = {"film": (64, 48), "camera": presets.default_camera(), "objects": objs}
    S = po.Scene(scene)
    rays = S.rays()
    ref = S.closest_hits(rays)
    on_mesh = np.all(np.abs(ref[:, 9:12] - np.array([0.9, 0.5, 0.1], np.float32)) < 1e-6, axis=1) & (ref[:, 0] == 1)
    assert on_mesh.sum() > 50
    oa, _ = S.render(2, 4, seed=2)
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("PTAMD_NO_LDS_GEOM", env)
        pt, cam = gpu.build_scene(scene)
        got = pt.ClosestHits(rays)
        assert (got[:, 0] == ref[:, 0]).mean() >= 0.998
        same = (got[:, 0] == 1) & (ref[:, 0] == 1) & (np.abs(got[:, 1] - ref[:, 1]) <= 2e-5 * np.abs(ref[:, 1])) & np.all(got[:, 9:12] == ref[:, 9:12], axis=1)
        assert same.sum() >= 0.995 * (ref[:, 0] == 1).sum()
        assert np.array_equal(got[same, 8], ref[same, 8])
        film = gpu.Film(64, 48)
        gpu.Render(cam, pt, film, 2, 4, seed=2)
        close = np.abs(film.Accum() - oa).max(axis=-1) <= 2e-3 + 2e-3 * np.abs(oa).max(axis=-1)
        assert close.mean() >= 0.95, close.mean()


def test_config3_full_combination(gpu, golden, dragon_field_glb):
    """Config 3 as benchmarked: the 1 M-triangle dragon field in the closed room at 1920x1080, without and with the cubemap
    sky. Exact sample counts; launches of under 16 samples add up bit for bit (the kept camera-ray hit of the one-launch render
    included), larger ones to float rounding;
    the frame agrees with the reference's 96x54 golden render of the same view (mean luminance, and pixel by pixel after
    averaging 20x20 blocks, no further from the mean of the reference's two 256-spp runs than those are from each other)."""
    from scenes import dragon
    w, h = 1920, 1080
    scene = dragon.config3_scene(dragon_field_glb, w, h)
    pt, cam = gpu.build_scene(scene)
    assert pt.Info()["triangles"] == 1_000_000 and pt.Info()["wide_nodes"] > 0
    g = golden("dragon_field")
    spp = 48
    film = gpu.Film(w, h)
    gpu.Render(cam, pt, film, spp, 8, count=True)
    st = gpu.RenderStats()
    assert st["samples"] == w * h * spp and st["steal_rounds"] > 0
    assert np.array_equal(film.Samples(), np.full((h, w), spp, np.uint32))
    f1, f2 = gpu.Film(w, h), gpu.Film(w, h)
    gpu.Render(cam, pt, f1, 15, 8)
    gpu.Render(cam, pt, f2, 1, 8, spp_begin=0)
    gpu.Render(cam, pt, f2, 6, 8, spp_begin=1)
    gpu.Render(cam, pt, f2, 8, 8, spp_begin=7)
    assert np.array_equal(f2.Accum()[::5].view(np.uint32), f1.Accum()[::5].view(np.uint32))
    gpu.Render(cam, pt, f2, spp - 15, 8, spp_begin=15)  # the rest in one sample-major launch: the one-launch film to rounding
    # (the two kernels are separate compilations: a last-bit difference now and then sends one sample of a pixel another way)
    close = np.isclose(f2.Accum()[::5], film.Accum()[::5], rtol=1e-5, atol=1e-6).all(axis=-1)
    assert close.mean() >= 0.995, float(close.mean())
    assert abs(float(f2.Accum().mean()) - float(film.Accum().mean())) <= 2e-4 * float(film.Accum().mean())
    A = film.Average()
    assert np.isfinite(A).all()
    ref = 0.5 * (g["avg_a"].astype(np.float32) + g["avg_b"].astype(np.float32))
    ref_lum = 0.5 * (float(g["mean_lum_a"]) + float(g["mean_lum_b"]))
    assert abs(lum(A).mean() - ref_lum) <= 0.03 * ref_lum
    blocks = A.reshape(54, 20, 96, 20, 3).mean(axis=(1, 3))
    err = np.abs(blocks - ref).max(axis=-1)
    own = np.abs(g["avg_a"].astype(np.float32) - g["avg_b"].astype(np.float32)).max(axis=-1)  # the reference against itself
    assert np.median(err) <= np.median(own) and (err <= 0.15).mean() >= (own <= 0.15).mean(), (np.median(err), (err <= 0.15).mean())
    # with the sky behind the open front: paths that leave the room pick up the cubemap instead of the constant
    m = 16
    sky = np.full((6, m, m, 4), 255, np.uint8); sky[..., 0] = 60; sky[..., 1] = 120
    pt.SetSkybox(sky, 1.5)
    fs = gpu.Film(w, h)
    gpu.Render(cam, pt, fs, 8, 8)
    S = fs.Average()
    assert np.isfinite(S).all() and np.array_equal(fs.Samples(), np.full((h, w), 8, np.uint32))
    assert S[..., 2].mean() > 1.05 * A[..., 2].mean()  # a bluer, brighter exit than the 0.2 grey


def test_camera_rays_at_4k(gpu, golden):
    """Camera::GetRay at 3840x2160 against the reference's rays (tests/golden/camera_4k.npz: three cameras, a lattice of
    rows and columns with the frame's edges): 2e-6, as at the small sizes."""
    from test_oracle_golden import cameras_4k
    g = golden("camera_4k")
    for scene, want in cameras_4k(g):
        scene = dict(scene, objects=presets.sphere_cornell(8, 8)["objects"])
        _, cam = gpu.build_scene(scene)
        rays = cam.GetRays().reshape(2160, 3840, 6)[g["ys"]][:, g["xs"]]
        assert np.abs(rays - want).max() <= 2e-6
        assert np.abs(np.linalg.norm(rays[..., 3:], axis=-1) - 1.0).max() <= 1e-6


def test_wide_nodes_agree_with_binary_nodes(gpu, monkeypatch):
    """Large meshes are walked through eight-wide nodes with 8-bit child boxes, smaller ones through binary nodes. With
    the size threshold taken away (PTAMD_WIDE_MIN_FACES) the wide walk must find the same closest hits and give the same
    per-sample radiance as the binary walk on small scenes too: an alpha-masked, textured mesh; two materials with
    vertex normals; a 270-triangle sphere; an instanced mesh under rotation and non-uniform scale."""
    monkeypatch.setenv("PTAMD_NO_LDS_GEOM", "1")
    sphere = soup(gltf_io.bumpy_sphere_primitive(10, 14, radius=1.0, bump=0.05, freq=5))
    inst = presets.cornell_room_boxes()
    inst.append({"type": "mesh", "data": sphere, "pos": (1.6, 1.2, 2.6), "rot": (15.0, 40.0, -10.0), "scale": (0.8, 0.5, 0.6)})
    inst.append({"type": "mesh", "data": sphere, "pos": (3.7, 2.6, 3.2), "rot": (0.0, 25.0, 65.0), "scale": (0.7, 1.2, 0.6)})
    scenes = [presets.mixed("tests/golden/textured.glb", 96, 96, mesh_pos=(2.7, 1.6, 2.6), mesh_rot=(0.0, 25.0, 0.0), mesh_scale=(1.4, 1.4, 1.4)),
              presets.mixed("tests/golden/blob.glb", 96, 96, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0)),
              {"film": (96, 72), "camera": presets.default_camera(), "objects": inst}]
    rng = np.random.default_rng(12)
    for scene in scenes:
        w, h = scene["film"]
        out = []
        o = (rng.random((4000, 3)) * 4.5 + 0.5).astype(np.float32)
        d = rng.normal(size=(4000, 3)).astype(np.float32)
        extra = np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], 1).astype(np.float32)
        for faces in ("1000000000", "1"):
            monkeypatch.setenv("PTAMD_WIDE_MIN_FACES", faces)
            pt, cam = gpu.build_scene(scene)
            rays = cam.GetRays().reshape(-1, 6)
            hits, cnt = pt.TraversalCounts(np.concatenate([rays, extra]))
            film = gpu.Film(w, h)
            gpu.Render(cam, pt, film, 3, 6, seed=31, count=True)
            out.append((hits, cnt, film.Accum(), gpu.RenderStats()))
            assert (pt.Info()["wide_nodes"] > 0) == (faces == "1")  # the wide walk really ran, or really did not
        (hb, cb, fb, sb), (hw, cw, fw, sw) = out
        assert cw[:, 0].sum() < 0.9 * cb[:, 0].sum()  # fewer node visits, work sharing's speculation included
        assert (hb[:, 0] == hw[:, 0]).mean() >= 0.9995
        hit = (hb[:, 0] == 1) & (hw[:, 0] == 1)
        same = hit & (np.abs(hb[:, 1] - hw[:, 1]) <= 1e-6 * np.abs(hb[:, 1]))
        assert same.sum() >= 0.999 * hit.sum()  # rays through shared edges may settle on the neighbour
        assert np.abs(hb[same, 2:8] - hw[same, 2:8]).max() <= 1e-5
        assert np.array_equal(hb[same, 8:], hw[same, 8:])
        close = np.abs(fb - fw).max(axis=-1) <= 1e-4 + 1e-4 * np.abs(fb).max(axis=-1)
        assert close.mean() >= 0.99, close.mean()
        assert sb["samples"] == sw["samples"] and sw["steal_rounds"] > 0
        # the planned collapse (least summed node area) against the older greedy one: never more nodes, the same hits
        planned = pt.Info()["wide_nodes"]
        monkeypatch.setenv("PTAMD_WIDE_GREEDY", "1")
        pg, _ = gpu.build_scene(scene)
        monkeypatch.delenv("PTAMD_WIDE_GREEDY")
        assert planned <= pg.Info()["wide_nodes"]
        hg, _ = pg.TraversalCounts(np.concatenate([rays, extra]))
        assert (hg[:, 0] == hw[:, 0]).mean() >= 0.9995
        both = (hg[:, 0] == 1) & (hw[:, 0] == 1)
        assert (np.abs(hg[both, 1] - hw[both, 1]) <= 1e-6 * np.abs(hw[both, 1])).mean() >= 0.999


def test_leaf_pass_rule_changes_the_schedule_not_the_image(gpu, monkeypatch):
    """The work-sharing walk holds its leaf passes back for mesh instances that few rays reach (the host's estimate per
    instance; PTAMD_SPARSE_MESHES forces it off or on). Closest hits are settled by an atomic minimum with a fixed tie
    rule, so the two schedules must give the same film bit for bit; on a small mesh among primitives the estimate picks
    the held-back form by itself, which runs fewer (fuller) leaf passes."""
    scene = presets.mixed("tests/golden/blob.glb", 160, 120, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0))
    out = {}
    for mode in ("0", "1", None):
        if mode is None:
            monkeypatch.delenv("PTAMD_SPARSE_MESHES", raising=False)
        else:
            monkeypatch.setenv("PTAMD_SPARSE_MESHES", mode)
        pt, cam = gpu.build_scene(scene)
        film = gpu.Film(160, 120)
        gpu.Render(cam, pt, film, 6, 8, seed=5, count=True)
        out[mode] = (film.Accum(), gpu.RenderStats())
    assert np.array_equal(out["0"][0].view(np.uint32), out["1"][0].view(np.uint32))
    assert np.array_equal(out["0"][0].view(np.uint32), out[None][0].view(np.uint32))
    assert out["1"][1]["tris"] > 0 and out["1"][1]["tri_trips"] < 0.8 * out["0"][1]["tri_trips"]
    assert out[None][1]["tri_trips"] == out["1"][1]["tri_trips"]  # the estimate marks this instance by itself


def test_quads_far_from_the_origin(gpu, monkeypatch):
    """The Cornell mesh moved 10^4 units from the origin, where a float coordinate resolves 10^-3: faces whose fourth
    corner no longer sits within rounding of b + c - a must not be paired, and whatever is paired must give the closest
    hits of the separate triangles."""
    src = gltf_io.load_gltf("tests/golden/cornell_mesh.glb")
    shift = np.array([1.0e4, -7.0e3, 4.0e3], np.float32)
    data = dict(src, positions=(src["positions"].reshape(-1, 3) + shift).reshape(-1, 9).astype(np.float32))
    cam = dict(presets.default_camera())
    cam["pos"] = tuple(float(x) for x in (np.array(cam["pos"], np.float32) + shift))
    scene = {"film": (96, 54), "camera": cam, "objects": [{"type": "mesh", "data": data, "pos": (0, 0, 0), "rot": (0, 0, 0), "scale": (1, 1, 1)}]}
    pt, camera = gpu.build_scene(scene)
    rays = camera.GetRays().reshape(-1, 6)
    a = pt.ClosestHits(rays)
    monkeypatch.setenv("PTAMD_NO_QUADS", "1")
    pt2, _ = gpu.build_scene(scene)
    b = pt2.ClosestHits(rays)
    assert a[:, 0].mean() > 0.5 and (a[:, 0] == b[:, 0]).mean() >= 0.999
    hit = (a[:, 0] == 1) & (b[:, 0] == 1)
    # at these coordinates t itself carries ~1e-3 of rounding from the hit point's position
    assert (np.abs(a[hit, 1] - b[hit, 1]) <= 4e-3).mean() >= 0.999
    assert np.array_equal(a[hit, 9:12], b[hit, 9:12]) or (np.all(a[hit, 9:12] == b[hit, 9:12], axis=1)).mean() >= 0.999
