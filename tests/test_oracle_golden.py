"""CPU tests (no GPU): the C oracle (oracle/pt_oracle.c) against the golden buffers that the reference's
own sources produced (tests/golden/*.npz, made by tests/golden/make_golden.py), and -- when the reference
binary oracle/_ref/ref_oracle exists (build container only) -- against fresh runs of it."""
import numpy as np
import pytest

from conftest import lum, rmse
from oracle import pyoracle as po

NAMES = ["sphere_cornell", "glass_mirror", "box_cornell", "mesh_cornell", "blob_mixed", "textured", "textured_jpeg", "dragon", "dragon_field"]


@pytest.fixture(scope="module")
def oracle_scenes(golden_scenes):
    return {k: po.Scene(v[0]) for k, v in golden_scenes.items()}


@pytest.mark.parametrize("name", NAMES)
def test_camera_rays_bit_exact(name, oracle_scenes, golden):
    """Camera::GetRay (Camera.cpp:38-64): the restatement reproduces the reference bit for bit."""
    g = golden(name)
    r = oracle_scenes[name].rays()
    assert r.shape == g["rays"].shape
    assert np.array_equal(r.view(np.uint32), g["rays"].view(np.uint32))


def cameras_4k(g):
    """(scene dict, expected rays on the lattice) for the cameras of tests/golden/camera_4k.npz."""
    for i in range(3):
        c = g["cam%d" % i]
        cam = {"pos": tuple(map(float, c[0:3])), "rot": tuple(map(float, c[3:6])), "fov": float(c[6]), "near": float(c[7]), "far": float(c[8])}
        yield {"film": (3840, 2160), "camera": cam, "objects": []}, g["rays%d" % i]


def test_camera_rays_at_4k_bit_exact(golden):
    """Camera::GetRay at 3840x2160 for three cameras (default, raised and turned, rolled wide-angle), on a lattice of
    rows and columns that includes the frame's edges: bit for bit the reference's rays."""
    g = golden("camera_4k")
    for scene, want in cameras_4k(g):
        r = po.Scene(scene).rays_at(g["ys"], g["xs"])
        assert np.array_equal(r.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("name", NAMES)
def test_closest_hit_records_bit_exact(name, oracle_scenes, golden):
    """Sphere/Box/Mesh::RayIntersect + the closest-hit loop: t, p, n, frontFace and material, bit for bit."""
    g = golden(name)
    h = oracle_scenes[name].closest_hits(g["rays"]).reshape(g["hits"].shape)
    assert np.array_equal(h.view(np.uint32), g["hits"].view(np.uint32))


@pytest.mark.parametrize("name", NAMES)
def test_albedo_and_depth1_images_bit_exact(name, oracle_scenes, golden):
    """Deterministic render modes (no random draw reaches the film): albedo-only and depth 1."""
    g = golden(name)
    S = oracle_scenes[name]
    a, s = S.render(1, 10, albedo_only=True, threads=4)
    assert np.array_equal(a.view(np.uint32), g["albedo_accum"].view(np.uint32))
    assert np.array_equal(po.resolve(a, s), g["albedo_rgba"])
    a, s = S.render(2, 1, albedo_only=False, threads=4)
    assert np.array_equal(a.view(np.uint32), g["depth1_accum"].view(np.uint32))
    assert np.array_equal(s, g["depth1_samples"])


def test_film_known_answers(golden):
    """Film::AddSample clamp, AverageAt and the four ResolveToRGBA8 variants (Film.cpp:34-104)."""
    g = golden("film_kat")
    import ctypes as C
    L = po.lib()
    n = len(g["values"])
    acc = np.zeros((n, 3), np.float32)
    cnt = np.zeros(n, np.uint32)
    fp = C.POINTER(C.c_float)
    for i in range(n):
        v = np.ascontiguousarray(g["values"][i])
        half = np.ascontiguousarray(v * np.float32(0.5))
        cp = C.cast(cnt[i:].ctypes.data, C.POINTER(C.c_uint32))
        L.po_film_add_sample(acc[i].ctypes.data_as(fp), cp, v.ctypes.data_as(fp))
        L.po_film_add_sample(acc[i].ctypes.data_as(fp), cp, half.ctypes.data_as(fp))
    assert np.array_equal(acc.view(np.uint32), g["accum"].view(np.uint32))
    assert np.array_equal(cnt, g["samples"])
    assert np.array_equal(po.average(acc, cnt).view(np.uint32), g["avg"].view(np.uint32))
    for cs in (0, 1):
        for tm in (0, 1):
            assert np.array_equal(po.resolve(acc, cnt, tm, cs), g["rgba_cs%d_tm%d" % (cs, tm)])


@pytest.mark.parametrize("name", NAMES)
def test_stochastic_render_matches_reference_statistics(name, oracle_scenes, golden):
    """The oracle draws from a counter-based PCG, the reference from an unseedable mt19937, so agreement is
    statistical: RMSE of the clamped-mean HDR buffer against a reference run must not exceed 1.25 x the
    reference's own run-to-run RMSE at the same spp, and mean luminance must agree within 2 %."""
    g = golden(name)
    spp, depth = int(g["spp"]), int(g["depth"])
    a, s = oracle_scenes[name].render(spp, depth)
    A = po.average(a, s)
    ra, rb = g["avg_a"].astype(np.float32), g["avg_b"].astype(np.float32)
    self_rmse = float(g["self_rmse"])
    assert rmse(A, ra) <= 1.25 * self_rmse
    assert rmse(A, rb) <= 1.25 * self_rmse
    ref_lum = 0.5 * (float(g["mean_lum_a"]) + float(g["mean_lum_b"]))
    assert abs(lum(A).mean() - ref_lum) <= 0.02 * ref_lum


def test_rng_is_counter_based_and_uniform():
    L = po.lib()
    assert L.po_pcg_hash(0) == 129708002  # PCG RXS-M-XS of 0 (fixed vector: guards the constants)
    k = L.po_path_key(0x5EED, 1234, 7)
    assert k == L.po_path_key(0x5EED, 1234, 7) and k != L.po_path_key(0x5EED, 1235, 7)
    assert (k >> 32) != 0 and (k >> 32) != (k & 0xFFFFFFFF)  # a 64-bit key: two hash chains
    assert (k >> 32) != (L.po_path_key(0x5EED, 1234, 8) >> 32) and (k >> 32) != (L.po_path_key(0x5EEE, 1234, 7) >> 32)
    u = np.array([L.po_rand01(L.po_path_key(1, p, s), b, d) for p in range(40) for s in range(10) for b in range(4) for d in range(5)])
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005


def test_split_invariance_of_the_oracle(oracle_scenes):
    """Any split over sample ranges or rectangles gives the same film (what multi-GPU sharding relies on)."""
    S = oracle_scenes["sphere_cornell"]
    w, h = S.w, S.h_px
    full_a, full_s = S.render(6, 4)
    a = np.zeros((h, w, 3), np.float32); s = np.zeros((h, w), np.uint32)
    S.render(2, 4, accum=a, samples=s, spp_begin=0)
    S.render(4, 4, accum=a, samples=s, spp_begin=2)
    assert np.array_equal(a, full_a) and np.array_equal(s, full_s)
    a[:] = 0; s[:] = 0
    S.render(6, 4, accum=a, samples=s, rect=(0, 0, 40, h))
    S.render(6, 4, accum=a, samples=s, rect=(40, 0, w, h))
    assert np.array_equal(a, full_a) and np.array_equal(s, full_s)


@pytest.mark.skipif(not po.have_ref(), reason="reference binary only exists in the build container")
def test_live_reference_dragon_hits_bit_exact():
    """100k-triangle chinese_dragon.glb straight from the reference's assets: loader, median-split BVH,
    traversal and hit interpolation of the restatement agree bit for bit with the reference."""
    path = "/root/reference/assets/models/chinese_dragon.glb"
    sc = {"film": (96, 72), "camera": {"pos": (28., 12., -60.), "rot": (0., 180., 0.), "fov": 45., "near": .1, "far": 100.},
          "objects": [{"type": "mesh", "path": path, "pos": (0, 0, 0), "rot": (-90., 0, 0), "scale": (1, 1.2, 0.8)}]}
    rr, rh = po.ref_rays(sc), po.ref_hits(sc)
    S = po.Scene(sc)
    assert np.array_equal(S.rays().view(np.uint32), rr.view(np.uint32))
    h = S.closest_hits(rr).reshape(rh.shape)
    assert rh[..., 0].mean() > 0.3
    assert np.array_equal(h.view(np.uint32), rh.view(np.uint32))
    import ctypes as C
    n, d = C.c_uint32(), C.c_uint32()
    po.lib().po_mesh_bvh_stats(S.h, 0, C.byref(n), C.byref(d))
    assert n.value == 131071  # the node count the reference's median split yields for 100 000 faces
