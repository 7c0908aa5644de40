"""GPU side of the two options the reference does not have (Russian roulette, cubemap sky): compared with the
CPU oracle's version of the same options sample for sample, plus the analytic properties that pin them
(see test_extensions.py). Tolerances: miss-only images are bit-exact (the lookup is integer work and
correctly rounded divisions); same-seed radiance follows test_gpu_parity's bar."""
import numpy as np
import pytest

from conftest import lum
from oracle import pyoracle as po
from scenes import presets
from test_extensions import dim_light, empty_scene, gradient_faces

pytestmark = pytest.mark.gpu


def test_uniform_cubemap_is_the_constant_background(gpu):
    scene = presets.glass_mirror(64, 64)
    pt, cam = gpu.build_scene(scene)
    f0, f1, f2 = gpu.Film(64, 64), gpu.Film(64, 64), gpu.Film(64, 64)
    gpu.Render(cam, pt, f0, 3, 6, seed=11)
    pt.SetSkybox(np.full((6, 4, 4, 4), 51, np.uint8), 1.0)
    gpu.Render(cam, pt, f1, 3, 6, seed=11)
    assert np.array_equal(f0.Accum().view(np.uint32), f1.Accum().view(np.uint32))
    pt.SetSkybox(np.full((6, 4, 4, 4), 255, np.uint8), 2.0)
    pt.SetSkybox(None)
    gpu.Render(cam, pt, f2, 3, 6, seed=11)
    assert np.array_equal(f0.Accum().view(np.uint32), f2.Accum().view(np.uint32))


def test_float_cubemap(gpu):
    """ptamd_scene_set_skybox_f32: a constant 0.2 float map is the constant background bit for bit, a float map holding
    texel / 255 * intensity is the RGBA8 map bit for bit, radiance above 1 arrives unclipped, and an RGBA8 map can
    replace a float one and back."""
    scene = presets.glass_mirror(64, 64)
    pt, cam = gpu.build_scene(scene)
    films = [gpu.Film(64, 64) for _ in range(5)]
    gpu.Render(cam, pt, films[0], 3, 6, seed=11)
    pt.SetSkyboxFloat(np.full((6, 4, 4, 3), 0.2, np.float32))
    gpu.Render(cam, pt, films[1], 3, 6, seed=11)
    assert np.array_equal(films[0].Accum().view(np.uint32), films[1].Accum().view(np.uint32))
    u8 = gradient_faces(16, seed=9)
    pt.SetSkybox(u8, 1.75)
    gpu.Render(cam, pt, films[2], 3, 6, seed=11)
    pt.SetSkyboxFloat((u8[..., :3].astype(np.float32) / np.float32(255)) * np.float32(1.75))
    gpu.Render(cam, pt, films[3], 3, 6, seed=11)
    assert np.array_equal(films[2].Accum().view(np.uint32), films[3].Accum().view(np.uint32))
    assert not np.array_equal(films[0].Accum(), films[2].Accum())
    pt.SetSkyboxFloat(None)
    gpu.Render(cam, pt, films[4], 3, 6, seed=11)
    assert np.array_equal(films[0].Accum().view(np.uint32), films[4].Accum().view(np.uint32))
    hdr = np.zeros((6, 2, 2, 3), np.float32); hdr[..., 0] = 7.5; hdr[..., 1] = 3.0; hdr[..., 2] = 40.0
    pt, cam = gpu.build_scene(dict(empty_scene(16, 8, (0, 0, 0)), skybox={"faces": hdr}))
    f = gpu.Film(16, 8)
    gpu.Render(cam, pt, f, 1, 2)
    a = f.Accum()
    assert np.array_equal(a, np.broadcast_to(np.float32([7.5, 3.0, 40.0]), a.shape))


def test_float_cubemap_scene_same_seed(gpu):
    """A scene lit through its open front by a float map with radiance up to 6: GPU and oracle, same random stream."""
    rng = np.random.default_rng(12)
    faces = (rng.random((6, 32, 32, 3)) ** 3 * 6.0).astype(np.float32)
    scene = dict(presets.glass_mirror(96, 96), skybox={"faces": faces})
    scene["camera"] = dict(scene["camera"], pos=(2.78, 2.73, -2.0), fov=70.0)
    pt, cam = gpu.build_scene(scene)
    film = gpu.Film(96, 96)
    gpu.Render(cam, pt, film, 2, 8, seed=4)
    a = film.Accum()
    oa, _ = po.Scene(scene).render(2, 8, seed=4)
    close = np.abs(a - oa).max(axis=-1) <= 2e-3 + 2e-3 * np.abs(oa).max(axis=-1)
    assert close.mean() >= 0.95, close.mean()
    assert abs(lum(a).mean() - lum(oa).mean()) <= 0.03 * lum(oa).mean()


def test_cubemap_lookup_matches_oracle_exactly(gpu):
    faces = gradient_faces(32)
    for rot in [(0, 0, 0), (0, 90, 0), (0, 180, 0), (80, 30, 0), (-80, 200, 10), (35, 45, 20)]:
        scene = dict(empty_scene(160, 90, rot), skybox={"faces": faces, "intensity": 1.5})
        pt, cam = gpu.build_scene(scene)
        film = gpu.Film(160, 90)
        gpu.Render(cam, pt, film, 1, 4, seed=1)
        a = film.Accum()
        oa, _ = po.Scene(scene).render(1, 4, seed=1)
        # camera rays agree to ~1e-7, so a direction a hair from a texel edge may pick the neighbour
        same = (a == oa).all(axis=-1)
        assert same.mean() >= 0.995, (rot, same.mean())
        assert np.isin(a[~same].view(np.uint32), oa.view(np.uint32)).all()  # and then it is still a texel of the map


def test_skybox_scene_same_seed(gpu):
    scene = dict(presets.glass_mirror(96, 96), skybox={"faces": gradient_faces(64, seed=8), "intensity": 1.0})
    scene["camera"] = dict(scene["camera"], pos=(2.78, 2.73, -2.0), fov=70.0)  # inside the open front: the sky is seen through it
    pt, cam = gpu.build_scene(scene)
    film = gpu.Film(96, 96)
    gpu.Render(cam, pt, film, 2, 8, seed=4)
    a = film.Accum()
    oa, _ = po.Scene(scene).render(2, 8, seed=4)
    close = np.abs(a - oa).max(axis=-1) <= 2e-3 + 2e-3 * np.abs(oa).max(axis=-1)
    assert close.mean() >= 0.95, close.mean()
    assert abs(lum(a).mean() - lum(oa).mean()) <= 0.03 * lum(oa).mean()


def test_russian_roulette_same_seed_and_statistics(gpu):
    scene = presets.glass_mirror(96, 96)
    pt, cam = gpu.build_scene(scene)
    S = po.Scene(scene)
    film = gpu.Film(96, 96)
    gpu.Render(cam, pt, film, 2, 16, seed=6, russian_roulette=True, count=True)
    st = gpu.RenderStats()
    a = film.Accum()
    oa, _, oc = S.render(2, 16, seed=6, russian_roulette=True, counters=True)
    close = np.abs(a - oa).max(axis=-1) <= 2e-3 + 2e-3 * np.abs(oa).max(axis=-1)
    assert close.mean() >= 0.93, close.mean()
    assert abs(st["rays"] - oc["rays"]) <= 0.01 * oc["rays"]
    # unbiased: the long-run image with and without roulette agree (below the film's luminance clamp, see
    # dim_light), and roulette casts fewer rays
    pt, cam = gpu.build_scene(dim_light(scene))
    spp = 512
    f0, f1 = gpu.Film(96, 96), gpu.Film(96, 96)
    gpu.Render(cam, pt, f0, spp, 16, seed=30, count=True)
    r0 = gpu.RenderStats()["rays"]
    gpu.Render(cam, pt, f1, spp, 16, seed=30, russian_roulette=True, count=True)
    r1 = gpu.RenderStats()["rays"]
    assert r1 < 0.8 * r0
    m0, m1 = lum(f0.Average()).mean(), lum(f1.Average()).mean()
    assert abs(m1 - m0) <= 0.005 * m0, (m0, m1)
    # a depth below the roulette start is untouched
    g0, g1 = gpu.Film(96, 96), gpu.Film(96, 96)
    gpu.Render(cam, pt, g0, 2, 3, seed=5)
    gpu.Render(cam, pt, g1, 2, 3, seed=5, russian_roulette=True)
    assert np.array_equal(g0.Accum().view(np.uint32), g1.Accum().view(np.uint32))
