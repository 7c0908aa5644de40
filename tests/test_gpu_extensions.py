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
