"""GPU: the HIP path against the reference's own converged images (tests/golden/converged_*.npz), per pixel and per
material region, through the C ABI. The GPU renders four times the reference's 32 768 samples per pixel on scenes that are
traversed from LDS (its own noise is then half the reference's) and as many as the reference on the others; see
converged_common.py for the statistics and tolerances."""
import numpy as np
import pytest

import converged_common as cc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", cc.NAMES)
def test_gpu_converges_to_the_reference_image(gpu, name, tmp_path_factory):
    scene, depth = cc.scene_of(name, tmp_path_factory.getbasetemp())
    g = cc.load(name)
    pt, cam = gpu.build_scene(scene)
    w, h = scene["film"]
    spp = 131072 if pt.Info()["lds_resident"] else 32768
    film = gpu.Film(w, h)
    done = 0
    while done < spp:  # launches of 16 384 samples: a 2 300-pixel film fills a fraction of the GPU, so a launch takes a while
        gpu.Render(cam, pt, film, 16384, depth, spp_begin=done, seed=0xBEEF)
        done += 16384
    assert int(film.Samples().min()) == spp
    frac, report = cc.check(name, film.Average(), spp, g)
    assert len(report) >= 3
