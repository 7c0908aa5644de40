"""CPU: the plain-C restatement (oracle/pt_oracle.c) against the reference's own converged images, per pixel and per
material region, for every golden scene and for seven scenes that each isolate one lobe of the BSDF
(PathTracer.cpp:73-302) or one quirk (a transmissive Box, Box.cpp:86-89; a textured emitter, Mesh.cpp:476-477).
See converged_common.py for the statistics. The restatement draws from its own generator (the reference's cannot be
seeded, PathTracer.cpp:8-16), so the comparison is between two independent estimates of the same image."""
import numpy as np
import pytest

import converged_common as cc
from oracle import pyoracle as po

SPP = 3072  # the oracle's own samples per pixel: about four seconds per scene on eight cores


@pytest.mark.parametrize("name", cc.NAMES)
def test_oracle_converges_to_the_reference_image(name, tmp_path_factory):
    scene, depth = cc.scene_of(name, tmp_path_factory.getbasetemp())
    g = cc.load(name)
    assert depth == int(g["depth"]) and tuple(g["mean"].shape[:2]) == (scene["film"][1], scene["film"][0])
    S = po.Scene(scene)
    a, s = S.render(SPP, depth, seed=0xC0FFEE)
    frac, report = cc.check(name, po.average(a, s), SPP, g)
    assert len(report) >= 3  # walls and at least one object each form a region
