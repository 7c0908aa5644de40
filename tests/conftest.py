import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "path-tracer_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    os.chdir(ROOT)  # golden scenes name their meshes relative to the repo root


@pytest.fixture(scope="session")
def dragon_field_glb(tmp_path_factory):
    """The ~1 M-triangle .glb of BASELINE.json's config 3: ten baked instances of tests/golden/chinese_dragon.glb
    (a byte copy of the reference's assets/models/chinese_dragon.glb), written by scenes/dragon.py (26 MB, 0.2 s)."""
    from scenes import dragon
    path = str(tmp_path_factory.mktemp("dragon") / "dragon_field.glb")
    assert dragon.write_dragon_field(path) == 1_000_000
    return path


@pytest.fixture(scope="session")
def golden_scenes(dragon_field_glb):
    """name -> (scene dict, depth), the same table tests/golden/make_golden.py rendered with the reference."""
    from scenes import dragon, presets
    return {
        "dragon": ({"film": (96, 72), "camera": {"pos": (28., 12., -60.), "rot": (0., 180., 0.), "fov": 45., "near": .1, "far": 100.},
                    "objects": [{"type": "mesh", "path": "tests/golden/chinese_dragon.glb", "pos": (0, 0, 0), "rot": (-90., 0, 0),
                                 "scale": (1, 1.2, 0.8)}]}, 6),
        "dragon_field": (dragon.config3_scene(dragon_field_glb, 96, 54), 8),
        "sphere_cornell": (presets.sphere_cornell(96, 96), 5),
        "glass_mirror": (presets.glass_mirror(96, 96), 12),
        "box_cornell": (presets.box_cornell(96, 64), 6),
        "mesh_cornell": (presets.mesh_cornell("tests/golden/cornell_mesh.glb", 96, 54), 8),
        "blob_mixed": (presets.mixed("tests/golden/blob.glb", 96, 96, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0),
                                     mesh_scale=(1.1, 0.9, 1.0)), 8),
        "textured": (presets.mixed("tests/golden/textured.glb", 96, 96, mesh_pos=(2.7, 1.6, 2.6), mesh_rot=(0.0, 25.0, 0.0),
                                   mesh_scale=(1.4, 1.4, 1.4)), 6),
        "textured_jpeg": (presets.mixed("tests/golden/textured_jpeg.glb", 96, 96, mesh_pos=(2.7, 1.6, 2.6), mesh_rot=(0.0, 25.0, 0.0),
                                        mesh_scale=(1.4, 1.4, 1.4)), 6),
        "mesh_cornell_128x72": (presets.mesh_cornell("tests/golden/cornell_mesh.glb", 128, 72), 8),
    }


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(scope="session")
def gpu():
    """Initialises the product library on cuda:0; fails loudly if the HIP extension is missing."""
    import ptamd
    ptamd.init(0)
    return ptamd


def lum(a):
    return (a * np.array([0.2126, 0.7152, 0.0722], np.float32)).sum(-1)


def rmse(a, b):
    return float(np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean()))
