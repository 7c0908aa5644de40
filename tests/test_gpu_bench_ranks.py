"""`python3 bench.py --gpus 2` as the driver types it, with no launcher around it: the command starts its own two ranks
(main.cpp:30-47 is the reference's one fan-out call; here the fan-out is one process per GPU). The box has one GPU, so the
two ranks share it and join over gloo (PTAMD_BENCH_BACKEND / PTAMD_BENCH_DEVICE); everything else -- the spawn, the
spp-slice split, the film sum, the max-over-ranks timing, rank 0's JSON line -- is the path an 8-GPU run takes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from scenes import presets

pytestmark = pytest.mark.gpu


def run_bench(tmp_path, *args, **env):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, cwd=str(tmp_path), env=e, timeout=400)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_plain_bench_command_starts_two_ranks(gpu, tmp_path):
    dump = str(tmp_path / "film.npy")
    r = run_bench(tmp_path, "--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", "9", "--no-cpu",
                  PTAMD_BENCH_BACKEND="gloo", PTAMD_BENCH_DEVICE="0", PTAMD_BENCH_DUMP=dump)
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["scaling"] == "strong" and r["backend"] == "gloo"
    assert r["time_to_1024spp_s"] is not None and r["time_to_1024spp_s"] > 0 and r["value"] > 50.0
    assert "roofline" not in r  # single-GPU diagnostics stay out of multi-rank lines
    # the summed film of the two ranks (5 + 4 samples per pixel) is the single-process film of 9 samples
    accum = np.load(dump)
    scene = presets.mesh_cornell("tests/golden/cornell_mesh.glb", 1920, 1080)
    pt, cam = gpu.build_scene(scene)
    film = gpu.Film(1920, 1080)
    gpu.Render(cam, pt, film, 9, 8)
    ref = film.Accum()
    assert np.allclose(accum, ref, rtol=1e-5, atol=1e-6)  # same per-sample values, another order of float additions
    assert float(ref.mean()) > 0.05


def test_weak_scaling_and_tile_sharding_with_two_ranks(gpu, tmp_path):
    r = run_bench(tmp_path, "--gpus", "2", "--steps", "1", "--warmup", "1", "--spp", "4", "--no-cpu", "--scaling", "weak",
                  PTAMD_BENCH_BACKEND="gloo", PTAMD_BENCH_DEVICE="0")
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and "4 spp per GPU" in r["config"]["workload"]
    dump = str(tmp_path / "film4.npy")
    r = run_bench(tmp_path, "--gpus", "2", "--steps", "1", "--warmup", "0", "--spp", "2", "--no-cpu", "--config", "4",
                  PTAMD_BENCH_BACKEND="gloo", PTAMD_BENCH_DEVICE="0", PTAMD_BENCH_DUMP=dump)
    assert r["n_gpus"] == 2 and r["config"]["baseline_config"] == 4 and "tiles" in r["config"]["sharding"]
    accum = np.load(dump)
    scene = presets.mixed("tests/golden/blob.glb", 3840, 2160, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0))
    pt, cam = gpu.build_scene(scene)
    film = gpu.Film(3840, 2160)
    gpu.Render(cam, pt, film, 2, 8)
    assert np.array_equal(accum.view(np.uint32), film.Accum().view(np.uint32))  # disjoint tiles: the sum only adds zeros
