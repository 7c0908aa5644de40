"""The N > 1 path with the GPU library as the renderer: two processes (both on the one GPU of the box, joined over gloo)
render their shares of a frame -- a sample range each, interleaved tiles launch by launch, or interleaved tiles in one
launch per rank -- and sum their films; the result is compared with one process rendering the whole frame."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", ["spp", "tile", "tiles1"])
def test_two_gpu_ranks_equal_one(gpu, mode, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_shard_worker as wk
    port = str(free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gpu_shard_worker.py"), str(r), "2", mode, port, str(tmp_path)],
                              cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        try:
            _, err = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, err[-2000:]
    accum = np.load(tmp_path / "accum.npy")
    samples = np.load(tmp_path / "samples.npy")
    assert int(np.load(tmp_path / "n.npy")[0]) == wk.W * wk.H * wk.SPP  # every (pixel, sample) rendered exactly once
    pt, cam = gpu.build_scene(wk.scene())
    film = gpu.Film(wk.W, wk.H)
    gpu.Render(cam, pt, film, wk.SPP, wk.DEPTH, seed=wk.SEED)
    assert np.array_equal(samples, film.Samples().astype(np.int32))
    ref = film.Accum()
    if mode == "spp":
        # the per-sample values are identical; only the order of the float additions differs
        assert np.allclose(accum, ref, rtol=1e-5, atol=1e-6)
    else:
        assert np.array_equal(accum.view(np.uint32), ref.view(np.uint32))  # disjoint pixels: the sum only adds zeros
