"""GPU smoke test of the multi-process launch path on ONE GPU: bench.py started by torch.distributed.run with a world of
one rank, which takes every step of the N > 1 path -- the `nccl` (RCCL) process group, the shared stream, the in-place
reduce of the caller-owned film, the barrier / all-reduce timing -- with a group of one. A scaling curve needs a whole
node; this only makes sure that the first multi-GPU run does not stop in initialisation. The child is a fresh process:
it initialises the GPU itself."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_under_torchrun_world_of_one(gpu, scaling):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--spp", "8", "--no-cpu", "--scaling", scaling]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 1 and r["scaling"] == scaling and r["value"] > 100.0  # Msamples/s: far above any CPU path
    assert r["time_to_1024spp_s"] is not None and r["time_to_1024spp_s"] > 0
    assert r["roofline"] is not None and 0 < r["roofline"]["frac"] < 1
    assert 5.5 < r["rays_per_sample"] < 7.5
