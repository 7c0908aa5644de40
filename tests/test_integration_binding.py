"""The reference-side binding (integration/GpuBackend.cpp: the translation unit a maintainer of the reference adds to
send RayTraceParallel, main.cpp:30-47, to the GPU) compiles against the reference's own headers and include/ptamd.h, links
with the reference's objects and the product library, and its host-side behaviour is right. Build container only: needs
/root/reference. On a GPU box the same binary (built beforehand) renders frames through the binding."""
import os
import subprocess

import pytest

from conftest import ROOT

CHECK = os.path.join(ROOT, "oracle", "_ref", "binding_check")


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/PathTracer"), reason="the reference's headers only exist in the build container")
def test_binding_compiles_against_the_reference_and_behaves_on_the_host():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-j4", "binding"], stdout=subprocess.DEVNULL)
    out = subprocess.run([CHECK, "tests/golden/cornell_mesh.glb"], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failure(s), 108 faces" in out.stdout
    assert "no GPU here" in out.stdout and "no CPU fallback" in out.stdout  # this container has no device: the library says so


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(CHECK), reason="binding_check is built in the build container (make -C oracle binding)")
def test_binding_renders_frames_on_the_gpu(gpu):
    out = subprocess.run([CHECK, "tests/golden/cornell_mesh.glb"], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rendered 24 frames on the GPU" in out.stdout and "0 failure(s)" in out.stdout
