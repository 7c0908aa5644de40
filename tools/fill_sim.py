"""Estimates what sequential multi-ray-per-lane traversal would gain: per-ray traversal work of random interior rays
(GPU probe), grouped into waves of 64 lanes with k rays per lane traced back to back."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "path-tracer_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import ptamd
from scenes import presets
os.chdir(ROOT)
ptamd.init(0)
scene = presets.mesh_cornell("tests/golden/cornell_mesh.glb", 64, 64)
pt, cam = ptamd.build_scene(scene)
rng = np.random.default_rng(1)
n = 64 * 4096
o = rng.random((n, 3)).astype(np.float32) * 5.0 + 0.25
d = rng.normal(size=(n, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
_, c = pt.TraversalCounts(np.concatenate([o, d], 1))
work = c[:, 0].astype(np.float64) * 48 + c[:, 1] * 30  # vector instructions per ray, roughly
for name, w in (("nodes", c[:, 0].astype(np.float64)), ("weighted", work)):
    print(name, "mean %.1f" % w.mean())
    for k in (1, 2, 3, 4):
        m = w[: (n // (64 * k)) * 64 * k].reshape(-1, k, 64).sum(axis=1)  # k rays per lane, back to back
        print("  k=%d: wave cost per ray %.2f (lane mean %.2f) -> fill %.0f%%" % (k, m.max(axis=1).mean() / k, w.mean(), 100 * w.mean() * k / m.max(axis=1).mean()))
