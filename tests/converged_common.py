"""Shared by the converged-image tests (CPU oracle and GPU): the fixtures tests/golden/converged_<scene>.npz hold, per scene,
the reference renderer's own clamped-mean image at 8 x 4096 samples per pixel (tests/golden/make_converged.py ran the
reference's sources), the standard error of that mean from the spread of its eight independent runs, a map of which material
the camera ray meets, and each material region's mean luminance in every run.

What is compared, and why these numbers:
  * per pixel: |image - reference mean| <= 4 sigma on at least 99 % of the pixels, sigma^2 = stderr^2 * (1 + reference
    samples / own samples) (the image under test has its own noise, scaled from the reference's by the sample counts),
    with a floor of 2e-4 of the pixel's value + 1e-5 for pixels without noise (a directly seen light). With eight runs the
    standard error has seven degrees of freedom: a correct renderer leaves about 0.5 % of the pixels outside 4 sigma.
  * per material region (the pixels whose camera ray meets one material; at least 24 pixels): mean luminance within
    0.5 % of the reference's, or within 4 sigma of the region mean where the reference's own run-to-run spread is larger
    than that (small or dim regions).
A bias in one lobe's weight (PathTracer.cpp:99-300) moves the region of the object that isolates the lobe by far more than
0.5 %: a 10 % error in the specular weight of the rough-metal sphere moves its region by about 10 %."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_converged  # noqa: E402  (the scene table; its main() needs the reference, the table does not)

LUM = np.array([0.2126, 0.7152, 0.0722], np.float64)
NAMES = ["sphere_cornell", "glass_mirror", "box_cornell", "mesh_cornell", "blob_mixed", "textured", "dragon_field",
         "lobe_rough_metal", "lobe_mirror", "lobe_plastic", "lobe_half_glass", "lobe_rough_glass", "lobe_glass_box", "lobe_emitter_panel"]


def load(name):
    return np.load(os.path.join(HERE, "golden", "converged_%s.npz" % name))


def scene_of(name, tmpdir):
    table, field = make_converged.scenes(str(tmpdir))
    scene, depth = table[name]
    if callable(scene):
        from scenes import dragon
        if not os.path.exists(field):
            dragon.write_dragon_field(field)
        scene = scene()
    return scene, depth


def check(name, image, own_spp, g, min_fraction=0.99):
    """image: (H, W, 3) clamped-mean radiance at own_spp samples per pixel. Returns a short report; raises on disagreement."""
    mean, stderr, region = g["mean"].astype(np.float64), g["stderr"].astype(np.float64), g["region"]
    ref_spp = float(g["spp"]) * float(g["chunks"])
    image = image.astype(np.float64)
    sigma = np.sqrt(stderr ** 2 * (1.0 + ref_spp / float(own_spp)))
    tol = 4.0 * sigma + 2e-4 * np.abs(mean) + 1e-5
    ok = (np.abs(image - mean) <= tol).all(axis=-1)
    assert ok.mean() >= min_fraction, "%s: %.2f %% of the pixels within 4 sigma of the reference" % (name, 100 * ok.mean())
    lum = (image * LUM).sum(-1)
    chunks = g["region_lum"].astype(np.float64)  # (regions, runs)
    report = []
    for r in range(1, chunks.shape[0]):
        m = region == r
        if m.sum() < 24:
            continue
        ref = chunks[r].mean()
        s_run = chunks[r].std(ddof=1)                       # spread of one run's region mean (spp samples per pixel)
        s = np.sqrt(s_run ** 2 / chunks.shape[1] + s_run ** 2 * float(g["spp"]) / float(own_spp))
        got = lum[m].mean()
        allowed = max(0.005 * ref, 4.0 * s)
        report.append((r, int(m.sum()), ref, got, (got - ref) / ref, allowed / ref))
        assert abs(got - ref) <= allowed, "%s region %d (%d pixels): mean luminance %.5f against the reference's %.5f (%.2f %%, allowed %.2f %%)" % (
            name, r, int(m.sum()), got, ref, 100 * (got - ref) / ref, 100 * allowed / ref)
    total_ref, total = (mean * LUM).sum(-1).mean(), lum.mean()
    assert abs(total - total_ref) <= 0.005 * total_ref, "%s: mean luminance %.5f against %.5f" % (name, total, total_ref)
    return ok.mean(), report
