"""Shared by the converged-image tests (CPU oracle and GPU): the fixtures tests/golden/converged_<scene>.npz hold, per scene,
the reference renderer's own clamped-mean image at 8 x 4096 samples per pixel (tests/golden/make_converged.py ran the
reference's sources), the standard error of that mean from the spread of its eight independent runs, a map of which material
the camera ray meets, and each material region's mean luminance in every run.

What is compared, and why these numbers:
  * per pixel (pixels whose eight neighbours show the same material): |image - reference mean| <= 4 sigma on at least
    99 % of them and <= 5 sigma on at least 99.7 %,
    sigma^2 = stderr^2 * (1 + reference samples / own samples) (the image under test has its own noise, scaled from the
    reference's by the sample counts), with a floor of 2e-4 of the pixel's value + 1e-5 for pixels without noise (a directly
    seen light). A pixel's standard error comes from eight runs (seven degrees of freedom) of samples with a heavy, clamped
    tail (Film.cpp:38-43), so on its own it is uncertain: with it two independent correct estimates leave 0.8-2.1 % of the
    pixels outside 4 sigma. stderr^2 is therefore the larger of the pixel's own and the mean over the pixel and its eight
    neighbours (all of one material): measured with the restatement at three seeds on all 14 scenes, 99.65-100 % of the
    pixels then lie within 4 sigma and 99.94-100 % within 5.
  * per material region (the pixels whose camera ray meets one material; at least 24 pixels): mean luminance within
    0.5 % of the reference's, or within 4 sigma of the region mean where the run-to-run spread of the two estimates is
    larger than that (small or dim regions, or few own samples); the same for the whole image. The spread of one run's
    region mean is taken as the larger of two estimates: the standard deviation of the eight runs' region means (seven
    degrees of freedom on a heavy-tailed quantity: for the side walls of the sphere room it came out at half of what 24
    runs of the restatement show), and the one pooled from the per-pixel standard errors of the region's pixels
    (independent pixels, channels taken as fully correlated: hundreds of pixels x seven degrees of freedom).
The lobe scenes are lit by a ceiling-wide panel, so their region means are known to 0.1-0.3 % from the reference's 32 768
samples per pixel. A bias in one lobe's weight (PathTracer.cpp:99-300) moves the region of the object that isolates the
lobe by about as much: a 10 % error in the specular weight of the rough-metal sphere moves its region by about 10 %."""
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


def check(name, image, own_spp, g):
    """image: (H, W, 3) clamped-mean radiance at own_spp samples per pixel. Returns (fraction of pixels within 4 sigma,
    region report); raises on disagreement."""
    mean, stderr, region = g["mean"].astype(np.float64), g["stderr"].astype(np.float64), g["region"]
    ref_spp = float(g["spp"]) * float(g["chunks"])
    image = image.astype(np.float64)
    var = stderr ** 2
    floor = 2e-4 * np.abs(mean) + 1e-5
    dev = np.abs(image - mean)
    # Pixels on a material boundary are left out of the per-pixel count (not out of the region means): the camera has no
    # jitter, so a centre ray that grazes a silhouette meets one surface or the other on a last-bit difference of the
    # intersection arithmetic, and the whole pixel follows (the deterministic-image tests bound how many do).
    pad = np.pad(region, 1, mode="edge")
    h, w = region.shape
    interior = np.ones((h, w), bool)
    vpad, near = np.pad(var, ((1, 1), (1, 1), (0, 0)), mode="edge"), np.zeros_like(var)
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            interior &= pad[dy:dy + h, dx:dx + w] == region
            near += vpad[dy:dy + h, dx:dx + w] / 9.0
    sigma = np.sqrt(np.maximum(var, near) * (1.0 + ref_spp / float(own_spp)))
    in4 = (dev <= 4.0 * sigma + floor).all(axis=-1)[interior].mean()
    in5 = (dev <= 5.0 * sigma + floor).all(axis=-1)[interior].mean()
    assert in4 >= 0.99 and in5 >= 0.997, "%s: %.2f %% of the pixels within 4 sigma of the reference, %.2f %% within 5" % (name, 100 * in4, 100 * in5)
    lum = (image * LUM).sum(-1)
    chunks = g["region_lum"].astype(np.float64)  # (regions, runs); region 0 = camera rays that miss
    runs = chunks.shape[1]
    own = float(g["spp"]) / float(own_spp)       # variance of the image under test in units of one run's

    lum_se = (stderr * LUM).sum(-1)              # per-pixel standard error of the luminance of the reference mean (upper bound)

    def compare(what, got, per_run, mask):
        ref = per_run.mean()
        pooled = np.sqrt(runs * (lum_se[mask] ** 2).sum()) / mask.sum()
        s_run = max(per_run.std(ddof=1), pooled)
        allowed = max(0.005 * abs(ref), 4.0 * s_run * np.sqrt(1.0 / runs + own))
        assert abs(got - ref) <= allowed, "%s %s: mean luminance %.5f against the reference's %.5f (%+.2f %%, allowed %.2f %%)" % (
            name, what, got, ref, 100 * (got - ref) / ref, 100 * allowed / abs(ref))
        return ref, got, (got - ref) / ref, allowed / abs(ref)

    report = []
    sizes = np.array([(region == r).sum() for r in range(chunks.shape[0])], np.float64)
    for r in range(1, chunks.shape[0]):
        if sizes[r] >= 24:
            report.append((r, int(sizes[r])) + compare("region %d (%d pixels)" % (r, int(sizes[r])), lum[region == r].mean(), chunks[r], region == r))
    compare("whole image", lum.mean(), (chunks * sizes[:, None]).sum(axis=0) / sizes.sum(), np.ones_like(region, bool))
    return in4, report
