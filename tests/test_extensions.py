"""The two options the reference does not have -- Russian roulette and the cubemap sky -- checked on the CPU
oracle against analytic properties (there is no reference output to compare with: parity for these two is
"unpinned" by construction, so the properties below are what pins them):
  * a cubemap whose every texel is (51, 51, 51) is the constant background 0.2 bit for bit;
  * the face / orientation rule is the OpenGL cubemap table, restated independently in numpy here;
  * Russian roulette leaves the expected image unchanged and shortens paths.
The GPU versions are then compared with the oracle sample for sample in test_gpu_extensions.py."""
import numpy as np

from conftest import lum
from oracle import pyoracle as po
from scenes import presets


def six_colour_faces(n=8):
    cols = [(255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 0), (255, 0, 255), (0, 255, 255)]
    f = np.zeros((6, n, n, 4), np.uint8)
    for i, c in enumerate(cols):
        f[i, :, :, :3] = c
        f[i, :, :, 3] = 255
    return f


def gradient_faces(n=16, seed=4):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (6, n, n, 4), dtype=np.uint8)


def gl_cubemap_lookup(faces, d):
    """OpenGL 4.6 core spec table 8.19 (major axis, sc, tc, ma), nearest texel; d: (N, 3) float32."""
    n = faces.shape[1]
    d = d.astype(np.float32)
    a = np.abs(d)
    out = np.zeros((len(d), 3), np.float32)
    for i in range(len(d)):
        x, y, z = d[i]
        ax, ay, az = a[i]
        if ax >= ay and ax >= az:
            face, sc, tc, ma = (0, -z, -y, ax) if x > 0 else (1, z, -y, ax)
        elif ay >= az:
            face, sc, tc, ma = (2, x, z, ay) if y > 0 else (3, x, -z, ay)
        else:
            face, sc, tc, ma = (4, x, -y, az) if z > 0 else (5, -x, -y, az)
        u = (np.float32(sc) / np.float32(ma) + np.float32(1)) * np.float32(0.5)
        v = (np.float32(tc) / np.float32(ma) + np.float32(1)) * np.float32(0.5)
        px = min(max(int(u * np.float32(n)), 0), n - 1)
        py = min(max(int(v * np.float32(n)), 0), n - 1)
        out[i] = faces[face, py, px, :3].astype(np.float32) / np.float32(255)
    return out


def empty_scene(w, h, rot, fov=90.0):
    cam = dict(presets.default_camera(), rot=rot, fov=fov)
    return {"film": (w, h), "camera": cam, "background": (0.2, 0.2, 0.2), "objects": []}


def test_uniform_cubemap_is_the_constant_background():
    scene = presets.glass_mirror(48, 48)
    a0, _ = po.Scene(scene).render(2, 6, seed=11)
    sky = np.full((6, 4, 4, 4), 51, np.uint8)  # 51 / 255 rounds to the float 0.2
    a1, _ = po.Scene(dict(scene, skybox={"faces": sky, "intensity": 1.0})).render(2, 6, seed=11)
    assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32))


def test_float_cubemap_is_returned_as_it_is():
    """The float (HDR) form of the sky map: a constant 0.2 map is the constant background bit for bit; a float map holding
    texel / 255 * intensity is the RGBA8 map with that intensity bit for bit; values above 1 come through unclipped (only
    Film::AddSample's luminance clamp at 12 applies, Film.cpp:38-43)."""
    scene = presets.glass_mirror(48, 48)
    a0, _ = po.Scene(scene).render(2, 6, seed=11)
    a1, _ = po.Scene(dict(scene, skybox={"faces": np.full((6, 4, 4, 3), 0.2, np.float32)})).render(2, 6, seed=11)
    assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32))
    u8 = gradient_faces(16, seed=9)
    f32 = (u8[..., :3].astype(np.float32) / np.float32(255)) * np.float32(1.75)
    b0, _ = po.Scene(dict(scene, skybox={"faces": u8, "intensity": 1.75})).render(2, 6, seed=3)
    b1, _ = po.Scene(dict(scene, skybox={"faces": f32})).render(2, 6, seed=3)
    assert np.array_equal(b0.view(np.uint32), b1.view(np.uint32))
    hdr = np.zeros((6, 2, 2, 3), np.float32); hdr[..., 0] = 7.5; hdr[..., 1] = 3.0; hdr[..., 2] = 40.0  # luminance 6.63: below the clamp
    S = po.Scene(dict(empty_scene(8, 6, (0, 0, 0)), skybox={"faces": hdr}))
    c, _ = S.render(1, 2)
    assert np.array_equal(c, np.broadcast_to(np.float32([7.5, 3.0, 40.0]), c.shape))


def test_cubemap_faces_and_orientation():
    faces = gradient_faces()
    for rot in [(0, 0, 0), (0, 90, 0), (0, 180, 0), (0, 270, 0), (80, 30, 0), (-80, 200, 10), (35, 45, 20)]:
        scene = dict(empty_scene(40, 30, rot), skybox={"faces": faces, "intensity": 2.0})
        S = po.Scene(scene)
        a, n = S.render(1, 3, seed=1)
        rays = S.rays().reshape(-1, 6)
        want = gl_cubemap_lookup(faces, rays[:, 3:6]) * np.float32(2.0)
        assert np.array_equal(a.reshape(-1, 3), want), rot
    # the six axes land on the six faces
    cols = six_colour_faces()
    for rot, face in [((0, -90, 0), 0), ((0, 90, 0), 1), ((90, 0, 0), 2), ((-90, 0, 0), 3), ((0, 180, 0), 4), ((0, 0, 0), 5)]:
        S = po.Scene(dict(empty_scene(9, 9, rot, fov=20.0), skybox={"faces": cols, "intensity": 1.0}))
        d = S.rays().reshape(9, 9, 6)[4, 4, 3:6]
        axis = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)][face]
        assert np.dot(d, axis) > 0.99, (rot, d)
        a, _ = S.render(1, 2)
        assert np.array_equal(a[4, 4], cols[face, 0, 0, :3].astype(np.float32) / 255), (rot, face)


def test_skybox_lights_the_scene_and_can_be_removed():
    scene = presets.sphere_cornell(32, 32)
    S = po.Scene(scene)
    a0, _ = S.render(4, 5, seed=2)
    S.set_skybox(np.full((6, 2, 2, 4), 255, np.uint8), 3.0)
    a1, _ = S.render(4, 5, seed=2)
    assert lum(a1).mean() > lum(a0).mean()
    S.set_skybox(None)
    a2, _ = S.render(4, 5, seed=2)
    assert np.array_equal(a0, a2)


def dim_light(scene, strength=3.0):
    """Film::AddSample clips a sample's luminance at 12 (Film.cpp:38-43). Roulette rescales surviving samples, so
    where samples reach the clamp more of them are clipped and the clipped image gets darker: the clamp's bias,
    not the roulette's. Two things push samples there: the room's 15x light, and materials with
    0 < transmission < 1, whose two lobes are each weighted by 1 / their selection probability
    (PathTracer.cpp:99-204, 213-300) so that throughput grows along a path (measured: a 0.7-transmission sphere
    gains 50 % radiance from depth 4 to 16). The unbiasedness check therefore uses a 3x light and drops those
    objects; roulette then matches to 0.1 % (glass, mirror and rough-metal spheres, depth 16, 2048 spp)."""
    objs = [dict(o) for o in scene["objects"] if not (o.get("mat") is not None and 0.0 < o["mat"][10] < 1.0)]
    for o in objs:
        if o.get("mat") is not None and o["mat"][8] > 0:
            o["mat"] = list(o["mat"][:8]) + [float(strength)] + list(o["mat"][9:])
    return dict(scene, objects=objs)


def test_russian_roulette_is_unbiased_and_shortens_paths():
    scene = dim_light(presets.glass_mirror(40, 40))
    S = po.Scene(scene)
    spp, depth = 192, 16
    a0, _, c0 = S.render(spp, depth, seed=21, counters=True)
    a1, _, c1 = S.render(spp, depth, seed=21, counters=True, russian_roulette=True)
    b0, _ = S.render(spp, depth, seed=22)
    assert c1["rays"] < 0.8 * c0["rays"]
    m0, m1, mb = lum(a0 / spp).mean(), lum(a1 / spp).mean(), lum(b0 / spp).mean()
    noise = abs(m0 - mb)  # seed-to-seed spread of the same estimator
    assert abs(m1 - m0) <= max(4 * noise, 0.01 * m0), (m0, m1, mb)
    # paths shorter than the roulette start are untouched: depth 3 renders are identical
    d0, _ = S.render(2, 3, seed=5)
    d1, _ = S.render(2, 3, seed=5, russian_roulette=True)
    assert np.array_equal(d0, d1)
