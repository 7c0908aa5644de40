"""Cubemap sky maps for the miss radiance (an option outside the reference, which returns a constant: PathTracer.cpp:59-60).

scenes/sky/{p,n}{x,y,z}.png are byte copies of the reference's assets/skyboxes/sky/ faces (1024 x 1024 RGBA8; data that
no source file of the reference reads), decoded here by the product library's own image reader."""
import os

import numpy as np

FACES = ("px", "nx", "py", "ny", "pz", "nz")  # +x -x +y -y +z -z, the order ptamd_scene_set_skybox expects


def load_faces(directory):
    """(6, N, N, 4) uint8, row 0 at the top."""
    import ptamd
    out = []
    for n in FACES:
        with open(os.path.join(directory, n + ".png"), "rb") as f:
            out.append(ptamd.decode_image(f.read()))
    return np.ascontiguousarray(np.stack(out))


def to_radiance(faces_u8, peak=4.0):
    """RGBA8 faces as float radiance: sRGB-decoded texel, then the highlight expansion x / (1 - x * (1 - 1 / peak)), which
    leaves dark texels alone and takes white to `peak` (an LDR photograph stretched into an HDR map)."""
    c = faces_u8[..., :3].astype(np.float32) / 255.0
    lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4).astype(np.float32)
    return np.ascontiguousarray(lin / (1.0 - lin * np.float32(1.0 - 1.0 / peak)), np.float32)


def load_hdr_faces(directory, peak=4.0):
    """(6, N, N, 3) float32 radiance for ptamd_scene_set_skybox_f32."""
    return to_radiance(load_faces(directory), peak)
