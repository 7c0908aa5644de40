"""TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/libpt_oracle.so (the CPU restatement) and a
runner for oracle/_ref/ref_oracle (the reference's own sources, when that binary has been built).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from scenes import gltf_io, presets  # noqa: E402

LIB_PATH = os.path.join(HERE, "libpt_oracle.so")
REF_BIN = os.path.join(HERE, "_ref", "ref_oracle")


class Material(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("roughness", C.c_float), ("metallic", C.c_float),
                ("emission_colour", C.c_float * 3), ("emission_strength", C.c_float), ("ior", C.c_float),
                ("transmission", C.c_float)]


class Pbr(C.Structure):
    _fields_ = [("base_color_tex", C.c_int32), ("normal_tex", C.c_int32), ("metallic_roughness_tex", C.c_int32),
                ("emissive_tex", C.c_int32), ("transmission_tex", C.c_int32), ("base_color_factor", C.c_float * 4),
                ("metallic_factor", C.c_float), ("roughness_factor", C.c_float), ("alpha_mode", C.c_int32),
                ("alpha_cutoff", C.c_float), ("normal_scale", C.c_float), ("emissive_factor", C.c_float * 3),
                ("transmission_factor", C.c_float), ("ior", C.c_float)]


class Image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("data", C.c_void_p)]


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("rotation_deg", C.c_float * 3), ("fov_deg", C.c_float),
                ("near_plane", C.c_float), ("far_plane", C.c_float)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "samples", "sphere_tests", "box_tests", "mesh_tests",
                                          "nodes_visited", "tris_tested", "shades")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.check_call(["make", "-C", HERE, "oracle"])
        L = C.CDLL(LIB_PATH)
        L.po_scene_create.restype = C.c_void_p
        L.po_scene_destroy.argtypes = [C.c_void_p]
        L.po_scene_set_background.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.po_scene_add_sphere.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_float, C.POINTER(Material)]
        L.po_scene_add_box.argtypes = [C.c_void_p] + [C.POINTER(C.c_float)] * 3 + [C.POINTER(Material)]
        L.po_scene_add_mesh.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_uint32, C.POINTER(Pbr), C.c_uint32, C.POINTER(Image)] + [C.POINTER(C.c_float)] * 3
        L.po_camera_ray.argtypes = [C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.po_closest_hit.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.po_trace_sample.argtypes = [C.c_void_p, C.POINTER(Camera)] + [C.c_int] * 4 + [C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_float)]
        L.po_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint32] + [C.c_int] * 5 + [C.POINTER(Counters)]
        L.po_render_ext.argtypes = L.po_render.argtypes + [C.c_uint32]
        L.po_scene_set_skybox.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_float]
        L.po_scene_set_skybox_f32.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.po_film_add_sample.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        L.po_film_resolve.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.po_pcg_hash.argtypes = [C.c_uint32]
        L.po_pcg_hash.restype = C.c_uint32
        L.po_path_key.argtypes = [C.c_uint32] * 3
        L.po_path_key.restype = C.c_uint64
        L.po_rand01.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.po_rand01.restype = C.c_float
        L.po_mesh_bvh_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _lib = L
    return _lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def make_material(m):
    out = Material()
    out.albedo[:] = m[0:3]
    out.roughness, out.metallic = m[3], m[4]
    out.emission_colour[:] = m[5:8]
    out.emission_strength, out.ior, out.transmission = m[8], m[9], m[10]
    return out


def make_camera(c):
    cam = Camera()
    cam.position[:] = c["pos"]
    cam.rotation_deg[:] = c["rot"]
    cam.fov_deg, cam.near_plane, cam.far_plane = c["fov"], c.get("near", 0.1), c.get("far", 100.0)
    return cam


def make_pbr_array(groups):
    arr = (Pbr * max(1, len(groups)))()
    for i, g in enumerate(groups):
        p = arr[i]
        for k in ("base_color_tex", "normal_tex", "metallic_roughness_tex", "emissive_tex", "transmission_tex",
                  "metallic_factor", "roughness_factor", "alpha_mode", "alpha_cutoff", "normal_scale",
                  "transmission_factor", "ior"):
            setattr(p, k, g[k])
        p.base_color_factor[:] = g["base_color_factor"]
        p.emissive_factor[:] = g["emissive_factor"]
    return arr


class Scene:
    """CPU-oracle scene built from a scenes/presets.py dict."""

    def __init__(self, scene, mesh_cache=None):
        L = lib()
        self.h = L.po_scene_create()
        self.scene = scene
        self._keep = []
        bg = scene.get("background", (0.2, 0.2, 0.2))
        L.po_scene_set_background(self.h, _f3(bg))
        for o in scene["objects"]:
            if o["type"] == "sphere":
                L.po_scene_add_sphere(self.h, _f3(o["pos"]), float(o["radius"]), C.byref(make_material(o["mat"])))
            elif o["type"] == "box":
                L.po_scene_add_box(self.h, _f3(o["pos"]), _f3(o["rot"]), _f3(o["size"]), C.byref(make_material(o["mat"])))
            elif o["type"] == "mesh":
                md = o.get("data")
                if md is None:
                    if mesh_cache is not None and o["path"] in mesh_cache:
                        md = mesh_cache[o["path"]]
                    else:
                        md = gltf_io.load_gltf(o["path"])
                        if mesh_cache is not None:
                            mesh_cache[o["path"]] = md
                self.add_mesh(md, o["pos"], o["rot"], o["scale"])
            else:
                raise ValueError(o["type"])
        self.cam = make_camera(scene["camera"])
        self.w, self.h_px = scene["film"]
        if scene.get("skybox") is not None:
            self.set_skybox(scene["skybox"]["faces"], scene["skybox"].get("intensity", 1.0))

    def set_skybox(self, faces, intensity=1.0):
        """Extension (no reference counterpart): faces = (6, N, N, 4) uint8 in +x -x +y -y +z -z order, (6, N, N, 3) float32
        radiance (intensity unused), or None."""
        if faces is None:
            lib().po_scene_set_skybox(self.h, 0, None, 0.0)
            return
        if np.asarray(faces).dtype != np.uint8:
            f = np.ascontiguousarray(faces, np.float32)
            ptrs = (C.c_void_p * 6)(*[f[i].ctypes.data for i in range(6)])
            lib().po_scene_set_skybox_f32(self.h, f.shape[1], ptrs)
            return
        f = np.ascontiguousarray(faces, np.uint8)
        ptrs = (C.c_void_p * 6)(*[f[i].ctypes.data for i in range(6)])
        lib().po_scene_set_skybox(self.h, f.shape[1], ptrs, float(intensity))

    def add_mesh(self, md, pos, rot, scale):
        L = lib()
        nf = md["positions"].shape[0]
        imgs = (Image * max(1, len(md["images"])))()
        for i, im in enumerate(md["images"]):
            im = np.ascontiguousarray(im)
            self._keep.append(im)
            imgs[i].height, imgs[i].width, imgs[i].channels = im.shape
            imgs[i].data = im.ctypes.data
        rc = L.po_scene_add_mesh(self.h, nf, md["positions"].ctypes.data, md["normals"].ctypes.data,
                                 md["uvs"].ctypes.data, md["group"].ctypes.data, len(md["groups"]),
                                 make_pbr_array(md["groups"]), len(md["images"]), imgs, _f3(pos), _f3(rot), _f3(scale))
        if rc < 0:
            raise RuntimeError("Mesh: no faces in model")

    def close(self):
        if self.h:
            lib().po_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rays(self):
        w, h = self.w, self.h_px
        out = np.zeros((h, w, 6), np.float32)
        buf = (C.c_float * 6)()
        for y in range(h):
            for x in range(w):
                lib().po_camera_ray(C.byref(self.cam), x, y, w, h, buf)
                out[y, x] = buf[:]
        return out

    def rays_at(self, ys, xs):
        """Camera rays of the pixels (y, x) for y in ys, x in xs: [len(ys), len(xs), 6]."""
        w, h = self.w, self.h_px
        out = np.zeros((len(ys), len(xs), 6), np.float32)
        buf = (C.c_float * 6)()
        for i, y in enumerate(ys):
            for j, x in enumerate(xs):
                lib().po_camera_ray(C.byref(self.cam), int(x), int(y), w, h, buf)
                out[i, j] = buf[:]
        return out

    def closest_hits(self, rays):
        r = np.ascontiguousarray(rays.reshape(-1, 6), np.float32)
        out = np.zeros((r.shape[0], 20), np.float32)
        fp = C.POINTER(C.c_float)
        for i in range(r.shape[0]):
            lib().po_closest_hit(self.h, r[i, 0:3].ctypes.data_as(fp), r[i, 3:6].ctypes.data_as(fp), out[i].ctypes.data_as(fp))
        return out

    def render(self, spp, depth, albedo_only=False, seed=0x5EED, spp_begin=0, rect=None, threads=None,
               accum=None, samples=None, counters=False, russian_roulette=False):
        w, h = self.w, self.h_px
        if accum is None:
            accum = np.zeros((h, w, 3), np.float32)
            samples = np.zeros((h, w), np.uint32)
        x0, y0, x1, y1 = rect if rect else (0, 0, w, h)
        cnt = Counters()
        lib().po_render_ext(self.h, C.byref(self.cam), w, h, accum.ctypes.data, samples.ctypes.data, spp_begin, spp,
                            depth, int(albedo_only), seed, x0, y0, x1, y1, threads or os.cpu_count() or 1,
                            C.byref(cnt) if counters else None, 2 if russian_roulette else 0)
        if counters:
            return accum, samples, cnt.as_dict()
        return accum, samples


def average(accum, samples):
    """Film::AverageAt over the whole film (Film.cpp:48-53)."""
    s = samples.astype(np.float32)[..., None]
    return np.where(s > 0, accum / np.maximum(s, 1), 0).astype(np.float32)


def resolve(accum, samples, tonemap=1, colourspace=1):
    n = samples.size
    out = np.zeros((n, 4), np.uint8)
    a = np.ascontiguousarray(accum, np.float32)
    s = np.ascontiguousarray(samples, np.uint32)
    lib().po_film_resolve(a.ctypes.data, s.ctypes.data, n, tonemap, colourspace, out.ctypes.data)
    return out.reshape(samples.shape + (4,))


# ----------------------------------------------------------------------------- reference binary (build container only)
def have_ref():
    return os.path.exists(REF_BIN)


def ref_run(mode, scene, *args, workdir=None):
    """Runs oracle/_ref/ref_oracle on a preset dict; returns (stdout, tmpdir)."""
    d = workdir or tempfile.mkdtemp(prefix="ptref_")
    sp = os.path.join(d, "scene.txt")
    presets.write_scene_txt(scene, sp)
    out = subprocess.run([REF_BIN, mode, sp] + [str(a) for a in args], check=True, capture_output=True, text=True)
    return out.stdout, d


def ref_render(scene, spp, depth, albedo_only=False, threads=8, tasks=128):
    w, h = scene["film"]
    d = tempfile.mkdtemp(prefix="ptref_")
    pre = os.path.join(d, "out")
    stdout, _ = ref_run("render", scene, pre, spp, depth, int(albedo_only), threads, tasks, workdir=d)
    accum = np.fromfile(pre + ".accum.f32", np.float32).reshape(h, w, 3)
    samples = np.fromfile(pre + ".samples.u32", np.uint32).reshape(h, w)
    rgba = np.fromfile(pre + ".rgba8", np.uint8).reshape(h, w, 4)
    info = json.loads(stdout.strip().splitlines()[-1])
    return accum, samples, rgba, info


def ref_rays(scene):
    w, h = scene["film"]
    d = tempfile.mkdtemp(prefix="ptref_")
    ref_run("rays", scene, os.path.join(d, "rays.f32"), workdir=d)
    return np.fromfile(os.path.join(d, "rays.f32"), np.float32).reshape(h, w, 6)


def ref_hits(scene):
    w, h = scene["film"]
    d = tempfile.mkdtemp(prefix="ptref_")
    ref_run("hits", scene, os.path.join(d, "hits.f32"), workdir=d)
    return np.fromfile(os.path.join(d, "hits.f32"), np.float32).reshape(h, w, 20)
