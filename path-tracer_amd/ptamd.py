"""Python host mirror of the reference's scene / camera / film vocabulary over the ptamd C ABI.

Class and method names follow the reference (RayObject.h, Sphere.h, Box.h, Mesh.h, Camera.h, Film.h,
PathTracer.h; frame driver main.cpp:30-47) so that a script written against the reference's C++
classes reads the same here:

    tracer = PathTracer(); tracer.AddRayObject(Sphere("s", (0, 1, 0), 1.0)); ...
    camera = Camera((w, h)); camera.SetFov(45); film = Film(w, h)
    RayTraceParallel((w, h), camera, tracer, film, depth, albedoOnly)       # one spp pass

Everything is computed by libptamd.so on the GPU; there is no CPU fallback: loading fails loudly when
the library is missing, and ptamd_init fails when no HIP device is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTAMD_LIB") or os.path.join(_HERE, "libptamd.so")  # PTAMD_LIB: development builds for A/B timing


class PtamdError(RuntimeError):
    pass


class CMaterial(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("roughness", C.c_float), ("metallic", C.c_float),
                ("emission_colour", C.c_float * 3), ("emission_strength", C.c_float), ("ior", C.c_float),
                ("transmission", C.c_float)]


class CPbr(C.Structure):
    _fields_ = [("base_color_tex", C.c_int32), ("normal_tex", C.c_int32), ("metallic_roughness_tex", C.c_int32),
                ("emissive_tex", C.c_int32), ("transmission_tex", C.c_int32), ("base_color_factor", C.c_float * 4),
                ("metallic_factor", C.c_float), ("roughness_factor", C.c_float), ("alpha_mode", C.c_int32),
                ("alpha_cutoff", C.c_float), ("normal_scale", C.c_float), ("emissive_factor", C.c_float * 3),
                ("transmission_factor", C.c_float), ("ior", C.c_float)]


class CImage(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("data", C.c_void_p)]


class CMeshDesc(C.Structure):
    _fields_ = [("num_faces", C.c_uint32), ("positions", C.c_void_p), ("normals", C.c_void_p),
                ("texcoords", C.c_void_p), ("group", C.c_void_p), ("num_groups", C.c_uint32),
                ("groups", C.POINTER(CPbr)), ("num_images", C.c_uint32), ("images", C.POINTER(CImage))]


class CCamera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("rotation_deg", C.c_float * 3), ("fov_deg", C.c_float),
                ("near_plane", C.c_float), ("far_plane", C.c_float)]


class CRect(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32)]


class CStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "rays", "nodes", "tris", "ray_trips", "node_trips", "tri_trips",
                                          "shade_lanes", "shade_trips", "steal_rounds")]


class CSceneInfo(C.Structure):
    _fields_ = [("objects", C.c_uint32), ("meshes", C.c_uint32), ("triangles", C.c_uint32), ("nodes", C.c_uint32),
                ("wide_nodes", C.c_uint32), ("lds_resident", C.c_uint32), ("geometry_uploads", C.c_uint64),
                ("row_updates", C.c_uint64), ("last_commit_seconds", C.c_double), ("last_bvh_seconds", C.c_double),
                ("d_nodes", C.c_void_p), ("d_tris", C.c_void_p), ("d_shade", C.c_void_p),
                ("leading_boxes", C.c_uint32), ("walk_cut", C.c_uint32)]


FLAG_COUNT = 1
FLAG_RUSSIAN_ROULETTE = 2

# every symbol include/ptamd.h declares (tests check that the library exports all of them)
EXPORTS = [
    "ptamd_init", "ptamd_set_stream", "ptamd_reset_stream", "ptamd_synchronize", "ptamd_last_error", "ptamd_material_default",
    "ptamd_pbr_default", "ptamd_scene_create", "ptamd_scene_destroy", "ptamd_scene_clear",
    "ptamd_scene_object_count", "ptamd_scene_add_sphere", "ptamd_scene_add_box", "ptamd_scene_add_mesh",
    "ptamd_scene_add_mesh_file", "ptamd_scene_set_transform", "ptamd_scene_set_radius",
    "ptamd_scene_set_material", "ptamd_scene_set_mesh_group", "ptamd_scene_set_background", "ptamd_scene_set_skybox", "ptamd_scene_set_skybox_f32",
    "ptamd_scene_commit", "ptamd_scene_get_info", "ptamd_model_load", "ptamd_model_load_ex", "ptamd_model_desc", "ptamd_model_group_name", "ptamd_model_free", "ptamd_image_decode", "ptamd_image_free",
    "ptamd_film_create", "ptamd_film_create_external", "ptamd_film_destroy", "ptamd_film_resize",
    "ptamd_film_reset", "ptamd_film_size", "ptamd_film_read", "ptamd_film_average",
    "ptamd_film_resolve_rgba8", "ptamd_film_device_ptrs", "ptamd_render", "ptamd_render_tiles", "ptamd_render_stats", "ptamd_render_counted_noalloc",
    "ptamd_camera_rays", "ptamd_closest_hits", "ptamd_traversal_counts",
]

_lib = None
_inited = None


def lib():
    """Loads libptamd.so (no GPU needed for loading; compute calls need ptamd_init)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PtamdError("libptamd.so is not built (run `make -C path-tracer_amd` or __graft_entry__.build())")
        try:
            # PyTorch wheels carry their own libamdhip64; two HIP runtimes in one process do not share devices.
            # Importing torch first makes the dynamic linker resolve this library's libamdhip64 dependency to the
            # copy that is already loaded, so tensors and ptamd buffers live in one runtime.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, f3 = C.c_void_p, C.POINTER(C.c_float)
        L.ptamd_last_error.restype = C.c_char_p
        L.ptamd_init.argtypes = [C.c_int]
        L.ptamd_set_stream.argtypes = [vp]
        L.ptamd_material_default.argtypes = [C.POINTER(CMaterial)]
        L.ptamd_pbr_default.argtypes = [C.POINTER(CPbr)]
        L.ptamd_scene_create.argtypes = [C.POINTER(vp)]
        L.ptamd_scene_destroy.argtypes = [vp]
        L.ptamd_scene_clear.argtypes = [vp]
        L.ptamd_scene_object_count.argtypes = [vp]
        L.ptamd_scene_add_sphere.argtypes = [vp, f3, C.c_float, C.POINTER(CMaterial)]
        L.ptamd_scene_add_box.argtypes = [vp, f3, f3, f3, C.POINTER(CMaterial)]
        L.ptamd_scene_add_mesh.argtypes = [vp, C.POINTER(CMeshDesc), f3, f3, f3]
        L.ptamd_scene_add_mesh_file.argtypes = [vp, C.c_char_p, f3, f3, f3]
        L.ptamd_scene_set_transform.argtypes = [vp, C.c_int, f3, f3, f3]
        L.ptamd_scene_set_radius.argtypes = [vp, C.c_int, C.c_float]
        L.ptamd_scene_set_material.argtypes = [vp, C.c_int, C.POINTER(CMaterial)]
        L.ptamd_scene_set_mesh_group.argtypes = [vp, C.c_int, C.c_uint32, C.POINTER(CPbr)]
        L.ptamd_scene_set_background.argtypes = [vp, f3]
        L.ptamd_scene_set_skybox.argtypes = [vp, C.c_int, C.POINTER(C.c_void_p), C.c_float]
        L.ptamd_scene_set_skybox_f32.argtypes = [vp, C.c_int, C.POINTER(C.c_void_p)]
        L.ptamd_image_decode.argtypes = [vp, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(vp)]
        L.ptamd_image_free.argtypes = [vp]
        L.ptamd_image_free.restype = None
        L.ptamd_scene_commit.argtypes = [vp]
        L.ptamd_scene_get_info.argtypes = [vp, C.POINTER(CSceneInfo)]
        L.ptamd_model_load.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.ptamd_model_load_ex.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(vp)]
        L.ptamd_model_desc.argtypes = [vp, C.POINTER(CMeshDesc)]
        L.ptamd_model_group_name.argtypes = [vp, C.c_uint32]
        L.ptamd_model_group_name.restype = C.c_char_p
        L.ptamd_model_free.argtypes = [vp]
        L.ptamd_model_free.restype = None
        L.ptamd_film_create.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
        L.ptamd_film_create_external.argtypes = [C.c_int, C.c_int, vp, vp, C.POINTER(vp)]
        L.ptamd_film_destroy.argtypes = [vp]
        L.ptamd_film_resize.argtypes = [vp, C.c_int, C.c_int]
        L.ptamd_film_reset.argtypes = [vp]
        L.ptamd_film_size.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ptamd_film_read.argtypes = [vp, vp, vp]
        L.ptamd_film_average.argtypes = [vp, vp]
        L.ptamd_film_resolve_rgba8.argtypes = [vp, C.c_int, C.c_int, vp]
        L.ptamd_film_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
        L.ptamd_render.argtypes = [vp, C.POINTER(CCamera), vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                   C.c_uint32, C.POINTER(CRect), C.c_uint32]
        L.ptamd_render_tiles.argtypes = [vp, C.POINTER(CCamera), vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint32,
                                         C.c_int, C.c_int, C.c_int, C.c_uint32]
        L.ptamd_render_stats.argtypes = [C.POINTER(CStats)]
        L.ptamd_render_counted_noalloc.argtypes = [vp, C.POINTER(CCamera), vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                                   C.c_uint32, C.POINTER(CRect), C.c_uint32, vp]
        L.ptamd_camera_rays.argtypes = [C.POINTER(CCamera), C.c_int, C.c_int, vp]
        L.ptamd_closest_hits.argtypes = [vp, vp, C.c_uint32, vp]
        L.ptamd_traversal_counts.argtypes = [vp, vp, C.c_uint32, vp, vp]
        _lib = L
    return _lib


def check(rc):
    if rc < 0:
        raise PtamdError("ptamd error %d: %s" % (rc, lib().ptamd_last_error().decode(errors="replace")))
    return rc


def init(device=0):
    global _inited
    if _inited != device:
        check(lib().ptamd_init(device))
        _inited = device


def set_stream(stream_ptr):
    check(lib().ptamd_set_stream(C.c_void_p(stream_ptr)))


def reset_stream():
    check(lib().ptamd_reset_stream())


def synchronize():
    check(lib().ptamd_synchronize())


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Material:
    """RayObject.h:10-20 (same field names, same defaults)."""

    def __init__(self, albedo=(1.0, 1.0, 1.0), roughness=1.0, metallic=0.0, emissionColour=(1.0, 1.0, 1.0),
                 emissionStrength=0.0, IOR=1.5, transmission=0.0):
        self.albedo = tuple(albedo)
        self.roughness = roughness
        self.metallic = metallic
        self.emissionColour = tuple(emissionColour)
        self.emissionStrength = emissionStrength
        self.IOR = IOR
        self.transmission = transmission

    @staticmethod
    def from_list(m):
        return Material(m[0:3], m[3], m[4], m[5:8], m[8], m[9], m[10])

    def to_c(self):
        c = CMaterial()
        c.albedo[:] = self.albedo
        c.roughness, c.metallic = self.roughness, self.metallic
        c.emission_colour[:] = self.emissionColour
        c.emission_strength, c.ior, c.transmission = self.emissionStrength, self.IOR, self.transmission
        return c


def make_pbr(d):
    p = CPbr()
    lib().ptamd_pbr_default(C.byref(p))
    for k, v in d.items():
        if k in ("base_color_factor", "emissive_factor"):
            getattr(p, k)[:] = v
        elif k != "name":
            setattr(p, k, v)
    return p


class RayObject:
    """RayObject.h:32-63."""

    def __init__(self, name="Object"):
        self.mName = name
        self.mPosition = (0.0, 0.0, 0.0)
        self.mRotation = (0.0, 0.0, 0.0)
        self.mMaterial = Material()
        self._scene = None
        self._index = -1

    def SetName(self, n): self.mName = n
    def GetName(self): return self.mName
    def GetPosition(self): return self.mPosition
    def GetRotation(self): return self.mRotation
    def GetMaterial(self): return self.mMaterial

    def SetPosition(self, p):
        self.mPosition = tuple(p)
        self._push()

    def SetRotation(self, r):
        self.mRotation = tuple(r)
        self._push()

    def SetMaterial(self, m):
        self.mMaterial = m
        if self._scene is not None and not isinstance(self, Mesh):
            check(lib().ptamd_scene_set_material(self._scene._h, self._index, C.byref(m.to_c())))

    def _size3(self):
        return None

    def _push(self):
        if self._scene is not None:
            s = self._size3()
            check(lib().ptamd_scene_set_transform(self._scene._h, self._index, _f3(self.mPosition), _f3(self.mRotation),
                                                  _f3(s) if s is not None else None))


class Sphere(RayObject):
    """Sphere.h:7-25."""

    def __init__(self, name="Sphere", position=(0.0, 0.0, 0.0), radius=1.0, albedo=None):
        super().__init__(name)
        self.mPosition = tuple(position)
        self.mRadius = radius
        if albedo is not None:
            self.mMaterial.albedo = tuple(albedo)

    def SetRadius(self, r):
        self.mRadius = r
        if self._scene is not None:
            check(lib().ptamd_scene_set_radius(self._scene._h, self._index, float(r)))

    def GetRadius(self): return self.mRadius

    def _add(self, h):
        return lib().ptamd_scene_add_sphere(h, _f3(self.mPosition), float(self.mRadius), C.byref(self.mMaterial.to_c()))


class Box(RayObject):
    """Box.h:7-23."""

    def __init__(self, name="Box"):
        super().__init__(name)
        self.mSize = (1.0, 1.0, 1.0)

    def SetSize(self, s):
        self.mSize = tuple(s)
        self._push()

    def GetSize(self): return self.mSize
    def _size3(self): return self.mSize

    def _add(self, h):
        return lib().ptamd_scene_add_box(h, _f3(self.mPosition), _f3(self.mRotation), _f3(self.mSize), C.byref(self.mMaterial.to_c()))


class Mesh(RayObject):
    """Mesh.h:11-63: constructed from a .glb/.gltf path (Mesh.cpp:7-13) or from triangle arrays
    (dict with positions (F,9), normals, uvs (F,6), group (F,), groups [pbr dicts], images [(H,W,C) uint8])."""

    def __init__(self, file_path=None, name="Mesh", data=None, node_transforms=False, allow_obj=False):
        super().__init__(name)
        self.mScale = (1.0, 1.0, 1.0)
        self._path = file_path
        self._data = data
        if file_path is None and data is None:
            raise PtamdError("Mesh needs a file path or triangle data")
        if data is None and (node_transforms or allow_obj):  # loader options the reference does not have
            self._data = load_model(file_path, node_transforms=node_transforms, allow_obj=allow_obj)

    def SetScale(self, s):
        self.mScale = tuple(s)
        self._push()

    def GetScale(self): return self.mScale
    def _size3(self): return self.mScale

    def SetGroupMaterial(self, group, pbr):
        """Edits one material group (the ImGui editor of Mesh.cpp:267-284)."""
        check(lib().ptamd_scene_set_mesh_group(self._scene._h, self._index, group, C.byref(make_pbr(pbr))))

    def _add(self, h):
        if self._data is None:
            return lib().ptamd_scene_add_mesh_file(h, self._path.encode(), _f3(self.mPosition), _f3(self.mRotation), _f3(self.mScale))
        d = self._data
        desc = CMeshDesc()
        P = np.ascontiguousarray(d["positions"], np.float32)
        N = np.ascontiguousarray(d["normals"], np.float32) if d.get("normals") is not None else None
        T = np.ascontiguousarray(d["uvs"], np.float32) if d.get("uvs") is not None else None
        G = np.ascontiguousarray(d["group"], np.int32) if d.get("group") is not None else None
        groups = d.get("groups", [])
        images = [np.ascontiguousarray(im, np.uint8) for im in d.get("images", [])]
        garr = (CPbr * max(1, len(groups)))(*[make_pbr(g) for g in groups])
        iarr = (CImage * max(1, len(images)))()
        for i, im in enumerate(images):
            iarr[i].height, iarr[i].width, iarr[i].channels = im.shape
            iarr[i].data = im.ctypes.data
        desc.num_faces = P.shape[0]
        desc.positions = P.ctypes.data
        desc.normals = N.ctypes.data if N is not None else None
        desc.texcoords = T.ctypes.data if T is not None else None
        desc.group = G.ctypes.data if G is not None else None
        desc.num_groups, desc.groups = len(groups), garr
        desc.num_images, desc.images = len(images), iarr
        return lib().ptamd_scene_add_mesh(h, C.byref(desc), _f3(self.mPosition), _f3(self.mRotation), _f3(self.mScale))


class PathTracer:
    """PathTracer.h:8-24: the scene list. Commit() uploads it (BVH build + SoA flattening)."""

    def __init__(self):
        init()
        h = C.c_void_p()
        check(lib().ptamd_scene_create(C.byref(h)))
        self._h = h
        self.rayObjects = []

    def AddRayObject(self, obj):
        idx = check(obj._add(self._h))
        obj._scene, obj._index = self, idx
        self.rayObjects.append(obj)

    def GetRayObjects(self): return self.rayObjects
    def GetSizeOfRayObjects(self): return check(lib().ptamd_scene_object_count(self._h))

    def ClearScene(self):
        check(lib().ptamd_scene_clear(self._h))
        for o in self.rayObjects:
            o._scene, o._index = None, -1
        self.rayObjects = []

    def SetBackgroundColour(self, rgb):
        check(lib().ptamd_scene_set_background(self._h, _f3(rgb)))

    def SetSkybox(self, faces, intensity=1.0):
        """Cubemap miss radiance (the reference has only the constant background): faces is a (6, N, N, 4) uint8
        array in +x -x +y -y +z -z order, row 0 at the top; None removes it."""
        if faces is None:
            check(lib().ptamd_scene_set_skybox(self._h, 0, None, 0.0))
            return
        f = np.ascontiguousarray(faces, np.uint8)
        if f.ndim != 4 or f.shape[0] != 6 or f.shape[1] != f.shape[2] or f.shape[3] != 4:
            raise ValueError("skybox faces must be (6, N, N, 4) uint8")
        ptrs = (C.c_void_p * 6)(*[f[i].ctypes.data for i in range(6)])
        check(lib().ptamd_scene_set_skybox(self._h, f.shape[1], ptrs, float(intensity)))

    def SetSkyboxFloat(self, faces):
        """The same with float radiance (an HDR map): faces is a (6, N, N, 3) float32 array, returned by a miss as it
        is; None removes it."""
        if faces is None:
            check(lib().ptamd_scene_set_skybox_f32(self._h, 0, None))
            return
        f = np.ascontiguousarray(faces, np.float32)
        if f.ndim != 4 or f.shape[0] != 6 or f.shape[1] != f.shape[2] or f.shape[3] != 3:
            raise ValueError("float skybox faces must be (6, N, N, 3) float32")
        ptrs = (C.c_void_p * 6)(*[f[i].ctypes.data for i in range(6)])
        check(lib().ptamd_scene_set_skybox_f32(self._h, f.shape[1], ptrs))

    def Commit(self):
        check(lib().ptamd_scene_commit(self._h))

    def Info(self):
        """Upload-path diagnostics (ptamd_scene_get_info) as a dict."""
        i = CSceneInfo()
        check(lib().ptamd_scene_get_info(self._h, C.byref(i)))
        return {n: getattr(i, n) for n, _ in CSceneInfo._fields_}

    def ClosestHits(self, rays):
        """Closest-hit records (PathTracer.cpp:38-57) for an (N,6) ray array."""
        r = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros((r.shape[0], 20), np.float32)
        check(lib().ptamd_closest_hits(self._h, r.ctypes.data, r.shape[0], out.ctypes.data))
        return out

    def TraversalCounts(self, rays):
        """(records, counts): counts[:, 0] BVH nodes visited, counts[:, 1] triangle tests per ray on the GPU."""
        r = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros((r.shape[0], 20), np.float32)
        cnt = np.zeros((r.shape[0], 2), np.uint32)
        check(lib().ptamd_traversal_counts(self._h, r.ctypes.data, r.shape[0], out.ctypes.data, cnt.ctypes.data))
        return out, cnt

    def close(self):
        if self._h:
            lib().ptamd_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Camera:
    """Camera.h:5-51."""

    def __init__(self, winSize, position=(0.0, 0.0, 0.0), rotation=(0.0, 0.0, 0.0)):
        self.mLastWinSize = tuple(winSize)
        self.mPosition, self.mRotation = tuple(position), tuple(rotation)
        self.mFov, self.mNearPlane, self.mFarPlane = 40.0, 0.1, 100.0

    def SetPosition(self, p): self.mPosition = tuple(p)
    def GetPosition(self): return self.mPosition
    def SetRotation(self, r): self.mRotation = tuple(r)
    def GetRotation(self): return self.mRotation
    def SetFov(self, f): self.mFov = f
    def GetFov(self): return self.mFov
    def SetNearPlane(self, n): self.mNearPlane = n
    def GetNearPlane(self): return self.mNearPlane
    def SetFarPlane(self, f): self.mFarPlane = f
    def GetFarPlane(self): return self.mFarPlane

    def to_c(self):
        c = CCamera()
        c.position[:] = self.mPosition
        c.rotation_deg[:] = self.mRotation
        c.fov_deg, c.near_plane, c.far_plane = self.mFov, self.mNearPlane, self.mFarPlane
        return c

    def GetRays(self, winSize=None):
        """Camera::GetRay for every pixel: (H, W, 6)."""
        init()
        w, h = winSize or self.mLastWinSize
        out = np.zeros((h, w, 6), np.float32)
        check(lib().ptamd_camera_rays(C.byref(self.to_c()), w, h, out.ctypes.data))
        return out


class Film:
    """Film.h:19-61. ColourSpace: 0 Linear, 1 sRGB; ToneMap: 0 None, 1 Reinhard (defaults sRGB + Reinhard)."""
    Linear, sRGB = 0, 1
    NoToneMap, Reinhard = 0, 1

    def __init__(self, width, height, external=None):
        init()
        h = C.c_void_p()
        if external is None:
            check(lib().ptamd_film_create(width, height, C.byref(h)))
        else:
            check(lib().ptamd_film_create_external(width, height, C.c_void_p(external[0]), C.c_void_p(external[1]), C.byref(h)))
        self._h = h
        self.mColourSpace, self.mToneMap = Film.sRGB, Film.Reinhard

    def Resize(self, w, h): check(lib().ptamd_film_resize(self._h, w, h))
    def Reset(self): check(lib().ptamd_film_reset(self._h))
    def SetColourSpace(self, cs): self.mColourSpace = cs
    def GetColourSpace(self): return self.mColourSpace
    def SetToneMap(self, tm): self.mToneMap = tm
    def GetToneMap(self): return self.mToneMap

    def _size(self):
        w, h = C.c_int(), C.c_int()
        check(lib().ptamd_film_size(self._h, C.byref(w), C.byref(h)))
        return w.value, h.value

    def Width(self): return self._size()[0]
    def Height(self): return self._size()[1]
    def PixelCount(self): return self.Width() * self.Height()

    def Accum(self):
        w, h = self._size()
        a = np.zeros((h, w, 3), np.float32)
        check(lib().ptamd_film_read(self._h, a.ctypes.data, None))
        return a

    def Samples(self):
        w, h = self._size()
        s = np.zeros((h, w), np.uint32)
        check(lib().ptamd_film_read(self._h, None, s.ctypes.data))
        return s

    def Average(self):
        """Film::AverageAt for every pixel: (H, W, 3)."""
        w, h = self._size()
        a = np.zeros((h, w, 3), np.float32)
        check(lib().ptamd_film_average(self._h, a.ctypes.data))
        return a

    def ResolveToRGBA8(self):
        w, h = self._size()
        out = np.zeros((h, w, 4), np.uint8)
        check(lib().ptamd_film_resolve_rgba8(self._h, self.mToneMap, self.mColourSpace, out.ctypes.data))
        return out

    def close(self):
        if self._h:
            lib().ptamd_film_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def Render(camera, pathTracer, film, spp_count, depth, albedoOnly=False, spp_begin=0, seed=0x5EED, rect=None, count=False,
           russian_roulette=False, tiles=None):
    """spp_count passes of RayTraceParallel in one launch (asynchronous). russian_roulette is an option the
    reference does not have (off: every path runs to `depth` as in PathTracer.cpp:35-36). tiles = (size, phase,
    stride): only the square tiles phase, phase + stride, ... of the film (one rank's share of a tile split)."""
    r = CRect(*rect) if rect is not None else None
    flags = (FLAG_COUNT if count else 0) | (FLAG_RUSSIAN_ROULETTE if russian_roulette else 0)
    if tiles is not None:
        if rect is not None:
            raise ValueError("rect and tiles exclude each other")
        size, phase, stride = tiles
        check(lib().ptamd_render_tiles(pathTracer._h, C.byref(camera.to_c()), film._h, spp_begin, spp_count, depth,
                                       int(bool(albedoOnly)), seed, int(size), int(phase), int(stride), flags))
        return
    check(lib().ptamd_render(pathTracer._h, C.byref(camera.to_c()), film._h, spp_begin, spp_count, depth,
                             int(bool(albedoOnly)), seed, C.byref(r) if r is not None else None, flags))


def RayTraceParallel(winSize, camera, pathTracer, film, depth, albedoOnly, frame=0, seed=0x5EED):
    """main.cpp:30-47: one sample per pixel over the whole frame. The thread pool and task count of the
    reference have no meaning here; `frame` is the pass number (the sample index)."""
    Render(camera, pathTracer, film, 1, depth, albedoOnly, spp_begin=frame, seed=seed)


def RenderStats():
    s = CStats()
    check(lib().ptamd_render_stats(C.byref(s)))
    return {n: int(getattr(s, n)) for n, _ in CStats._fields_}


def build_scene(scene, mesh_data=None):
    """PathTracer + Camera from a scenes/presets.py dict. mesh_data: {path: triangle dict} to bypass the file loader."""
    pt = PathTracer()
    for o in scene["objects"]:
        if o["type"] == "sphere":
            s = Sphere("Sphere", o["pos"], o["radius"])
            s.mMaterial = Material.from_list(o["mat"])
            pt.AddRayObject(s)
        elif o["type"] == "box":
            b = Box("Box")
            b.mPosition, b.mRotation, b.mSize = tuple(o["pos"]), tuple(o["rot"]), tuple(o["size"])
            b.mMaterial = Material.from_list(o["mat"])
            pt.AddRayObject(b)
        elif o["type"] == "mesh":
            data = o.get("data")
            if data is None and mesh_data is not None:
                data = mesh_data.get(o["path"])
            m = Mesh(o.get("path"), "Mesh", data=data)
            m.mPosition, m.mRotation, m.mScale = tuple(o["pos"]), tuple(o["rot"]), tuple(o["scale"])
            pt.AddRayObject(m)
        else:
            raise ValueError(o["type"])
    if "background" in scene:
        pt.SetBackgroundColour(scene["background"])
    if scene.get("skybox") is not None:
        if np.asarray(scene["skybox"]["faces"]).dtype == np.uint8:
            pt.SetSkybox(scene["skybox"]["faces"], scene["skybox"].get("intensity", 1.0))
        else:
            pt.SetSkyboxFloat(scene["skybox"]["faces"])
    pt.Commit()
    c = scene["camera"]
    cam = Camera(scene["film"], c["pos"], c["rot"])
    cam.SetFov(c["fov"])
    cam.SetNearPlane(c.get("near", 0.1))
    cam.SetFarPlane(c.get("far", 100.0))
    return pt, cam


def decode_image(data):
    """An encoded PNG or JPEG (bytes) as the loader decodes a model's images: (H, W, 4) uint8. Needs no GPU."""
    w, h, px = C.c_int(), C.c_int(), C.c_void_p()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    check(lib().ptamd_image_decode(C.cast(buf, C.c_void_p), len(data), C.byref(w), C.byref(h), C.byref(px)))
    try:
        return np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 4)).copy()
    finally:
        lib().ptamd_image_free(px)


LOAD_NODE_TRANSFORMS = 1
LOAD_OBJ = 2


def load_model(path, node_transforms=False, allow_obj=False):
    """ModelLoader(path) through the C ABI: returns the triangle dict used by Mesh(data=...). The two options are
    not in the reference: applying the glTF scene graph, and reading Wavefront .obj files."""
    h = C.c_void_p()
    flags = (LOAD_NODE_TRANSFORMS if node_transforms else 0) | (LOAD_OBJ if allow_obj else 0)
    check(lib().ptamd_model_load_ex(path.encode(), flags, C.byref(h)))
    try:
        d = CMeshDesc()
        check(lib().ptamd_model_desc(h, C.byref(d)))
        nf = d.num_faces

        def arr(ptr, n, dt):
            if not ptr or n == 0:
                return np.zeros((0,), dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).copy()
        out = {"positions": arr(d.positions, nf * 9, np.float32).reshape(nf, 9),
               "normals": arr(d.normals, nf * 9, np.float32).reshape(nf, 9),
               "uvs": arr(d.texcoords, nf * 6, np.float32).reshape(nf, 6),
               "group": arr(d.group, nf, np.int32), "groups": [], "images": []}
        for i in range(d.num_groups):
            g = d.groups[i]
            out["groups"].append({
                "name": (lib().ptamd_model_group_name(h, i) or b"").decode(),
                "base_color_tex": g.base_color_tex, "normal_tex": g.normal_tex,
                "metallic_roughness_tex": g.metallic_roughness_tex, "emissive_tex": g.emissive_tex,
                "transmission_tex": g.transmission_tex, "base_color_factor": list(g.base_color_factor),
                "metallic_factor": g.metallic_factor, "roughness_factor": g.roughness_factor,
                "alpha_mode": g.alpha_mode, "alpha_cutoff": g.alpha_cutoff, "normal_scale": g.normal_scale,
                "emissive_factor": list(g.emissive_factor), "transmission_factor": g.transmission_factor, "ior": g.ior})
        for i in range(d.num_images):
            im = d.images[i]
            n = im.width * im.height * im.channels
            px = arr(im.data, n, np.uint8).reshape(im.height, im.width, im.channels) if n else np.zeros((0, 0, 4), np.uint8)
            out["images"].append(px)
        return out
    finally:
        lib().ptamd_model_free(h)
