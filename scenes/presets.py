"""Scene presets shared by the tests, the golden-fixture generator and bench.py.

A scene is a plain dict:
    {"film": (W, H),
     "camera": {"pos": (x,y,z), "rot": (rx,ry,rz) degrees, "fov": deg, "near": n, "far": f},
     "background": (r,g,b),
     "objects": [{"type": "sphere", "pos", "radius", "mat"},
                 {"type": "box", "pos", "rot", "size", "mat"},
                 {"type": "mesh", "path", "pos", "rot", "scale"}]}
"mat" is the 11-float material of the reference (RayObject.h:10-20):
albedo(3) roughness metallic emissionColour(3) emissionStrength IOR transmission.

The literals restate the scene built in the reference's main() (main.cpp:60-63, 218-305);
all arithmetic is done in float32 in the reference's order so the numbers match bit for bit.
"""
import numpy as np

f32 = np.float32


def material(albedo=(1, 1, 1), roughness=1.0, metallic=0.0, emission_colour=(1, 1, 1),
             emission_strength=0.0, ior=1.5, transmission=0.0):
    """Material with the reference defaults (RayObject.h:12-19)."""
    return [float(f32(v)) for v in (*albedo, roughness, metallic, *emission_colour,
                                    emission_strength, ior, transmission)]


# main.cpp:225-227
RED = (0.651, 0.051, 0.051)
GREEN = (0.122, 0.451, 0.149)
WHITE = (0.729, 0.729, 0.729)
LIGHT_COLOUR = (1.0, 0.84, 0.59)
LIGHT_STRENGTH = 15.0


def _v(*xs):
    return tuple(float(f32(x)) for x in xs)


def default_camera():
    """main.cpp:60-63 (double literals narrowed to float by glm::vec3)."""
    return {"pos": _v(f32(278 * 0.01), f32(273 * 0.01), f32(-800 * 0.01)), "rot": (0.0, 180.0, 0.0),
            "fov": 45.0, "near": 0.1, "far": 100.0}


def cornell_room_boxes():
    """Six thin wall boxes + the emissive ceiling light, main.cpp:221-305 (float32 arithmetic)."""
    S = f32(0.01)
    THICK = f32(0.01)
    half_t = THICK * f32(0.5) / S  # THICK * 0.5f / S
    objs = []

    def box(pos, size, mat):
        objs.append({"type": "box", "pos": _v(*pos), "rot": (0.0, 0.0, 0.0), "size": _v(*size), "mat": mat})

    def sv(x, y, z):  # S * glm::vec3(x, y, z)
        return (S * f32(x), S * f32(y), S * f32(z))

    long_z = S * f32(555.0) * f32(10)  # S * 555.0f * 10
    # Floor, Ceiling (main.cpp:232-247)
    box(sv(277.5, f32(0.0) + half_t, 277.5), (S * f32(555.0), THICK, long_z), material(WHITE))
    box(sv(277.5, f32(555.0) - half_t, 277.5), (S * f32(555.0), THICK, long_z), material(WHITE))
    # BackWall and the "FrontWall" that duplicates it (main.cpp:250-265)
    box(sv(277.5, 277.5, f32(555.0) - half_t), (S * f32(555.0), S * f32(555.0), THICK), material(WHITE))
    box(sv(277.5, 277.5, f32(555.0) - half_t), (S * f32(555.0), S * f32(555.0), THICK), material(WHITE))
    # LeftWall (red), RightWall (green) (main.cpp:268-283)
    box(sv(f32(555.0) - half_t, 277.5, 277.5), (THICK, S * f32(555.0), long_z), material(RED))
    box(sv(f32(0.0) + half_t, 277.5, 277.5), (THICK, S * f32(555.0), long_z), material(GREEN))
    # CeilingLight (main.cpp:290-304)
    ly = (f32(555.0) - half_t) - f32(100) * S
    box(sv(278.0, ly, 279.5), (S * f32(130.0), THICK * f32(0.6), S * f32(105.0)),
        material(albedo=(0, 0, 0), emission_colour=LIGHT_COLOUR, emission_strength=LIGHT_STRENGTH))
    return objs


def inner_blocks():
    """Short and tall Cornell blocks, main.cpp:313-334 (commented out there; part of the mesh preset)."""
    S = f32(0.01)

    def sv(x, y, z):
        return _v(S * f32(x), S * f32(y), S * f32(z))

    return [
        {"type": "box", "pos": sv(f32(130.0) + f32(165.0) * f32(0.5), f32(0.0) + f32(165.0) * f32(0.5), f32(65.0) + f32(165.0) * f32(0.5)),
         "rot": (0.0, -18.0, 0.0), "size": sv(165.0, 165.0, 165.0), "mat": material(WHITE)},
        {"type": "box", "pos": sv(f32(265.0) + f32(165.0) * f32(0.5), f32(0.0) + f32(330.0) * f32(0.5), f32(295.0) + f32(165.0) * f32(0.5)),
         "rot": (0.0, 22.5, 0.0), "size": sv(165.0, 330.0, 165.0), "mat": material(WHITE)},
    ]


def sphere_cornell(w=512, h=512):
    """BASELINE config 1: the reference room + a diffuse and a mirror sphere."""
    objs = cornell_room_boxes()
    objs.append({"type": "sphere", "pos": (1.8, 1.0, 3.5), "radius": 1.0, "mat": material(albedo=(0.8, 0.8, 0.8))})
    objs.append({"type": "sphere", "pos": (3.9, 0.8, 1.8), "radius": 0.8,
                 "mat": material(albedo=(0.9, 0.9, 0.9), roughness=0.0, metallic=1.0)})
    return {"film": (w, h), "camera": default_camera(), "background": (0.2, 0.2, 0.2), "objects": objs}


def box_cornell(w=512, h=512):
    """The room plus the two inner blocks as Box primitives (geometric twin of the mesh Cornell)."""
    return {"film": (w, h), "camera": default_camera(), "background": (0.2, 0.2, 0.2),
            "objects": cornell_room_boxes() + inner_blocks()}


def mesh_cornell(glb_path, w=1920, h=1080):
    """BASELINE config 2: the same room and blocks as one triangle mesh (see scenes/gltf_gen.py)."""
    return {"film": (w, h), "camera": default_camera(), "background": (0.2, 0.2, 0.2),
            "objects": [{"type": "mesh", "path": glb_path, "pos": (0.0, 0.0, 0.0), "rot": (0.0, 0.0, 0.0),
                         "scale": (1.0, 1.0, 1.0)}]}


def glass_mirror(w=512, h=512):
    """BASELINE config 5: glass, mirror and rough-metal spheres in the room (divergence stress)."""
    objs = cornell_room_boxes()
    objs.append({"type": "sphere", "pos": (1.6, 0.9, 2.6), "radius": 0.9,
                 "mat": material(albedo=(1, 1, 1), roughness=0.0, ior=1.5, transmission=1.0)})
    objs.append({"type": "sphere", "pos": (4.0, 0.8, 3.4), "radius": 0.8,
                 "mat": material(albedo=(0.95, 0.95, 0.95), roughness=0.0, metallic=1.0)})
    objs.append({"type": "sphere", "pos": (2.9, 0.55, 1.0), "radius": 0.55,
                 "mat": material(albedo=(0.9, 0.6, 0.2), roughness=0.35, metallic=1.0)})
    objs.append({"type": "sphere", "pos": (2.7, 2.6, 3.6), "radius": 0.6,
                 "mat": material(albedo=(0.8, 0.9, 1.0), roughness=0.2, ior=1.33, transmission=0.7)})
    objs.append({"type": "box", "pos": (4.3, 2.2, 2.0), "rot": (20.0, 35.0, 10.0), "size": (0.9, 0.9, 0.9),
                 "mat": material(albedo=(1, 1, 1), roughness=0.05, ior=1.5, transmission=0.9)})
    return {"film": (w, h), "camera": default_camera(), "background": (0.2, 0.2, 0.2), "objects": objs}


def mixed(mesh_path, w=512, h=512, mesh_pos=(2.775, 0.02, 2.8), mesh_rot=(-90.0, 0.0, 160.0), mesh_scale=(0.06, 0.06, 0.06)):
    """BASELINE config 4 style: room + spheres + one file mesh."""
    objs = cornell_room_boxes()
    objs.append({"type": "sphere", "pos": (0.9, 0.6, 1.2), "radius": 0.6,
                 "mat": material(albedo=(0.95, 0.95, 0.95), roughness=0.0, metallic=1.0)})
    objs.append({"type": "sphere", "pos": (4.6, 0.7, 1.6), "radius": 0.7,
                 "mat": material(albedo=(1, 1, 1), roughness=0.0, ior=1.5, transmission=1.0)})
    objs.append({"type": "mesh", "path": mesh_path, "pos": mesh_pos, "rot": mesh_rot, "scale": mesh_scale})
    return {"film": (w, h), "camera": default_camera(), "background": (0.2, 0.2, 0.2), "objects": objs}


def room_without_light():
    """The six wall boxes of the reference room without its ceiling light (for scenes lit by something else)."""
    return cornell_room_boxes()[:-1]


def lobe_scene(kind, w=64, h=36, mesh_path=None):
    """Small scenes that each put ONE branch of PathTracer::TraceRay's lobe choice (PathTracer.cpp:99-300) in front of the
    camera, for converged-image comparisons with the reference:
      rough_metal   GGX specular lobe alone (metallic 1: the diffuse weight is zero), PathTracer.cpp:213-283
      mirror        roughness 0 specular lobe (the hL = (0,0,1) shortcut), PathTracer.cpp:218-224
      plastic       diffuse + dielectric specular split by specProb, PathTracer.cpp:207-211, 284-300
      half_glass    transmission 0.5, roughness 0: transmission / opaque lobes each divided by their selection
                    probability, Fresnel split, refraction and total internal reflection, PathTracer.cpp:99-204
      rough_glass   transmission 1, roughness 0.3: GGX interface reflection, PathTracer.cpp:122-180
      glass_box     a transmissive Box: its hits never report a front face (Box.cpp:86-89), so eta_t = 1
      emitter_panel a mesh whose only light is an emissive texture: radiance = e * |e| (Mesh.cpp:476-477), room light removed
    """
    big = {"type": "sphere", "pos": (2.78, 1.45, 2.6), "radius": 1.45}
    side = {"type": "sphere", "pos": (4.55, 0.7, 1.5), "radius": 0.7, "mat": material(albedo=(0.2, 0.4, 0.9))}
    # A light that covers most of the ceiling at a sixth of the reference light's strength: paths find it often, so the
    # images converge about ten times faster than under the reference's small lamp and a lobe's weight shows in the means.
    panel = {"type": "box", "pos": (2.78, 5.44, 2.8), "rot": (0.0, 0.0, 0.0), "size": (4.4, 0.006, 4.4),
             "mat": material(albedo=(0, 0, 0), emission_colour=LIGHT_COLOUR, emission_strength=2.5)}
    objs = room_without_light() + [panel]
    if kind == "rough_metal":
        objs += [dict(big, mat=material(albedo=(0.95, 0.7, 0.3), roughness=0.4, metallic=1.0)), side]
    elif kind == "mirror":
        objs += [dict(big, mat=material(albedo=(0.9, 0.95, 0.9), roughness=0.0, metallic=1.0)), side]
    elif kind == "plastic":
        objs += [dict(big, mat=material(albedo=(0.8, 0.25, 0.2), roughness=0.25, metallic=0.0)), side]
    elif kind == "half_glass":
        objs += [dict(big, mat=material(albedo=(0.9, 0.8, 0.6), roughness=0.0, ior=1.5, transmission=0.5)), side]
    elif kind == "rough_glass":
        objs += [dict(big, mat=material(albedo=(1, 1, 1), roughness=0.3, ior=1.45, transmission=1.0)), side]
    elif kind == "glass_box":
        objs += [{"type": "box", "pos": (2.5, 1.3, 2.0), "rot": (15.0, 30.0, 5.0), "size": (2.2, 2.2, 1.2),
                  "mat": material(albedo=(1, 1, 1), roughness=0.0, ior=1.5, transmission=1.0)},
                 dict(side, pos=(2.9, 0.9, 3.9), radius=0.9)]
    elif kind == "emitter_panel":
        objs = room_without_light() + [dict(big, radius=1.0, pos=(2.78, 1.0, 2.8), mat=material(albedo=(0.8, 0.8, 0.8))),
                                       {"type": "mesh", "path": mesh_path, "pos": (2.78, 2.9, 4.9), "rot": (0.0, 0.0, 0.0), "scale": (1.0, 1.0, 1.0)}]
    else:
        raise ValueError(kind)
    cam = dict(default_camera(), pos=(2.78, 2.0, -2.5))  # closer than the reference's camera: the object fills a sixth of the frame
    return {"film": (w, h), "camera": cam, "background": (0.2, 0.2, 0.2), "objects": objs}


LOBE_KINDS = ("rough_metal", "mirror", "plastic", "half_glass", "rough_glass", "glass_box", "emitter_panel")


def write_scene_txt(scene, path):
    """Text form read by oracle/ref_driver.cpp. %.9g round-trips float32 exactly."""
    def fl(xs):
        return " ".join("%.9g" % float(f32(x)) for x in xs)
    lines = ["film %d %d" % tuple(scene["film"])]
    c = scene["camera"]
    lines.append("camera %s %s %s %s" % (fl(c["pos"]), fl(c["rot"]), fl([c["fov"]]), fl([c["near"], c["far"]])))
    bg = tuple(scene.get("background", (0.2, 0.2, 0.2)))
    if bg != (0.2, 0.2, 0.2):  # the reference's background is a private constant (PathTracer.h:21): its driver rejects this record
        lines.append("background %s" % fl(bg))
    for o in scene["objects"]:
        if o["type"] == "sphere":
            lines.append("sphere %s %s %s" % (fl(o["pos"]), fl([o["radius"]]), fl(o["mat"])))
        elif o["type"] == "box":
            lines.append("box %s %s %s %s" % (fl(o["pos"]), fl(o["rot"]), fl(o["size"]), fl(o["mat"])))
        elif o["type"] == "mesh":
            lines.append("mesh %s %s %s %s" % (o["path"], fl(o["pos"]), fl(o["rot"]), fl(o["scale"])))
        else:
            raise ValueError(o["type"])
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
