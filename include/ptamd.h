/* ptamd -- MI355X-native path-tracing core, C ABI.
 *
 * Drop-in boundary for the frame-render seam of jtothec67/Path-Tracer:
 *   RayTraceParallel (src/PathTracer/main.cpp:30-47) = Camera::GetRay (Camera.h:12)
 *   -> PathTracer::TraceRay (PathTracer.h:11) -> Film::AddSample (Film.h:29),
 * plus the scene / camera / film state those calls read.  The reference has no FFI of its
 * own; each entry point below names the reference interface it replaces.  Plain pointers and
 * sizes only; no exceptions cross this boundary; every call returns PTAMD_OK (0) or a negative
 * error code and ptamd_last_error() holds the message (thread-local).
 *
 * Ownership: handles own their host and device memory until *_destroy.  Array arguments are
 * copied before the call returns.  One host thread per device context; calls on one handle
 * must not overlap.
 */
#ifndef PTAMD_H
#define PTAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTAMD_OK 0
#define PTAMD_ERR_ARG (-1)      /* bad argument (null handle, bad index, bad size) */
#define PTAMD_ERR_IO (-2)       /* file missing / unreadable / unsupported (loader errors of ModelLoader.h:137-142,209-217,253-256,403) */
#define PTAMD_ERR_STATE (-3)    /* scene not committed, film size mismatch, ... */
#define PTAMD_ERR_DEVICE (-4)   /* HIP runtime error */
#define PTAMD_ERR_LIMIT (-5)    /* a documented limit exceeded (BVH depth, object count) */

/* Material, RayObject.h:10-20 (same field order, same defaults when built with ptamd_material_default). */
typedef struct ptamd_material {
    float albedo[3];
    float roughness;
    float metallic;
    float emission_colour[3];
    float emission_strength;
    float ior;
    float transmission;
} ptamd_material;

/* ModelLoader::PBRMaterial, ModelLoader.h:56-85 (occlusion fields omitted: the reference never reads them while tracing). */
typedef struct ptamd_pbr {
    int32_t base_color_tex, normal_tex, metallic_roughness_tex, emissive_tex, transmission_tex; /* image index or -1 */
    float base_color_factor[4];
    float metallic_factor, roughness_factor;
    int32_t alpha_mode; /* 0 opaque, 1 mask, 2 blend (blend is treated as opaque, as the reference does) */
    float alpha_cutoff;
    float normal_scale;
    float emissive_factor[3];
    float transmission_factor;
    float ior;
} ptamd_pbr;

/* ModelLoader::EmbeddedImage, ModelLoader.h:48-54: 8-bit interleaved texels, row 0 first. */
typedef struct ptamd_image {
    int32_t width, height, channels;
    const uint8_t* data;
} ptamd_image;

/* De-indexed triangle soup as ModelLoader::Face (ModelLoader.h:29-40) stores it. */
typedef struct ptamd_mesh_desc {
    uint32_t num_faces;
    const float* positions;   /* num_faces*9: a.xyz b.xyz c.xyz */
    const float* normals;     /* num_faces*9 or NULL (zeros, the Vertex default) */
    const float* texcoords;   /* num_faces*6 or NULL */
    const int32_t* group;     /* num_faces material-group indices, -1 = none; NULL = all -1 */
    uint32_t num_groups;
    const ptamd_pbr* groups;
    uint32_t num_images;
    const ptamd_image* images;
} ptamd_mesh_desc;

/* Camera state, Camera.h:37-42. Rotation is Euler XYZ in degrees, fov is vertical, degrees. */
typedef struct ptamd_camera {
    float position[3];
    float rotation_deg[3];
    float fov_deg;
    float near_plane, far_plane;
} ptamd_camera;

typedef struct ptamd_rect { int32_t x0, y0, x1, y1; } ptamd_rect; /* [x0,x1) x [y0,y1), y = 0 is the bottom row */

/* Counters of the last ptamd_render / ptamd_render_tiles call that passed PTAMD_FLAG_COUNT (each scene tallies into a
 * buffer of its own; ptamd_render_stats reads the one used last). */
typedef struct ptamd_stats {
    uint64_t samples; /* paths started (top-level TraceRay calls) */
    uint64_t rays;    /* path segments answered, primary + bounces (scenes with meshes read from global memory keep the
                         camera ray's hit per pixel after the first sample: those segments count, their traversal is not redone) */
    uint64_t nodes;   /* BVH inner-node visits (each tests both child boxes) */
    uint64_t tris;    /* ray-triangle tests */
    /* Wave-level loop trips (64 lanes issue together): lanes-per-trip = rays / ray_trips, nodes / node_trips,
     * tris / tri_trips, shade_lanes / shade_trips is how full the wave was in that loop. */
    uint64_t ray_trips, node_trips, tri_trips;
    uint64_t shade_lanes, shade_trips; /* hits shaded, and wave-level passes through the shading block */
    uint64_t steal_rounds;             /* wave-level rounds in which idle lanes took over pending subtrees of busy lanes */
} ptamd_stats;

typedef struct ptamd_scene ptamd_scene;
typedef struct ptamd_film ptamd_film;
typedef struct ptamd_model ptamd_model;

#define PTAMD_FLAG_COUNT 1u  /* fill ptamd_stats (a counting build of the kernel: slower, not for timing) */
/* No reference counterpart (its bounce loop always runs to maxDepth, PathTracer.cpp:35-36): from the third
 * bounce on a path continues with probability q = clamp(max(throughput), 0.05, 1) and is divided by q.
 * Leaves the expected radiance unchanged; what reaches the film can still differ a little because
 * Film::AddSample clips each sample's luminance at 12 (Film.cpp:38-43) and rescaled samples reach that
 * clamp more often. Off by default so that default renders follow the reference. */
#define PTAMD_FLAG_RUSSIAN_ROULETTE 2u

/* ---- context ------------------------------------------------------------------------------ */
/* Selects the HIP device for this process and creates the library's stream.  No reference
 * counterpart (the reference is CPU-only).  Fails with PTAMD_ERR_DEVICE when no GPU is usable:
 * there is no CPU fallback.  One process drives one GPU: a second call must name the same device.
 * The library keeps process-wide state (device, stream); calls that touch the same scene or
 * film handle, or that change the stream, must not overlap (the last-error string is per thread). */
int ptamd_init(int device);
/* Use a caller-owned hipStream_t (e.g. PyTorch's current stream) for all later launches, copies and memsets.
 * NULL is HIP's default (null) stream. ptamd_reset_stream goes back to the library's own non-blocking stream. */
int ptamd_set_stream(void* hip_stream);
int ptamd_reset_stream(void);
/* Blocks until all work queued by this library has finished. */
int ptamd_synchronize(void);
const char* ptamd_last_error(void);
void ptamd_material_default(ptamd_material* out); /* Material{} defaults, RayObject.h:12-19 */
void ptamd_pbr_default(ptamd_pbr* out);           /* PBRMaterial{} defaults, ModelLoader.h:60-84 */

/* ---- scene: PathTracer::AddRayObject / GetSizeOfRayObjects / ClearScene (PathTracer.h:13-18) -- */
int ptamd_scene_create(ptamd_scene** out);
int ptamd_scene_destroy(ptamd_scene* s);
int ptamd_scene_clear(ptamd_scene* s);
int ptamd_scene_object_count(const ptamd_scene* s);
/* Sphere(name,pos,radius)+SetMaterial (Sphere.h:10-20). Returns the object index (>= 0). */
int ptamd_scene_add_sphere(ptamd_scene* s, const float pos[3], float radius, const ptamd_material* mat);
/* Box + SetPosition/SetRotation/SetSize/SetMaterial (Box.h:10-18, RayObject.h:44-51). */
int ptamd_scene_add_box(ptamd_scene* s, const float pos[3], const float rot_deg[3], const float size[3],
                        const ptamd_material* mat);
/* Mesh from triangle data + SetPosition/SetRotation/SetScale (Mesh.h:14-22). */
int ptamd_scene_add_mesh(ptamd_scene* s, const ptamd_mesh_desc* mesh, const float pos[3], const float rot_deg[3],
                         const float scale[3]);
/* Mesh(path, name) (Mesh.cpp:7-13): loads a .glb/.gltf exactly as ptamd_model_load does. */
int ptamd_scene_add_mesh_file(ptamd_scene* s, const char* path, const float pos[3], const float rot_deg[3],
                              const float scale[3]);
/* Edits between frames (RayObject::SetPosition/SetRotation/SetMaterial, Sphere::SetRadius, Box::SetSize,
 * Mesh::SetScale; the ImGui editors of Sphere.cpp:72-88, Box.cpp:94-110, Mesh.cpp:259-287). Pass NULL to
 * leave a field unchanged. A commit is required before the next render. */
int ptamd_scene_set_transform(ptamd_scene* s, int object, const float pos[3], const float rot_deg[3],
                              const float size_or_scale[3]);
int ptamd_scene_set_radius(ptamd_scene* s, int object, float radius);
int ptamd_scene_set_material(ptamd_scene* s, int object, const ptamd_material* mat);
int ptamd_scene_set_mesh_group(ptamd_scene* s, int object, uint32_t group, const ptamd_pbr* pbr);
/* PathTracer::mBackgroundColour (PathTracer.h:21); default (0.2, 0.2, 0.2). */
int ptamd_scene_set_background(ptamd_scene* s, const float rgb[3]);
/* No reference counterpart (a miss returns the constant above, PathTracer.cpp:59-60): a cubemap whose
 * texel replaces that constant. faces: +x, -x, +y, -y, +z, -z, each face_size x face_size RGBA8, row 0 at
 * the top, OpenGL cubemap orientation; radiance = texel / 255 * intensity (nearest texel, no sRGB decode).
 * faces == NULL or face_size <= 0 removes it. Takes effect at once; no commit needed. Limit 4096 x 4096 per face (the map
 * is kept as float radiance on the device, 1.6 GB at that size). */
int ptamd_scene_set_skybox(ptamd_scene* s, int face_size, const uint8_t* const faces[6], float intensity);
/* The same with float radiance (an HDR map; no reference counterpart either): each face is face_size x face_size texels
 * of three floats (r, g, b), row 0 at the top; a miss returns the nearest texel as it is -- no intensity factor, no upper
 * bound (Film::AddSample still clips each sample's luminance at 12, Film.cpp:38-43). Replaces an RGBA8 map and vice
 * versa. faces == NULL or face_size <= 0 removes it. Same limit. */
int ptamd_scene_set_skybox_f32(ptamd_scene* s, int face_size, const float* const faces[6]);
/* Flattens the object list into SoA device buffers, builds and linearises the per-mesh BVHs
 * (replaces Mesh::BuildBVH, Mesh.cpp:289-397) and uploads. Re-callable after edits; trees are built once per
 * mesh, and edits that leave the geometry alone rewrite single table rows (see ptamd_scene_get_info). */
int ptamd_scene_commit(ptamd_scene* s);
/* No reference counterpart (diagnostics of the upload path). A commit after edits of transforms, radii, sizes, materials
 * or material-group factors rewrites only the rows concerned (objects / instances / materials / groups): geometry
 * buffers keep their addresses and geometry_uploads does not move. Adding or removing objects, or making a group start
 * or stop sampling images or masking, uploads everything again. */
typedef struct ptamd_scene_info {
    uint32_t objects, meshes, triangles, nodes, wide_nodes, lds_resident;
    uint64_t geometry_uploads;   /* full flatten + upload passes so far */
    uint64_t row_updates;        /* table rows rewritten by incremental commits so far */
    double last_commit_seconds;  /* host time of the last ptamd_scene_commit */
    double last_bvh_seconds;     /* of which BVH construction (0 when the trees were still valid) */
    const void* d_nodes;         /* device addresses of the geometry buffers (identity checks only) */
    const void* d_tris;
    const void* d_shade;
    uint32_t leading_boxes;      /* unrotated boxes at the head of the object list whose tests share one set of direction
                                    reciprocals (0 when fewer than two lead the list) */
    uint32_t walk_cut;           /* LDS-resident scenes whose last object is a mesh: the 64ths of a wave's walking lanes at which a walk
                                    is cut and resumed in the wave's next search. Scenes read from global memory whose only mesh is
                                    their last object: the lanes of a wave that wait at the mesh's bounds before the wave walks it
                                    for all of them. (0: neither; 0xffffffff: not measured yet, the first launch of 16 spp or more
                                    measures it.) The film does not depend on it */
} ptamd_scene_info;
int ptamd_scene_get_info(const ptamd_scene* s, ptamd_scene_info* out);

/* ---- model ingest: ModelLoader::LoadGLTF (ModelLoader.h:171-420) ---------------------------- */
int ptamd_model_load(const char* path, ptamd_model** out);
/* The same with options that go beyond the reference (flags = 0 is ptamd_model_load):
 * PTAMD_LOAD_NODE_TRANSFORMS applies the glTF scene graph to the vertices (the reference ignores nodes,
 * ModelLoader.h:204-206): each node of the default scene that refers to a mesh emits it under the product of the
 * node matrices from its root. PTAMD_LOAD_OBJ also accepts Wavefront .obj (+ .mtl) files, which the reference
 * rejects (ModelLoader.h:139-142). */
#define PTAMD_LOAD_NODE_TRANSFORMS 1u
#define PTAMD_LOAD_OBJ 2u
int ptamd_model_load_ex(const char* path, uint32_t flags, ptamd_model** out);
int ptamd_model_desc(const ptamd_model* m, ptamd_mesh_desc* out); /* pointers stay valid until ptamd_model_free */
const char* ptamd_model_group_name(const ptamd_model* m, uint32_t group);
void ptamd_model_free(ptamd_model* m);
/* One encoded image (PNG or JPEG) decoded as the loader decodes the images of a model: RGBA8, row 0 first -- what
 * tinygltf's LoadImageData gets from stb_image with four requested channels (contrib/include/tiny_gltf.h:2617-2722,
 * compiled by ModelLoader.cpp:1-7). *rgba (width * height * 4 bytes) belongs to the caller until ptamd_image_free.
 * Needs no GPU. */
int ptamd_image_decode(const uint8_t* bytes, size_t size, int* width, int* height, uint8_t** rgba);
void ptamd_image_free(uint8_t* rgba);

/* ---- film: Film.h:19-61 -------------------------------------------------------------------- */
int ptamd_film_create(int width, int height, ptamd_film** out);            /* Film(w,h) */
/* Film over caller-owned device buffers (accum: w*h*3 float, samples: w*h uint32), e.g. tensors that a
 * collective reduces in place. The caller zeroes and frees them. */
int ptamd_film_create_external(int width, int height, float* d_accum, uint32_t* d_samples, ptamd_film** out);
int ptamd_film_destroy(ptamd_film* f);
int ptamd_film_resize(ptamd_film* f, int width, int height);               /* Film::Resize (clears) */
int ptamd_film_reset(ptamd_film* f);                                       /* Film::Reset */
int ptamd_film_size(const ptamd_film* f, int* width, int* height);
int ptamd_film_read(ptamd_film* f, float* accum, uint32_t* samples);       /* Film::Accum / Film::Samples, to host; either may be NULL */
int ptamd_film_average(ptamd_film* f, float* avg);                         /* Film::AverageAt for every pixel, w*h*3 floats, to host */
/* Film::ResolveToRGBA8 (Film.cpp:55-104): tonemap 0 none / 1 Reinhard, colourspace 0 linear / 1 sRGB. w*h*4 bytes to host. */
int ptamd_film_resolve_rgba8(ptamd_film* f, int tonemap, int colourspace, uint8_t* rgba);
int ptamd_film_device_ptrs(ptamd_film* f, void** d_accum, void** d_samples);

/* ---- render: RayTraceParallel (main.cpp:30-47) ---------------------------------------------- */
/* Adds spp_count samples per pixel (sample indices spp_begin .. spp_begin+spp_count-1) for every pixel in
 * rect (NULL = whole film); equals spp_count successive RayTraceParallel passes.  depth and albedo_only are
 * TraceRay's arguments (PathTracer.h:11).  The random stream is keyed by (seed, pixel, sample index, bounce,
 * draw), so any split of the work over calls, rects or GPUs produces the same per-sample values.
 * How a pixel's samples are added: a call of fewer than 16 samples adds them one by one in the reference's order
 * (Film::AddSample, Film.cpp:34-46), so passes of that size give the same film bit for bit however they are grouped; a
 * call of 16 samples and more adds the exactly rounded sum of its samples in one float addition, so regrouping such
 * calls moves the film by float rounding (1e-7 relative).
 * Asynchronous on the library stream; film reads synchronise. */
int ptamd_render(ptamd_scene* s, const ptamd_camera* cam, ptamd_film* f, uint32_t spp_begin, uint32_t spp_count,
                 int depth, int albedo_only, uint32_t seed, const ptamd_rect* rect, uint32_t flags);
/* The same for one rank's share of an interleaved tile split, in ONE launch: the film is cut into square tiles of
 * tile_size pixels (a multiple of 8), numbered row-major, and the call renders the tiles tile_phase,
 * tile_phase + tile_stride, ... (rank and world size of a multi-GPU job).  It generalises the row-band task split of
 * RayTraceParallel (main.cpp:33-46: numTasks bands handed to a thread pool) to ranks; per-sample values are those of
 * ptamd_render, so the films of all phases sum to the single-launch film bit for bit. */
int ptamd_render_tiles(ptamd_scene* s, const ptamd_camera* cam, ptamd_film* f, uint32_t spp_begin, uint32_t spp_count,
                       int depth, int albedo_only, uint32_t seed, int tile_size, int tile_phase, int tile_stride, uint32_t flags);
int ptamd_render_stats(ptamd_stats* out); /* synchronises */
/* No counterpart in the reference. ptamd_render with PTAMD_FLAG_COUNT for callers that cannot allocate (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): the launch tallies into a statically allocated
 * device array of the library (the 10 counters of ptamd_stats, 80 bytes per GPU) and the totals are then copied, on the
 * library stream, to d_stats, a DEVICE pointer to one ptamd_stats that the caller owns. The call allocates and frees
 * nothing and does not synchronise; film and sample values are those of ptamd_render. All calls on one GPU share the
 * array (it is cleared at the start of each call): calls that overlap on one GPU race. */
int ptamd_render_counted_noalloc(ptamd_scene* s, const ptamd_camera* cam, ptamd_film* f, uint32_t spp_begin, uint32_t spp_count,
                                 int depth, int albedo_only, uint32_t seed, const ptamd_rect* rect, uint32_t flags, ptamd_stats* d_stats);

/* ---- probes used by the parity tests ------------------------------------------------------- */
/* Camera::GetRay (Camera.cpp:38-64) for every pixel: out[(y*w+x)*6] = origin xyz, direction xyz (host). */
int ptamd_camera_rays(const ptamd_camera* cam, int width, int height, float* out);
/* Closest-hit loop of PathTracer::TraceRay (PathTracer.cpp:38-57) for n host rays (6 floats each);
 * rec: n*20 floats {hit, t, p xyz, n xyz, frontFace, material 11 floats}. */
int ptamd_closest_hits(ptamd_scene* s, const float* rays, uint32_t n, float* rec);
/* The same, plus the work the GPU traversal did per ray: counts[2*i] = BVH nodes visited (each tests two child
 * boxes), counts[2*i+1] = ray-triangle tests. No reference counterpart; used to report executed next to
 * algorithmic work. */
int ptamd_traversal_counts(ptamd_scene* s, const float* rays, uint32_t n, float* rec, uint32_t* counts);

#ifdef __cplusplus
}
#endif
#endif
