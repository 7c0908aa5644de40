/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the path-tracing hot path.
 *
 * Plain-C restatement of the reference renderer's algorithm (jtothec67/Path-Tracer,
 * src/PathTracer/*).  Every function cites the reference file:line it follows.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product path (libptamd.so) never does.
 *
 * Parity is pinned: the deterministic parts (Camera::GetRay, the three
 * intersectors, albedo-only and depth-1 images, Film) are compared against the
 * reference's own sources compiled into oracle/_ref/ref_oracle, and against
 * golden buffers that binary produced (tests/golden/, made by
 * tests/golden/make_golden.py).  The one deliberate difference is the random
 * source: the reference draws from an unseedable thread_local std::mt19937
 * (PathTracer.cpp:8-16); this oracle draws from a counter-based PCG hash keyed
 * by (seed, pixel, sample, bounce, draw) so that a GPU render can be compared
 * sample-for-sample.  Stochastic agreement with the reference is therefore
 * statistical (RMSE vs spp, mean luminance), never bitwise.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* RayObject.h:10-20 */
typedef struct po_material {
    float albedo[3];
    float roughness;
    float metallic;
    float emission_colour[3];
    float emission_strength;
    float ior;
    float transmission;
} po_material;

/* ModelLoader.h:56-85 (occlusion fields dropped: never read on the hot path) */
typedef struct po_pbr {
    int32_t base_color_tex, normal_tex, metallic_roughness_tex, emissive_tex, transmission_tex;
    float base_color_factor[4];
    float metallic_factor, roughness_factor;
    int32_t alpha_mode; /* 0 opaque, 1 mask, 2 blend */
    float alpha_cutoff;
    float normal_scale;
    float emissive_factor[3];
    float transmission_factor;
    float ior;
} po_pbr;

/* ModelLoader.h:48-54 */
typedef struct po_image {
    int32_t width, height, channels;
    const uint8_t* data;
} po_image;

/* Camera.h:37-42 */
typedef struct po_camera {
    float position[3];
    float rotation_deg[3];
    float fov_deg;
    float near_plane, far_plane;
} po_camera;

typedef struct po_counters {
    uint64_t rays;          /* TraceRay calls that reach the closest-hit loop */
    uint64_t samples;       /* top-level TraceRay calls */
    uint64_t sphere_tests, box_tests, mesh_tests;
    uint64_t nodes_visited; /* RayAabb calls */
    uint64_t tris_tested;   /* RayTriMT calls */
    uint64_t shades;        /* hits that reach the BSDF stage */
} po_counters;

typedef struct po_scene po_scene;

po_scene* po_scene_create(void);
void po_scene_destroy(po_scene*);
void po_scene_set_background(po_scene*, const float rgb[3]);
int po_scene_add_sphere(po_scene*, const float pos[3], float radius, const po_material*);
int po_scene_add_box(po_scene*, const float pos[3], const float rot_deg[3], const float size[3], const po_material*);
/* positions/normals: num_faces*9 floats (a,b,c), texcoords: num_faces*6, group: num_faces ints (-1 = none).
 * normals/texcoords may be NULL (zeros, as ModelLoader::Vertex defaults). Arrays are copied. */
int po_scene_add_mesh(po_scene*, uint32_t num_faces, const float* positions, const float* normals,
                      const float* texcoords, const int32_t* group, uint32_t num_groups, const po_pbr* groups,
                      uint32_t num_images, const po_image* images,
                      const float pos[3], const float rot_deg[3], const float scale[3]);

/* Camera::GetRay -> out[0..2] origin, out[3..5] direction */
void po_camera_ray(const po_camera*, int x, int y, int w, int h, float out[6]);

/* Closest-hit loop of PathTracer::TraceRay (PathTracer.cpp:38-57). rec: 20 floats
 * {hit, t, p(3), n(3), frontFace, material(11)}; returns hit flag. */
int po_closest_hit(const po_scene*, const float origin[3], const float dir[3], float rec[20]);

/* One top-level PathTracer::TraceRay for pixel (x,y), sample index s. */
void po_trace_sample(const po_scene*, const po_camera*, int w, int h, int x, int y, uint32_t sample,
                     int depth, int albedo_only, uint32_t seed, float rgb[3]);

/* RayTraceParallel equivalent: adds spp_count samples (indices spp_begin..) to accum/samples for the
 * pixel rectangle [x0,x1) x [y0,y1) (pass 0,0,w,h for the full frame). accum: w*h*3, samples: w*h.
 * nthreads worker threads pull row bands from a shared counter. counters may be NULL. */
void po_render(const po_scene*, const po_camera*, int w, int h, float* accum, uint32_t* samples,
               uint32_t spp_begin, uint32_t spp_count, int depth, int albedo_only, uint32_t seed,
               int x0, int y0, int x1, int y1, int nthreads, po_counters* counters);

/* ---- extensions with NO counterpart in the reference (it has no skybox code and no Russian roulette; both are
 * off unless asked for). They exist so that the GPU implementation of the same extensions can be checked
 * sample for sample; their own correctness is tested analytically (uniform cubemap == constant background,
 * roulette leaves the expected image unchanged). */
#define PO_FLAG_RUSSIAN_ROULETTE 2u
/* Cubemap: 6 faces (+x, -x, +y, -y, +z, -z), each size x size RGBA8, row 0 = top; OpenGL face orientation,
 * nearest texel, texel / 255 * intensity replaces the constant miss radiance. faces == NULL disables it. */
void po_scene_set_skybox(po_scene*, int size, const uint8_t* const faces[6], float intensity);
void po_scene_set_skybox_f32(po_scene*, int size, const float* const faces[6]); /* float rgb texels, returned as they are */
/* po_render with flags (PO_FLAG_RUSSIAN_ROULETTE: from the 3rd bounce on a path survives with probability
 * q = clamp(max(throughput), 0.05, 1) drawn from slot 7 of the bounce's random numbers, and is divided by q). */
void po_render_ext(const po_scene*, const po_camera*, int w, int h, float* accum, uint32_t* samples,
                   uint32_t spp_begin, uint32_t spp_count, int depth, int albedo_only, uint32_t seed,
                   int x0, int y0, int x1, int y1, int nthreads, po_counters* counters, uint32_t flags);

/* Film::AddSample (Film.cpp:34-46) on one pixel. */
void po_film_add_sample(float accum_px[3], uint32_t* samples_px, const float rgb[3]);
/* Film::ResolveToRGBA8 (Film.cpp:55-104). tonemap: 0 none, 1 Reinhard; colourspace: 0 linear, 1 sRGB. */
void po_film_resolve(const float* accum, const uint32_t* samples, int n, int tonemap, int colourspace, uint8_t* rgba);

/* Random source shared with the GPU kernels. */
uint32_t po_pcg_hash(uint32_t v);
uint64_t po_path_key(uint32_t seed, uint32_t pixel, uint32_t sample); /* lower half | upper half << 32 */
float po_rand01(uint64_t key, uint32_t bounce, uint32_t draw);

/* BVH statistics of mesh object `obj` (node count, max depth) for tests. */
int po_mesh_bvh_stats(const po_scene*, int obj, uint32_t* nodes, uint32_t* depth);

#ifdef __cplusplus
}
#endif
#endif
