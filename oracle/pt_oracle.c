/* TEST INFRASTRUCTURE ONLY -- see pt_oracle.h.  Plain-C restatement of the reference hot path.
 * Citations are relative to /root/reference/src/PathTracer/ unless they name contrib/include/glm.
 * Built with -ffp-contract=off so that float expressions round exactly as the reference's
 * g++ -O2 x86-64 build does (no fused multiply-add). */
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ glm-equivalent helpers */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;
typedef struct { float m[4][4]; } m4; /* m[col][row], as glm */

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scl3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 div3s(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* glm compute_dot<vec3>: tmp = a*b; tmp.x + tmp.y + tmp.z */
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* glm detail/func_geometric.inl compute_cross */
static inline v3 cross3(v3 x, v3 y) { return V3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
/* glm normalize: x * inversesqrt(dot(x,x)), inversesqrt = 1/sqrt */
static inline v3 norm3(v3 a) { return scl3(a, 1.0f / sqrtf(dot3(a, a))); }
static inline float len3(v3 a) { return sqrtf(dot3(a, a)); }
static inline float fminr(float x, float y) { return (y < x) ? y : x; } /* glm::min / std::min */
static inline float fmaxr(float x, float y) { return (x < y) ? y : x; } /* glm::max / std::max */
static inline float clampr(float x, float lo, float hi) { return fminr(fmaxr(x, lo), hi); }
static inline v3 min3(v3 a, v3 b) { return V3(fminr(a.x, b.x), fminr(a.y, b.y), fminr(a.z, b.z)); }
static inline v3 max3(v3 a, v3 b) { return V3(fmaxr(a.x, b.x), fmaxr(a.y, b.y), fmaxr(a.z, b.z)); }
/* glm::reflect: I - N * dot(N, I) * 2 */
static inline v3 reflect3(v3 I, v3 N) { return sub3(I, scl3(scl3(N, dot3(N, I)), 2.0f)); }
/* glm::refract (detail/func_geometric.inl:129-138) */
static inline v3 refract3(v3 I, v3 N, float eta)
{
    float d = dot3(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k >= 0.0f) return sub3(scl3(I, eta), scl3(N, eta * d + sqrtf(k)));
    return V3(0.f, 0.f, 0.f);
}
static inline float radiansf(float deg) { return deg * 0.01745329251994329576923690768489f; }

static m4 m4_identity(void)
{
    m4 r; memset(&r, 0, sizeof r);
    r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0f;
    return r;
}
static inline v4 col(const m4* a, int c) { v4 r = { a->m[c][0], a->m[c][1], a->m[c][2], a->m[c][3] }; return r; }
static inline v4 add4(v4 a, v4 b) { v4 r = { a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; return r; }
static inline v4 scl4(v4 a, float s) { v4 r = { a.x * s, a.y * s, a.z * s, a.w * s }; return r; }
static inline void setcol(m4* a, int c, v4 v) { a->m[c][0] = v.x; a->m[c][1] = v.y; a->m[c][2] = v.z; a->m[c][3] = v.w; }

/* glm ext/matrix_transform.inl: translate */
static m4 m4_translate(m4 m, v3 v)
{
    m4 r = m;
    v4 c = add4(add4(add4(scl4(col(&m, 0), v.x), scl4(col(&m, 1), v.y)), scl4(col(&m, 2), v.z)), col(&m, 3));
    setcol(&r, 3, c);
    return r;
}
/* glm ext/matrix_transform.inl:18-46 rotate */
static m4 m4_rotate(m4 m, float angle, v3 v)
{
    float c = cosf(angle), s = sinf(angle);
    v3 axis = norm3(v);
    v3 temp = scl3(axis, 1.0f - c);
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;            R[0][1] = temp.x * axis.y + s * axis.z;   R[0][2] = temp.x * axis.z - s * axis.y;
    R[1][0] = temp.y * axis.x - s * axis.z;   R[1][1] = c + temp.y * axis.y;            R[1][2] = temp.y * axis.z + s * axis.x;
    R[2][0] = temp.z * axis.x + s * axis.y;   R[2][1] = temp.z * axis.y - s * axis.x;   R[2][2] = c + temp.z * axis.z;
    m4 r;
    for (int i = 0; i < 3; ++i)
        setcol(&r, i, add4(add4(scl4(col(&m, 0), R[i][0]), scl4(col(&m, 1), R[i][1])), scl4(col(&m, 2), R[i][2])));
    setcol(&r, 3, col(&m, 3));
    return r;
}
static m4 m4_scale(m4 m, v3 v)
{
    m4 r;
    setcol(&r, 0, scl4(col(&m, 0), v.x));
    setcol(&r, 1, scl4(col(&m, 1), v.y));
    setcol(&r, 2, scl4(col(&m, 2), v.z));
    setcol(&r, 3, col(&m, 3));
    return r;
}
/* glm detail/type_mat4x4.inl operator*(mat4, mat4) */
static m4 m4_mul(m4 a, m4 b)
{
    m4 r;
    for (int c = 0; c < 4; ++c) {
        v4 t = scl4(col(&a, 0), b.m[c][0]);
        t = add4(t, scl4(col(&a, 1), b.m[c][1]));
        t = add4(t, scl4(col(&a, 2), b.m[c][2]));
        t = add4(t, scl4(col(&a, 3), b.m[c][3]));
        setcol(&r, c, t);
    }
    return r;
}
/* glm detail/type_mat4x4.inl operator*(mat4, vec4): (m0*x + m1*y) + (m2*z + m3*w) */
static v4 m4_mulv(const m4* m, v4 v)
{
    v4 a0 = add4(scl4(col(m, 0), v.x), scl4(col(m, 1), v.y));
    v4 a1 = add4(scl4(col(m, 2), v.z), scl4(col(m, 3), v.w));
    return add4(a0, a1);
}
/* glm detail/func_matrix.inl:388-446 compute_inverse<4,4> */
static m4 m4_inverse(m4 mm)
{
    float (*m)[4] = mm.m;
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    float Fac0[4] = { Coef00, Coef00, Coef02, Coef03 }, Fac1[4] = { Coef04, Coef04, Coef06, Coef07 };
    float Fac2[4] = { Coef08, Coef08, Coef10, Coef11 }, Fac3[4] = { Coef12, Coef12, Coef14, Coef15 };
    float Fac4[4] = { Coef16, Coef16, Coef18, Coef19 }, Fac5[4] = { Coef20, Coef20, Coef22, Coef23 };
    float Vec0[4] = { m[1][0], m[0][0], m[0][0], m[0][0] }, Vec1[4] = { m[1][1], m[0][1], m[0][1], m[0][1] };
    float Vec2[4] = { m[1][2], m[0][2], m[0][2], m[0][2] }, Vec3[4] = { m[1][3], m[0][3], m[0][3], m[0][3] };
    const float SignA[4] = { +1, -1, +1, -1 }, SignB[4] = { -1, +1, -1, +1 };
    m4 inv;
    for (int i = 0; i < 4; ++i) {
        float Inv0 = Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i] + Vec3[i] * Fac2[i];
        float Inv1 = Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i] + Vec3[i] * Fac4[i];
        float Inv2 = Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i] + Vec3[i] * Fac5[i];
        float Inv3 = Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i] + Vec2[i] * Fac5[i];
        inv.m[0][i] = Inv0 * SignA[i];
        inv.m[1][i] = Inv1 * SignB[i];
        inv.m[2][i] = Inv2 * SignA[i];
        inv.m[3][i] = Inv3 * SignB[i];
    }
    float d0 = m[0][0] * inv.m[0][0], d1 = m[0][1] * inv.m[1][0], d2 = m[0][2] * inv.m[2][0], d3 = m[0][3] * inv.m[3][0];
    float Dot1 = (d0 + d1) + (d2 + d3);
    float ood = 1.0f / Dot1;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) inv.m[c][r] = inv.m[c][r] * ood;
    return inv;
}
/* glm ext/matrix_clip_space.inl:249-262 perspectiveRH_NO */
static m4 m4_perspective(float fovy, float aspect, float zn, float zf)
{
    float th = tanf(fovy / 2.0f);
    m4 r; memset(&r, 0, sizeof r);
    r.m[0][0] = 1.0f / (aspect * th);
    r.m[1][1] = 1.0f / th;
    r.m[2][2] = -(zf + zn) / (zf - zn);
    r.m[2][3] = -1.0f;
    r.m[3][2] = -(2.0f * zf * zn) / (zf - zn);
    return r;
}
/* R = I * Rx * Ry * Rz as in Camera.cpp:24-27, Box.cpp:16-19, Mesh.cpp:49-51 */
static m4 euler_xyz(m4 m, v3 rot_deg)
{
    m = m4_rotate(m, radiansf(rot_deg.x), V3(1, 0, 0));
    m = m4_rotate(m, radiansf(rot_deg.y), V3(0, 1, 0));
    m = m4_rotate(m, radiansf(rot_deg.z), V3(0, 0, 1));
    return m;
}

/* ------------------------------------------------------------------ random source */
uint32_t po_pcg_hash(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
/* 64-bit key of a path: the lower half is one hash chain over (seed, pixel, sample), the upper half a second, independent
 * one. Two paths draw the same numbers only if both halves agree. */
uint64_t po_path_key(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    uint32_t sh = po_pcg_hash(seed);
    uint32_t k = po_pcg_hash(sh ^ pixel);
    k = po_pcg_hash(k + sample * 0x9E3779B9u);
    uint32_t h = po_pcg_hash((sh ^ 0x68E31DA4u) + pixel * 0x9E3779B9u);
    h = po_pcg_hash(h ^ (sample * 0x85EBCA6Bu));
    return (uint64_t)k | ((uint64_t)h << 32);
}
/* Replaces Rand01 (PathTracer.cpp:13-16): uniform in [0,1) with 24 random bits. The draw counter, offset by the upper half
 * of the key, enters the lower half through an exclusive-or, not an addition: key + counter would make the streams of two
 * paths whose keys differ by less than 256 share draws in shifted positions. */
float po_rand01(uint64_t key, uint32_t bounce, uint32_t draw)
{
    uint32_t r = po_pcg_hash((uint32_t)key ^ ((bounce * 8u + draw + 1u) * 0x9E3779B9u + (uint32_t)(key >> 32)));
    return (float)(r >> 8) * 5.9604644775390625e-8f;
}
typedef struct { uint64_t key; uint32_t bounce, draw; uint32_t flags; v3 thr; } rng_t; /* flags/thr: extension state (Russian roulette) */
static inline float Rand01(rng_t* g) { return po_rand01(g->key, g->bounce, g->draw++); }

/* ------------------------------------------------------------------ scene types */
typedef struct { v3 o, d; float ior; } ray_t;                         /* Ray.h:5-13 */
typedef struct { float t; v3 p, n; po_material mat; int frontFace; } hit_t; /* RayObject.h:22-30 (mat by value) */

typedef struct { v3 bmin, bmax; uint32_t leftFirst, rightChild, count; } bvh_node; /* Mesh.h:42-49 */
typedef struct { v3 pos; float uv[2]; v3 nrm; } vertex_t;              /* ModelLoader.h:29-34 */
typedef struct { vertex_t a, b, c; int group; } face_t;                /* ModelLoader.h:36-40 */
typedef struct { int w, h, ch; uint8_t* data; } image_t;

typedef struct {
    uint32_t nfaces; face_t* faces;
    uint32_t ngroups; po_pbr* groups;
    uint32_t nimages; image_t* images;
    bvh_node* nodes; uint32_t nnodes, cap_nodes, depth;
    uint32_t* faceIdx;
    v3 *fbmin, *fbmax, *fcent;
    v3 pos, rot, scale;
    m4 M, Minv; /* Mesh.cpp:47-54, computed once (same inputs every call) */
} mesh_t;

enum { OBJ_SPHERE = 0, OBJ_BOX = 1, OBJ_MESH = 2 };
typedef struct {
    int kind;
    v3 pos, rot, size; /* size: box extents; sphere: size.x = radius */
    po_material mat;
    mesh_t* mesh;
    float W[3][3], L[3][3]; /* box: worldFromLocal and its transpose (Box.cpp:16-23), computed once: the
                               reference rebuilds them on every call from the same inputs */
} object_t;

struct po_scene {
    object_t* objs; int nobjs, cap;
    v3 background; /* PathTracer.h:21 */
    int sky_size; uint8_t* sky; float sky_intensity; /* extension: cubemap, 6 * size * size * 4 bytes */
    float* sky_f32;                                  /* or float radiance, 6 * size * size * 3 floats */
};

typedef struct { po_counters c; } tls_counters;
static __thread po_counters* g_cnt = NULL;
#define COUNT(field) do { if (g_cnt) g_cnt->field++; } while (0)

po_scene* po_scene_create(void)
{
    po_scene* s = (po_scene*)calloc(1, sizeof *s);
    s->background = V3(0.2f, 0.2f, 0.2f);
    return s;
}
static void mesh_free(mesh_t* m)
{
    if (!m) return;
    for (uint32_t i = 0; i < m->nimages; ++i) free(m->images[i].data);
    free(m->faces); free(m->groups); free(m->images); free(m->nodes); free(m->faceIdx);
    free(m->fbmin); free(m->fbmax); free(m->fcent); free(m);
}
void po_scene_destroy(po_scene* s)
{
    if (!s) return;
    for (int i = 0; i < s->nobjs; ++i) mesh_free(s->objs[i].mesh);
    free(s->objs); free(s->sky); free(s->sky_f32); free(s);
}
void po_scene_set_skybox(po_scene* s, int size, const uint8_t* const faces[6], float intensity)
{
    free(s->sky); s->sky = NULL; s->sky_size = 0;
    free(s->sky_f32); s->sky_f32 = NULL;
    if (!faces || size <= 0) return;
    size_t fb = (size_t)size * size * 4;
    s->sky = (uint8_t*)malloc(6 * fb);
    for (int f = 0; f < 6; ++f) memcpy(s->sky + f * fb, faces[f], fb);
    s->sky_size = size; s->sky_intensity = intensity;
}
void po_scene_set_skybox_f32(po_scene* s, int size, const float* const faces[6])
{
    free(s->sky); s->sky = NULL; s->sky_size = 0;
    free(s->sky_f32); s->sky_f32 = NULL;
    if (!faces || size <= 0) return;
    size_t fn = (size_t)size * size * 3;
    s->sky_f32 = (float*)malloc(6 * fn * sizeof(float));
    for (int f = 0; f < 6; ++f) memcpy(s->sky_f32 + f * fn, faces[f], fn * sizeof(float));
    s->sky_size = size;
}
/* Extension (no reference counterpart): miss radiance from a cubemap, OpenGL face selection and orientation. */
static v3 sky_lookup(const po_scene* s, v3 d)
{
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int face; float sc, tc, ma;
    if (ax >= ay && ax >= az) { ma = ax; if (d.x > 0) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; } }
    else if (ay >= az) { ma = ay; if (d.y > 0) { face = 2; sc = d.x; tc = d.z; } else { face = 3; sc = d.x; tc = -d.z; } }
    else { ma = az; if (d.z > 0) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; } }
    float u = (sc / ma + 1.0f) * 0.5f, v = (tc / ma + 1.0f) * 0.5f;
    int n = s->sky_size;
    int x = (int)(u * (float)n), y = (int)(v * (float)n);
    x = x < 0 ? 0 : (x > n - 1 ? n - 1 : x); y = y < 0 ? 0 : (y > n - 1 ? n - 1 : y);
    if (s->sky_f32) { const float* r = s->sky_f32 + (((size_t)face * n + y) * n + x) * 3; return V3(r[0], r[1], r[2]); }
    const uint8_t* t = s->sky + (((size_t)face * n + y) * n + x) * 4;
    return scl3(V3((float)t[0] / 255.0f, (float)t[1] / 255.0f, (float)t[2] / 255.0f), s->sky_intensity);
}
void po_scene_set_background(po_scene* s, const float rgb[3]) { s->background = V3(rgb[0], rgb[1], rgb[2]); }
static object_t* push_obj(po_scene* s)
{
    if (s->nobjs == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 8;
        s->objs = (object_t*)realloc(s->objs, (size_t)s->cap * sizeof(object_t));
    }
    object_t* o = &s->objs[s->nobjs++];
    memset(o, 0, sizeof *o);
    return o;
}
int po_scene_add_sphere(po_scene* s, const float pos[3], float radius, const po_material* m)
{
    object_t* o = push_obj(s);
    o->kind = OBJ_SPHERE; o->pos = V3(pos[0], pos[1], pos[2]); o->size = V3(radius, 0, 0); o->mat = *m;
    return s->nobjs - 1;
}
int po_scene_add_box(po_scene* s, const float pos[3], const float rot[3], const float size[3], const po_material* m)
{
    object_t* o = push_obj(s);
    o->kind = OBJ_BOX; o->pos = V3(pos[0], pos[1], pos[2]); o->rot = V3(rot[0], rot[1], rot[2]);
    o->size = V3(size[0], size[1], size[2]); o->mat = *m;
    m4 R = euler_xyz(m4_identity(), o->rot);
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) { o->W[c][r] = R.m[c][r]; o->L[r][c] = R.m[c][r]; }
    return s->nobjs - 1;
}

/* ------------------------------------------------------------------ BVH build (Mesh.cpp:289-397) */
/* Mesh::RangeBounds, Mesh.cpp:379-397 */
static void range_bounds(const mesh_t* m, uint32_t start, uint32_t count, v3* omin, v3* omax)
{
    v3 bmin = m->fbmin[m->faceIdx[start]], bmax = m->fbmax[m->faceIdx[start]];
    for (uint32_t i = start + 1; i < start + count; ++i) {
        uint32_t fi = m->faceIdx[i];
        bmin = min3(bmin, m->fbmin[fi]);
        bmax = max3(bmax, m->fbmax[fi]);
    }
    *omin = bmin; *omax = bmax;
}
static inline float axis_of(v3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
/* std::nth_element stand-in: after the call idx[lo..nth) <= idx[nth] <= idx(nth..hi) by centroid[axis].
 * (Which of several equal-keyed faces lands on which side may differ from libstdc++; the closest hit
 * does not depend on it.) */
static void nth_element_idx(uint32_t* idx, const v3* cent, int axis, long lo, long nth, long hi)
{
    long l = lo, r = hi - 1;
    while (l < r) {
        float a = axis_of(cent[idx[l]], axis), b = axis_of(cent[idx[(l + r) / 2]], axis), c = axis_of(cent[idx[r]], axis);
        float pivot = fmaxr(fminr(a, b), fminr(fmaxr(a, b), c)); /* median of three */
        long i = l, lt = l, gt = r; /* three-way partition: [l,lt) < pivot, [lt,i) == pivot, (gt,r] > pivot */
        while (i <= gt) {
            float k = axis_of(cent[idx[i]], axis);
            if (k < pivot) { uint32_t t = idx[lt]; idx[lt] = idx[i]; idx[i] = t; ++lt; ++i; }
            else if (k > pivot) { uint32_t t = idx[gt]; idx[gt] = idx[i]; idx[i] = t; --gt; }
            else ++i;
        }
        if (nth < lt) r = lt - 1;
        else if (nth > gt) l = gt + 1;
        else return;
    }
}
/* Mesh::BuildNode, Mesh.cpp:328-377 */
static uint32_t build_node(mesh_t* m, uint32_t start, uint32_t count, uint32_t depth)
{
    const uint32_t leafThreshold = 2; /* Mesh.h:59 */
    uint32_t ni = m->nnodes++;
    if (depth > m->depth) m->depth = depth;
    v3 bmin, bmax;
    range_bounds(m, start, count, &bmin, &bmax);
    m->nodes[ni].bmin = bmin; m->nodes[ni].bmax = bmax;
    if (count <= leafThreshold) {
        m->nodes[ni].leftFirst = start; m->nodes[ni].count = count; m->nodes[ni].rightChild = 0;
        return ni;
    }
    v3 e = sub3(bmax, bmin);
    int axis = 0;
    if (e.y > e.x && e.y >= e.z) axis = 1;
    else if (e.z > e.x && e.z >= e.y) axis = 2;
    uint32_t mid = start + count / 2;
    nth_element_idx(m->faceIdx, m->fcent, axis, (long)start, (long)mid, (long)start + count);
    uint32_t leftCount = mid - start, rightCount = count - leftCount;
    if (leftCount == 0 || rightCount == 0) { leftCount = count / 2; rightCount = count - leftCount; }
    m->nodes[ni].count = 0;
    uint32_t l = build_node(m, start, leftCount, depth + 1);
    uint32_t r = build_node(m, start + leftCount, rightCount, depth + 1);
    m->nodes[ni].leftFirst = l; m->nodes[ni].rightChild = r;
    return ni;
}
/* Mesh::BuildBVH, Mesh.cpp:289-326 */
static void build_bvh(mesh_t* m)
{
    uint32_t N = m->nfaces;
    m->faceIdx = (uint32_t*)malloc((size_t)N * 4);
    m->fbmin = (v3*)malloc((size_t)N * sizeof(v3));
    m->fbmax = (v3*)malloc((size_t)N * sizeof(v3));
    m->fcent = (v3*)malloc((size_t)N * sizeof(v3));
    for (uint32_t i = 0; i < N; ++i) {
        m->faceIdx[i] = i;
        v3 p0 = m->faces[i].a.pos, p1 = m->faces[i].b.pos, p2 = m->faces[i].c.pos;
        m->fbmin[i] = min3(p0, min3(p1, p2));
        m->fbmax[i] = max3(p0, max3(p1, p2));
        m->fcent[i] = div3s(add3(add3(p0, p1), p2), 3.0f);
    }
    m->cap_nodes = 2 * N;
    m->nodes = (bvh_node*)malloc((size_t)m->cap_nodes * sizeof(bvh_node));
    m->nnodes = 0; m->depth = 0;
    build_node(m, 0, N, 1);
}

int po_scene_add_mesh(po_scene* s, uint32_t nf, const float* P, const float* Nn, const float* T, const int32_t* grp,
                      uint32_t ngroups, const po_pbr* groups, uint32_t nimages, const po_image* images,
                      const float pos[3], const float rot[3], const float scale[3])
{
    if (nf == 0) return -1; /* Mesh.cpp:293 throws */
    mesh_t* m = (mesh_t*)calloc(1, sizeof *m);
    m->nfaces = nf;
    m->faces = (face_t*)calloc(nf, sizeof(face_t));
    for (uint32_t i = 0; i < nf; ++i) {
        vertex_t* vs[3] = { &m->faces[i].a, &m->faces[i].b, &m->faces[i].c };
        for (int k = 0; k < 3; ++k) {
            vs[k]->pos = V3(P[i * 9 + k * 3], P[i * 9 + k * 3 + 1], P[i * 9 + k * 3 + 2]);
            if (Nn) vs[k]->nrm = V3(Nn[i * 9 + k * 3], Nn[i * 9 + k * 3 + 1], Nn[i * 9 + k * 3 + 2]);
            if (T) { vs[k]->uv[0] = T[i * 6 + k * 2]; vs[k]->uv[1] = T[i * 6 + k * 2 + 1]; }
        }
        m->faces[i].group = grp ? grp[i] : -1;
    }
    m->ngroups = ngroups;
    if (ngroups) { m->groups = (po_pbr*)malloc(ngroups * sizeof(po_pbr)); memcpy(m->groups, groups, ngroups * sizeof(po_pbr)); }
    m->nimages = nimages;
    if (nimages) {
        m->images = (image_t*)calloc(nimages, sizeof(image_t));
        for (uint32_t i = 0; i < nimages; ++i) {
            m->images[i].w = images[i].width; m->images[i].h = images[i].height; m->images[i].ch = images[i].channels;
            size_t bytes = (size_t)images[i].width * images[i].height * images[i].channels;
            if (images[i].data && bytes) { m->images[i].data = (uint8_t*)malloc(bytes); memcpy(m->images[i].data, images[i].data, bytes); }
        }
    }
    m->pos = V3(pos[0], pos[1], pos[2]); m->rot = V3(rot[0], rot[1], rot[2]); m->scale = V3(scale[0], scale[1], scale[2]);
    m->M = m4_scale(euler_xyz(m4_translate(m4_identity(), m->pos), m->rot), m->scale);
    m->Minv = m4_inverse(m->M);
    build_bvh(m);
    object_t* o = push_obj(s);
    o->kind = OBJ_MESH; o->mesh = m;
    return s->nobjs - 1;
}
int po_mesh_bvh_stats(const po_scene* s, int obj, uint32_t* nodes, uint32_t* depth)
{
    if (obj < 0 || obj >= s->nobjs || s->objs[obj].kind != OBJ_MESH) return -1;
    *nodes = s->objs[obj].mesh->nnodes; *depth = s->objs[obj].mesh->depth;
    return 0;
}

/* ------------------------------------------------------------------ camera (Camera.cpp:22-64) */
typedef struct { m4 invView, invProj; } cam_state;
/* CalculateMatrices, Camera.cpp:22-35 */
static cam_state camera_prepare(const po_camera* cam, int W, int H)
{
    cam_state cs;
    m4 R = euler_xyz(m4_identity(), V3(cam->rotation_deg[0], cam->rotation_deg[1], cam->rotation_deg[2]));
    m4 worldFromCam = m4_mul(m4_translate(m4_identity(), V3(cam->position[0], cam->position[1], cam->position[2])), R);
    cs.invView = worldFromCam;
    m4 proj = m4_perspective(radiansf(cam->fov_deg), (float)W / (float)H, cam->near_plane, cam->far_plane);
    cs.invProj = m4_inverse(proj);
    return cs;
}
/* GetRay, Camera.cpp:46-63 */
static void camera_ray(const cam_state* cs, int px, int py, int W, int H, float out[6])
{
    const m4 invView = cs->invView, invProj = cs->invProj;
    float nx = (((float)px + 0.5f) / (float)W) * 2.f - 1.f;
    float ny = (((float)py + 0.5f) / (float)H) * 2.f - 1.f;
    v4 clipNear = { nx, ny, -1.f, 1.f }, clipFar = { nx, ny, 1.f, 1.f };
    v4 camNear = m4_mulv(&invProj, clipNear);
    camNear.x /= camNear.w; camNear.y /= camNear.w; camNear.z /= camNear.w; camNear.w /= camNear.w;
    v4 camFar = m4_mulv(&invProj, clipFar);
    camFar.x /= camFar.w; camFar.y /= camFar.w; camFar.z /= camFar.w; camFar.w /= camFar.w;
    v4 zero1 = { 0, 0, 0, 1 };
    v4 ow = m4_mulv(&invView, zero1);
    v3 dirCam = norm3(V3(camFar.x - camNear.x, camFar.y - camNear.y, camFar.z - camNear.z));
    /* mat3(mInvView) * dirCam */
    v3 dw = V3(invView.m[0][0] * dirCam.x + invView.m[1][0] * dirCam.y + invView.m[2][0] * dirCam.z,
               invView.m[0][1] * dirCam.x + invView.m[1][1] * dirCam.y + invView.m[2][1] * dirCam.z,
               invView.m[0][2] * dirCam.x + invView.m[1][2] * dirCam.y + invView.m[2][2] * dirCam.z);
    dw = norm3(dw);
    /* Ray(origin, direction) normalises once more (Ray.h:8) */
    dw = norm3(dw);
    out[0] = ow.x; out[1] = ow.y; out[2] = ow.z; out[3] = dw.x; out[4] = dw.y; out[5] = dw.z;
}
void po_camera_ray(const po_camera* cam, int px, int py, int W, int H, float out[6])
{
    cam_state cs = camera_prepare(cam, W, H);
    camera_ray(&cs, px, py, W, H, out);
}

/* ------------------------------------------------------------------ Sphere (Sphere.cpp:32-70) */
static int sphere_intersect(const object_t* o, const ray_t* ray, float tMin, float tMax, hit_t* out)
{
    COUNT(sphere_tests);
    float radius = o->size.x;
    v3 oc = sub3(ray->o, o->pos);
    float a = dot3(ray->d, ray->d);
    float h = dot3(oc, ray->d);
    float c = dot3(oc, oc) - radius * radius;
    float disc = h * h - a * c;
    if (disc < 0.0f) return 0;
    float sq = sqrtf(disc);
    float t = (-h - sq) / a;
    if (t < tMin || t > tMax) {
        t = (-h + sq) / a;
        if (t < tMin || t > tMax) return 0;
    }
    out->t = t;
    out->p = add3(ray->o, scl3(ray->d, t));
    v3 outward = div3s(sub3(out->p, o->pos), radius);
    int ff = dot3(ray->d, outward) < 0.0f;
    out->frontFace = ff;
    out->n = ff ? outward : neg3(outward);
    out->mat = o->mat;
    return 1;
}

/* ------------------------------------------------------------------ Box (Box.cpp:13-92) */
static inline v3 m3_mulv(const float (*M)[3], v3 v) /* M[col][row] */
{
    return V3(M[0][0] * v.x + M[1][0] * v.y + M[2][0] * v.z,
              M[0][1] * v.x + M[1][1] * v.y + M[2][1] * v.z,
              M[0][2] * v.x + M[1][2] * v.y + M[2][2] * v.z);
}
static int box_intersect(const object_t* o, const ray_t* ray, float tMin, float tMax, hit_t* out)
{
    COUNT(box_tests);
    const float (*W)[3] = o->W; const float (*L)[3] = o->L; /* worldFromLocal, localFromWorld */
    v3 ro = m3_mulv(L, sub3(ray->o, o->pos));
    v3 rd = m3_mulv(L, ray->d);
    v3 he = scl3(o->size, 0.5f);
    const float BIG = 1e30f;
    v3 invD = V3(rd.x != 0.0f ? 1.0f / rd.x : (rd.x > 0.0f ? BIG : -BIG),
                 rd.y != 0.0f ? 1.0f / rd.y : (rd.y > 0.0f ? BIG : -BIG),
                 rd.z != 0.0f ? 1.0f / rd.z : (rd.z > 0.0f ? BIG : -BIG));
    v3 t1 = mul3(sub3(neg3(he), ro), invD);
    v3 t2 = mul3(sub3(he, ro), invD);
    v3 tMin3 = min3(t1, t2), tMax3 = max3(t1, t2);
    float tEntry = fmaxr(fmaxr(tMin3.x, tMin3.y), tMin3.z);
    float tExit = fminr(fminr(tMax3.x, tMax3.y), tMax3.z);
    if (tExit < tEntry) return 0;
    if (tExit < tMin) return 0;
    float tHit = tEntry;
    if (tHit < tMin) tHit = tExit;
    if (tHit < tMin || tHit > tMax) return 0;
    v3 pL = add3(ro, scl3(rd, tHit));
    v3 pW = add3(ray->o, scl3(ray->d, tHit));
    v3 nL = V3(0, 0, 0);
    float eps = 1e-4f * fmaxr(fmaxr(he.x, he.y), he.z);
    if (fabsf(pL.x - he.x) <= eps) nL = V3(1, 0, 0);
    else if (fabsf(pL.x + he.x) <= eps) nL = V3(-1, 0, 0);
    else if (fabsf(pL.y - he.y) <= eps) nL = V3(0, 1, 0);
    else if (fabsf(pL.y + he.y) <= eps) nL = V3(0, -1, 0);
    else if (fabsf(pL.z - he.z) <= eps) nL = V3(0, 0, 1);
    else if (fabsf(pL.z + he.z) <= eps) nL = V3(0, 0, -1);
    else {
        if (tEntry == tMin3.x) nL = V3((ro.x + tEntry * rd.x) > 0.0f ? 1.f : -1.f, 0, 0);
        else if (tEntry == tMin3.y) nL = V3(0, (ro.y + tEntry * rd.y) > 0.0f ? 1.f : -1.f, 0);
        else nL = V3(0, 0, (ro.z + tEntry * rd.z) > 0.0f ? 1.f : -1.f);
    }
    v3 nW = norm3(m3_mulv(W, nL));
    int ff = dot3(ray->d, nW) < 0.0f;
    out->t = tHit; out->p = pW; out->n = ff ? nW : neg3(nW);
    out->mat = o->mat;
    /* Box.cpp:86-89 never writes frontFace: it keeps the caller's zero-initialised value. */
    return 1;
}

/* ------------------------------------------------------------------ Mesh (Mesh.cpp:15-257, 401-486) */
/* SampleImageNearest, Mesh.cpp:15-37 */
static void sample_nearest(const image_t* img, float u, float v, float out[4])
{
    if (img->w <= 0 || img->h <= 0 || img->ch <= 0 || !img->data) { out[0] = out[1] = out[2] = out[3] = 1.f; return; }
    u = u - floorf(u); v = v - floorf(v); /* glm::fract */
    if (u < 0) u += 1.0f;
    if (v < 0) v += 1.0f;
    int x = (int)(u * (float)img->w), y = (int)(v * (float)img->h);
    int ix = x < 0 ? 0 : (x > img->w - 1 ? img->w - 1 : x);
    int iy = y < 0 ? 0 : (y > img->h - 1 ? img->h - 1 : y);
    size_t idx = ((size_t)iy * (size_t)img->w + (size_t)ix) * (size_t)img->ch;
    for (int c = 0; c < 4; ++c)
        out[c] = (c < img->ch) ? ((float)img->data[idx + (size_t)c] / 255.0f) : (c == 3 ? 1.0f : 0.0f);
}
/* Mesh::RayAabb, Mesh.cpp:401-416 */
static int ray_aabb(const ray_t* r, v3 bmin, v3 bmax, float tMax, float* t0, float* t1)
{
    COUNT(nodes_visited);
    v3 invD = V3(1.0f / r->d.x, 1.0f / r->d.y, 1.0f / r->d.z);
    v3 t0s = mul3(sub3(bmin, r->o), invD);
    v3 t1s = mul3(sub3(bmax, r->o), invD);
    v3 ts = min3(t0s, t1s), tb = max3(t0s, t1s);
    *t0 = fmaxr(fmaxr(ts.x, ts.y), fmaxr(ts.z, 0.0f));
    *t1 = fminr(fminr(tb.x, tb.y), fminr(tb.z, tMax));
    return *t1 >= *t0;
}
/* Mesh::RayTriMT, Mesh.cpp:418-445 */
static int ray_tri(const ray_t* r, const face_t* f, float* t, float* u, float* v)
{
    COUNT(tris_tested);
    v3 v0 = f->a.pos, e1 = sub3(f->b.pos, v0), e2 = sub3(f->c.pos, v0);
    v3 p = cross3(r->d, e2);
    float det = dot3(e1, p);
    const float eps = 1e-8f;
    if (fabsf(det) < eps) return 0;
    float invDet = 1.0f / det;
    v3 tvec = sub3(r->o, v0);
    *u = dot3(tvec, p) * invDet;
    if (*u < 0.0f || *u > 1.0f) return 0;
    v3 q = cross3(tvec, e1);
    *v = dot3(r->d, q) * invDet;
    if (*v < 0.0f || *u + *v > 1.0f) return 0;
    *t = dot3(e2, q) * invDet;
    return *t > eps;
}
/* Mesh::FillMaterialAt, Mesh.cpp:447-486 */
static void fill_material(const mesh_t* m, int group, float u, float v, po_material* out)
{
    const po_pbr* g = &m->groups[group];
    float base[4] = { g->base_color_factor[0], g->base_color_factor[1], g->base_color_factor[2], g->base_color_factor[3] };
    float tx[4];
    if (g->base_color_tex >= 0) { sample_nearest(&m->images[g->base_color_tex], u, v, tx); for (int i = 0; i < 4; ++i) base[i] *= tx[i]; }
    out->albedo[0] = base[0]; out->albedo[1] = base[1]; out->albedo[2] = base[2];
    float rough = g->roughness_factor, metal = g->metallic_factor;
    if (g->metallic_roughness_tex >= 0) {
        sample_nearest(&m->images[g->metallic_roughness_tex], u, v, tx);
        rough = clampr(tx[1] * rough, 0.001f, 1.0f);
        metal = clampr(tx[2] * metal, 0.0f, 1.0f);
    }
    out->roughness = rough; out->metallic = metal;
    v3 em = V3(g->emissive_factor[0], g->emissive_factor[1], g->emissive_factor[2]);
    if (g->emissive_tex >= 0) { sample_nearest(&m->images[g->emissive_tex], u, v, tx); em = mul3(em, V3(tx[0], tx[1], tx[2])); }
    out->emission_colour[0] = em.x; out->emission_colour[1] = em.y; out->emission_colour[2] = em.z;
    out->emission_strength = len3(em);
    float tr = g->transmission_factor;
    if (g->transmission_tex >= 0) { sample_nearest(&m->images[g->transmission_tex], u, v, tx); tr *= tx[0]; }
    out->transmission = clampr(tr, 0.0f, 1.0f);
    out->ior = g->ior;
}
/* Mesh::RayIntersect, Mesh.cpp:39-257 */
static int mesh_intersect(const object_t* o, const ray_t* ray, float tMin, float tMax, hit_t* out)
{
    COUNT(mesh_tests);
    const mesh_t* m = o->mesh;
    const m4 M = m->M, Minv = m->Minv;
    v4 o4 = { ray->o.x, ray->o.y, ray->o.z, 1.0f }, d4 = { ray->d.x, ray->d.y, ray->d.z, 0.0f };
    v4 oo = m4_mulv(&Minv, o4), dd = m4_mulv(&Minv, d4);
    ray_t rObj; rObj.o = V3(oo.x, oo.y, oo.z); rObj.d = V3(dd.x, dd.y, dd.z); rObj.ior = 1.0f;
    float dirLen = len3(rObj.d);
    if (dirLen == 0.0f) return 0;
    rObj.d = div3s(rObj.d, dirLen);
    if (m->nnodes == 0) return 0;
    const float tMinObj = tMin * dirLen, tMaxObj = tMax * dirLen;
    float closestT = tMaxObj;
    int bestFace = -1; float bestU = 0.f, bestV = 0.f;
    uint32_t stack[64]; int sp = 0;
    stack[sp++] = 0u;
    while (sp) {
        const bvh_node* node = &m->nodes[stack[--sp]];
        float t0, t1;
        if (!ray_aabb(&rObj, node->bmin, node->bmax, closestT, &t0, &t1)) continue;
        if (node->count > 0) {
            uint32_t start = node->leftFirst, end = start + node->count;
            for (uint32_t i = start; i < end; ++i) {
                uint32_t fi = m->faceIdx[i];
                const face_t* f = &m->faces[fi];
                float t, u, v;
                if (!ray_tri(&rObj, f, &t, &u, &v)) continue;
                if (t < tMinObj || t >= closestT) continue;
                if (f->group >= 0) { /* alpha MASK cutout, Mesh.cpp:106-125 */
                    const po_pbr* pbr = &m->groups[f->group];
                    if (pbr->alpha_mode == 1) {
                        float w = 1.0f - u - v;
                        float uu = w * f->a.uv[0] + u * f->b.uv[0] + v * f->c.uv[0];
                        float vv = w * f->a.uv[1] + u * f->b.uv[1] + v * f->c.uv[1];
                        float alpha = pbr->base_color_factor[3];
                        if (pbr->base_color_tex >= 0) { float tx[4]; sample_nearest(&m->images[pbr->base_color_tex], uu, vv, tx); alpha *= tx[3]; }
                        if (alpha < pbr->alpha_cutoff) continue;
                    }
                }
                closestT = t; bestFace = (int)fi; bestU = u; bestV = v;
            }
        } else {
            uint32_t left = node->leftFirst, right = node->rightChild;
            float lt0, lt1, rt0, rt1;
            int hitL = ray_aabb(&rObj, m->nodes[left].bmin, m->nodes[left].bmax, closestT, &lt0, &lt1);
            int hitR = ray_aabb(&rObj, m->nodes[right].bmin, m->nodes[right].bmax, closestT, &rt0, &rt1);
            if (hitL && hitR) {
                if (lt0 < rt0) { stack[sp++] = right; stack[sp++] = left; }
                else { stack[sp++] = left; stack[sp++] = right; }
            } else if (hitL) stack[sp++] = left;
            else if (hitR) stack[sp++] = right;
        }
    }
    if (bestFace < 0) return 0;
    const face_t* f = &m->faces[bestFace];
    const float u = bestU, v = bestV, w = 1.0f - u - v;
    v3 pObj = add3(add3(scl3(f->a.pos, w), scl3(f->b.pos, u)), scl3(f->c.pos, v));
    v3 nObjS = norm3(add3(add3(scl3(f->a.nrm, w), scl3(f->b.nrm, u)), scl3(f->c.nrm, v)));
    v3 e1 = sub3(f->b.pos, f->a.pos), e2 = sub3(f->c.pos, f->a.pos);
    v3 nObjG = norm3(cross3(e1, e2));
    float uvx = w * f->a.uv[0] + u * f->b.uv[0] + v * f->c.uv[0];
    float uvy = w * f->a.uv[1] + u * f->b.uv[1] + v * f->c.uv[1];
    v4 p4 = { pObj.x, pObj.y, pObj.z, 1.0f };
    v4 pw4 = m4_mulv(&M, p4);
    v3 pW = V3(pw4.x, pw4.y, pw4.z);
    v3 nObj = norm3(nObjS);
    if (f->group >= 0) {
        const po_pbr* pbr = &m->groups[f->group];
        if (pbr->normal_tex >= 0) { /* Mesh.cpp:174-226 */
            v3 dp1 = sub3(f->b.pos, f->a.pos), dp2 = sub3(f->c.pos, f->a.pos);
            float du1 = f->b.uv[0] - f->a.uv[0], dv1 = f->b.uv[1] - f->a.uv[1];
            float du2 = f->c.uv[0] - f->a.uv[0], dv2 = f->c.uv[1] - f->a.uv[1];
            float det = du1 * dv2 - dv1 * du2;
            v3 T, B;
            if (fabsf(det) > 1e-8f) {
                float r = 1.0f / det;
                v3 t = scl3(sub3(scl3(dp1, dv2), scl3(dp2, dv1)), r);
                t = norm3(sub3(t, scl3(nObj, dot3(nObj, t))));
                T = t;
                B = norm3(cross3(nObj, T));
            } else {
                v3 up = (fabsf(nObj.z) < 0.999f) ? V3(0, 0, 1) : V3(1, 0, 0);
                T = norm3(cross3(up, nObj));
                B = cross3(nObj, T);
            }
            float wu = uvx - floorf(uvx), wv = uvy - floorf(uvy);
            float tx[4];
            sample_nearest(&m->images[pbr->normal_tex], wu, wv, tx);
            v3 nts = V3(tx[0] * 2.0f - 1.0f, tx[1] * 2.0f - 1.0f, tx[2] * 2.0f - 1.0f);
            nts.x *= pbr->normal_scale; nts.y *= pbr->normal_scale;
            nts = norm3(nts);
            /* mat3(T,B,N) * n_ts */
            v3 nn = V3(T.x * nts.x + B.x * nts.y + nObj.x * nts.z,
                       T.y * nts.x + B.y * nts.y + nObj.y * nts.z,
                       T.z * nts.x + B.z * nts.y + nObj.z * nts.z);
            nObj = norm3(nn);
        }
    }
    /* MinvT = transpose(mat3(Minv)); MinvT * n -> rows of mat3(Minv) */
    v3 nW = norm3(V3(Minv.m[0][0] * nObj.x + Minv.m[0][1] * nObj.y + Minv.m[0][2] * nObj.z,
                     Minv.m[1][0] * nObj.x + Minv.m[1][1] * nObj.y + Minv.m[1][2] * nObj.z,
                     Minv.m[2][0] * nObj.x + Minv.m[2][1] * nObj.y + Minv.m[2][2] * nObj.z));
    v3 ngW = norm3(V3(Minv.m[0][0] * nObjG.x + Minv.m[0][1] * nObjG.y + Minv.m[0][2] * nObjG.z,
                      Minv.m[1][0] * nObjG.x + Minv.m[1][1] * nObjG.y + Minv.m[1][2] * nObjG.z,
                      Minv.m[2][0] * nObjG.x + Minv.m[2][1] * nObjG.y + Minv.m[2][2] * nObjG.z));
    int ff = dot3(ray->d, ngW) < 0.0f;
    if (!ff) nW = neg3(nW);
    if (f->group >= 0) fill_material(m, f->group, uvx, uvy, &out->mat);
    else {
        po_material d = { { 1, 1, 1 }, 1.f, 0.f, { 1, 1, 1 }, 0.f, 1.5f, 0.f }; /* Material{} defaults, RayObject.h:12-19 */
        out->mat = d;
    }
    out->t = closestT / dirLen; out->p = pW; out->n = nW; out->frontFace = ff;
    return 1;
}

/* ------------------------------------------------------------------ closest hit (PathTracer.cpp:38-57) */
static int closest_hit(const po_scene* s, const ray_t* ray, hit_t* best, float* closestOut)
{
    const float kTMin = 1e-4f, kTMax = 1e30f;
    int hitSomething = 0;
    float closestT = kTMax;
    for (int i = 0; i < s->nobjs; ++i) {
        const object_t* o = &s->objs[i];
        hit_t h; memset(&h, 0, sizeof h); /* Hit h{} */
        int ok = o->kind == OBJ_SPHERE ? sphere_intersect(o, ray, kTMin, closestT, &h)
               : o->kind == OBJ_BOX    ? box_intersect(o, ray, kTMin, closestT, &h)
                                       : mesh_intersect(o, ray, kTMin, closestT, &h);
        if (ok) {
            if (h.t >= closestT) continue;
            hitSomething = 1; closestT = h.t; *best = h;
        }
    }
    *closestOut = closestT;
    return hitSomething;
}
int po_closest_hit(const po_scene* s, const float origin[3], const float dir[3], float rec[20])
{
    ray_t r; r.o = V3(origin[0], origin[1], origin[2]); r.d = V3(dir[0], dir[1], dir[2]); r.ior = 1.0f;
    hit_t h; memset(&h, 0, sizeof h);
    float ct;
    memset(rec, 0, 20 * sizeof(float));
    if (!closest_hit(s, &r, &h, &ct)) return 0;
    rec[0] = 1.f; rec[1] = h.t; rec[2] = h.p.x; rec[3] = h.p.y; rec[4] = h.p.z;
    rec[5] = h.n.x; rec[6] = h.n.y; rec[7] = h.n.z; rec[8] = h.frontFace ? 1.f : 0.f;
    memcpy(&rec[9], &h.mat, 11 * sizeof(float));
    return 1;
}

/* ------------------------------------------------------------------ integrator (PathTracer.cpp:19-303) */
/* SampleCosineHemisphereLocal, PathTracer.cpp:19-31 */
static v3 sample_cos_hemi(rng_t* g)
{
    float u1 = Rand01(g), u2 = Rand01(g);
    float r = sqrtf(u1);
    float phi = 2.0f * 3.1415926535f * u2;
    return V3(r * cosf(phi), r * sinf(phi), sqrtf(fmaxr(0.0f, 1.0f - u1)));
}
static float G1f(float c, float alpha) /* PathTracer.cpp:157-163, 259-266 */
{
    c = clampr(c, 0.0f, 1.0f);
    float s = sqrtf(fmaxr(0.0f, 1.0f - c * c));
    float t = (c > 0.0f) ? (s / c) : 0.0f;
    float r = sqrtf(1.0f + (alpha * alpha) * (t * t));
    return 2.0f / (1.0f + r);
}
/* GGX half-vector sample, PathTracer.cpp:125-137 and 218-235 */
static v3 sample_half(float roughness, float alpha, rng_t* g)
{
    if (roughness <= 1e-4f) return V3(0, 0, 1);
    float u1 = Rand01(g), u2 = Rand01(g);
    float phi = 2.0f * 3.1415926535f * u1;
    float a2 = alpha * alpha;
    float tan2t = a2 * u2 / fmaxr(1e-6f, 1.0f - u2);
    float cosT = 1.0f / sqrtf(1.0f + tan2t);
    float sinT = sqrtf(fmaxr(0.0f, 1.0f - cosT * cosT));
    return norm3(V3(sinT * cosf(phi), sinT * sinf(phi), cosT));
}
/* Extension (no reference counterpart): Russian roulette on the path throughput. Returns 0 when the path is
 * cut; otherwise rescales *weight and stores the new throughput in the child's state. */
static int roulette(rng_t* g, rng_t* child, v3* weight)
{
    v3 thr = mul3(g->thr, *weight);
    if ((g->flags & PO_FLAG_RUSSIAN_ROULETTE) && g->bounce + 1 >= 3) {
        float q = clampr(fmaxr(fmaxr(thr.x, thr.y), thr.z), 0.05f, 1.0f);
        if (!(po_rand01(g->key, g->bounce, 7) < q)) return 0;
        float iq = 1.0f / q;
        *weight = scl3(*weight, iq);
        thr = scl3(thr, iq);
    }
    child->thr = thr;
    return 1;
}
static v3 trace_ray(const po_scene* s, ray_t ray, int depth, int albedoOnly, rng_t* g)
{
    if (depth <= 0) return V3(0, 0, 0);
    COUNT(rays);
    const float kTMin = 1e-4f;
    hit_t best; memset(&best, 0, sizeof best);
    float closestT;
    if (!closest_hit(s, &ray, &best, &closestT)) return s->sky_size > 0 ? sky_lookup(s, ray.d) : s->background;
    const po_material* m = &best.mat;
    v3 albedo = V3(m->albedo[0], m->albedo[1], m->albedo[2]);
    if (albedoOnly) { /* PathTracer.cpp:62-71 */
        float dist = clampr(closestT / 20.f, 0.0f, 0.8f);
        return scl3(albedo, 1.0f - dist);
    }
    COUNT(shades);
    v3 L = scl3(V3(m->emission_colour[0], m->emission_colour[1], m->emission_colour[2]), m->emission_strength);
    v3 n = norm3(best.n);
    v3 t, b;
    if (fabsf(n.z) < 0.999f) t = norm3(cross3(V3(0, 0, 1), n));
    else t = norm3(cross3(V3(1, 0, 0), n));
    b = cross3(n, t);
    v3 wo = norm3(neg3(ray.d));
    v3 woL = V3(dot3(wo, t), dot3(wo, b), dot3(wo, n));
    float cosNo = fmaxr(0.0f, woL.z);
    /* glm::mix(x, y, a) = x*(1-a) + y*a */
    v3 F0 = V3(0.04f * (1.0f - m->metallic) + albedo.x * m->metallic,
               0.04f * (1.0f - m->metallic) + albedo.y * m->metallic,
               0.04f * (1.0f - m->metallic) + albedo.z * m->metallic);
    float roughness = clampr(m->roughness, 0.0f, 1.0f);
    float alpha = fmaxr(1e-4f, roughness * roughness);
    rng_t child = { g->key, g->bounce + 1, 0, g->flags, g->thr };
    const float pT = clampr(m->transmission, 0.0f, 1.0f);
    if (pT > 0.0f && Rand01(g) < pT) { /* PathTracer.cpp:99-204 */
        float eta_i = ray.ior, eta_m = m->ior;
        float eta_t = best.frontFace ? eta_m : 1.0f;
        float eta = eta_i / eta_t;
        float cos_i = clampr(dot3(neg3(ray.d), n), 0.0f, 1.0f);
        float r0 = (eta_i - eta_t) / (eta_i + eta_t);
        r0 *= r0;
        float F = r0 + (1.0f - r0) * powf(1.0f - cos_i, 5.0f);
        float sin2_t = eta * eta * fmaxr(0.0f, 1.0f - cos_i * cos_i);
        int TIR = (sin2_t > 1.0f);
        float pR = TIR ? 1.0f : F;
        if (Rand01(g) < pR) {
            v3 hL = sample_half(roughness, alpha, g);
            v3 wiL = reflect3(neg3(woL), norm3(hL));
            if (wiL.z <= 0.0f) return L;
            v3 wi = norm3(add3(add3(scl3(t, wiL.x), scl3(b, wiL.y)), scl3(n, wiL.z)));
            float cosNi = fmaxr(0.0f, wiL.z), cosNh = fmaxr(0.0f, hL.z), cosVh = fmaxr(0.0f, dot3(woL, hL));
            float fp = powf(1.0f - clampr(cosVh, 0.0f, 1.0f), 5.0f);
            v3 Fm = V3(F0.x + (1.0f - F0.x) * fp, F0.y + (1.0f - F0.y) * fp, F0.z + (1.0f - F0.z) * fp);
            float G = G1f(cosNo, alpha) * G1f(cosNi, alpha);
            float denom = fmaxr(1e-6f, cosNo * cosNh);
            v3 weight = div3s(scl3(Fm, G * cosVh), denom);
            float selPdf = fmaxr(1e-3f, pT * pR);
            weight = div3s(weight, selPdf);
            ray_t next; next.o = add3(best.p, scl3(wi, kTMin)); next.d = wi; next.ior = ray.ior;
            if (!roulette(g, &child, &weight)) return L;
            L = add3(L, mul3(weight, trace_ray(s, next, depth - 1, 0, &child)));
            return L;
        } else if (!TIR) {
            v3 tdir = norm3(refract3(ray.d, n, eta));
            float selPdf = fmaxr(1e-3f, pT * (1.0f - pR));
            float weight = (1.0f - F) / selPdf;
            ray_t next; next.o = add3(best.p, scl3(tdir, kTMin)); next.d = tdir; next.ior = eta_t;
            v3 w3 = V3(weight, weight, weight);
            if (!roulette(g, &child, &w3)) return L;
            L = add3(L, mul3(w3, trace_ray(s, next, depth - 1, 0, &child)));
            return L;
        }
    }
    float cosThetaV = clampr(cosNo, 0.0f, 1.0f);
    float fv = powf(1.0f - cosThetaV, 5.0f);
    v3 Fv = V3(F0.x + (1.0f - F0.x) * fv, F0.y + (1.0f - F0.y) * fv, F0.z + (1.0f - F0.z) * fv);
    float specProb = clampr((Fv.x + Fv.y + Fv.z) * (1.0f / 3.0f), 0.05f, 0.95f);
    if (Rand01(g) < specProb) { /* PathTracer.cpp:213-283 */
        v3 hL = sample_half(roughness, alpha, g);
        v3 wiL = reflect3(neg3(woL), norm3(hL));
        if (wiL.z <= 0.0f) return L;
        v3 wi = norm3(add3(add3(scl3(t, wiL.x), scl3(b, wiL.y)), scl3(n, wiL.z)));
        float cosNi = fmaxr(0.0f, wiL.z), cosNh = fmaxr(0.0f, hL.z), cosVh = fmaxr(0.0f, dot3(woL, hL));
        float fp = powf(1.0f - clampr(cosVh, 0.0f, 1.0f), 5.0f);
        v3 F = V3(F0.x + (1.0f - F0.x) * fp, F0.y + (1.0f - F0.y) * fp, F0.z + (1.0f - F0.z) * fp);
        float G = G1f(cosNo, alpha) * G1f(cosNi, alpha);
        float denom = fmaxr(1e-6f, cosNo * cosNh);
        v3 weight = div3s(scl3(F, G * cosVh), denom);
        weight = div3s(weight, fmaxr(1e-3f, specProb));
        weight = div3s(weight, fmaxr(1e-3f, 1.0f - pT));
        ray_t next; next.o = add3(best.p, scl3(n, kTMin)); next.d = wi; next.ior = 1.0f;
        if (!roulette(g, &child, &weight)) return L;
        L = add3(L, mul3(weight, trace_ray(s, next, depth - 1, 0, &child)));
    } else { /* PathTracer.cpp:284-300 */
        v3 dl = sample_cos_hemi(g);
        v3 dw = norm3(add3(add3(scl3(t, dl.x), scl3(b, dl.y)), scl3(n, dl.z)));
        ray_t next; next.o = add3(best.p, scl3(n, kTMin)); next.d = dw; next.ior = 1.0f;
        v3 weight = div3s(scl3(albedo, 1.0f - m->metallic), fmaxr(1e-3f, 1.0f - specProb));
        weight = div3s(weight, fmaxr(1e-3f, 1.0f - pT));
        if (!roulette(g, &child, &weight)) return L;
        L = add3(L, mul3(weight, trace_ray(s, next, depth - 1, 0, &child)));
    }
    return L;
}

static void trace_sample(const po_scene* s, const cam_state* cs, int w, int h, int x, int y, uint32_t sample,
                         int depth, int albedo_only, uint32_t seed, uint32_t flags, float rgb[3])
{
    float rr[6];
    camera_ray(cs, x, y, w, h, rr);
    ray_t ray; ray.o = V3(rr[0], rr[1], rr[2]); ray.d = V3(rr[3], rr[4], rr[5]); ray.ior = 1.0f;
    rng_t g = { po_path_key(seed, (uint32_t)(y * w + x), sample), 0, 0, flags, { 1.0f, 1.0f, 1.0f } };
    if (g_cnt) g_cnt->samples++;
    v3 c = trace_ray(s, ray, depth, albedo_only, &g);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}
void po_trace_sample(const po_scene* s, const po_camera* cam, int w, int h, int x, int y, uint32_t sample,
                     int depth, int albedo_only, uint32_t seed, float rgb[3])
{
    cam_state cs = camera_prepare(cam, w, h);
    trace_sample(s, &cs, w, h, x, y, sample, depth, albedo_only, seed, 0u, rgb);
}

/* ------------------------------------------------------------------ film (Film.cpp:34-104) */
void po_film_add_sample(float a[3], uint32_t* n, const float rgb[3])
{
    v3 c = V3(rgb[0], rgb[1], rgb[2]);
    const float maxLum = 12.0f;
    float lum = dot3(c, V3(0.2126f, 0.7152f, 0.0722f));
    if (lum > maxLum) c = scl3(c, maxLum / lum);
    a[0] += c.x; a[1] += c.y; a[2] += c.z;
    *n += 1u;
}
static float srgb_oetf(float c)
{
    float u = fmaxr(0.0f, c);
    return (u <= 0.0031308f) ? (12.92f * u) : (1.055f * powf(u, 1.0f / 2.4f) - 0.055f);
}
void po_film_resolve(const float* accum, const uint32_t* samples, int n, int tonemap, int colourspace, uint8_t* rgba)
{
    for (int p = 0; p < n; ++p) {
        uint32_t s = samples[p];
        v3 c = s ? div3s(V3(accum[3 * p], accum[3 * p + 1], accum[3 * p + 2]), (float)s) : V3(0, 0, 0);
        if (tonemap == 1) c = V3(c.x / (1.0f + c.x), c.y / (1.0f + c.y), c.z / (1.0f + c.z));
        v3 e = c;
        if (colourspace == 1) e = V3(srgb_oetf(c.x), srgb_oetf(c.y), srgb_oetf(c.z));
        e = V3(clampr(e.x, 0.f, 1.f), clampr(e.y, 0.f, 1.f), clampr(e.z, 0.f, 1.f));
        rgba[4 * p + 0] = (uint8_t)lroundf(e.x * 255.0f);
        rgba[4 * p + 1] = (uint8_t)lroundf(e.y * 255.0f);
        rgba[4 * p + 2] = (uint8_t)lroundf(e.z * 255.0f);
        rgba[4 * p + 3] = 255u;
    }
}

/* ------------------------------------------------------------------ frame driver (main.cpp:17-47) */
typedef struct {
    const po_scene* s; cam_state cs; int w, h; float* accum; uint32_t* samples;
    uint32_t spp_begin, spp_count; int depth, albedo_only; uint32_t seed, flags; int x0, y0, x1, y1;
    int next_row; pthread_mutex_t mu; po_counters total; int want_counters;
} job_t;
static void* worker(void* arg)
{
    job_t* j = (job_t*)arg;
    po_counters local; memset(&local, 0, sizeof local);
    g_cnt = j->want_counters ? &local : NULL;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        int y = j->next_row++;
        pthread_mutex_unlock(&j->mu);
        if (y >= j->y1) break;
        for (int x = j->x0; x < j->x1; ++x) {
            int p = y * j->w + x;
            /* one pixel's samples in index order: the same add order as spp successive frame passes */
            for (uint32_t k = 0; k < j->spp_count; ++k) {
                float rgb[3];
                trace_sample(j->s, &j->cs, j->w, j->h, x, y, j->spp_begin + k, j->depth, j->albedo_only, j->seed, j->flags, rgb);
                po_film_add_sample(&j->accum[3 * p], &j->samples[p], rgb);
            }
        }
    }
    if (j->want_counters) {
        pthread_mutex_lock(&j->mu);
        uint64_t* d = (uint64_t*)&j->total; const uint64_t* sft = (const uint64_t*)&local;
        for (size_t i = 0; i < sizeof(po_counters) / sizeof(uint64_t); ++i) d[i] += sft[i];
        pthread_mutex_unlock(&j->mu);
    }
    g_cnt = NULL;
    return NULL;
}
void po_render(const po_scene* s, const po_camera* cam, int w, int h, float* accum, uint32_t* samples,
               uint32_t spp_begin, uint32_t spp_count, int depth, int albedo_only, uint32_t seed,
               int x0, int y0, int x1, int y1, int nthreads, po_counters* counters)
{
    po_render_ext(s, cam, w, h, accum, samples, spp_begin, spp_count, depth, albedo_only, seed, x0, y0, x1, y1, nthreads, counters, 0u);
}
void po_render_ext(const po_scene* s, const po_camera* cam, int w, int h, float* accum, uint32_t* samples,
                   uint32_t spp_begin, uint32_t spp_count, int depth, int albedo_only, uint32_t seed,
                   int x0, int y0, int x1, int y1, int nthreads, po_counters* counters, uint32_t flags)
{
    job_t j; memset(&j, 0, sizeof j);
    j.flags = flags;
    j.s = s; j.cs = camera_prepare(cam, w, h); j.w = w; j.h = h; j.accum = accum; j.samples = samples;
    j.spp_begin = spp_begin; j.spp_count = spp_count; j.depth = depth; j.albedo_only = albedo_only; j.seed = seed;
    j.x0 = x0 < 0 ? 0 : x0; j.y0 = y0 < 0 ? 0 : y0; j.x1 = x1 > w ? w : x1; j.y1 = y1 > h ? h : y1;
    j.next_row = j.y0; j.want_counters = counters != NULL;
    pthread_mutex_init(&j.mu, NULL);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    pthread_t* th = (pthread_t*)malloc((size_t)nthreads * sizeof(pthread_t));
    for (int i = 0; i < nthreads; ++i) pthread_create(&th[i], NULL, worker, &j);
    for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
    free(th);
    pthread_mutex_destroy(&j.mu);
    if (counters) *counters = j.total;
}
