// Device functions of the path tracer: intersection, BVH traversal, hit reconstruction, BSDF sampling.
// gfx950 only (wave64). Reference citations are to /root/reference/src/PathTracer/.
#pragma once
#include <hip/hip_runtime.h>
#include "pt_types.h"

namespace ptamd {

#define PT_DEV __device__ __forceinline__

struct f3 { float x, y, z; };
PT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
PT_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV f3 cross(f3 x, f3 y) { return mk3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
PT_DEV float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
PT_DEV float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
PT_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
PT_DEV f3 normalize(f3 a) { return a * frsq(dot(a, a)); }
PT_DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
PT_DEV float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
// The minimum instructions as they are: fminf() makes the compiler quieten an operand it cannot prove canonical with an
// extra v_max x, x, x (the closest-hit distance, on every trip of the node loop); a NaN operand yields the other one
// either way.
PT_DEV float min_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
PT_DEV float min3_raw(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// ---------------------------------------------------------------- random source
// Counter-based replacement for the reference's thread_local mt19937 (PathTracer.cpp:8-16): a PCG
// output hash keyed by (seed, pixel, sample, bounce, draw), with a 64-bit key per path. Bit-identical to oracle/pt_oracle.c. The draw counter is
// mixed in by a multiply and an exclusive-or: added to the key it would let paths with nearby keys share draws.
PT_DEV uint32_t pcg_hash(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
PT_DEV uint32_t path_key(uint32_t seed_hash, uint32_t pixel, uint32_t sample)
{
    uint32_t k = pcg_hash(seed_hash ^ pixel);
    return pcg_hash(k + sample * 0x9E3779B9u);
}
// The upper half of the 64-bit path key: a second, independent hash chain over (seed, pixel, sample). It offsets the draw
// counter, so two paths draw the same numbers only if both halves agree (2^-64 per pair; with the lower half alone a 4K
// render at 16 384 samples per pixel would run every stream of random numbers about 32 times).
PT_DEV uint32_t path_key_hi(uint32_t seed_hash, uint32_t pixel, uint32_t sample)
{
    uint32_t k = pcg_hash((seed_hash ^ 0x68E31DA4u) + pixel * 0x9E3779B9u);
    return pcg_hash(k ^ (sample * 0x85EBCA6Bu));
}
constexpr uint32_t kDrawStep = 0x9E3779B9u;
// Draw number `draw` of bounce `bounce` hashes key ^ ((bounce * 8 + draw + 1) * kDrawStep + key_hi); as a running counter
// ctr = draw_counter(bounce, key_hi), advanced by kDrawStep per draw, a bounce costs one multiply-add and each draw an
// add and an exclusive-or.
PT_DEV uint32_t draw_counter(uint32_t bounce, uint32_t key_hi) { return (bounce * 8u + 1u) * kDrawStep + key_hi; }
PT_DEV float rand01c(uint32_t key, uint32_t ctr)
{
    uint32_t r = pcg_hash(key ^ ctr);
    return (float)(r >> 8) * 5.9604644775390625e-8f;
}
PT_DEV float rand01(uint32_t key, uint32_t key_hi, uint32_t bounce, uint32_t draw)
{
    uint32_t r = pcg_hash(key ^ ((bounce * 8u + draw + 1u) * 0x9E3779B9u + key_hi));
    return (float)(r >> 8) * 5.9604644775390625e-8f;
}

// ---------------------------------------------------------------- closest-hit search state
struct Ray { f3 o, d; };

// What the search keeps per lane: enough to rebuild the full Hit for the winner only.
constexpr int kMeshHit = 1 << 30; // Best::obj of a mesh hit: kMeshHit | index of the mesh instance (DMesh), not the object index
struct Best {
    float t;       // world-space distance (PathTracer.cpp:43 closestT)
    int obj;       // object index (sphere, box), kMeshHit | mesh instance index, or -1 = miss
    int tri;       // mesh hit (obj has kMeshHit): triangle slot in the scene-wide arrays (tri_base included), | kFrontBit when the ray meets its front side
    float u, v;    // mesh: barycentrics in that triangle's own record
};

// Sphere::RayIntersect (Sphere.cpp:32-53): returns t or -1.
PT_DEV float sphere_t(const DObject& ob, const Ray& r, float tMin, float tMax)
{
    f3 oc = r.o - mk3(ob.c[0], ob.c[1], ob.c[2]);
    // + eps (zero for a sphere, opaque to the compiler): keeps a and its reciprocal, which do not depend on the
    // sphere, from being hoisted in front of the object loop, where they would cost a register through every mesh
    // traversal and four instructions per ray in scenes without spheres
    float a = dot(r.d, r.d) + ob.eps;
    float h = dot(oc, r.d);
    float c = dot(oc, oc) - ob.radius * ob.radius;
    float disc = h * h - a * c;
    if (disc < 0.0f) return -1.0f;
    float sq = fsqrt(disc);
    float ia = frcp(a);
    float t = (-h - sq) * ia;
    if (t < tMin || t > tMax) {
        t = (-h + sq) * ia;
        if (t < tMin || t > tMax) return -1.0f;
    }
    return t;
}

struct BoxLocal { f3 ro, rd, tMin3; float tEntry, tExit; };
// Box::RayIntersect steps 1-4 (Box.cpp:16-48). SHORTCUT (the search loop; not the one call per hit of box_normal): an
// unrotated box (DObject::aligned, the walls of a room) has the world axes as its own, so its local ray is
// (origin - centre, direction); the six dot products with unit axes that this skips return these same values.
template <bool SHORTCUT>
PT_DEV BoxLocal box_slabs(const DObject& ob, const Ray& r)
{
    BoxLocal b;
    f3 v = r.o - mk3(ob.c[0], ob.c[1], ob.c[2]);
    if (SHORTCUT && ob.aligned) {
        // + pad1 (a zero the compiler cannot see through): without it the reciprocals below are loop-invariant and
        // get hoisted out of the object loop, where they cost three registers and three v_rcp per ray in scenes
        // that have no box at all
        const float zero = ob.pad1;
        b.ro = v; b.rd = mk3(r.d.x + zero, r.d.y + zero, r.d.z + zero);
    } else {
        f3 a0 = mk3(ob.ax0[0], ob.ax0[1], ob.ax0[2]), a1 = mk3(ob.ax1[0], ob.ax1[1], ob.ax1[2]), a2 = mk3(ob.ax2[0], ob.ax2[1], ob.ax2[2]);
        b.ro = mk3(dot(a0, v), dot(a1, v), dot(a2, v));
        b.rd = mk3(dot(a0, r.d), dot(a1, r.d), dot(a2, r.d));
    }
    const float BIG = 1e30f;
    f3 inv = mk3(b.rd.x != 0.0f ? frcp(b.rd.x) : -BIG, b.rd.y != 0.0f ? frcp(b.rd.y) : -BIG, b.rd.z != 0.0f ? frcp(b.rd.z) : -BIG);
    f3 he = mk3(ob.half[0], ob.half[1], ob.half[2]);
    f3 t1 = (-he - b.ro) * inv, t2 = (he - b.ro) * inv;
    b.tMin3 = mk3(fminf(t1.x, t2.x), fminf(t1.y, t2.y), fminf(t1.z, t2.z));
    f3 tMax3 = mk3(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y), fmaxf(t1.z, t2.z));
    b.tEntry = fmaxf(fmaxf(b.tMin3.x, b.tMin3.y), b.tMin3.z);
    b.tExit = fminf(fminf(tMax3.x, tMax3.y), tMax3.z);
    return b;
}
// Box.cpp:50-54: returns tHit or -1.
PT_DEV float box_t(const DObject& ob, const Ray& r, float tMin, float tMax)
{
    BoxLocal b = box_slabs<true>(ob, r);
    if (b.tExit < b.tEntry) return -1.0f;
    if (b.tExit < tMin) return -1.0f;
    float tHit = b.tEntry;
    if (tHit < tMin) tHit = b.tExit;
    if (tHit < tMin || tHit > tMax) return -1.0f;
    return tHit;
}
// Box.cpp:56-83: world normal (face-forwarded) for the winning hit.
PT_DEV f3 box_normal(const DObject& ob, const Ray& r, float tHit)
{
    BoxLocal b = box_slabs<false>(ob, r);
    f3 pL = b.ro + b.rd * tHit;
    f3 he = mk3(ob.half[0], ob.half[1], ob.half[2]);
    float eps = ob.eps;
    int axis; float sgn;
    if (fabsf(pL.x - he.x) <= eps) { axis = 0; sgn = 1.f; }
    else if (fabsf(pL.x + he.x) <= eps) { axis = 0; sgn = -1.f; }
    else if (fabsf(pL.y - he.y) <= eps) { axis = 1; sgn = 1.f; }
    else if (fabsf(pL.y + he.y) <= eps) { axis = 1; sgn = -1.f; }
    else if (fabsf(pL.z - he.z) <= eps) { axis = 2; sgn = 1.f; }
    else if (fabsf(pL.z + he.z) <= eps) { axis = 2; sgn = -1.f; }
    else {
        if (b.tEntry == b.tMin3.x) { axis = 0; sgn = (b.ro.x + b.tEntry * b.rd.x) > 0.0f ? 1.f : -1.f; }
        else if (b.tEntry == b.tMin3.y) { axis = 1; sgn = (b.ro.y + b.tEntry * b.rd.y) > 0.0f ? 1.f : -1.f; }
        else { axis = 2; sgn = (b.ro.z + b.tEntry * b.rd.z) > 0.0f ? 1.f : -1.f; }
    }
    f3 ax = axis == 0 ? mk3(ob.ax0[0], ob.ax0[1], ob.ax0[2]) : (axis == 1 ? mk3(ob.ax1[0], ob.ax1[1], ob.ax1[2]) : mk3(ob.ax2[0], ob.ax2[1], ob.ax2[2]));
    f3 nW = normalize(ax * sgn);
    return dot(r.d, nW) < 0.0f ? nW : -nW;
}

// Affine transform by the 3x4 rows stored in DMesh (glm's (m0*x + m1*y) + (m2*z + m3*w) grouping, Mesh.cpp:59-60,166).
PT_DEV f3 xform_point(const float* m, f3 p)
{
    return mk3((m[0] * p.x + m[1] * p.y) + (m[2] * p.z + m[3]),
               (m[4] * p.x + m[5] * p.y) + (m[6] * p.z + m[7]),
               (m[8] * p.x + m[9] * p.y) + (m[10] * p.z + m[11]));
}
PT_DEV f3 xform_dir(const float* m, f3 d)
{
    return mk3((m[0] * d.x + m[1] * d.y) + m[2] * d.z, (m[4] * d.x + m[5] * d.y) + m[6] * d.z, (m[8] * d.x + m[9] * d.y) + m[10] * d.z);
}
// transpose(mat3(Minv)) * n (Mesh.cpp:55,229-230): dot of n with column r of mat3(Minv) = minv[4*c + r] over c.
PT_DEV f3 xform_normal(const float* minv, f3 n)
{
    return mk3(minv[0] * n.x + minv[4] * n.y + minv[8] * n.z, minv[1] * n.x + minv[5] * n.y + minv[9] * n.z, minv[2] * n.x + minv[6] * n.y + minv[10] * n.z);
}

// Mesh::RayTriMT (Mesh.cpp:418-445) on the plane form of DTri: same determinant threshold, same t, u, v up to
// rounding, with the distance window of the caller (Mesh.cpp:104) folded in. INCLUSIVE: accept t == tHi.
// Straight-line code: every stage is computed and the eight conditions are combined at the end. The staged form (leave
// after the determinant, after the distance window, after u) skips work only when all 64 rays of the wave leave together,
// which incoherent rays rarely do, and pays three nested EXEC-mask regions per test: measured 6 % slower end to end on an
// LDS-resident scene.
PT_DEV float plane_coord(float4 q, f3 h) { return fmaf(q.x, h.x, fmaf(q.y, h.y, fmaf(q.z, h.z, q.w))); }      // q.xyz . h + q.w
PT_DEV float plane_offset(float4 q, f3 o) { return fmaf(-q.x, o.x, fmaf(-q.y, o.y, fmaf(-q.z, o.z, q.w))); } // q.w - q.xyz . o
template <bool INCLUSIVE>
PT_DEV bool tri_test_record(float4 q0, float4 q1, float4 q2, f3 o, f3 d, float tLo, float tHi, bool quad, float& t, float& u, float& v, float& dn)
{
    const f3 n = mk3(q0.x, q0.y, q0.z);
    dn = dot(n, d);
    t = plane_offset(q0, o) * frcp(dn);
    const f3 h = o + d * t;
    u = plane_coord(q1, h);
    v = plane_coord(q2, h);
    // a quad (two triangles forming a parallelogram, this record = the one at corner a) covers the whole unit square
    return !(fabsf(dn) < 1e-8f) & (t > 1e-8f) & !(t < tLo) & !(INCLUSIVE ? t > tHi : t >= tHi) & !(u < 0.0f) & !(u > 1.0f) & !(v < 0.0f)
           & !(quad ? v > 1.0f : u + v > 1.0f);
}
template <bool INCLUSIVE>
PT_DEV bool tri_test(const float4* __restrict__ tp, f3 o, f3 d, float tLo, float tHi, bool quad, float& t, float& u, float& v, float& dn)
{
    return tri_test_record<INCLUSIVE>(tp[0], tp[1], tp[2], o, d, tLo, tHi, quad, t, u, v, dn);
}

PT_DEV float4 sample_nearest(const DScene& sc, int img, float u, float v);
// Alpha MASK cutout (Mesh.cpp:106-125): false when the texel is below the cutoff. u, v: barycentrics in the
// triangle's own record.
PT_DEV bool alpha_pass(const DScene& sc, const DMesh& me, int tri, float u, float v)
{
    const float4* sh = sc.shade + ((size_t)me.tri_base + (size_t)tri) * 7;
    const int group = __float_as_int(sh[6].x);
    if (group < 0) return true;
    const DPbr& g = sc.groups[me.group_base + group];
    if (g.alpha_mode != 1) return true;
    float alpha = g.base_color_factor[3];
    if (g.base_color_tex >= 0) {
        float4 s4 = sh[4], s5 = sh[5];
        float w = 1.0f - u - v;
        float tu = w * s4.z + u * s5.x + v * s5.z, tv = w * s4.w + u * s5.y + v * s5.w;
        alpha *= sample_nearest(sc, g.base_color_tex, tu, tv).w;
    }
    return !(alpha < g.alpha_cutoff);
}
// Barycentrics of the point where the ray meets the plane of the record (no range checks); dn = n . d of that record.
PT_DEV void tri_uv(const float4* __restrict__ tp, f3 o, f3 d, float& u, float& v, float& dn)
{
    float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
    f3 n = mk3(q0.x, q0.y, q0.z);
    dn = dot(n, d);
    float t = plane_offset(q0, o) * frcp(dn);
    f3 h = o + d * t;
    u = plane_coord(q1, h);
    v = plane_coord(q2, h);
}
// The side of a triangle that a ray meets travels with the triangle index (bit 30; a mesh has fewer than 2^25
// faces): Mesh.cpp:160-162,230-233 take it from the sign of direction . MinvT * normalize(e1 x e2), which is the
// sign of n . d that the triangle test has just computed in object space (positive scale factors apart).
constexpr int kFrontBit = 1 << 30;
PT_DEV int with_side(int tri, float dn) { return dn < 0.0f ? (tri | kFrontBit) : tri; }

// The triangle of a hit on primitive record `slot`: the record's own triangle, or for a quad hit beyond the diagonal the
// next slot, whose barycentrics (and side) come from its own record.
PT_DEV int hit_triangle(const float4* tris, int slot, bool quad, f3 o, f3 d, float& u, float& v, float& dn)
{
    if (!(quad && u + v > 1.0f)) return slot;
    tri_uv(tris + (size_t)(slot + 1) * 3, o, d, u, v, dn);
    return slot + 1;
}

// Where the traversal reads BVH nodes and hot triangles (3 float4 each) from: the global arrays of the scene
// (DSignNode, 7 float4 per node, read by the work-sharing traversal), or the copy that the workgroup staged into LDS
// for a scene of a few KB (nodes in the LDS format of kLdsNodeBytes, read by mesh_traverse).
struct Geom { const float4* nodes; const float4* tris; };

// Iterative BVH traversal (replaces the stack walk of Mesh.cpp:79-149), "while-while" form: all lanes of a
// wave first descend through inner nodes until each holds a leaf (or has run out of work), then all of them
// test their leaf's triangles, so the two code bodies are never issued for a partial wave inside one loop
// trip. Near child first (Mesh.cpp:142-145); the far child goes on a per-lane stack kept in LDS (column
// `slot` of a [depth][BLOCK] array, so a wave's pushes and pops hit 64 different banks). A popped inner node
// that has fallen behind the closest hit costs one visit (both child tests fail against closestT); storing
// entry distances to drop such entries on pop was measured slower (8-byte stack traffic, more LDS per block).
// nodes / tris: per-lane work. node_trips / tri_trips: loop trips the whole wave paid for (counted by the first
// active lane of each trip), so lanes-per-trip = nodes / node_trips is the lane utilisation of that loop.
struct TraversalCounts { uint32_t nodes, tris, node_trips, tri_trips, steal_rounds; };
PT_DEV bool wave_leader() { return (int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1; }
typedef uint32_t StackEntry;

#ifndef PT_CUT_STEPS
#define PT_CUT_STEPS 2
#endif
// A search that cuts its tail (sample-major kernel, LDS-resident scenes whose last object is a mesh). A wave pays for the longest of
// its 64 walks: 26 node trips for 9 visits per lane on the mesh Cornell. With a cut the wave leaves the walk when the lanes
// still walking are down to `cut64` 64ths of those that entered it; a lane whose walk is not finished parks its traversal
// state (three words in LDS; its stack column stays where it is) and goes on in the wave's next search, beside the new rays
// of the lanes that finished, which are shaded in between. The walk of a lane is the same sequence of steps either way, so
// its closest hit (and which of two equal hits it keeps) does not change.
struct WalkBudget {
    uint32_t* park;   // LDS, [3][BLOCK]: node reference with the stack height above it, closest distance, triangle
    uint32_t cut64;   // 1..63 (wave-uniform)
    bool resume;      // in: the lane goes on with its parked walk
    bool unfinished;  // out: the lane's walk is parked again
};

template <int BLOCK, bool COUNT, bool BUDGET = false>
PT_DEV void mesh_traverse(const DScene& sc, const Geom& geom, const DMesh& me, f3 o, f3 d, float tMinObj, float& closestT,
                          int& bestTri, float& bu, float& bv, StackEntry* __restrict__ stack, int slot, TraversalCounts& cnt, WalkBudget* wb = nullptr)
{
    const char* __restrict__ nodes = reinterpret_cast<const char*>(geom.nodes) + (size_t)me.node_base * kLdsNodeBytes;
    const float4* __restrict__ tris = geom.tris + (size_t)me.tri_base * 3;
    const float BIG = 1e30f;
    f3 inv = mk3(fabsf(d.x) > 1e-30f ? frcp(d.x) : BIG, fabsf(d.y) > 1e-30f ? frcp(d.y) : BIG, fabsf(d.z) > 1e-30f ? frcp(d.z) : BIG);
    f3 oi = o * inv;
    // where this ray reads each axis of a node from: the four floats that start with the planes it enters through
    const char* __restrict__ ax = nodes + (inv.x < 0.0f ? 16 : 0);
    const char* __restrict__ ay = nodes + 32 + (inv.y < 0.0f ? 16 : 0);
    const char* __restrict__ az = nodes + 64 + (inv.z < 0.0f ? 16 : 0);
    constexpr int kDone = 0x7fffffff; // never a valid node reference
    int sp = 0;
    int cur = me.root >= 0 ? me.root * (int)kLdsNodeBytes : me.root; // inner nodes: byte offset
    if (BUDGET && wb->resume) {
        const uint32_t* pk = wb->park + slot;
        const uint32_t w = pk[0];
        cur = (int)(w << 5) >> 5; sp = (int)(w >> 27); closestT = __uint_as_float(pk[BLOCK]); bestTri = (int)pk[2 * BLOCK];
    }
    auto pop = [&]() {
        if (sp > 0) { --sp; cur = (int)stack[sp * BLOCK + slot]; } else cur = kDone;
    };
    auto node_step = [&]() {
        if (COUNT) { ++cnt.nodes; if (wave_leader()) ++cnt.node_trips; }
        const float4 x = *reinterpret_cast<const float4*>(ax + cur); // l.near, r.near, l.far, r.far
        const float4 y = *reinterpret_cast<const float4*>(ay + cur);
        const float4 z = *reinterpret_cast<const float4*>(az + cur);
        const int2 ch = *reinterpret_cast<const int2*>(nodes + cur + 96);
        float lmin = fmaxf(fmaxf(fmaf(x.x, inv.x, -oi.x), fmaf(y.x, inv.y, -oi.y)), fmaxf(fmaf(z.x, inv.z, -oi.z), 0.0f));
        float lmax = min_raw(min3_raw(fmaf(x.z, inv.x, -oi.x), fmaf(y.z, inv.y, -oi.y), fmaf(z.z, inv.z, -oi.z)), closestT);
        float rmin = fmaxf(fmaxf(fmaf(x.y, inv.x, -oi.x), fmaf(y.y, inv.y, -oi.y)), fmaxf(fmaf(z.y, inv.z, -oi.z), 0.0f));
        float rmax = min_raw(min3_raw(fmaf(x.w, inv.x, -oi.x), fmaf(y.w, inv.y, -oi.y), fmaf(z.w, inv.z, -oi.z)), closestT);
        bool hl = lmax >= lmin, hr = rmax >= rmin;
        const int left = ch.x, right = ch.y;
        if (hl && hr) {
            bool leftNear = lmin < rmin; // Mesh.cpp:143
            stack[sp * BLOCK + slot] = (uint32_t)(leftNear ? right : left);
            ++sp;
            cur = leftNear ? left : right;
        } else if (hl || hr) {
            cur = hl ? left : right;
        } else {
            pop();
        }
    };
    auto leaf_pass = [&]() { // leaf: 1..4 primitives (the builder makes at most 2); ~ref = first_slot << 6 | quad_mask << 2 | (primitives - 1)
        uint32_t code = ~(uint32_t)cur;
        uint32_t slotIdx = code >> 6, mask = (code >> 2) & 15u, prims = (code & 3u) + 1u;
        for (uint32_t i = 0; i < prims; ++i, mask >>= 1) {
            if (COUNT) { ++cnt.tris; if (wave_leader()) ++cnt.tri_trips; }
            const bool quad = (mask & 1u) != 0u;
            const uint32_t rec = slotIdx;
            slotIdx += quad ? 2u : 1u;
            float t, u, v, dn;
            // (records staged in LDS: their index fits 24 bits, and a 24-bit multiply runs at the full rate)
            if (!tri_test<false>(tris + __umul24(rec, 3u), o, d, tMinObj, closestT, quad, t, u, v, dn)) continue; // Mesh.cpp:104
            const int tri = hit_triangle(tris, (int)rec, quad, o, d, u, v, dn);
            if ((me.flags & 1u) && !alpha_pass(sc, me, tri, u, v)) continue; // Mesh.cpp:106-125
            closestT = t; bestTri = with_side(tri, dn);
            if (!BUDGET) { bu = u; bv = v; } // (a walk with a budget rebuilds them at its end: two registers fewer through the loops)
        }
        pop();
    };
    if (!BUDGET) {
        while (cur != kDone) {
            while ((uint32_t)cur < (uint32_t)kDone) node_step(); // inner node
            if (cur < 0) leaf_pass();
        }
    } else {
        // The same two phases with wave-uniform loop conditions (the number of lanes still walking is a property of the wave).
        // At least one walk has ended when the cut comes, and a wave of fewer than 64 / cut64 lanes is never cut, so every
        // search makes progress.
        const uint32_t started = (uint32_t)__popcll(__ballot(cur != kDone));
        const uint32_t floorLanes = (started * wb->cut64) >> 6;
        bool cut = false;
        for (;;) {
            bool go = (uint32_t)cur < (uint32_t)kDone;
            while (__any(go)) { // PT_CUT_STEPS node steps per pass: the loop's own tests and branches cost a twelfth of a step each
                if ((uint32_t)__popcll(__ballot(cur != kDone)) <= floorLanes) { cut = true; break; }
                if (go) node_step();
#pragma unroll
                for (int extra = 1; extra < PT_CUT_STEPS; ++extra)
                    if ((uint32_t)cur < (uint32_t)kDone) node_step();
                go = (uint32_t)cur < (uint32_t)kDone;
            }
            if (cur < 0) leaf_pass();
            const uint32_t walking = (uint32_t)__popcll(__ballot(cur != kDone));
            if (cut || walking == 0u || walking <= floorLanes) break;
        }
        wb->unfinished = cur != kDone;
        if (wb->unfinished) {
            // a node reference is a byte offset into the staged nodes or a leaf code (pt_types.h): 27 bits with its sign; the
            // stack height rides above it (the launcher grants a budget only to trees of at most 31 levels)
            uint32_t* pk = wb->park + slot;
            pk[0] = ((uint32_t)cur & 0x07ffffffu) | ((uint32_t)sp << 27); pk[BLOCK] = __float_as_uint(closestT); pk[2 * BLOCK] = (uint32_t)bestTri;
        } else if (bestTri >= 0) {
            // the barycentrics of the closest hit: the same formulas on the same record and ray return the bits that the
            // triangle test had (a quad's second triangle was resolved to its own slot when the hit was accepted)
            float dn;
            tri_uv(tris + (size_t)(bestTri & (kFrontBit - 1)) * 3, o, d, bu, bv, dn);
        }
    }
}

// ---------------------------------------------------------------- cooperative traversal (meshes read from global memory)
//
// Work sharing inside the wave. A wave pays for the longest traversal among its 64 rays, and with incoherent
// rays most lanes finish early (measured: 37 % of lanes busy in the node loop on a one-mesh scene, 8 % when a
// few rays of the wave enter a large mesh). Lanes that run out of work therefore take over pending subtrees of
// busy lanes: a lane traverses a *task* = (ray, subtree), not necessarily its own ray. The pieces:
//   * res[lane]: one 64-bit word per ray in LDS, (bits of closest t) << 32 | (triangle + 1). Every accepted hit
//     is published with an atomic minimum, so the closest hit of a ray is the minimum over all tasks that worked
//     on it, ties go to the lower triangle index whatever the visiting order, and the word doubles as the
//     shared pruning bound that every task re-reads when it pops its stack.
//   * the per-lane stack is used as a deque: its owner pops at the top, a thief takes the oldest entry (the
//     largest pending subtree) at the bottom.
//   * a steal round runs when at least PT_STEAL_MIN lanes are idle and some lane has pending entries: idle lane
//     number r pairs with pending lane number r through a small mailbox, copies that lane's ray with cross-lane
//     reads and starts on the stolen subtree.
// Every loop trip some lane advances (node step, leaf, or a steal that hands out a pending subtree), and the
// loop ends when no lane holds a node, which also means all stacks are empty: stealing happens before a lane
// with entries can go idle.
struct Coop { unsigned long long* res; uint32_t* mail; }; // LDS, one entry per thread of the block each
#ifndef PT_STEAL_MIN
#define PT_STEAL_MIN 8   // idle lanes that make a steal round worth its cost
#endif
#ifndef PT_WIDE_NODE_BIAS
#define PT_WIDE_NODE_BIAS 2 // nodes are opened as long as twice the lanes that want one reach the number waiting for a leaf test
#endif
#ifndef PT_WIDE_LEAF_BIAS
#define PT_WIDE_LEAF_BIAS 1 // a leaf trip runs when more than 1/PT_WIDE_LEAF_BIAS as many lanes wait for one as for a node
#endif
#ifndef PT_STEAL_AFTER_LEAVES
#define PT_STEAL_AFTER_LEAVES 1 // the eight-wide walk runs steal rounds after leaf trips too, not only after node trips
#endif
#ifndef PT_SHARE_LEAVES
#define PT_SHARE_LEAVES 1 // ... and hands out single hit leaves of lanes that hold several
#endif
#ifndef PT_SPLIT_GROUPS
#define PT_SPLIT_GROUPS 1 // the eight-wide walk also hands out parts of current groups
#endif
#ifndef PT_LEAD_BOXES
#define PT_LEAD_BOXES 1 // the leading run of unrotated boxes shares one set of direction reciprocals
#endif
#ifndef PT_STEAL_COOL
#define PT_STEAL_COOL 0  // node trips to wait after a steal round of the two-child walk (1 costs 6 %, 3 costs 16 %: the rounds pay)
#endif

PT_DEV int rank_in(unsigned long long mask) // set bits of mask below this lane
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// `valid`: this lane has a ray of its own to trace through the mesh (every lane of the wave must make the call:
// lanes without a ray still help). Returns the closest triangle (or -1) and its distance in closestT.
template <int BLOCK, bool COUNT>
PT_DEV int mesh_traverse_coop(const DScene& sc, const Geom& geom, const DMesh& me, bool valid, f3 o, f3 d, float tMinObj, float& closestT,
                         StackEntry* __restrict__ stack, const Coop& coop, int slot, TraversalCounts& cnt)
{
    const char* __restrict__ nodes = reinterpret_cast<const char*>(geom.nodes) + (size_t)me.node_base * sizeof(DSignNode);
    const float4* __restrict__ tris = geom.tris + (size_t)me.tri_base * 3;
    const float BIG = 1e30f;
    const int lane = slot & 63, waveBase = slot & ~63;
    auto inverse = [&](f3 dir) { return mk3(fabsf(dir.x) > 1e-30f ? frcp(dir.x) : BIG, fabsf(dir.y) > 1e-30f ? frcp(dir.y) : BIG, fabsf(dir.z) > 1e-30f ? frcp(dir.z) : BIG); };
    f3 inv = inverse(d);
    f3 oi = o * inv;
    constexpr int kDone = 0x7fffffff; // never a valid node index
    coop.res[slot] = ((unsigned long long)__float_as_uint(closestT) << 32); // triangle field 0 = none: an equal t does not replace it
    int owner = lane;        // whose ray this lane is working on
    float bound = closestT;  // pruning distance of the current task
    int sp = 0, bot = 0;     // own stack entries are rows [bot, sp)
    int cur = valid ? me.root : kDone;
    int cool = 0;
    const int leafMin = (me.flags & 16u) ? 65 : 0; // an instance that few rays reach: leaf holders wait for the node holders
    auto pop = [&]() {
        if (sp > bot) {
            --sp; cur = (int)stack[sp * BLOCK + slot];
            // hits that other tasks of this ray have published (high half of the 64-bit word)
            bound = fminf(bound, __uint_as_float(__hip_atomic_load(reinterpret_cast<uint32_t*>(&coop.res[waveBase + owner]) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)));
        } else cur = kDone;
    };
    // One loop trip is either a node step for every lane that holds an inner node or a leaf pass for every lane that
    // holds a leaf, whichever group is larger: the branch is wave-uniform, so the two bodies are never issued for a
    // partial wave in the same trip (the point of "while-while"). For an instance that few of the scene's rays reach (the
    // host's estimate, DMesh::flags bit 4) the leaf holders wait until no lane holds an inner node: a leaf pass for three
    // lanes costs more than it lets them gain (+6 % on a 7680-triangle mesh among primitives), whereas holding the
    // leaves back on a model that fills the frame costs 28 %.
    for (;;) {
        const bool isInner = (uint32_t)cur < (uint32_t)kDone, isLeaf = cur < 0;
        const int nInner = __popcll(__ballot(isInner)), nLeaf = __popcll(__ballot(isLeaf));
        if (nInner + nLeaf == 0) break;
        if (nInner > 0 && (nInner >= nLeaf || nLeaf < leafMin)) {
            if (isInner) {
                if (COUNT) { ++cnt.nodes; if (wave_leader()) ++cnt.node_trips; }
                // DSignNode: per axis the 16 bytes that start with the planes this task's ray enters through (the sign bit of
                // the reciprocal direction picks the half: a thief's ray is not the lane's own, so it is taken per visit)
                const char* n = nodes + (size_t)cur * sizeof(DSignNode);
                const float4 x = *reinterpret_cast<const float4*>(n + ((__float_as_uint(inv.x) >> 27) & 16u));      // l.near, r.near, l.far, r.far
                const float4 y = *reinterpret_cast<const float4*>(n + 32 + ((__float_as_uint(inv.y) >> 27) & 16u));
                const float4 z = *reinterpret_cast<const float4*>(n + 64 + ((__float_as_uint(inv.z) >> 27) & 16u));
                const int2 ch = *reinterpret_cast<const int2*>(n + 96);
                float lmin = fmaxf(fmaxf(fmaf(x.x, inv.x, -oi.x), fmaf(y.x, inv.y, -oi.y)), fmaxf(fmaf(z.x, inv.z, -oi.z), 0.0f));
                float lmax = fminf(fminf(fmaf(x.z, inv.x, -oi.x), fmaf(y.z, inv.y, -oi.y)), fminf(fmaf(z.z, inv.z, -oi.z), bound));
                float rmin = fmaxf(fmaxf(fmaf(x.y, inv.x, -oi.x), fmaf(y.y, inv.y, -oi.y)), fmaxf(fmaf(z.y, inv.z, -oi.z), 0.0f));
                float rmax = fminf(fminf(fmaf(x.w, inv.x, -oi.x), fmaf(y.w, inv.y, -oi.y)), fminf(fmaf(z.w, inv.z, -oi.z), bound));
                bool hl = lmax >= lmin, hr = rmax >= rmin;
                int left = ch.x, right = ch.y;
                if (hl && hr) {
                    bool leftNear = lmin < rmin; // Mesh.cpp:143
                    stack[sp * BLOCK + slot] = (uint32_t)(leftNear ? right : left);
                    ++sp;
                    cur = leftNear ? left : right;
                } else if (hl || hr) {
                    cur = hl ? left : right;
                } else {
                    pop();
                }
            }
            // ---- steal round
            const bool idle = cur == kDone, rich = sp > bot;
            const unsigned long long idleMask = __ballot(idle), richMask = __ballot(rich);
            --cool;
            if (cool < 0 && richMask != 0ull && __popcll(idleMask) >= PT_STEAL_MIN) {
                cool = PT_STEAL_COOL;
                if (COUNT) { if (wave_leader()) ++cnt.steal_rounds; }
                const int pairs = min(__popcll(idleMask), __popcll(richMask));
                const int r = rank_in(idle ? idleMask : richMask);
                const bool victim = rich && r < pairs, thief = idle && r < pairs;
                if (victim) coop.mail[waveBase + r] = (uint32_t)lane;
                __builtin_amdgcn_wave_barrier(); // LDS executes a wave's operations in order; this keeps the compiler from reordering them
                int src = lane;
                if (thief) src = (int)coop.mail[waveBase + r];
                // every lane takes part in the cross-lane reads; a lane that is not a thief reads itself
                o.x = __shfl(o.x, src, 64); o.y = __shfl(o.y, src, 64); o.z = __shfl(o.z, src, 64);
                d.x = __shfl(d.x, src, 64); d.y = __shfl(d.y, src, 64); d.z = __shfl(d.z, src, 64);
                tMinObj = __shfl(tMinObj, src, 64); bound = __shfl(bound, src, 64); owner = __shfl(owner, src, 64);
                // (the reciprocals come along too: cross-lane reads run beside the vector ALU, which is the busy unit)
                inv.x = __shfl(inv.x, src, 64); inv.y = __shfl(inv.y, src, 64); inv.z = __shfl(inv.z, src, 64);
                oi.x = __shfl(oi.x, src, 64); oi.y = __shfl(oi.y, src, 64); oi.z = __shfl(oi.z, src, 64);
                const int sBot = __shfl(bot, src, 64);
                if (thief) {
                    cur = (int)stack[sBot * BLOCK + waveBase + src];
                    sp = 0; bot = 0;
                }
                if (victim) ++bot;
            }
        } else if (isLeaf) {
            uint32_t code = ~(uint32_t)cur;
            uint32_t slotIdx = code >> 6, mask = (code >> 2) & 15u, prims = (code & 3u) + 1u;
            for (uint32_t i = 0; i < prims; ++i, mask >>= 1) {
                if (COUNT) { ++cnt.tris; if (wave_leader()) ++cnt.tri_trips; }
                const bool quad = (mask & 1u) != 0u;
                const uint32_t rec = slotIdx;
                slotIdx += quad ? 2u : 1u;
                float t, u, v, dn;
                if (!tri_test<true>(tris + (size_t)rec * 3, o, d, tMinObj, bound, quad, t, u, v, dn)) continue; // Mesh.cpp:104; an equal distance is settled by the published minimum
                int tri = (int)rec + ((quad && u + v > 1.0f) ? 1 : 0);
                if (me.flags & 1u) { // Mesh.cpp:106-125
                    tri = hit_triangle(tris, (int)rec, quad, o, d, u, v, dn);
                    if (!alpha_pass(sc, me, tri, u, v)) continue;
                }
                bound = t;
                atomicMin(&coop.res[waveBase + owner], ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)((uint32_t)tri + 1u));
            }
            pop();
        }
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned long long best = __hip_atomic_load(&coop.res[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    closestT = __uint_as_float((uint32_t)(best >> 32));
    return (int)(uint32_t)best - 1;
}

// ---------------------------------------------------------------- eight-wide traversal (meshes read from global memory)
//
// Walks DWideNode trees (pt_types.h). One loop trip = one node visit: five 16-byte reads bring the frame and the
// eight quantised child boxes, the slab tests run on bytes converted in registers (v_cvt_f32_ubyteN), and the hits
// come out as two 8-bit masks in traversal order (bit i = slot i ^ octant of the ray): leaf children, whose
// primitives are tested at once, and inner children, which stay in the current group {child_base, hits << 8 | imask}.
// The next node is the lowest set bit of the current group; a group that still has bits when a new node is opened goes
// on the per-lane LDS stack (two 4-byte rows per entry), so the stack holds at most one entry per tree level. Pending
// children carry no distance: one that has fallen behind the closest hit costs its visit, in which all eight boxes fail.
PT_DEV uint32_t permute_slots16(uint32_t m, uint32_t oct) // bit (s ^ oct) of each byte of the result = bit s of that byte of m
{
    if (oct & 1u) m = ((m & 0x5555u) << 1) | ((m >> 1) & 0x5555u);
    if (oct & 2u) m = ((m & 0x3333u) << 2) | ((m >> 2) & 0x3333u);
    if (oct & 4u) m = ((m & 0x0f0fu) << 4) | ((m >> 4) & 0x0f0fu);
    return m;
}
template <int S> PT_DEV float byte_of(uint32_t lo, uint32_t hi) // byte S (0..7) of the 8-byte pair as a float
{
    const uint32_t w = S < 4 ? lo : hi;
    return (float)((w >> (8 * (S & 3))) & 255u);
}
template <int S>
PT_DEV uint32_t wide_slab(uint32_t nx0, uint32_t nx1, uint32_t ny0, uint32_t ny1, uint32_t nz0, uint32_t nz1,
                          uint32_t fx0, uint32_t fx1, uint32_t fy0, uint32_t fy1, uint32_t fz0, uint32_t fz1,
                          f3 sc, f3 of, float bound)
{
    const float tn = fmaxf(fmaxf(fmaf(byte_of<S>(nx0, nx1), sc.x, of.x), fmaf(byte_of<S>(ny0, ny1), sc.y, of.y)), fmaxf(fmaf(byte_of<S>(nz0, nz1), sc.z, of.z), 0.0f));
    const float tf = fminf(fminf(fmaf(byte_of<S>(fx0, fx1), sc.x, of.x), fmaf(byte_of<S>(fy0, fy1), sc.y, of.y)), fminf(fmaf(byte_of<S>(fz0, fz1), sc.z, of.z), bound));
    return tf >= tn ? (1u << S) : 0u;
}

// Work sharing as in mesh_traverse_coop: a lane works on a task = (some lane's ray, a group of pending children);
// hits are published with an atomic minimum on the owner's 64-bit word, which is also the pruning bound that a task
// re-reads when it pops; an idle lane takes over the oldest stack entry (a whole group) of a lane that has one. A
// loop trip is wave-uniform: every lane that can opens one node, or every lane that holds hit leaf children tests
// the primitives of one of them, whichever group is larger. `valid`: this lane has a ray of its own (all lanes of
// the wave make the call). Returns the closest triangle slot (mesh-relative, or -1) and its distance in closestT.
template <int BLOCK, bool COUNT>
PT_DEV int mesh_traverse_wide(const DScene& sc, const DMesh& me, bool valid, f3 o, f3 d, float tMinObj, float& closestT,
                              StackEntry* __restrict__ stack, const Coop& coop, int slot, TraversalCounts& cnt)
{
    const float4* __restrict__ nodes = sc.wide_nodes + (size_t)me.node_base * 5;
    const float4* __restrict__ tris = sc.tris + (size_t)me.tri_base * 3;
    const float BIG = 1e30f;
    const int lane = slot & 63, waveBase = slot & ~63;
    auto inverse = [&](f3 dir) { return mk3(fabsf(dir.x) > 1e-30f ? frcp(dir.x) : BIG, fabsf(dir.y) > 1e-30f ? frcp(dir.y) : BIG, fabsf(dir.z) > 1e-30f ? frcp(dir.z) : BIG); };
    auto octant = [&](f3 iv) { return (iv.x < 0.0f ? 1u : 0u) | (iv.y < 0.0f ? 2u : 0u) | (iv.z < 0.0f ? 4u : 0u); };
    f3 inv = inverse(d);
    uint32_t oct = octant(inv);
    coop.res[slot] = ((unsigned long long)__float_as_uint(closestT) << 32); // triangle field 0 = none: an equal t does not replace it
    int owner = lane;        // whose ray this lane is working on
    float bound = closestT;  // pruning distance of the current task
    int sp = 0, bot = 0;     // own stack entries are [bot, sp), two rows each
    uint32_t gBase = 0u, gBits = 0u; // current group: first inner child, (pending inner hits in traversal order) << 8 | imask
    int node = valid ? 0 : -1;       // node to open next, or -1: take it from the current group
    uint32_t pLeaves = 0u, pTriBase = 0u, pMeta0 = 0u, pMeta1 = 0u; // hit leaf children of the last node, not yet tested
    for (;;) {
        const bool wantsLeaf = pLeaves != 0u;
        const bool wantsNode = !wantsLeaf && (node >= 0 || (gBits >> 8) != 0u || sp > bot);
        const int nNode = __popcll(__ballot(wantsNode)), nLeaf = __popcll(__ballot(wantsLeaf));
        if (nNode + nLeaf == 0) break;
        const bool nodeTrip = PT_WIDE_NODE_BIAS * nNode >= PT_WIDE_LEAF_BIAS * nLeaf;
        if (nodeTrip) {
            if (wantsNode) {
                if (node < 0) {
                    if ((gBits >> 8) == 0u) {
                        --sp;
                        gBase = stack[(2 * sp) * BLOCK + slot]; gBits = stack[(2 * sp + 1) * BLOCK + slot];
                        // hits that other tasks of this ray have published (high half of the 64-bit word)
                        bound = fminf(bound, __uint_as_float(__hip_atomic_load(reinterpret_cast<uint32_t*>(&coop.res[waveBase + owner]) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)));
                    }
                    const uint32_t i = (uint32_t)__ffs((int)(gBits >> 8)) - 1u;
                    gBits &= ~(0x100u << i);
                    const uint32_t s = i ^ oct;
                    node = (int)(gBase + (uint32_t)__popc(gBits & 0xffu & ((1u << s) - 1u)));
                }
                if (COUNT) { ++cnt.nodes; if (wave_leader()) ++cnt.node_trips; }
                const float4* n = nodes + (size_t)node * 5;
                const float4 h0 = n[0], h1 = n[1], q0 = n[2], q1 = n[3], q2 = n[4];
                node = -1;
                const uint32_t em = __float_as_uint(h0.w);
                const uint32_t imask = em >> 24;
                // this node's grid as the ray sees it: t = q * (pitch * inv) + (origin - o) * inv
                const f3 scl = mk3(__uint_as_float((em & 255u) << 23) * inv.x, __uint_as_float(((em >> 8) & 255u) << 23) * inv.y, __uint_as_float(((em >> 16) & 255u) << 23) * inv.z);
                const f3 off = mk3((h0.x - o.x) * inv.x, (h0.y - o.y) * inv.y, (h0.z - o.z) * inv.z);
                const uint32_t lx0 = __float_as_uint(q0.x), lx1 = __float_as_uint(q0.y), ly0 = __float_as_uint(q0.z), ly1 = __float_as_uint(q0.w);
                const uint32_t lz0 = __float_as_uint(q1.x), lz1 = __float_as_uint(q1.y), hx0 = __float_as_uint(q1.z), hx1 = __float_as_uint(q1.w);
                const uint32_t hy0 = __float_as_uint(q2.x), hy1 = __float_as_uint(q2.y), hz0 = __float_as_uint(q2.z), hz1 = __float_as_uint(q2.w);
                const bool nx = (oct & 1u) != 0u, ny = (oct & 2u) != 0u, nz = (oct & 4u) != 0u;
                const uint32_t ax0 = nx ? hx0 : lx0, ax1 = nx ? hx1 : lx1, bx0 = nx ? lx0 : hx0, bx1 = nx ? lx1 : hx1; // a: entry planes, b: exit planes
                const uint32_t ay0 = ny ? hy0 : ly0, ay1 = ny ? hy1 : ly1, by0 = ny ? ly0 : hy0, by1 = ny ? ly1 : hy1;
                const uint32_t az0 = nz ? hz0 : lz0, az1 = nz ? hz1 : lz1, bz0 = nz ? lz0 : hz0, bz1 = nz ? lz1 : hz1;
                uint32_t hit = 0u;
#define PT_SLAB(S) hit |= wide_slab<S>(ax0, ax1, ay0, ay1, az0, az1, bx0, bx1, by0, by1, bz0, bz1, scl, off, bound)
                PT_SLAB(0); PT_SLAB(1); PT_SLAB(2); PT_SLAB(3); PT_SLAB(4); PT_SLAB(5); PT_SLAB(6); PT_SLAB(7);
#undef PT_SLAB
                const uint32_t pm = permute_slots16(((hit & imask) << 8) | (hit & ~imask & 255u), oct);
                if ((gBits >> 8) != 0u) { // the group that was being worked on waits below the new one
                    stack[(2 * sp) * BLOCK + slot] = gBase; stack[(2 * sp + 1) * BLOCK + slot] = gBits;
                    ++sp;
                }
                gBase = __float_as_uint(h1.x); gBits = (pm & 0xff00u) | imask;
                pLeaves = pm & 255u; pTriBase = __float_as_uint(h1.y); pMeta0 = __float_as_uint(h1.z); pMeta1 = __float_as_uint(h1.w);
            }
        } else if (wantsLeaf) {
            const uint32_t i = (uint32_t)__ffs((int)pLeaves) - 1u;
            pLeaves &= pLeaves - 1u;
            const uint32_t s = i ^ oct;
            const uint32_t m = ((s & 4u ? pMeta1 : pMeta0) >> (8u * (s & 3u))) & 255u;
            // both records of a two-primitive leaf are requested before the first test: their latencies overlap
            const uint32_t rec0 = pTriBase + (m & 31u), mask = m >> 6;
            const bool two = ((m >> 5) & 1u) != 0u;
            const uint32_t rec1 = two ? rec0 + ((mask & 1u) ? 2u : 1u) : rec0;
            const float4* __restrict__ p0 = tris + (size_t)rec0 * 3;
            const float4* __restrict__ p1 = tris + (size_t)rec1 * 3;
            const float4 a0 = p0[0], a1 = p0[1], a2 = p0[2], b0 = p1[0], b1 = p1[1], b2 = p1[2];
            auto accept = [&](uint32_t rec, bool quad, float t, float u, float v, float dn) {
                int tri = (int)rec + ((quad && u + v > 1.0f) ? 1 : 0);
                if (me.flags & 1u) { // Mesh.cpp:106-125
                    tri = hit_triangle(tris, (int)rec, quad, o, d, u, v, dn);
                    if (!alpha_pass(sc, me, tri, u, v)) return;
                }
                bound = t;
                atomicMin(&coop.res[waveBase + owner], ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)((uint32_t)tri + 1u));
            };
            {
                if (COUNT) { ++cnt.tris; if (wave_leader()) ++cnt.tri_trips; }
                float t, u, v, dn;
                const bool quad = (mask & 1u) != 0u;
                if (tri_test_record<true>(a0, a1, a2, o, d, tMinObj, bound, quad, t, u, v, dn)) accept(rec0, quad, t, u, v, dn); // Mesh.cpp:104; an equal distance is settled by the published minimum
            }
            if (two) {
                if (COUNT) { ++cnt.tris; if (wave_leader()) ++cnt.tri_trips; }
                float t, u, v, dn;
                const bool quad = (mask & 2u) != 0u;
                if (tri_test_record<true>(b0, b1, b2, o, d, tMinObj, bound, quad, t, u, v, dn)) accept(rec1, quad, t, u, v, dn);
            }
        }
        if (PT_STEAL_AFTER_LEAVES || nodeTrip) {
            // ---- steal round: idle lanes take the oldest pending group of lanes that have one on their stack, or the
            // far part of the current group of a lane whose stack is empty (its upper four slots when both halves hold
            // hits, else its farthest hit): near the end of a wave's search the pending work sits in a few current groups
            const bool idle = pLeaves == 0u && node < 0 && (gBits >> 8) == 0u && sp == bot;
            const uint32_t pend = gBits >> 8;
            const bool rich = sp > bot || (PT_SPLIT_GROUPS && (pend & (pend - 1u)) != 0u) || (PT_SHARE_LEAVES && (pLeaves & (pLeaves - 1u)) != 0u);
            const unsigned long long idleMask = __ballot(idle), richMask = __ballot(rich);
            if (richMask != 0ull && __popcll(idleMask) >= PT_STEAL_MIN) {
                if (COUNT) { if (wave_leader()) ++cnt.steal_rounds; }
                const int pairs = min(__popcll(idleMask), __popcll(richMask));
                const int r = rank_in(idle ? idleMask : richMask);
                const bool victim = rich && r < pairs, thief = idle && r < pairs;
                uint32_t oBase = gBase, oBits = 0u; // what a victim hands over
                if (victim) {
                    coop.mail[waveBase + r] = (uint32_t)lane;
                    if (sp > bot) {
                        oBase = stack[(2 * bot) * BLOCK + slot]; oBits = stack[(2 * bot + 1) * BLOCK + slot];
                        ++bot;
                    } else if ((pend & (pend - 1u)) != 0u) {
                        const uint32_t up = pend & 0xf0u;
                        const uint32_t give = (up != 0u && (pend & 0x0fu) != 0u) ? up : (0x80000000u >> __clz((int)pend));
                        oBits = (give << 8) | (gBits & 255u);
                        gBits &= ~(give << 8);
                    } else { // the farthest of several hit leaf children, as (first triangle slot, primitive count and quad mask)
                        const uint32_t i = 31u - (uint32_t)__clz((int)pLeaves);
                        pLeaves &= ~(1u << i);
                        const uint32_t sl = i ^ oct;
                        const uint32_t m = ((sl & 4u ? pMeta1 : pMeta0) >> (8u * (sl & 3u))) & 255u;
                        oBase = pTriBase + (m & 31u);
                        oBits = 0x80000000u | (m >> 5);
                    }
                }
                __builtin_amdgcn_wave_barrier(); // LDS executes a wave's operations in order; this keeps the compiler from reordering them
                int src = lane;
                if (thief) src = (int)coop.mail[waveBase + r];
                // every lane takes part in the cross-lane reads; a lane that is not a thief reads itself
                o.x = __shfl(o.x, src, 64); o.y = __shfl(o.y, src, 64); o.z = __shfl(o.z, src, 64);
                d.x = __shfl(d.x, src, 64); d.y = __shfl(d.y, src, 64); d.z = __shfl(d.z, src, 64);
                tMinObj = __shfl(tMinObj, src, 64); bound = __shfl(bound, src, 64); owner = __shfl(owner, src, 64);
                oBase = __shfl(oBase, src, 64); oBits = __shfl(oBits, src, 64);
                // (the reciprocals and the octant come along too: cross-lane reads run beside the vector ALU, the busy unit)
                inv.x = __shfl(inv.x, src, 64); inv.y = __shfl(inv.y, src, 64); inv.z = __shfl(inv.z, src, 64);
                oct = (uint32_t)__shfl((int)oct, src, 64);
                if (thief) {

                    sp = 0; bot = 0;
                    if (oBits & 0x80000000u) { // one leaf: every meta byte describes it, so any slot number finds it
                        gBase = 0u; gBits = 0u;
                        pTriBase = oBase; pMeta0 = pMeta1 = ((oBits & 7u) << 5) * 0x01010101u; pLeaves = 1u;
                    } else { gBase = oBase; gBits = oBits; }
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned long long best = __hip_atomic_load(&coop.res[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    closestT = __uint_as_float((uint32_t)(best >> 32));
    return (int)(uint32_t)best - 1;
}

// One mesh object of the closest-hit loop through the work-sharing walks (Mesh::Hit, Mesh.cpp:57-77, 249): every lane of the
// wave makes the call together; a lane that does not want an answer (`wants` false) enters without a ray and takes over
// pending subtrees of the others. A closer hit replaces `best`.
template <int BLOCK, bool COUNT>
PT_DEV void shared_walk_hit(const DScene& sc, const Geom& geom, const DObject& ob, const Ray& ray, bool wants, Best& best,
                            StackEntry* __restrict__ stack, const Coop& coop, int slot, TraversalCounts& cnt)
{
    const DMesh& me = sc.meshes[ob.mesh];
    const bool identity = (me.flags & 8u) != 0u; // wave-uniform: multiplying by the identity returns its argument exactly
    f3 oo = identity ? ray.o : xform_point(me.minv, ray.o);
    f3 dd = identity ? ray.d : xform_dir(me.minv, ray.d);
    float len2 = dot(dd, dd);
    const bool valid = len2 != 0.0f && wants; // Mesh.cpp:64
    float dirLen = fsqrt(len2);
    float invLen = frcp(dirLen);
    dd = dd * invLen;
    float tMinObj = 1e-4f * dirLen;
    float closestT = best.t * dirLen;      // Mesh.cpp:72-75
    float u = 0.f, v = 0.f;
    int tri = (me.flags & 4u) ? mesh_traverse_wide<BLOCK, COUNT>(sc, me, valid, oo, dd, tMinObj, closestT, stack, coop, slot, cnt)
                              : mesh_traverse_coop<BLOCK, COUNT>(sc, geom, me, valid, oo, dd, tMinObj, closestT, stack, coop, slot, cnt);
    if (!valid) tri = -1;
    if (tri >= 0) { // the winner's own record, this lane's own ray
        float dn;
        tri_uv(sc.tris + ((size_t)me.tri_base + (size_t)tri) * 3, oo, dd, u, v, dn);
        tri = with_side(tri, dn);
        float tW = closestT * invLen;      // Mesh.cpp:249
        if (tW < best.t) { best.t = tW; best.obj = kMeshHit | ob.mesh; best.tri = tri + (int)me.tri_base; best.u = u; best.v = v; }
    }
}

// Does the ray meet the bounds of the scene's last object, a mesh, in front of the hit at tMax? In the mesh's own frame with
// the direction as transformed (not normalised: its parameter is the world one); the box of the positions comes padded from
// the host in DObject::c / half, and every doubtful case answers yes.
PT_DEV bool meets_tail_mesh(const DScene& sc, const Ray& ray, float tMax)
{
    const DObject& ob = sc.objects[sc.num_objects - 1];
    const DMesh& me = sc.meshes[ob.mesh];
    const bool identity = (me.flags & 8u) != 0u;
    const f3 oo = identity ? ray.o : xform_point(me.minv, ray.o);
    const f3 dd = identity ? ray.d : xform_dir(me.minv, ray.d);
    const float BIG = 1e30f;
    const f3 inv = mk3(dd.x != 0.0f ? frcp(dd.x) : BIG, dd.y != 0.0f ? frcp(dd.y) : BIG, dd.z != 0.0f ? frcp(dd.z) : BIG);
    const f3 ro = oo - mk3(ob.c[0], ob.c[1], ob.c[2]);
    const f3 he = mk3(ob.half[0], ob.half[1], ob.half[2]);
    const f3 t1 = (-he - ro) * inv, t2 = (he - ro) * inv;
    const float tEntry = fmaxf(fmaxf(fminf(t1.x, t2.x), fminf(t1.y, t2.y)), fminf(t1.z, t2.z));
    const float tExit = fminf(fminf(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y)), fmaxf(t1.z, t2.z));
    return !(tExit < tEntry) & !(tExit < 0.0f) & !(tEntry > tMax * 1.0001f);
}

// Closest-hit loop of PathTracer::TraceRay (PathTracer.cpp:38-57) over the flat object list. COOP selects the
// work-sharing traversal (every lane of the wave must then make the call together). `helper` (COOP only): this lane
// needs no answer of its own (its result is discarded by the caller); it enters the meshes without a ray and takes
// over pending subtrees of the other lanes. PRIMS: every object but the last (a scene whose only mesh is its last object:
// the caller runs that walk itself, shared_walk_hit).
template <int BLOCK, bool COUNT, bool COOP, bool BUDGET = false, bool PRIMS = false>
PT_DEV Best closest_hit(const DScene& sc, const Geom& geom, const Ray& ray, StackEntry* __restrict__ stack, const Coop& coop, int slot, TraversalCounts* counts = nullptr, bool helper = false,
                        WalkBudget* wb = nullptr)
{
    TraversalCounts cnt; cnt.nodes = 0; cnt.tris = 0; cnt.node_trips = 0; cnt.tri_trips = 0; cnt.steal_rounds = 0;
    const float kTMin = 1e-4f, kTMax = 1e30f;
    Best best; best.t = kTMax; best.obj = -1; best.tri = -1; best.u = 0.f; best.v = 0.f;
    if (BUDGET) wb->unfinished = false;
    int i = 0;
#if PT_LEAD_BOXES
    // The scene's leading run of unrotated boxes (the walls of a room; DScene::lead_aligned, 0 when shorter than two): their
    // local ray direction is the world one, so its reciprocals are formed once for the run instead of once per box (three
    // quarter-rate v_rcp each). Same values, same order of tests as the general loop below.
    if (sc.lead_aligned > 0) {
        const float BIG = 1e30f;
        const f3 inv = mk3(ray.d.x != 0.0f ? frcp(ray.d.x) : -BIG, ray.d.y != 0.0f ? frcp(ray.d.y) : -BIG, ray.d.z != 0.0f ? frcp(ray.d.z) : -BIG);
        for (; i < sc.lead_aligned; ++i) {
            const DObject& ob = sc.objects[i];
            const f3 ro = ray.o - mk3(ob.c[0], ob.c[1], ob.c[2]);
            const f3 he = mk3(ob.half[0], ob.half[1], ob.half[2]);
            const f3 t1 = (-he - ro) * inv, t2 = (he - ro) * inv;
            const float tEntry = fmaxf(fmaxf(fminf(t1.x, t2.x), fminf(t1.y, t2.y)), fminf(t1.z, t2.z));
            const float tExit = fminf(fminf(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y)), fmaxf(t1.z, t2.z));
            float tHit = tEntry < kTMin ? tExit : tEntry; // Box.cpp:50-54
            const bool hit = !(tExit < tEntry) & !(tExit < kTMin) & !(tHit < kTMin) & !(tHit > best.t) & (tHit < best.t);
            if (hit) { best.t = tHit; best.obj = i; }
        }
    }
    // The spheres that follow that run directly (DScene::sphere_run, 0 when fewer than two): a = d . d and its reciprocal
    // do not depend on the sphere (Sphere.cpp:36,46-48) and are formed once for the run.
    if (sc.sphere_run > 0) {
        const float a = dot(ray.d, ray.d), ia = frcp(a);
        for (const int end = i + sc.sphere_run; i < end; ++i) {
            const DObject& ob = sc.objects[i];
            const f3 oc = ray.o - mk3(ob.c[0], ob.c[1], ob.c[2]);
            const float h = dot(oc, ray.d);
            const float c = dot(oc, oc) - ob.radius * ob.radius;
            const float disc = h * h - a * c;
            if (disc < 0.0f) continue;
            const float sq = fsqrt(disc);
            float t = (-h - sq) * ia;
            if (t < kTMin || t > best.t) {
                t = (-h + sq) * ia;
                if (t < kTMin || t > best.t) continue;
            }
            if (t >= 0.0f && t < best.t) { best.t = t; best.obj = i; }
        }
    }
#endif
    const int numObjects = PRIMS ? sc.num_objects - 1 : sc.num_objects;
    for (; i < numObjects; ++i) {
        const DObject& ob = sc.objects[i];
        if (ob.kind == KIND_SPHERE) {
            float t = sphere_t(ob, ray, kTMin, best.t);
            if (t >= 0.0f && t < best.t) { best.t = t; best.obj = i; }
        } else if (ob.kind == KIND_BOX) {
            float t = box_t(ob, ray, kTMin, best.t);
            if (t >= 0.0f && t < best.t) { best.t = t; best.obj = i; }
        } else if (COOP) {
            shared_walk_hit<BLOCK, COUNT>(sc, geom, ob, ray, !helper, best, stack, coop, slot, cnt);
        } else if (!PRIMS) {
            const DMesh& me = sc.meshes[ob.mesh];
            const bool identity = (me.flags & 8u) != 0u; // wave-uniform: multiplying by the identity returns its argument exactly
            f3 oo = identity ? ray.o : xform_point(me.minv, ray.o);
            f3 dd = identity ? ray.d : xform_dir(me.minv, ray.d);
            float len2 = dot(dd, dd);
            if (len2 == 0.0f) continue; // Mesh.cpp:64
            float dirLen = fsqrt(len2);
            float invLen = frcp(dirLen);
            dd = dd * invLen;
            float tMinObj = kTMin * dirLen;
            float closestT = best.t * dirLen;      // Mesh.cpp:72-75
            int tri = -1; float u = 0.f, v = 0.f;
            if (BUDGET && i == sc.num_objects - 1) {
                // The cut covers the walk of the scene's last object. A lane that resumes has run the objects before it again
                // with its unchanged ray (beside the lanes that test them for the first time) and found the same hits.
                mesh_traverse<BLOCK, COUNT, true>(sc, geom, me, oo, dd, tMinObj, closestT, tri, u, v, stack, slot, cnt, wb);
                if (wb->unfinished) tri = -1; // not final yet
            } else {
                mesh_traverse<BLOCK, COUNT>(sc, geom, me, oo, dd, tMinObj, closestT, tri, u, v, stack, slot, cnt);
            }
            if (tri >= 0) {
                float tW = closestT * invLen;      // Mesh.cpp:249
                // the face is kept as an index into the scene-wide arrays: the hit reconstruction can then fetch its shade
                // record without waiting for the object and mesh records
                if (tW < best.t) { best.t = tW; best.obj = kMeshHit | ob.mesh; best.tri = tri + (int)me.tri_base; best.u = u; best.v = v; }
            }
        }
    }
    if (COUNT) *counts = cnt;
    return best;
}

// Full hit record for the winner (RayObject.h:22-30 with the material by value).
struct HitInfo {
    f3 p, n;
    bool frontFace;
    f3 albedo; float roughness, metallic;
    f3 emission;           // emissionColour * emissionStrength (PathTracer.cpp:76)
    float ior, transmission;
};

PT_DEV void load_material(const DScene& sc, int idx, HitInfo& h)
{
    const float4* m = reinterpret_cast<const float4*>(sc.materials + idx);
    float4 a = m[0], b = m[1], c = m[2];
    h.albedo = mk3(a.x, a.y, a.z); h.roughness = a.w; h.metallic = b.x;
    h.emission = mk3(b.y, b.z, b.w) * c.x; h.ior = c.y; h.transmission = c.z;
}

// SampleImageNearest (Mesh.cpp:15-37): RGBA8 arena, repeat wrap, no filtering, no sRGB decode.
PT_DEV float4 sample_nearest(const DScene& sc, int img, float u, float v)
{
    DImage im = sc.images[img];
    if (im.width <= 0 || im.height <= 0) return make_float4(1.f, 1.f, 1.f, 1.f);
    u = u - floorf(u); v = v - floorf(v);
    if (u < 0.f) u += 1.0f;
    if (v < 0.f) v += 1.0f;
    int x = (int)(u * (float)im.width), y = (int)(v * (float)im.height);
    x = min(max(x, 0), im.width - 1); y = min(max(y, 0), im.height - 1);
    uint32_t px = *reinterpret_cast<const uint32_t*>(sc.texels + im.offset + ((size_t)y * im.width + x) * 4);
    const float k = 255.0f;
    return make_float4((float)(px & 255u) / k, (float)((px >> 8) & 255u) / k, (float)((px >> 16) & 255u) / k, (float)(px >> 24) / k);
}

// Mesh::FillMaterialAt (Mesh.cpp:447-486) for a group that samples textures.
PT_DEV void fill_material_textured(const DScene& sc, const DPbr& g, float u, float v, HitInfo& h)
{
    float4 base = make_float4(g.base_color_factor[0], g.base_color_factor[1], g.base_color_factor[2], g.base_color_factor[3]);
    if (g.base_color_tex >= 0) { float4 t = sample_nearest(sc, g.base_color_tex, u, v); base.x *= t.x; base.y *= t.y; base.z *= t.z; }
    h.albedo = mk3(base.x, base.y, base.z);
    float rough = g.roughness_factor, metal = g.metallic_factor;
    if (g.metallic_roughness_tex >= 0) {
        float4 t = sample_nearest(sc, g.metallic_roughness_tex, u, v);
        rough = clampf(t.y * rough, 0.001f, 1.0f);
        metal = clampf(t.z * metal, 0.0f, 1.0f);
    }
    h.roughness = rough; h.metallic = metal;
    f3 em = mk3(g.emissive_factor[0], g.emissive_factor[1], g.emissive_factor[2]);
    if (g.emissive_tex >= 0) { float4 t = sample_nearest(sc, g.emissive_tex, u, v); em = em * mk3(t.x, t.y, t.z); }
    h.emission = em * fsqrt(dot(em, em)); // emissionColour * length(emissionColour)
    float tr = g.transmission_factor;
    if (g.transmission_tex >= 0) tr *= sample_nearest(sc, g.transmission_tex, u, v).x;
    h.transmission = clampf(tr, 0.0f, 1.0f);
    h.ior = g.ior;
}

// Rebuilds the Hit of the winning object: Sphere.cpp:55-68, Box.cpp:56-89, Mesh.cpp:153-254.
PT_DEV void fill_hit(const DScene& sc, const Ray& ray, const Best& best, HitInfo& h)
{
    // Every branch leaves its results in these locals and `h` is written once, after the branches join: stores
    // to different fields at the ends of the branches get merged by the compiler into stores through a selected
    // address, which forces those fields into scratch memory.
    // A mesh hit names its mesh instance and its face itself (kMeshHit): the instance record and the shade record of
    // the face are fetched side by side, with no object record before them.
    const bool isMesh = (best.obj & kMeshHit) != 0;
    const DObject& ob = sc.objects[isMesh ? 0 : best.obj];
    f3 p, n; bool frontFace;
    int matIndex = -1;
    if (!isMesh) load_material(sc, ob.material, h); // issued first: the loads overlap the normal computation
    if (!isMesh && ob.kind == KIND_SPHERE) {
        p = ray.o + ray.d * best.t;
        f3 outward = (p - mk3(ob.c[0], ob.c[1], ob.c[2])) * frcp(ob.radius);
        frontFace = dot(ray.d, outward) < 0.0f;
        n = frontFace ? outward : -outward;
    } else if (!isMesh) {
        p = ray.o + ray.d * best.t;
        n = box_normal(ob, ray, best.t);
        frontFace = false; // Box never reports a front face (Box.cpp:86-89 with PathTracer.cpp:47)
    } else {
        const DMesh& me = sc.meshes[best.obj & (kMeshHit - 1)];
        const float4* s = sc.shade + (size_t)(best.tri & (kFrontBit - 1)) * 7;
        float4 s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3], s4 = s[4], s5 = s[5], s6 = s[6];
        f3 pa = mk3(s0.x, s0.y, s0.z), pb = mk3(s0.w, s1.x, s1.y), pc = mk3(s1.z, s1.w, s2.x);
        f3 na = mk3(s2.y, s2.z, s2.w), nb = mk3(s3.x, s3.y, s3.z), nc = mk3(s3.w, s4.x, s4.y);
        float u = best.u, v = best.v;
        float w = 1.0f - u - v;
        f3 pObj = (pa * w + pb * u) + pc * v;
        f3 nObj = normalize((na * w + nb * u) + nc * v);
        float tu = w * s4.z + u * s5.x + v * s5.z, tv = w * s4.w + u * s5.y + v * s5.w;
        const bool identity = (me.flags & 8u) != 0u;
        p = identity ? pObj : xform_point(me.m, pObj);
        matIndex = __float_as_int(s6.y); // constant material slot of the face, or -1 (Mesh.cpp:242-246: slot 0 = Material{} defaults)
        if (matIndex < 0) {
            const int group = __float_as_int(s6.x);
            const DPbr& g = sc.groups[me.group_base + group];
            if (g.normal_tex >= 0) { // Mesh.cpp:174-226
                f3 dp1 = pb - pa, dp2 = pc - pa;
                float du1 = s5.x - s4.z, dv1 = s5.y - s4.w, du2 = s5.z - s4.z, dv2 = s5.w - s4.w;
                float det = du1 * dv2 - dv1 * du2;
                f3 T, B;
                if (fabsf(det) > 1e-8f) {
                    f3 t = (dp1 * dv2 - dp2 * dv1) * frcp(det);
                    T = normalize(t - nObj * dot(nObj, t));
                    B = normalize(cross(nObj, T));
                } else {
                    f3 up = (fabsf(nObj.z) < 0.999f) ? mk3(0, 0, 1) : mk3(1, 0, 0);
                    T = normalize(cross(up, nObj));
                    B = cross(nObj, T);
                }
                float4 tx = sample_nearest(sc, g.normal_tex, tu - floorf(tu), tv - floorf(tv));
                f3 nts = mk3((tx.x * 2.0f - 1.0f) * g.normal_scale, (tx.y * 2.0f - 1.0f) * g.normal_scale, tx.z * 2.0f - 1.0f);
                nts = normalize(nts);
                nObj = normalize(T * nts.x + B * nts.y + nObj * nts.z);
            }
            bool textured = (g.base_color_tex >= 0) | (g.metallic_roughness_tex >= 0) | (g.emissive_tex >= 0) | (g.transmission_tex >= 0);
            if (textured) fill_material_textured(sc, g, tu, tv, h);
            else matIndex = g.material; // normal map only: constant material slot
        }
        f3 nW = normalize(identity ? nObj : xform_normal(me.minv, nObj));
        frontFace = (best.tri & kFrontBit) != 0; // Mesh.cpp:230-233, see kFrontBit
        n = frontFace ? nW : -nW;
    }
    if (matIndex >= 0) load_material(sc, matIndex, h);
    h.p = p; h.n = n; h.frontFace = frontFace;
}

// GGX half-vector sample (PathTracer.cpp:125-137, 218-235) from the two uniforms of this bounce's direction
// sample; the azimuth 2*pi*u1 is taken in turns.
PT_DEV f3 sample_half(float roughness, float alpha, float u1, float u2)
{
    if (roughness <= 1e-4f) return mk3(0, 0, 1);
    float a2 = alpha * alpha;
    float tan2t = a2 * u2 * frcp(fmaxf(1e-6f, 1.0f - u2));
    float cosT = frsq(1.0f + tan2t);
    float sinT = fsqrt(fmaxf(0.0f, 1.0f - cosT * cosT));
    return normalize(mk3(sinT * __builtin_amdgcn_cosf(u1), sinT * __builtin_amdgcn_sinf(u1), cosT));
}
PT_DEV float smith_g1(float c, float alpha) // PathTracer.cpp:157-163
{
    c = clampf(c, 0.0f, 1.0f);
    float s = fsqrt(fmaxf(0.0f, 1.0f - c * c));
    float t = (c > 0.0f) ? s * frcp(c) : 0.0f;
    float r = fsqrt(1.0f + (alpha * alpha) * (t * t));
    return 2.0f * frcp(1.0f + r);
}

// One BSDF step of PathTracer::TraceRay (PathTracer.cpp:78-300) in throughput form.
// Returns false when the path ends here (specular sample below the surface, PathTracer.cpp:141,239-240).
// On true: ray/ior are the next segment and `weight` multiplies the throughput.
PT_DEV bool sample_bsdf(const HitInfo& h, Ray& ray, float& ior, f3& weight, uint32_t key, uint32_t key_hi, uint32_t bounce)
{
    const float kTMin = 1e-4f;
    f3 n = normalize(h.n);
    f3 t = (fabsf(n.z) < 0.999f) ? normalize(mk3(-n.y, n.x, 0.0f)) : normalize(mk3(0.0f, -n.z, n.y));
    f3 b = cross(n, t);
    f3 wo = normalize(-ray.d);
    f3 woL = mk3(dot(wo, t), dot(wo, b), dot(wo, n));
    float cosNo = fmaxf(0.0f, woL.z);
    float om = 1.0f - h.metallic;
    f3 F0 = mk3(0.04f * om + h.albedo.x * h.metallic, 0.04f * om + h.albedo.y * h.metallic, 0.04f * om + h.albedo.z * h.metallic);
    float roughness = clampf(h.roughness, 0.0f, 1.0f);
    float alpha = fmaxf(1e-4f, roughness * roughness);
    uint32_t ctr = draw_counter(bounce, key_hi); // draws of this bounce, in order
    auto next = [&]() { const float u = rand01c(key, ctr); ctr += kDrawStep; return u; };
    const float pT = clampf(h.transmission, 0.0f, 1.0f);
    bool interfaceReflect = false;
    float selPdf = 1.0f;
    if (pT > 0.0f && next() < pT) { // PathTracer.cpp:99-204
        float eta_i = ior;
        float eta_t = h.frontFace ? h.ior : 1.0f;
        float eta = eta_i * frcp(eta_t);
        float cos_i = clampf(dot(-ray.d, n), 0.0f, 1.0f);
        float r0 = (eta_i - eta_t) * frcp(eta_i + eta_t);
        r0 *= r0;
        float F = r0 + (1.0f - r0) * pow5(1.0f - cos_i);
        float sin2_t = eta * eta * fmaxf(0.0f, 1.0f - cos_i * cos_i);
        bool TIR = sin2_t > 1.0f;
        float pR = TIR ? 1.0f : F;
        if (next() < pR) {
            interfaceReflect = true;
            selPdf = fmaxf(1e-3f, pT * pR);
        } else if (!TIR) {
            // glm::refract (contrib/include/glm/detail/func_geometric.inl:129-138)
            float dn = dot(n, ray.d);
            float k = 1.0f - eta * eta * (1.0f - dn * dn);
            f3 tdir = (k >= 0.0f) ? (ray.d * eta - n * (eta * dn + fsqrt(k))) : mk3(0, 0, 0);
            tdir = normalize(tdir);
            float w = (1.0f - F) * frcp(fmaxf(1e-3f, pT * (1.0f - pR)));
            weight = mk3(w, w, w);
            ray.o = h.p + tdir * kTMin;
            ray.d = tdir;
            ior = eta_t;
            return true;
        }
    }
    bool specular = interfaceReflect;
    float specProb = 1.0f;
    if (!interfaceReflect) {
        float fv = pow5(1.0f - clampf(cosNo, 0.0f, 1.0f));
        f3 Fv = mk3(F0.x + (1.0f - F0.x) * fv, F0.y + (1.0f - F0.y) * fv, F0.z + (1.0f - F0.z) * fv);
        specProb = clampf((Fv.x + Fv.y + Fv.z) * (1.0f / 3.0f), 0.05f, 0.95f);
        specular = next() < specProb;
    }
    // Both lobes draw their direction from the same two counter slots, and both turn a local direction into a
    // world one through the same frame: the uniforms come before the branch and the frame after it, so a wave
    // whose lanes split between the lobes (nearly every wave: 5 % of a diffuse surface's samples are specular)
    // issues them once.
    const float u1 = rand01c(key, ctr), u2 = rand01c(key, ctr + kDrawStep);
    f3 local; // sampled direction in the (t, b, n) frame
    float k;
    if (specular) { // PathTracer.cpp:124-179 (interface reflection) and 213-283 (specular lobe): same GGX sample
        f3 hL = sample_half(roughness, alpha, u1, u2);
        float dwh = dot(woL, hL);
        local = hL * (2.0f * dwh) - woL; // reflect(-woL, hL)
        if (local.z <= 0.0f) return false;
        float cosNi = fmaxf(0.0f, local.z), cosNh = fmaxf(0.0f, hL.z), cosVh = fmaxf(0.0f, dwh);
        float fp = pow5(1.0f - clampf(cosVh, 0.0f, 1.0f));
        f3 F = mk3(F0.x + (1.0f - F0.x) * fp, F0.y + (1.0f - F0.y) * fp, F0.z + (1.0f - F0.z) * fp);
        float G = smith_g1(cosNo, alpha) * smith_g1(cosNi, alpha);
        k = G * cosVh * frcp(fmaxf(1e-6f, cosNo * cosNh));
        if (interfaceReflect) k *= frcp(selPdf);
        else k *= frcp(fmaxf(1e-3f, specProb)) * frcp(fmaxf(1e-3f, 1.0f - pT));
        weight = F * k;
    } else {
        // Lambert (PathTracer.cpp:284-300) with SampleCosineHemisphereLocal (PathTracer.cpp:19-31)
        float r = fsqrt(u1);
        local = mk3(r * __builtin_amdgcn_cosf(u2), r * __builtin_amdgcn_sinf(u2), fsqrt(fmaxf(0.0f, 1.0f - u1)));
        k = om * frcp(fmaxf(1e-3f, 1.0f - specProb)) * frcp(fmaxf(1e-3f, 1.0f - pT));
        weight = h.albedo * k;
    }
    f3 wi = normalize(t * local.x + b * local.y + n * local.z);
    if (interfaceReflect) {
        ray.o = h.p + wi * kTMin; // offset along the chosen direction, medium unchanged (PathTracer.cpp:174-176)
    } else {
        ray.o = h.p + n * kTMin;
        ior = 1.0f; // default-constructed Ray (PathTracer.cpp:278, 297)
    }
    ray.d = wi;
    return true;
}

// Camera::GetRay (Camera.cpp:46-63) with the matrices of CalculateMatrices.
PT_DEV Ray camera_ray(const DCamera& cam, int px, int py)
{
    const float* P = cam.inv_proj; const float* V = cam.inv_view;
    float nx = (((float)px + 0.5f) * frcp((float)cam.width)) * 2.f - 1.f;
    float ny = (((float)py + 0.5f) * frcp((float)cam.height)) * 2.f - 1.f;
    // invProj * (nx, ny, -+1, 1): (m0*x + m1*y) + (m2*z + m3*w)
    float nr[4], fr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float xy = P[r] * nx + P[4 + r] * ny;
        nr[r] = xy + (P[8 + r] * -1.f + P[12 + r]);
        fr[r] = xy + (P[8 + r] * 1.f + P[12 + r]);
    }
    float inr = frcp(nr[3]), ifr = frcp(fr[3]);
    f3 cn = mk3(nr[0] * inr, nr[1] * inr, nr[2] * inr);
    f3 cf = mk3(fr[0] * ifr, fr[1] * ifr, fr[2] * ifr);
    f3 dc = normalize(cf - cn);
    f3 dw = mk3(V[0] * dc.x + V[4] * dc.y + V[8] * dc.z, V[1] * dc.x + V[5] * dc.y + V[9] * dc.z, V[2] * dc.x + V[6] * dc.y + V[10] * dc.z);
    Ray ray;
    ray.o = mk3(V[12], V[13], V[14]);
    ray.d = normalize(normalize(dw));
    return ray;
}

// Miss radiance from the cubemap (an extension: the reference only has the constant of PathTracer.h:21).
// OpenGL face selection and orientation, nearest texel. The map is stored as float radiance, one 16-byte read per lookup:
// RGBA8 maps were converted at upload (texel / 255 * intensity, no sRGB decode), float maps are kept as they came.
PT_DEV f3 sky_lookup(const DScene& sc, f3 d)
{
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int face; float s, t, ma;
    if (ax >= ay && ax >= az) { ma = ax; if (d.x > 0.f) { face = 0; s = -d.z; t = -d.y; } else { face = 1; s = d.z; t = -d.y; } }
    else if (ay >= az) { ma = ay; if (d.y > 0.f) { face = 2; s = d.x; t = d.z; } else { face = 3; s = d.x; t = -d.z; } }
    else { ma = az; if (d.z > 0.f) { face = 4; s = d.x; t = -d.y; } else { face = 5; s = -d.x; t = -d.y; } }
    float u = (__fdiv_rn(s, ma) + 1.0f) * 0.5f, v = (__fdiv_rn(t, ma) + 1.0f) * 0.5f;
    const int n = sc.sky_size;
    int x = (int)(u * (float)n), y = (int)(v * (float)n);
    x = min(max(x, 0), n - 1); y = min(max(y, 0), n - 1);
    const float4 r = sc.sky[((size_t)face * n + y) * n + x];
    return mk3(r.x, r.y, r.z);
}

} // namespace ptamd
