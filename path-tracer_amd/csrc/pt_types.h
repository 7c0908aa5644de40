// Device-side data layout shared by the host flattening code and the HIP kernels.
//
// Everything the kernels read is one of:
//   * a wave-uniform table (objects, mesh instances, materials, PBR groups): read through scalar
//     loads, a few hundred bytes per scene;
//   * per-mesh BVH nodes (both child boxes in the parent: 112 B with the planes in both ray orders, or 80 B with eight
//     quantised children) and hot triangles (48 B, plane form), stored as float4 so that every per-lane fetch is a
//     16-byte load;
//   * cold per-triangle shading records (112 B) touched once per ray at the closest hit;
//   * the film (12 B accum + 4 B count per pixel), read and written once per launch.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace ptamd {

enum : int { KIND_SPHERE = 0, KIND_BOX = 1, KIND_MESH = 2 };

// RayObject.h:10-20 plus padding to 3 x float4.
struct DMaterial {
    float albedo[3]; float roughness;
    float metallic; float emission_colour[3];
    float emission_strength; float ior; float transmission; float pad;
};

// One entry per RayObject, in AddRayObject order (the closest-hit loop keeps the first of equal hits).
struct DObject {
    int32_t kind, material, mesh;
    int32_t aligned;                // box whose axes are the world axes (no rotation): its local ray is origin - centre, direction
    float c[3]; float radius;       // sphere/box centre; sphere radius. Mesh: centre of the box of its positions in its own frame
    float half[3]; float eps;       // box half extents; face epsilon = 1e-4 * max half extent (Box.cpp:63). Mesh: half extents of that
                                    // box, padded (negative when the mesh has no finite position)
    float ax0[3]; float pad1;       // box axes = columns of worldFromLocal (Box.cpp:16-23)
    float ax1[3]; float pad2;
    float ax2[3]; float pad3;
};

// Mesh instance (Mesh.cpp:47-55), rows of the affine 3x4 parts so a point transform is 3 dot products.
struct DMesh {
    float minv[12];   // minv[4*r + c] = Minv[c][r] (glm column-major), r = 0..2, c = 0..3
    float m[12];      // same for M
    uint32_t node_base;   // index of this mesh's first node in the node array
    uint32_t tri_base;    // index of its first hot triangle (in 48-B records)
    uint32_t group_base;  // index of its first PBR group
    uint32_t image_base;  // index of its first image
    uint32_t num_tris;
    int32_t root;         // child reference of the root (see DNode)
    uint32_t flags;       // bit 0: some group uses alpha MASK; bit 1: some group samples a texture; bit 2: the mesh is
                          // traversed through DWideNode (node_base then counts 80-byte nodes, the root is node 0); bit 3: the
                          // instance matrix is the identity (M x = x exactly: the kernels skip the products); bit 4: few of the
                          // scene's rays reach this instance (the work-sharing walk then holds its leaf passes back)
    uint32_t pad;
};

// BVH2 node holding both children's boxes, as the tree builder emits it (uploaded as DSignNode). Child reference: >= 0 -> node index (relative to node_base);
// < 0 -> leaf: ~ref = (first_slot << 6) | (quad_mask << 2) | (primitives - 1), slots relative to tri_base; bit i of
// quad_mask: primitive i is a quad (two adjacent slots tested through the first one's record).
struct DNode {
    float lmin[3]; float lmax0;     // f4 0: l.min.xyz, l.max.x
    float lmax12[2]; float rmin01[2]; // f4 1: l.max.yz, r.min.xy
    float rmin2; float rmax[3];     // f4 2: r.min.z, r.max.xyz
    int32_t left, right; int32_t pad[2]; // f4 3
};

// The form in which the two-child nodes are uploaded for the work-sharing walk over global memory (mesh_traverse_coop),
// one per DNode in the same order. Per axis eight floats [l.min, r.min, l.max, r.max | l.max, r.max, l.min, r.min]
// (l, r: the two children's boxes): a ray that travels along +axis reads the first four with one 16-byte load, one that
// travels along -axis the second four, and either way gets (l.near, r.near, l.far, r.far) in its own order, so the slab
// test needs no per-axis min/max (14 half-rate instructions fewer per visit) and computes the same values
// (a * inv - o * inv is monotonic in a). Child references as in DNode.
struct DSignNode {
    float axis[3][8];
    int32_t left, right; int32_t pad[2];
};
static_assert(sizeof(DSignNode) == 112, "DSignNode is 7 float4");

// Eight-wide node with quantised child boxes for meshes that are read from global memory, where a ray's cost is the
// bytes it pulls through L2 (a DNode spends 32 bytes per child box, this one 10). Child boxes are stored on a grid of
// 256 steps per axis that starts at `origin` and has a power-of-two pitch 2^(e - 127), rounded outwards. Children that
// are inner nodes are stored contiguously from child_base in slot order, so the k-th inner slot's node is child_base +
// (number of inner slots below k); leaf children name up to 2 primitives (4 triangle slots) inside the node's own range
// of at most 32 triangle slots from tri_base. Slots are assigned so that visiting them in the order of (slot ^ octant
// of the ray direction) goes roughly front to back. An empty slot has an inverted box (lo 255, hi 0) and is never hit.
struct DWideNode {
    float origin[3]; uint8_t ex, ey, ez, imask;       // f4 0; imask: bit s set when slot s is an inner node
    uint32_t child_base, tri_base; uint8_t meta[8];   // f4 1; meta of a leaf slot: bits 0-4 first triangle slot (relative to
                                                      //       tri_base), bit 5 primitives - 1, bits 6-7 quad mask
    uint8_t qlo_x[8], qlo_y[8];                       // f4 2
    uint8_t qlo_z[8], qhi_x[8];                       // f4 3
    uint8_t qhi_y[8], qhi_z[8];                       // f4 4
};
static_assert(sizeof(DWideNode) == 80, "DWideNode is 5 float4");

// Node of the LDS-resident traversal (scenes whose nodes and hot triangles fit kLdsBlockBytes), built by the host
// next to the DNode array, one per DNode in the same order, 28 dwords. Per axis eight floats
// [l.min, r.min, l.max, r.max | l.max, r.max, l.min, r.min] (l, r: the two children's boxes): a ray that travels
// along +axis reads the first four with one 16-byte read, one that travels along -axis the second four, and either
// way gets (l.near, r.near, l.far, r.far) in its own order, so the slab test needs no per-axis min/max and computes
// the same values (a * inv - o * inv is monotonic in a). Dwords 24, 25: child references, inner ones as byte
// offsets from the mesh's first node, leaves as in DNode; 26, 27 pad.
constexpr uint32_t kLdsNodeBytes = 112;
// LDS one workgroup of the render kernel may use when the scene's geometry is to live there: a third of the CU's
// 160 KB. Measured on a 108-triangle room with padded allocations: 5 workgroups per CU 38.8 ms, 4: 44.3, 3: 54.5,
// 2: 76.9, against 70 ms for the same scene read from global memory; so geometry stays in LDS down to three
// workgroups per CU. What has to fit: the traversal stack (1 KB per tree level), 3 KB of film sums, the LDS nodes
// and the hot triangles.
constexpr uint32_t kLdsBlockBytes = 160 * 1024 / 3;
constexpr uint32_t kLdsFixedBytes = 4 * 1024; // per workgroup, besides stack and geometry
// Launches of this many samples per pixel and more run the sample-major kernel (render_samples_kernel). Measured at 1080p,
// pixel-major against sample-major kernel in Msamples/s: mesh Cornell 3419 / 3184 at 8 samples, 3533 / 3770 at 16, 3625 / 4134
// at 32; the room around a 100 000-triangle model 1434 / 1514, 1479 / 1860, 1515 / 2093; glass and mirror spheres 6978 / 7040,
// 7208 / 7250, 7443 / 7756; a million triangles in the room 742 / 751, 761 / 784, 781 / 832.
constexpr uint32_t kSampleMajorMinSpp = 16;

// Hot triangle in plane form. With n = e1 x e2 (e1 = v1 - v0, e2 = v2 - v0): the determinant of Mesh::RayTriMT
// (Mesh.cpp:421-431) is -(n . d), its t is (n . v0 - n . o) / (n . d), and the barycentrics of the hit point h
// are u = m1 . h + d1, v = m2 . h + d2 with m1 = (e2 x n) / |n|^2, m2 = (n x e1) / |n|^2, d1 = -m1 . v0,
// d2 = -m2 . v0. Same 48 bytes as v0, e1, e2, about half the arithmetic per test. Computed in double on the host.
struct DTri {
    float n[3]; float nv0;    // f4 0
    float m1[3]; float d1;    // f4 1
    float m2[3]; float d2;    // f4 2
};
static_assert(sizeof(DTri) == 48, "DTri is 3 float4");

// Cold shading record in the same (leaf) order as DTri: ModelLoader::Face (ModelLoader.h:29-40).
struct DShade {
    float pa[3], pb[3], pc[3];
    float na[3], nb[3], nc[3];
    float ta[2], tb[2], tc[2];
    int32_t group;
    int32_t material;  // >= 0: constant material slot of the face (group without textures, or no group: slot 0), so the
                       // hit fill skips the group record; -1: the group samples textures or a normal map
    float pad[2];
};
static_assert(sizeof(DShade) == 112, "DShade is 7 float4");

// ModelLoader::PBRMaterial (ModelLoader.h:56-85), device form; image indices are global.
struct DPbr {
    int32_t base_color_tex, normal_tex, metallic_roughness_tex, emissive_tex;
    int32_t transmission_tex, alpha_mode; float alpha_cutoff, normal_scale;
    float base_color_factor[4];
    float emissive_factor[3]; float metallic_factor;
    float roughness_factor, transmission_factor, ior; int32_t material; // slot of the constant material when no texture is sampled
};

// ModelLoader::EmbeddedImage: texels always RGBA8 in one arena.
struct DImage { int32_t width, height; uint32_t offset; uint32_t pad; };

struct DScene {
    const DObject* objects; int32_t num_objects; int32_t max_stack;
    const DMesh* meshes;
    const DMaterial* materials;
    const float4* nodes;   // DSignNode as 7 float4
    const float4* wide_nodes; // DWideNode as 5 float4 (meshes with flag bit 2)
    const float4* tris;    // DTri as 3 float4
    const float4* shade;   // DShade as 7 float4
    const DPbr* groups;
    const DImage* images;
    const uint8_t* texels;
    float background[3];
    int32_t lead_aligned;     // objects [0, lead_aligned) are all unrotated boxes (DObject::aligned); 0 when fewer than two lead the list
    uint32_t tail_mesh;       // bit 0: the last object of the list is a mesh (its walk may cut its tail, WalkBudget); bit 1: it is the
                              // scene's only mesh (read from global memory, its shared walk may wait for more rays)
    uint32_t num_tris;        // total over all meshes
    const float4* lds_nodes;  // kLdsNodeBytes per node; present when lds_resident
    uint32_t lds_node_f4;     // their size in float4
    uint32_t lds_resident;    // the LDS nodes and the hot triangles fit kLdsBlockBytes with the stack: traverse them from LDS
    int32_t sky_size;      // cubemap miss radiance (no reference counterpart): 0 = constant background
    int32_t sphere_run;    // objects [lead_aligned, lead_aligned + sphere_run) are all spheres; 0 when fewer than two follow the boxes
    const float4* sky;     // 6 faces (+x -x +y -y +z -z) of sky_size^2 texels of float radiance (rgb, w unused), row 0 = top
};

// Camera matrices as Camera::CalculateMatrices leaves them (Camera.cpp:22-35), glm column-major.
struct DCamera {
    float inv_proj[16];
    float inv_view[16];
    int32_t width, height, pad0, pad1;
};

struct DRenderParams {
    DCamera cam;
    float* accum;          // w*h*3
    uint32_t* samples;     // w*h
    unsigned long long* counters; // the 10 fields of ptamd_stats in order, or null
    int32_t x0, y0, x1, y1;
    uint32_t spp_begin, spp_count;
    int32_t depth, albedo_only;
    uint32_t seed, flags;  // flags: PTAMD_FLAG_* of the public header
    int32_t tiles_x, tiles_y; // 8x8-pixel tiles covering the rect
    int32_t stack_depth;      // rows of the per-lane traversal stack in LDS (set by the launcher)
    uint32_t walk_budget;     // sample-major LDS kernel, scenes whose last object is a mesh: the 64ths of a wave's walking lanes at which
                              // it cuts a walk and parks what is left (WalkBudget; set by the host, 0 = never). Global-memory scenes
                              // whose only mesh is their last object: the lanes (1..64) that wait at its bounds before the wave
                              // walks it (0 = every search walks it)
    // Interleaved-tile launches (tile_stride > 0, ptamd_render_tiles) cover, instead of the rect, the square tiles of
    // tile_size pixels (a multiple of 8) whose row-major index is tile_phase + k * tile_stride, k < tiles_owned;
    // tile_cols tiles per film row, tile_side = tile_size / 8 wave-tiles per tile row, tile_waves = tile_side^2.
    int32_t tile_size, tile_phase, tile_stride, tile_cols;
    int32_t tile_side, tile_waves, tiles_owned, pad2;
};

} // namespace ptamd
