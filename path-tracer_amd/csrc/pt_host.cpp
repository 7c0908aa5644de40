// C ABI of the path-tracing core (include/ptamd.h): scene container, flattening + upload, film, launches.
// There is no CPU rendering path in this library: every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ptamd.h"
#include "bvh_build.h"
#include "gltf_loader.h"
#include "hostmath.h"
#include "pt_types.h"

namespace ptamd {
hipError_t launch_render(const DScene& sc, const DRenderParams& rp, hipStream_t stream);
hipError_t launch_camera_rays(const DCamera& cam, float* d_out, hipStream_t stream);
hipError_t launch_closest_hits(const DScene& sc, const float* d_rays, uint32_t n, float* d_rec, uint32_t* d_counts, hipStream_t stream);
hipError_t launch_average(const float* accum, const uint32_t* samples, int n, float* avg, hipStream_t stream);
hipError_t launch_resolve(const float* accum, const uint32_t* samples, int n, int tonemap, int colourspace, uint32_t* rgba, hipStream_t stream);
int max_stack_depth();
hipError_t scratch_counters(int device, unsigned long long** out);
} // namespace ptamd

using namespace ptamd;

namespace {

thread_local std::string g_error;
hipStream_t g_own_stream = nullptr;
hipStream_t g_stream = nullptr;
bool g_initialised = false;
int g_device = -1;
constexpr int kNumCounters = 10;
const unsigned long long* g_last_counters = nullptr; // counters of the last counting launch (a scene's own buffer): what ptamd_render_stats reads
unsigned long long* g_static_counters = nullptr; // the same, in the library's static device array (no allocation behind it)

int fail(int code, const std::string& msg) { g_error = msg; return code; }

#define PT_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(PTAMD_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

int need_init()
{
    if (!g_initialised) return fail(PTAMD_ERR_DEVICE, "ptamd_init has not succeeded: no HIP device selected (there is no CPU fallback)");
    return PTAMD_OK;
}

struct HostMesh {
    Model model;
    BvhResult bvh;
    bool bvh_valid = false;
    float bmin[3] = { 0, 0, 0 }, bmax[3] = { 0, 0, 0 }; // bounds of the positions, set with the tree
};

struct HostObject {
    int kind = KIND_SPHERE;
    float pos[3] = { 0, 0, 0 }, rot[3] = { 0, 0, 0 }, size[3] = { 1, 1, 1 }; // size: box extents or mesh scale
    float radius = 1.0f;
    ptamd_material mat;
    std::shared_ptr<HostMesh> mesh;
};

template <class T> struct DevBuf {
    T* ptr = nullptr; size_t count = 0;
    int upload(const std::vector<T>& v)
    {
        release();
        count = v.size();
        size_t bytes = std::max<size_t>(1, count) * sizeof(T);
        PT_HIP(hipMalloc(reinterpret_cast<void**>(&ptr), bytes));
        if (count) PT_HIP(hipMemcpy(ptr, v.data(), count * sizeof(T), hipMemcpyHostToDevice));
        return PTAMD_OK;
    }
    void release() { if (ptr) { (void)hipFree(ptr); ptr = nullptr; } count = 0; }
};

} // namespace

struct ptamd_scene {
    std::vector<HostObject> objects;
    float background[3] = { 0.2f, 0.2f, 0.2f };
    bool committed = false;
    // What the next commit has to do. structure_dirty: the object list, a mesh's triangles or a group's use of images
    // changed, so everything is flattened and uploaded again. Otherwise only the rows named by `edits` (per object:
    // 1 transform / shape, 2 material, 4 a mesh's material groups) are rewritten in the small tables; nodes, triangles,
    // shading records and texels stay where they are.
    bool structure_dirty = true;
    std::vector<uint8_t> edits;
    std::vector<DObject> h_objects; std::vector<DMesh> h_meshes; std::vector<DPbr> h_groups; // host mirrors of the last full commit
    uint64_t geometry_uploads = 0, row_updates = 0;
    double last_commit_seconds = 0.0, last_bvh_seconds = 0.0;
    DevBuf<DObject> d_objects; DevBuf<DMesh> d_meshes; DevBuf<DMaterial> d_materials;
    DevBuf<DSignNode> d_nodes; DevBuf<DWideNode> d_wide; DevBuf<float> d_lds_nodes; DevBuf<DTri> d_tris; DevBuf<DShade> d_shade; DevBuf<DPbr> d_groups;
    DevBuf<DImage> d_images; DevBuf<uint8_t> d_texels;
    DevBuf<float> d_sky; int sky_size = 0;
    unsigned long long* d_counters = nullptr; // this scene's PTAMD_FLAG_COUNT tallies (allocated by its first counting render)
    // Schedule of the sample-major kernel for this scene's mesh walks (DRenderParams::walk_budget): -1 = not measured yet for
    // (film size, depth); LDS-resident scenes: the 64ths of a wave's walking lanes at which a walk is cut; scenes read from
    // global memory: the waiting lanes that start a deferred shared walk; 0 = the plain schedule. Measured at the first
    // eligible launch after a commit (tune_walk_cut).
    int walk_cut = -1, walk_cut_w = 0, walk_cut_h = 0, walk_cut_depth = 0;
    uint32_t walk_cut_uses = 0; // launches that have used the measured schedule
    DScene dscene;
    void release() // what commit rebuilds; the sky map is uploaded by its own call and outlives commits
    {
        d_objects.release(); d_meshes.release(); d_materials.release(); d_nodes.release(); d_wide.release(); d_lds_nodes.release(); d_tris.release();
        d_shade.release(); d_groups.release(); d_images.release(); d_texels.release();
    }
};

struct ptamd_film {
    int w = 0, h = 0;
    float* accum = nullptr; uint32_t* samples = nullptr;
    bool external = false;
    float* avg = nullptr; uint32_t* rgba = nullptr; // lazily allocated staging
    void release_staging() { if (avg) (void)hipFree(avg); if (rgba) (void)hipFree(rgba); avg = nullptr; rgba = nullptr; }
    void release()
    {
        release_staging();
        if (!external) { if (accum) (void)hipFree(accum); if (samples) (void)hipFree(samples); }
        accum = nullptr; samples = nullptr;
    }
};

struct ptamd_model { Model model; };

namespace {

DMaterial to_dmaterial(const ptamd_material& m)
{
    DMaterial d; std::memset(&d, 0, sizeof d);
    std::memcpy(d.albedo, m.albedo, 12); d.roughness = m.roughness; d.metallic = m.metallic;
    std::memcpy(d.emission_colour, m.emission_colour, 12); d.emission_strength = m.emission_strength;
    d.ior = m.ior; d.transmission = m.transmission;
    return d;
}

// Mesh::FillMaterialAt (Mesh.cpp:447-486) for a group without textures: constant per group.
ptamd_material group_material(const ptamd_pbr& g)
{
    ptamd_material m;
    for (int k = 0; k < 3; ++k) m.albedo[k] = g.base_color_factor[k];
    m.roughness = g.roughness_factor; m.metallic = g.metallic_factor;
    for (int k = 0; k < 3; ++k) m.emission_colour[k] = g.emissive_factor[k];
    const float* e = g.emissive_factor;
    m.emission_strength = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    m.transmission = std::min(std::max(g.transmission_factor, 0.0f), 1.0f);
    m.ior = g.ior;
    return m;
}

void rows3x4(const Mat4& M, float out[12])
{
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) out[4 * r + c] = M.m[c][r];
}

int check_scene(const ptamd_scene* s) { return s ? PTAMD_OK : fail(PTAMD_ERR_ARG, "null scene handle"); }
int check_object(const ptamd_scene* s, int object)
{
    if (!s) return fail(PTAMD_ERR_ARG, "null scene handle");
    if (object < 0 || object >= (int)s->objects.size()) return fail(PTAMD_ERR_ARG, "object index out of range");
    return PTAMD_OK;
}

int copy_mesh_desc(const ptamd_mesh_desc* d, Model& m)
{
    if (!d || !d->positions) return fail(PTAMD_ERR_ARG, "mesh description without positions");
    if (d->num_faces == 0) return fail(PTAMD_ERR_ARG, "Mesh: no faces in model"); // Mesh.cpp:293
    if (d->num_faces >= (1u << 25)) return fail(PTAMD_ERR_LIMIT, "mesh has too many faces (limit 2^25 - 1)");
    if (d->num_groups && !d->groups) return fail(PTAMD_ERR_ARG, "mesh description counts material groups but has no group array");
    if (d->num_images && !d->images) return fail(PTAMD_ERR_ARG, "mesh description counts images but has no image array");
    const size_t nf = d->num_faces;
    m.positions.assign(d->positions, d->positions + nf * 9);
    if (d->normals) m.normals.assign(d->normals, d->normals + nf * 9); else m.normals.assign(nf * 9, 0.0f);
    if (d->texcoords) m.texcoords.assign(d->texcoords, d->texcoords + nf * 6); else m.texcoords.assign(nf * 6, 0.0f);
    if (d->group) m.group.assign(d->group, d->group + nf); else m.group.assign(nf, -1);
    for (size_t i = 0; i < nf; ++i)
        if (m.group[i] >= (int32_t)d->num_groups) return fail(PTAMD_ERR_ARG, "face material group index out of range");
    m.groups.assign(d->groups, d->groups + d->num_groups);
    m.group_names.assign(d->num_groups, std::string());
    m.image_data.clear(); m.images.clear();
    for (uint32_t i = 0; i < d->num_images; ++i) {
        const ptamd_image& im = d->images[i];
        std::vector<uint8_t> rgba;
        if (im.width > 0 && im.height > 0 && im.channels > 0 && im.data) {
            // SampleImageNearest (Mesh.cpp:33-35) fills missing channels with 0 and alpha with 1: expand to RGBA8 once
            size_t n = (size_t)im.width * im.height;
            rgba.resize(n * 4);
            for (size_t p = 0; p < n; ++p)
                for (int c = 0; c < 4; ++c)
                    rgba[4 * p + c] = c < im.channels ? im.data[p * im.channels + c] : (c == 3 ? 255 : 0);
        }
        m.image_data.push_back(std::move(rgba));
        ptamd_image o; o.width = m.image_data.back().empty() ? 0 : im.width; o.height = m.image_data.back().empty() ? 0 : im.height;
        o.channels = 4; o.data = nullptr;
        m.images.push_back(o);
    }
    for (size_t i = 0; i < m.images.size(); ++i) m.images[i].data = m.image_data[i].empty() ? nullptr : m.image_data[i].data();
    for (const auto& g : m.groups) {
        const int32_t tex[5] = { g.base_color_tex, g.normal_tex, g.metallic_roughness_tex, g.emissive_tex, g.transmission_tex };
        for (int32_t t : tex) if (t >= (int32_t)m.images.size()) return fail(PTAMD_ERR_ARG, "material group references an image that was not supplied");
    }
    return PTAMD_OK;
}

// Geometry fields of a sphere's or box's row (kind, material and mesh stay as they are).
void fill_object_row(const HostObject& ho, DObject& d)
{
    std::memcpy(d.c, ho.pos, 12);
    if (ho.kind == KIND_SPHERE) { d.radius = ho.radius; return; }
    for (int k = 0; k < 3; ++k) d.half[k] = ho.size[k] * 0.5f;                     // Box.cpp:30
    d.eps = 1e-4f * std::max(std::max(d.half[0], d.half[1]), d.half[2]);          // Box.cpp:63
    Mat4 R = rotate_euler_xyz(Mat4::identity(), ho.rot);                          // Box.cpp:16-19
    for (int k = 0; k < 3; ++k) { d.ax0[k] = R.m[0][k]; d.ax1[k] = R.m[1][k]; d.ax2[k] = R.m[2][k]; }
    d.aligned = d.ax0[0] == 1.f && d.ax0[1] == 0.f && d.ax0[2] == 0.f && d.ax1[0] == 0.f && d.ax1[1] == 1.f && d.ax1[2] == 0.f
                && d.ax2[0] == 0.f && d.ax2[1] == 0.f && d.ax2[2] == 1.f;
}
// Instance matrices of a mesh (Mesh.cpp:47-54).
void fill_mesh_matrices(const HostObject& ho, DMesh& dm)
{
    Mat4 M = translate(Mat4::identity(), ho.pos);
    M = rotate_euler_xyz(M, ho.rot);
    M = scale(M, ho.size);
    Mat4 Minv = inverse(M);
    rows3x4(M, dm.m); rows3x4(Minv, dm.minv);
    bool identity = true;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) identity = identity && dm.m[4 * r + c] == (r == c ? 1.0f : 0.0f) && dm.minv[4 * r + c] == (r == c ? 1.0f : 0.0f);
    dm.flags = identity ? (dm.flags | 8u) : (dm.flags & ~8u);
}
// Marks the mesh instances that few of a scene's rays reach (DMesh::flags bit 4), for which the work-sharing walk holds its
// leaf passes back. Estimate: a random line through a convex body meets a convex body inside it with probability
// area(inner) / area(outer); an instance is "sparse" when the surface area of its world bounds is under a quarter of that
// of the bounds of the whole scene. Returns true when a flag changed.
struct Bounds3 {
    float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    void add(float x, float y, float z) { const float p[3] = { x, y, z }; for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    void add(const Bounds3& b) { if (b.lo[0] <= b.hi[0]) { add(b.lo[0], b.lo[1], b.lo[2]); add(b.hi[0], b.hi[1], b.hi[2]); } }
    double area() const
    {
        if (!(lo[0] <= hi[0])) return 0.0;
        const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
};
void mark_sparse_meshes(const std::vector<HostObject>& objects, const std::vector<DObject>& rows, std::vector<DMesh>& meshes, std::vector<char>* changed)
{
    Bounds3 all; std::vector<double> area(meshes.size(), 0.0);
    for (size_t i = 0; i < objects.size() && i < rows.size(); ++i) {
        const HostObject& ho = objects[i];
        Bounds3 b;
        if (ho.kind == KIND_MESH) {
            const DMesh& dm = meshes[(size_t)rows[i].mesh];
            const float* lo = ho.mesh->bmin; const float* hi = ho.mesh->bmax;
            if (!(lo[0] <= hi[0])) continue;
            for (int c = 0; c < 8; ++c) {
                const float x = (c & 1) ? hi[0] : lo[0], y = (c & 2) ? hi[1] : lo[1], z = (c & 4) ? hi[2] : lo[2];
                b.add(dm.m[0] * x + dm.m[1] * y + dm.m[2] * z + dm.m[3], dm.m[4] * x + dm.m[5] * y + dm.m[6] * z + dm.m[7],
                      dm.m[8] * x + dm.m[9] * y + dm.m[10] * z + dm.m[11]);
            }
            area[(size_t)rows[i].mesh] = b.area();
        } else {
            // a sphere's radius, or half a box's diagonal (whatever its rotation), around the centre
            const float r = ho.kind == KIND_SPHERE ? std::fabs(ho.radius)
                                                   : 0.5f * std::sqrt(ho.size[0] * ho.size[0] + ho.size[1] * ho.size[1] + ho.size[2] * ho.size[2]);
            b.add(ho.pos[0] - r, ho.pos[1] - r, ho.pos[2] - r); b.add(ho.pos[0] + r, ho.pos[1] + r, ho.pos[2] + r);
        }
        all.add(b);
    }
    const double sceneArea = all.area();
    const char* force = std::getenv("PTAMD_SPARSE_MESHES"); // 0 | 1: no instance / every instance
    for (size_t m = 0; m < meshes.size(); ++m) {
        bool sparse = std::isfinite(sceneArea) && sceneArea > 0.0 && area[m] < 0.25 * sceneArea;
        if (force) sparse = force[0] == '1';
        const uint32_t flags = sparse ? (meshes[m].flags | 16u) : (meshes[m].flags & ~16u);
        if (changed && flags != meshes[m].flags) (*changed)[m] = 1;
        meshes[m].flags = flags;
    }
}
bool group_samples_images(const ptamd_pbr& g)
{
    return g.base_color_tex >= 0 || g.metallic_roughness_tex >= 0 || g.emissive_tex >= 0 || g.transmission_tex >= 0 || g.normal_tex >= 0;
}
DPbr make_dpbr(const ptamd_pbr& g, uint32_t image_base, int32_t material_row)
{
    DPbr dg; std::memset(&dg, 0, sizeof dg);
    auto img = [&](int32_t t) { return t >= 0 ? (int32_t)(image_base + (uint32_t)t) : -1; };
    dg.base_color_tex = img(g.base_color_tex); dg.normal_tex = img(g.normal_tex);
    dg.metallic_roughness_tex = img(g.metallic_roughness_tex); dg.emissive_tex = img(g.emissive_tex);
    dg.transmission_tex = img(g.transmission_tex);
    dg.alpha_mode = g.alpha_mode; dg.alpha_cutoff = g.alpha_cutoff; dg.normal_scale = g.normal_scale;
    std::memcpy(dg.base_color_factor, g.base_color_factor, 16);
    std::memcpy(dg.emissive_factor, g.emissive_factor, 12);
    dg.metallic_factor = g.metallic_factor; dg.roughness_factor = g.roughness_factor;
    dg.transmission_factor = g.transmission_factor; dg.ior = g.ior;
    dg.material = material_row;
    return dg;
}
void mark_edit(ptamd_scene* s, int object, uint8_t what)
{
    if (s->edits.size() < s->objects.size()) s->edits.resize(s->objects.size(), 0);
    s->edits[(size_t)object] |= what;
    s->committed = false;
}

DCamera make_camera(const ptamd_camera& c, int w, int h)
{
    // Camera::CalculateMatrices (Camera.cpp:22-35)
    Mat4 R = rotate_euler_xyz(Mat4::identity(), c.rotation_deg);
    Mat4 worldFromCam = mul(translate(Mat4::identity(), c.position), R);
    Mat4 proj = perspective(deg2rad(c.fov_deg), (float)w / (float)h, c.near_plane, c.far_plane);
    Mat4 invProj = inverse(proj);
    DCamera d; std::memset(&d, 0, sizeof d);
    std::memcpy(d.inv_proj, invProj.m, 64);
    std::memcpy(d.inv_view, worldFromCam.m, 64);
    d.width = w; d.height = h;
    return d;
}

} // namespace

extern "C" {

const char* ptamd_last_error(void) { return g_error.c_str(); }

void ptamd_material_default(ptamd_material* m)
{
    if (!m) return;
    m->albedo[0] = m->albedo[1] = m->albedo[2] = 1.0f; m->roughness = 1.0f; m->metallic = 0.0f;
    m->emission_colour[0] = m->emission_colour[1] = m->emission_colour[2] = 1.0f; m->emission_strength = 0.0f;
    m->ior = 1.5f; m->transmission = 0.0f;
}
void ptamd_pbr_default(ptamd_pbr* p)
{
    if (!p) return;
    p->base_color_tex = p->normal_tex = p->metallic_roughness_tex = p->emissive_tex = p->transmission_tex = -1;
    for (int k = 0; k < 4; ++k) p->base_color_factor[k] = 1.0f;
    p->metallic_factor = 0.0f; p->roughness_factor = 1.0f; p->alpha_mode = 0; p->alpha_cutoff = 0.5f; p->normal_scale = 1.0f;
    p->emissive_factor[0] = p->emissive_factor[1] = p->emissive_factor[2] = 0.0f;
    p->transmission_factor = 0.0f; p->ior = 1.5f;
}

int ptamd_init(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(PTAMD_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(PTAMD_ERR_ARG, "device index out of range");
    if (g_initialised && device != g_device) return fail(PTAMD_ERR_STATE, "already initialised on another device (one process drives one GPU)");
    PT_HIP(hipSetDevice(device));
    if (!g_own_stream) { PT_HIP(hipStreamCreateWithFlags(&g_own_stream, hipStreamNonBlocking)); g_stream = g_own_stream; }
    // the statically allocated counters of ptamd_render_counted_noalloc: their address on this device, resolved once here
    PT_HIP(scratch_counters(device, &g_static_counters));
    g_initialised = true; g_device = device;
    return PTAMD_OK;
}
int ptamd_set_stream(void* s)
{
    if (int rc = need_init()) return rc;
    g_stream = reinterpret_cast<hipStream_t>(s);
    return PTAMD_OK;
}
int ptamd_reset_stream(void)
{
    if (int rc = need_init()) return rc;
    g_stream = g_own_stream;
    return PTAMD_OK;
}
int ptamd_synchronize(void)
{
    if (int rc = need_init()) return rc;
    PT_HIP(hipStreamSynchronize(g_stream));
    return PTAMD_OK;
}

// ------------------------------------------------------------------------------- scene
int ptamd_scene_create(ptamd_scene** out)
{
    if (!out) return fail(PTAMD_ERR_ARG, "null output pointer");
    *out = new ptamd_scene();
    std::memset(&(*out)->dscene, 0, sizeof(DScene));
    return PTAMD_OK;
}
int ptamd_scene_destroy(ptamd_scene* s)
{
    if (!s) return PTAMD_OK;
    s->release(); s->d_sky.release();
    if (s->d_counters) { if (g_last_counters == s->d_counters) g_last_counters = nullptr; (void)hipFree(s->d_counters); }
    delete s;
    return PTAMD_OK;
}
int ptamd_scene_clear(ptamd_scene* s)
{
    if (int rc = check_scene(s)) return rc;
    s->objects.clear(); s->committed = false; s->structure_dirty = true;
    return PTAMD_OK;
}
int ptamd_scene_object_count(const ptamd_scene* s) { return s ? (int)s->objects.size() : fail(PTAMD_ERR_ARG, "null scene handle"); }

int ptamd_scene_add_sphere(ptamd_scene* s, const float pos[3], float radius, const ptamd_material* mat)
{
    if (int rc = check_scene(s)) return rc;
    if (!pos) return fail(PTAMD_ERR_ARG, "null position");
    HostObject o; o.kind = KIND_SPHERE; std::memcpy(o.pos, pos, 12); o.radius = radius;
    if (mat) o.mat = *mat; else ptamd_material_default(&o.mat);
    s->objects.push_back(o); s->committed = false; s->structure_dirty = true;
    return (int)s->objects.size() - 1;
}
int ptamd_scene_add_box(ptamd_scene* s, const float pos[3], const float rot[3], const float size[3], const ptamd_material* mat)
{
    if (int rc = check_scene(s)) return rc;
    if (!pos || !size) return fail(PTAMD_ERR_ARG, "null position or size");
    HostObject o; o.kind = KIND_BOX; std::memcpy(o.pos, pos, 12); std::memcpy(o.size, size, 12);
    if (rot) std::memcpy(o.rot, rot, 12);
    if (mat) o.mat = *mat; else ptamd_material_default(&o.mat);
    s->objects.push_back(o); s->committed = false; s->structure_dirty = true;
    return (int)s->objects.size() - 1;
}
static int add_mesh_object(ptamd_scene* s, std::shared_ptr<HostMesh> hm, const float pos[3], const float rot[3], const float scl[3])
{
    HostObject o; o.kind = KIND_MESH; o.mesh = std::move(hm);
    ptamd_material_default(&o.mat);
    if (pos) std::memcpy(o.pos, pos, 12);
    if (rot) std::memcpy(o.rot, rot, 12);
    if (scl) std::memcpy(o.size, scl, 12);
    s->objects.push_back(o); s->committed = false; s->structure_dirty = true;
    return (int)s->objects.size() - 1;
}
int ptamd_scene_add_mesh(ptamd_scene* s, const ptamd_mesh_desc* mesh, const float pos[3], const float rot[3], const float scl[3])
{
    if (int rc = check_scene(s)) return rc;
    auto hm = std::make_shared<HostMesh>();
    if (int rc = copy_mesh_desc(mesh, hm->model)) return rc;
    return add_mesh_object(s, hm, pos, rot, scl);
}
int ptamd_scene_add_mesh_file(ptamd_scene* s, const char* path, const float pos[3], const float rot[3], const float scl[3])
{
    if (int rc = check_scene(s)) return rc;
    if (!path) return fail(PTAMD_ERR_ARG, "null path");
    auto hm = std::make_shared<HostMesh>();
    try {
        load_gltf(path, hm->model);
    } catch (const std::exception& e) {
        return fail(PTAMD_ERR_IO, e.what());
    }
    if (hm->model.num_faces() == 0) return fail(PTAMD_ERR_IO, "Mesh: no faces in model");
    if (hm->model.num_faces() >= (1u << 25)) return fail(PTAMD_ERR_LIMIT, "mesh has too many faces (limit 2^25 - 1)");
    return add_mesh_object(s, hm, pos, rot, scl);
}
int ptamd_scene_set_transform(ptamd_scene* s, int object, const float pos[3], const float rot[3], const float size[3])
{
    if (int rc = check_object(s, object)) return rc;
    HostObject& o = s->objects[(size_t)object];
    if (pos) std::memcpy(o.pos, pos, 12);
    if (rot) std::memcpy(o.rot, rot, 12);
    if (size) std::memcpy(o.size, size, 12);
    mark_edit(s, object, 1);
    return PTAMD_OK;
}
int ptamd_scene_set_radius(ptamd_scene* s, int object, float radius)
{
    if (int rc = check_object(s, object)) return rc;
    if (s->objects[(size_t)object].kind != KIND_SPHERE) return fail(PTAMD_ERR_ARG, "object is not a sphere");
    s->objects[(size_t)object].radius = radius; mark_edit(s, object, 1);
    return PTAMD_OK;
}
int ptamd_scene_set_material(ptamd_scene* s, int object, const ptamd_material* mat)
{
    if (int rc = check_object(s, object)) return rc;
    if (!mat) return fail(PTAMD_ERR_ARG, "null material");
    if (s->objects[(size_t)object].kind == KIND_MESH) return fail(PTAMD_ERR_ARG, "mesh materials are edited per group (ptamd_scene_set_mesh_group)");
    s->objects[(size_t)object].mat = *mat; mark_edit(s, object, 2);
    return PTAMD_OK;
}
int ptamd_scene_set_mesh_group(ptamd_scene* s, int object, uint32_t group, const ptamd_pbr* pbr)
{
    if (int rc = check_object(s, object)) return rc;
    HostObject& o = s->objects[(size_t)object];
    if (o.kind != KIND_MESH) return fail(PTAMD_ERR_ARG, "object is not a mesh");
    if (!pbr || group >= o.mesh->model.groups.size()) return fail(PTAMD_ERR_ARG, "bad material group");
    const int32_t tex[5] = { pbr->base_color_tex, pbr->normal_tex, pbr->metallic_roughness_tex, pbr->emissive_tex, pbr->transmission_tex };
    for (int32_t t : tex) if (t >= (int32_t)o.mesh->model.images.size()) return fail(PTAMD_ERR_ARG, "material group references an image that was not supplied");
    const ptamd_pbr& old = o.mesh->model.groups[group];
    // a group that starts or stops sampling images, or masking, changes per-face records and mesh flags: full commit
    if (group_samples_images(old) != group_samples_images(*pbr) || (old.alpha_mode == 1) != (pbr->alpha_mode == 1)) s->structure_dirty = true;
    o.mesh->model.groups[group] = *pbr; mark_edit(s, object, 4);
    return PTAMD_OK;
}
int ptamd_scene_set_background(ptamd_scene* s, const float rgb[3])
{
    if (int rc = check_scene(s)) return rc;
    if (!rgb) return fail(PTAMD_ERR_ARG, "null colour");
    std::memcpy(s->background, rgb, 12);
    std::memcpy(s->dscene.background, rgb, 12); // passed by value with every launch: no commit needed
    return PTAMD_OK;
}

// Both sky entry points end here: `tex` holds the six faces as float radiance, four floats per texel (rgb + one unused:
// a texel is one 16-byte read).
static int upload_sky(ptamd_scene* s, int face_size, const std::vector<float>& tex)
{
    if (int rc = s->d_sky.upload(tex)) { // the old map is gone and the new one did not arrive: fall back to the constant background
        s->d_sky.release(); s->sky_size = 0;
        s->dscene.sky = nullptr; s->dscene.sky_size = 0;
        return rc;
    }
    s->sky_size = face_size;
    s->dscene.sky = reinterpret_cast<const float4*>(s->d_sky.ptr); s->dscene.sky_size = face_size;
    return PTAMD_OK;
}
static int clear_sky(ptamd_scene* s)
{
    s->d_sky.release(); s->sky_size = 0;
    s->dscene.sky = nullptr; s->dscene.sky_size = 0;
    return PTAMD_OK;
}
static int check_sky_faces(int face_size, const void* const* faces)
{
    if (face_size > 4096) return fail(PTAMD_ERR_LIMIT, "skybox faces are limited to 4096 x 4096");
    for (int f = 0; f < 6; ++f) if (!faces[f]) return fail(PTAMD_ERR_ARG, "null skybox face");
    return PTAMD_OK;
}

int ptamd_scene_set_skybox(ptamd_scene* s, int face_size, const uint8_t* const faces[6], float intensity)
{
    if (int rc = check_scene(s)) return rc;
    if (!faces || face_size <= 0) return clear_sky(s);
    if (int rc = need_init()) return rc;
    if (int rc = check_sky_faces(face_size, reinterpret_cast<const void* const*>(faces))) return rc;
    const size_t n = (size_t)face_size * face_size;
    try {
        std::vector<float> tex(6 * n * 4, 0.0f);
        for (int f = 0; f < 6; ++f)
            for (size_t p = 0; p < n; ++p)
                for (int k = 0; k < 3; ++k) tex[(f * n + p) * 4 + k] = ((float)faces[f][4 * p + k] / 255.0f) * intensity; // texel / 255 * intensity, rounded like that
        return upload_sky(s, face_size, tex);
    } catch (const std::exception& e) { return fail(PTAMD_ERR_LIMIT, e.what()); }
}

int ptamd_scene_set_skybox_f32(ptamd_scene* s, int face_size, const float* const faces[6])
{
    if (int rc = check_scene(s)) return rc;
    if (!faces || face_size <= 0) return clear_sky(s);
    if (int rc = need_init()) return rc;
    if (int rc = check_sky_faces(face_size, reinterpret_cast<const void* const*>(faces))) return rc;
    const size_t n = (size_t)face_size * face_size;
    try {
        std::vector<float> tex(6 * n * 4, 0.0f);
        for (int f = 0; f < 6; ++f)
            for (size_t p = 0; p < n; ++p) std::memcpy(&tex[(f * n + p) * 4], faces[f] + 3 * p, 12);
        return upload_sky(s, face_size, tex);
    } catch (const std::exception& e) { return fail(PTAMD_ERR_LIMIT, e.what()); }
}


// ------------------------------------------------------------------------------- LDS nodes
// Appends one LDS-format node (see kLdsNodeBytes in pt_types.h) per DNode.
static void append_lds_nodes(const std::vector<DNode>& nodes, std::vector<float>& out)
{
    for (const DNode& n : nodes) {
        const float lmin[3] = { n.lmin[0], n.lmin[1], n.lmin[2] }, lmax[3] = { n.lmax0, n.lmax12[0], n.lmax12[1] };
        const float rmin[3] = { n.rmin01[0], n.rmin01[1], n.rmin2 }, rmax[3] = { n.rmax[0], n.rmax[1], n.rmax[2] };
        for (int k = 0; k < 3; ++k) {
            const float fwd[4] = { lmin[k], rmin[k], lmax[k], rmax[k] }, back[4] = { lmax[k], rmax[k], lmin[k], rmin[k] };
            out.insert(out.end(), fwd, fwd + 4);
            out.insert(out.end(), back, back + 4);
        }
        auto ref = [](int32_t r) { int32_t v = r >= 0 ? r * (int32_t)kLdsNodeBytes : r; float f; std::memcpy(&f, &v, 4); return f; };
        out.push_back(ref(n.left)); out.push_back(ref(n.right)); out.push_back(0.0f); out.push_back(0.0f);
    }
}

// DScene::lead_aligned: how many objects at the head of the list are unrotated boxes (fewer than two count as none).
static int32_t leading_aligned_boxes(const std::vector<DObject>& objects)
{
    size_t n = 0;
    while (n < objects.size() && objects[n].kind == KIND_BOX && objects[n].aligned) ++n;
    return n >= 2 ? (int32_t)n : 0;
}

// DScene::sphere_run: how many spheres follow the leading boxes directly (fewer than two count as none).
static int32_t sphere_run_after(const std::vector<DObject>& objects, int32_t lead)
{
    size_t n = (size_t)lead;
    while (n < objects.size() && objects[n].kind == KIND_SPHERE) ++n;
    return n - (size_t)lead >= 2 ? (int32_t)(n - (size_t)lead) : 0;
}

// Appends one DSignNode per DNode (the upload format of the two-child trees, pt_types.h).
static void append_sign_nodes(const std::vector<DNode>& nodes, std::vector<DSignNode>& out)
{
    for (const DNode& n : nodes) {
        const float lmin[3] = { n.lmin[0], n.lmin[1], n.lmin[2] }, lmax[3] = { n.lmax0, n.lmax12[0], n.lmax12[1] };
        const float rmin[3] = { n.rmin01[0], n.rmin01[1], n.rmin2 }, rmax[3] = { n.rmax[0], n.rmax[1], n.rmax[2] };
        DSignNode d; std::memset(&d, 0, sizeof d);
        for (int k = 0; k < 3; ++k) {
            const float both[8] = { lmin[k], rmin[k], lmax[k], rmax[k], lmax[k], rmax[k], lmin[k], rmin[k] };
            std::memcpy(d.axis[k], both, sizeof both);
        }
        d.left = n.left; d.right = n.right;
        out.push_back(d);
    }
}

static int commit_scene(ptamd_scene* s);
int ptamd_scene_commit(ptamd_scene* s)
{
    if (int rc = check_scene(s)) return rc;
    if (int rc = need_init()) return rc;
    // the tree builders and the flattening allocate and start threads: nothing they throw may cross the C boundary
    try {
        return commit_scene(s);
    } catch (const std::bad_alloc&) {
        s->committed = false;
        return fail(PTAMD_ERR_LIMIT, "out of host memory while committing the scene");
    } catch (const std::exception& e) {
        s->committed = false;
        return fail(PTAMD_ERR_STATE, e.what());
    }
}
static int commit_scene(ptamd_scene* s)
{
    const auto t_begin = std::chrono::steady_clock::now();
    auto seconds_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    if (!s->structure_dirty && s->h_objects.size() == s->objects.size()) {
        // Only rows of the small tables changed since the last full commit: rewrite those, on the library stream, behind
        // whatever launch still reads the old values.
        s->edits.resize(s->objects.size(), 0);
        for (size_t i = 0; i < s->objects.size(); ++i) {
            const uint8_t what = s->edits[i];
            if (!what) continue;
            // A moved object changes which walks are long: the schedule is measured again at the next eligible launch (an edited
            // material leaves the walks as they are) -- unless the last measurement is under 32 launches old: a session that
            // moves an object before every frame measures once in 32 frames, not ten probe launches per frame.
            if ((what & 1) && s->walk_cut_uses >= 32u) s->walk_cut = -1;
            const HostObject& ho = s->objects[i];
            DObject& row = s->h_objects[i];
            if (ho.kind == KIND_MESH) {
                DMesh& dm = s->h_meshes[(size_t)row.mesh];
                if (what & 1) {
                    fill_mesh_matrices(ho, dm);
                    PT_HIP(hipMemcpyAsync(s->d_meshes.ptr + row.mesh, &dm, sizeof(DMesh), hipMemcpyHostToDevice, g_stream)); ++s->row_updates;
                }
                if (what & 4) {
                    const std::vector<ptamd_pbr>& gs = ho.mesh->model.groups;
                    for (size_t g = 0; g < gs.size(); ++g) {
                        DPbr& dg = s->h_groups[dm.group_base + g];
                        dg = make_dpbr(gs[g], dm.image_base, dg.material);
                        const DMaterial m = to_dmaterial(group_material(gs[g]));
                        PT_HIP(hipMemcpyAsync(s->d_groups.ptr + dm.group_base + g, &dg, sizeof(DPbr), hipMemcpyHostToDevice, g_stream));
                        PT_HIP(hipMemcpyAsync(s->d_materials.ptr + dg.material, &m, sizeof(DMaterial), hipMemcpyHostToDevice, g_stream));
                        s->row_updates += 2;
                    }
                }
            } else {
                if (what & 1) {
                    fill_object_row(ho, row);
                    PT_HIP(hipMemcpyAsync(s->d_objects.ptr + i, &row, sizeof(DObject), hipMemcpyHostToDevice, g_stream)); ++s->row_updates;
                }
                if (what & 2) {
                    const DMaterial m = to_dmaterial(ho.mat);
                    PT_HIP(hipMemcpyAsync(s->d_materials.ptr + row.material, &m, sizeof(DMaterial), hipMemcpyHostToDevice, g_stream)); ++s->row_updates;
                }
            }
            s->edits[i] = 0;
        }
        {
            std::vector<char> changed(s->h_meshes.size(), 0);
            mark_sparse_meshes(s->objects, s->h_objects, s->h_meshes, &changed);
            for (size_t m = 0; m < changed.size(); ++m)
                if (changed[m]) { PT_HIP(hipMemcpyAsync(s->d_meshes.ptr + m, &s->h_meshes[m], sizeof(DMesh), hipMemcpyHostToDevice, g_stream)); ++s->row_updates; }
        }
        PT_HIP(hipStreamSynchronize(g_stream)); // the copies read host rows that the next edit may change
        s->dscene.lead_aligned = leading_aligned_boxes(s->h_objects); // a rotation edit can end or extend the run
        s->dscene.sphere_run = sphere_run_after(s->h_objects, s->dscene.lead_aligned);
        s->committed = true;
        s->last_commit_seconds = seconds_since(t_begin);
        return PTAMD_OK;
    }
    s->committed = false; // until every buffer of the new state has arrived
    s->walk_cut = -1;     // new geometry: the walk schedule is measured again at the next eligible launch
    double bvhSeconds = 0.0;
    const int maxDepth = max_stack_depth();
    std::vector<DObject> objects; std::vector<DMesh> meshes; std::vector<DMaterial> materials;
    std::vector<DSignNode> nodes; std::vector<DTri> tris; std::vector<DShade> shade; std::vector<DPbr> groups;
    std::vector<DImage> images; std::vector<uint8_t> texels;
    ptamd_material defMat; ptamd_material_default(&defMat);
    materials.push_back(to_dmaterial(defMat)); // slot 0: Material{} (faces without a material group, Mesh.cpp:242-246)
    std::vector<std::pair<const HostMesh*, uint32_t>> uploaded; // geometry shared by several instances is stored once
    struct MeshSlots { uint32_t node_base, tri_base, group_base, image_base; uint32_t flags; };
    std::vector<float> ldsNodes; // LDS-format nodes of all meshes while they can still fit, kLdsNodeBytes each
    std::vector<MeshSlots> slots;
    std::vector<DWideNode> wideNodes;

    // Trees first: whether the scene's geometry is to live in LDS decides which node format (and with it which triangle
    // order) the meshes are flattened in.
    size_t ldsGeomBytes = 0; int stackLevels = 1;
    {
        std::vector<const HostMesh*> seen;
        for (const HostObject& ho : s->objects) {
            if (ho.kind != KIND_MESH) continue;
            HostMesh& hm = *ho.mesh;
            if (!hm.bvh_valid) {
                const Model& md = hm.model;
                const auto t0 = std::chrono::steady_clock::now();
                hm.bvh = build_bvh(md.positions.data(), md.group.data(), md.num_faces(), maxDepth);
                for (int k = 0; k < 3; ++k) { hm.bmin[k] = 3.0e38f; hm.bmax[k] = -3.0e38f; }
                for (size_t i = 0; i < md.positions.size(); ++i) { // (a NaN coordinate leaves the bounds alone)
                    const int k = (int)(i % 3);
                    hm.bmin[k] = std::min(hm.bmin[k], md.positions[i]); hm.bmax[k] = std::max(hm.bmax[k], md.positions[i]);
                }
                if (hm.bvh.depth > maxDepth) return fail(PTAMD_ERR_LIMIT, "BVH deeper than the traversal stack");
                build_wide(hm.bvh);
                hm.bvh_valid = true;
                bvhSeconds += seconds_since(t0);
            }
            stackLevels = std::max(stackLevels, hm.bvh.depth);
            if (std::find(seen.begin(), seen.end(), &hm) == seen.end()) {
                seen.push_back(&hm);
                ldsGeomBytes += hm.bvh.nodes.size() * kLdsNodeBytes + (size_t)hm.model.num_faces() * sizeof(DTri);
            }
        }
    }
    // small scenes are traversed from an LDS copy of their nodes (in LDS format) and hot triangles, scenes without
    // meshes included; PTAMD_NO_LDS_GEOM (read at commit) forces the global-memory kernel for A/B tests
    const bool ldsResident = (size_t)std::min(stackLevels, maxDepth) * 1024 + kLdsFixedBytes + ldsGeomBytes <= kLdsBlockBytes && !std::getenv("PTAMD_NO_LDS_GEOM");
    // Large meshes of scenes that are read from global memory are walked through eight-wide quantised nodes (a stack
    // entry of that walk takes two rows). Measured on one MI355X: a 1 M-triangle mesh, whose 64-byte binary nodes miss
    // L2, renders 8 % faster through them (38 % fewer bytes from the memory side, 45 % more vector instructions); a
    // 4 k-triangle mesh, which stays in L2, 20 % slower. PTAMD_NO_WIDE keeps the binary nodes and
    // PTAMD_WIDE_MIN_FACES moves the threshold, for A/B tests.
    const bool wideAllowed = !ldsResident && !std::getenv("PTAMD_NO_WIDE");
    const uint32_t wideMinFaces = std::getenv("PTAMD_WIDE_MIN_FACES") ? (uint32_t)std::atoi(std::getenv("PTAMD_WIDE_MIN_FACES")) : 200000u;
    auto uses_wide = [&](const HostMesh& hm) {
        return wideAllowed && !hm.bvh.wide.empty() && 2 * hm.bvh.wide_depth <= maxDepth && hm.model.num_faces() >= wideMinFaces;
    };

    for (const HostObject& ho : s->objects) {
        DObject d; std::memset(&d, 0, sizeof d);
        d.kind = ho.kind; d.mesh = -1; d.material = 0;
        if (ho.kind == KIND_SPHERE || ho.kind == KIND_BOX) {
            fill_object_row(ho, d);
            d.material = (int32_t)materials.size(); materials.push_back(to_dmaterial(ho.mat));
        } else {
            HostMesh& hm = *ho.mesh;
            const Model& md = hm.model;
            const bool wide = uses_wide(hm);
            const std::vector<uint32_t>& order = wide ? hm.bvh.wide_order : hm.bvh.order;
            size_t slot = 0;
            for (; slot < uploaded.size(); ++slot) if (uploaded[slot].first == &hm) break;
            if (slot == uploaded.size()) {
                MeshSlots ms;
                ms.tri_base = (uint32_t)tris.size();
                ms.group_base = (uint32_t)groups.size(); ms.image_base = (uint32_t)images.size(); ms.flags = wide ? 4u : 0u;
                if (wide) {
                    ms.node_base = (uint32_t)wideNodes.size();
                    wideNodes.insert(wideNodes.end(), hm.bvh.wide.begin(), hm.bvh.wide.end());
                } else {
                    ms.node_base = (uint32_t)nodes.size();
                    append_sign_nodes(hm.bvh.nodes, nodes);
                    if (ldsResident) append_lds_nodes(hm.bvh.nodes, ldsNodes);
                }
                std::vector<int32_t> faceMaterial; // per group: constant material slot, or -1 when the group samples images
                for (const ptamd_pbr& g : md.groups) {
                    const DPbr dg = make_dpbr(g, ms.image_base, (int32_t)materials.size());
                    materials.push_back(to_dmaterial(group_material(g)));
                    if (g.alpha_mode == 1) ms.flags |= 1u;
                    if (group_samples_images(g)) ms.flags |= 2u;
                    groups.push_back(dg);
                    faceMaterial.push_back(group_samples_images(g) ? -1 : dg.material);
                }
                for (uint32_t i = 0; i < md.num_faces(); ++i) {
                    uint32_t f = order[i];
                    // vertices in the order of the hot record: a quad's first triangle starts at the corner opposite the
                    // shared edge (a cyclic rotation: same triangle, same orientation)
                    const int rot = hm.bvh.rotation[f];
                    float p[9], nrm[9], uv[6];
                    for (int k = 0; k < 3; ++k) {
                        const int src = (k + rot) % 3;
                        std::memcpy(p + 3 * k, &md.positions[9 * (size_t)f + 3 * src], 12);
                        std::memcpy(nrm + 3 * k, &md.normals[9 * (size_t)f + 3 * src], 12);
                        std::memcpy(uv + 2 * k, &md.texcoords[6 * (size_t)f + 2 * src], 8);
                    }
                    DTri t; std::memset(&t, 0, sizeof t);
                    {
                        const double v0[3] = { p[0], p[1], p[2] };
                        const double e1[3] = { (double)(p[3] - p[0]), (double)(p[4] - p[1]), (double)(p[5] - p[2]) }; // float differences, as Mesh.cpp:421-422 forms them
                        const double e2[3] = { (double)(p[6] - p[0]), (double)(p[7] - p[1]), (double)(p[8] - p[2]) };
                        const double n[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
                        const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
                        if (nn > 0.0) {
                            const double m1[3] = { (e2[1] * n[2] - e2[2] * n[1]) / nn, (e2[2] * n[0] - e2[0] * n[2]) / nn, (e2[0] * n[1] - e2[1] * n[0]) / nn };
                            const double m2[3] = { (n[1] * e1[2] - n[2] * e1[1]) / nn, (n[2] * e1[0] - n[0] * e1[2]) / nn, (n[0] * e1[1] - n[1] * e1[0]) / nn };
                            for (int k = 0; k < 3; ++k) { t.n[k] = (float)n[k]; t.m1[k] = (float)m1[k]; t.m2[k] = (float)m2[k]; }
                            t.nv0 = (float)(n[0] * v0[0] + n[1] * v0[1] + n[2] * v0[2]);
                            t.d1 = (float)-(m1[0] * v0[0] + m1[1] * v0[1] + m1[2] * v0[2]);
                            t.d2 = (float)-(m2[0] * v0[0] + m2[1] * v0[1] + m2[2] * v0[2]);
                        } // a degenerate triangle keeps n = 0: the determinant test rejects it
                    }
                    tris.push_back(t);
                    DShade sh; std::memset(&sh, 0, sizeof sh);
                    std::memcpy(sh.pa, p, 36);
                    std::memcpy(sh.na, nrm, 36);
                    std::memcpy(sh.ta, uv, 24);
                    sh.group = md.group[f];
                    sh.material = md.group[f] >= 0 ? faceMaterial[(size_t)md.group[f]] : 0;
                    shade.push_back(sh);
                }
                for (size_t i = 0; i < md.images.size(); ++i) {
                    DImage di; di.width = md.images[i].width; di.height = md.images[i].height; di.offset = (uint32_t)texels.size(); di.pad = 0;
                    texels.insert(texels.end(), md.image_data[i].begin(), md.image_data[i].end());
                    images.push_back(di);
                }
                uploaded.emplace_back(&hm, (uint32_t)slots.size());
                slots.push_back(ms);
            }
            const MeshSlots& ms = slots[uploaded[slot].second];
            DMesh dm; std::memset(&dm, 0, sizeof dm);
            dm.node_base = ms.node_base; dm.tri_base = ms.tri_base; dm.group_base = ms.group_base; dm.image_base = ms.image_base;
            dm.num_tris = md.num_faces(); dm.root = wide ? 0 : hm.bvh.root; dm.flags = ms.flags;
            fill_mesh_matrices(ho, dm);
            // the box of the mesh's positions in its own frame, a little larger (pt_device.h: meets_tail_mesh)
            for (int k = 0; k < 3; ++k) {
                const float lo = hm.bmin[k], hi = hm.bmax[k];
                d.c[k] = 0.5f * lo + 0.5f * hi;
                d.half[k] = lo <= hi ? (0.5f * hi - 0.5f * lo) * 1.0002f + 1e-4f * std::max(std::fabs(lo), std::fabs(hi)) + 1e-30f : -1.0f;
            }
            d.mesh = (int32_t)meshes.size();
            meshes.push_back(dm);
        }
        objects.push_back(d);
    }

    // a hit names its face by its index in the scene-wide arrays with two flag bits above it (pt_device.h: kFrontBit)
    if (tris.size() >= (size_t(1) << 30)) return fail(PTAMD_ERR_LIMIT, "scene has too many triangles (limit 2^30 - 1 over all distinct meshes)");
    std::memset(&s->dscene, 0, sizeof(DScene)); // nothing may point at the buffers released next
    s->release();
    mark_sparse_meshes(s->objects, objects, meshes, nullptr);
    if (int rc = s->d_objects.upload(objects)) return rc;
    if (int rc = s->d_meshes.upload(meshes)) return rc;
    if (int rc = s->d_materials.upload(materials)) return rc;
    if (int rc = s->d_nodes.upload(nodes)) return rc;
    if (int rc = s->d_wide.upload(wideNodes)) return rc;
    if (ldsResident && !ldsNodes.empty()) { if (int rc = s->d_lds_nodes.upload(ldsNodes)) return rc; }
    if (int rc = s->d_tris.upload(tris)) return rc;
    if (int rc = s->d_shade.upload(shade)) return rc;
    if (int rc = s->d_groups.upload(groups)) return rc;
    if (int rc = s->d_images.upload(images)) return rc;
    if (int rc = s->d_texels.upload(texels)) return rc;
    DScene& ds = s->dscene;
    ds.objects = s->d_objects.ptr; ds.num_objects = (int32_t)objects.size();
    ds.meshes = s->d_meshes.ptr; ds.materials = s->d_materials.ptr;
    ds.nodes = reinterpret_cast<const float4*>(s->d_nodes.ptr);
    ds.wide_nodes = reinterpret_cast<const float4*>(s->d_wide.ptr);
    ds.tris = reinterpret_cast<const float4*>(s->d_tris.ptr);
    ds.lds_resident = ldsResident ? 1u : 0u;
    ds.lds_nodes = ldsResident ? reinterpret_cast<const float4*>(s->d_lds_nodes.ptr) : nullptr; ds.lds_node_f4 = ldsResident ? (uint32_t)(ldsNodes.size() / 4) : 0u;
    ds.shade = reinterpret_cast<const float4*>(s->d_shade.ptr);
    ds.groups = s->d_groups.ptr; ds.images = s->d_images.ptr; ds.texels = s->d_texels.ptr;
    std::memcpy(ds.background, s->background, 12);
    ds.sky = s->sky_size ? reinterpret_cast<const float4*>(s->d_sky.ptr) : nullptr; ds.sky_size = s->sky_size;
    ds.tail_mesh = (!s->objects.empty() && s->objects.back().kind == KIND_MESH) ? 1u : 0u; ds.num_tris = (uint32_t)tris.size();
    if (ds.tail_mesh && meshes.size() == 1) ds.tail_mesh |= 2u; // ... and it is the scene's only mesh
    ds.max_stack = 1; // rows of the per-lane traversal stack the deepest walk of this scene needs
    for (const HostObject& ho : s->objects)
        if (ho.kind == KIND_MESH) ds.max_stack = std::max(ds.max_stack, uses_wide(*ho.mesh) ? 2 * ho.mesh->bvh.wide_depth : ho.mesh->bvh.depth);
    ds.lead_aligned = leading_aligned_boxes(objects);
    ds.sphere_run = sphere_run_after(objects, ds.lead_aligned);
    s->h_objects = std::move(objects); s->h_meshes = std::move(meshes); s->h_groups = std::move(groups);
    s->edits.assign(s->objects.size(), 0);
    s->structure_dirty = false;
    ++s->geometry_uploads;
    s->committed = true;
    s->last_bvh_seconds = bvhSeconds;
    s->last_commit_seconds = seconds_since(t_begin);
    return PTAMD_OK;
}
int ptamd_scene_get_info(const ptamd_scene* s, ptamd_scene_info* out)
{
    if (!s || !out) return fail(PTAMD_ERR_ARG, "null argument");
    std::memset(out, 0, sizeof *out);
    out->objects = (uint32_t)s->objects.size(); out->meshes = (uint32_t)s->h_meshes.size();
    out->triangles = s->dscene.num_tris; out->nodes = (uint32_t)s->d_nodes.count; out->wide_nodes = (uint32_t)s->d_wide.count;
    out->lds_resident = s->dscene.lds_resident;
    out->leading_boxes = (uint32_t)s->dscene.lead_aligned;
    out->walk_cut = s->walk_cut < 0 ? 0xffffffffu : (uint32_t)s->walk_cut;
    out->geometry_uploads = s->geometry_uploads; out->row_updates = s->row_updates;
    out->last_commit_seconds = s->last_commit_seconds; out->last_bvh_seconds = s->last_bvh_seconds;
    out->d_nodes = s->d_wide.count ? (const void*)s->d_wide.ptr : (const void*)s->d_nodes.ptr; out->d_tris = s->d_tris.ptr; out->d_shade = s->d_shade.ptr;
    return PTAMD_OK;
}

// ------------------------------------------------------------------------------- model
int ptamd_model_load(const char* path, ptamd_model** out) { return ptamd_model_load_ex(path, 0u, out); }
int ptamd_model_load_ex(const char* path, uint32_t flags, ptamd_model** out)
{
    if (!path || !out) return fail(PTAMD_ERR_ARG, "null argument");
    if (flags & ~(PTAMD_LOAD_NODE_TRANSFORMS | PTAMD_LOAD_OBJ)) return fail(PTAMD_ERR_ARG, "unknown load flag");
    static_assert(PTAMD_LOAD_NODE_TRANSFORMS == kLoadNodeTransforms && PTAMD_LOAD_OBJ == kLoadObj, "flag values");
    auto m = std::make_unique<ptamd_model>();
    try {
        load_gltf(path, m->model, flags);
    } catch (const std::exception& e) {
        return fail(PTAMD_ERR_IO, e.what());
    }
    *out = m.release();
    return PTAMD_OK;
}
int ptamd_model_desc(const ptamd_model* m, ptamd_mesh_desc* out)
{
    if (!m || !out) return fail(PTAMD_ERR_ARG, "null argument");
    const Model& md = m->model;
    out->num_faces = md.num_faces();
    out->positions = md.positions.data(); out->normals = md.normals.data(); out->texcoords = md.texcoords.data();
    out->group = md.group.data();
    out->num_groups = (uint32_t)md.groups.size(); out->groups = md.groups.data();
    out->num_images = (uint32_t)md.images.size(); out->images = md.images.data();
    return PTAMD_OK;
}
const char* ptamd_model_group_name(const ptamd_model* m, uint32_t group)
{
    if (!m || group >= m->model.group_names.size()) return nullptr;
    return m->model.group_names[group].c_str();
}
void ptamd_model_free(ptamd_model* m) { delete m; }

int ptamd_image_decode(const uint8_t* bytes, size_t size, int* width, int* height, uint8_t** rgba)
{
    if (!bytes || !width || !height || !rgba) return fail(PTAMD_ERR_ARG, "null argument");
    *rgba = nullptr; *width = *height = 0;
    try {
        std::vector<uint8_t> px; int w = 0, h = 0;
        decode_image(bytes, size, w, h, px);
        uint8_t* out = static_cast<uint8_t*>(std::malloc(std::max<size_t>(1, px.size())));
        if (!out) return fail(PTAMD_ERR_LIMIT, "out of memory");
        std::memcpy(out, px.data(), px.size());
        *rgba = out; *width = w; *height = h;
        return PTAMD_OK;
    } catch (const std::exception& e) { return fail(PTAMD_ERR_IO, e.what()); }
}
void ptamd_image_free(uint8_t* rgba) { std::free(rgba); }

// ------------------------------------------------------------------------------- film
static int film_alloc(ptamd_film* f, int w, int h)
{
    f->w = std::max(0, w); f->h = std::max(0, h); // Film::Resize clamps at 0 (Film.cpp:18-19)
    size_t n = std::max<size_t>(1, (size_t)f->w * f->h);
    PT_HIP(hipMalloc(reinterpret_cast<void**>(&f->accum), n * 12));
    PT_HIP(hipMalloc(reinterpret_cast<void**>(&f->samples), n * 4));
    PT_HIP(hipMemsetAsync(f->accum, 0, n * 12, g_stream));
    PT_HIP(hipMemsetAsync(f->samples, 0, n * 4, g_stream));
    return PTAMD_OK;
}
int ptamd_film_create(int w, int h, ptamd_film** out)
{
    if (!out) return fail(PTAMD_ERR_ARG, "null output pointer");
    if (int rc = need_init()) return rc;
    auto f = std::make_unique<ptamd_film>();
    if (int rc = film_alloc(f.get(), w, h)) { f->release(); return rc; }
    *out = f.release();
    return PTAMD_OK;
}
int ptamd_film_create_external(int w, int h, float* d_accum, uint32_t* d_samples, ptamd_film** out)
{
    if (!out || !d_accum || !d_samples || w <= 0 || h <= 0) return fail(PTAMD_ERR_ARG, "bad external film arguments");
    if (int rc = need_init()) return rc;
    auto f = new ptamd_film();
    f->w = w; f->h = h; f->accum = d_accum; f->samples = d_samples; f->external = true;
    *out = f;
    return PTAMD_OK;
}
int ptamd_film_destroy(ptamd_film* f)
{
    if (!f) return PTAMD_OK;
    if (g_initialised) (void)hipStreamSynchronize(g_stream);
    f->release();
    delete f;
    return PTAMD_OK;
}
int ptamd_film_resize(ptamd_film* f, int w, int h)
{
    if (!f) return fail(PTAMD_ERR_ARG, "null film handle");
    if (f->external) return fail(PTAMD_ERR_STATE, "a film over caller-owned buffers cannot be resized");
    if (int rc = need_init()) return rc;
    PT_HIP(hipStreamSynchronize(g_stream));
    f->release();
    return film_alloc(f, w, h);
}
int ptamd_film_reset(ptamd_film* f)
{
    if (!f) return fail(PTAMD_ERR_ARG, "null film handle");
    if (int rc = need_init()) return rc;
    size_t n = (size_t)f->w * f->h;
    if (n) { PT_HIP(hipMemsetAsync(f->accum, 0, n * 12, g_stream)); PT_HIP(hipMemsetAsync(f->samples, 0, n * 4, g_stream)); }
    return PTAMD_OK;
}
int ptamd_film_size(const ptamd_film* f, int* w, int* h)
{
    if (!f) return fail(PTAMD_ERR_ARG, "null film handle");
    if (w) *w = f->w;
    if (h) *h = f->h;
    return PTAMD_OK;
}
int ptamd_film_read(ptamd_film* f, float* accum, uint32_t* samples)
{
    if (!f) return fail(PTAMD_ERR_ARG, "null film handle");
    if (int rc = need_init()) return rc;
    size_t n = (size_t)f->w * f->h;
    if (accum && n) PT_HIP(hipMemcpyAsync(accum, f->accum, n * 12, hipMemcpyDeviceToHost, g_stream));
    if (samples && n) PT_HIP(hipMemcpyAsync(samples, f->samples, n * 4, hipMemcpyDeviceToHost, g_stream));
    PT_HIP(hipStreamSynchronize(g_stream));
    return PTAMD_OK;
}
int ptamd_film_average(ptamd_film* f, float* avg)
{
    if (!f || !avg) return fail(PTAMD_ERR_ARG, "null argument");
    if (int rc = need_init()) return rc;
    size_t n = (size_t)f->w * f->h;
    if (!n) return PTAMD_OK;
    if (!f->avg) PT_HIP(hipMalloc(reinterpret_cast<void**>(&f->avg), n * 12));
    PT_HIP(launch_average(f->accum, f->samples, (int)n, f->avg, g_stream));
    PT_HIP(hipMemcpyAsync(avg, f->avg, n * 12, hipMemcpyDeviceToHost, g_stream));
    PT_HIP(hipStreamSynchronize(g_stream));
    return PTAMD_OK;
}
int ptamd_film_resolve_rgba8(ptamd_film* f, int tonemap, int colourspace, uint8_t* rgba)
{
    if (!f || !rgba) return fail(PTAMD_ERR_ARG, "null argument");
    if (tonemap < 0 || tonemap > 1 || colourspace < 0 || colourspace > 1) return fail(PTAMD_ERR_ARG, "tonemap / colourspace must be 0 or 1");
    if (int rc = need_init()) return rc;
    size_t n = (size_t)f->w * f->h;
    if (!n) return PTAMD_OK;
    if (!f->rgba) PT_HIP(hipMalloc(reinterpret_cast<void**>(&f->rgba), n * 4));
    PT_HIP(launch_resolve(f->accum, f->samples, (int)n, tonemap, colourspace, f->rgba, g_stream));
    PT_HIP(hipMemcpyAsync(rgba, f->rgba, n * 4, hipMemcpyDeviceToHost, g_stream));
    PT_HIP(hipStreamSynchronize(g_stream));
    return PTAMD_OK;
}
int ptamd_film_device_ptrs(ptamd_film* f, void** a, void** s)
{
    if (!f) return fail(PTAMD_ERR_ARG, "null film handle");
    if (a) *a = f->accum;
    if (s) *s = f->samples;
    return PTAMD_OK;
}

// ------------------------------------------------------------------------------- render
// Two schedules of the sample-major kernel leave the film as it is and pay on some scenes only, so the first launch that
// could use one measures it (DRenderParams::walk_budget carries the choice):
//  * the walk cut (WalkBudget in pt_device.h; LDS-resident scenes whose last object is a mesh) pays when a few long walks
//    hold a wave (the mesh Cornell: +9 %, the same behind a sphere +8 %) and costs when all walks are about as long, when
//    the kernel runs at low occupancy (a camera inside a 192-triangle blob: -18 %) or when the objects in front of the mesh
//    make every extra search expensive (a room of seven boxes and two spheres around the mesh: -20 %);
//  * the deferred shared walk (DEFER in render_samples_kernel; scenes read from global memory whose only mesh is their last
//    object): the lanes whose rays meet the mesh's bounds wait until 20 or 40 of them make the walk worth its trips. A
//    7680-triangle mesh among primitives: +28 % at 20 lanes, +27 % at 24, +17 % at 12; a million triangles that most rays
//    reach: -0.6 % at 24, +3.6 % at 40, +0.8 % at 48, -7 % at 56.
// Which of these a scene is shows in no count that the host could take at commit, so the first eligible launch renders a
// sample of the frame at 64 samples into a scratch film -- every n-th 16x16-pixel tile, four times the workgroups that the
// GPU keeps resident in all (a launch of fewer measures one workgroup's life, not the rate) -- once to warm the caches and
// then three times per candidate, and keeps the fastest candidate if its best time is 1.5 % under that of the plain
// schedule: 50 to 250 ms in all, once per commit and (film size, depth). Films of under two such samples, or of more than
// 64 Mpixels (the scratch film), are rendered with the plain schedule.
// PTAMD_WALK_BUDGET=<0..63> (the cut) and PTAMD_WALK_DEFER=<0..64> (the waiting lanes) set them without measuring.
constexpr uint32_t kWalkCut = 28; // of 64: cut when the lanes still walking are down to 7/16 of those that entered the walk
constexpr uint32_t kDeferFew = 20, kDeferMany = 40; // waiting lanes that start a deferred walk
static int tune_walk_cut(ptamd_scene* s, const DRenderParams& user)
{
    const int W = user.cam.width, H = user.cam.height;
    s->walk_cut_w = W; s->walk_cut_h = H; s->walk_cut_depth = user.depth;
    s->walk_cut = 0;
    const bool defer = !s->dscene.lds_resident;
    const uint32_t cand[3] = { 0u, defer ? kDeferFew : kWalkCut, kDeferMany };
    const int ncand = defer ? 3 : 2;
    const int kTile = 16;                          // pixels: four 8x8-pixel workgroups
    const long long kProbeTiles = 4 * 1280 / 4;    // four times 256 CUs x 5 resident workgroups
    const int cols = (W + kTile - 1) / kTile;
    const long long total = (long long)cols * ((H + kTile - 1) / kTile);
    if (total < 2 * kProbeTiles || (long long)W * H > (64ll << 20)) return PTAMD_OK;
    int stride = (int)(total / kProbeTiles);
    auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
    while (gcd(stride, cols) != 1) ++stride; // the sampled tiles fall in every column of tiles, not in a few
    float* accum = nullptr; uint32_t* samples = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() { if (accum) (void)hipFree(accum); if (samples) (void)hipFree(samples); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); };
    auto bail = [&](hipError_t e) { cleanup(); return fail(PTAMD_ERR_DEVICE, hipGetErrorString(e)); };
    hipError_t e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&accum), (size_t)W * H * 12)) != hipSuccess) return bail(e);
    if ((e = hipMalloc(reinterpret_cast<void**>(&samples), (size_t)W * H * 4)) != hipSuccess) return bail(e);
    if ((e = hipMemsetAsync(accum, 0, (size_t)W * H * 12, g_stream)) != hipSuccess) return bail(e);
    if ((e = hipMemsetAsync(samples, 0, (size_t)W * H * 4, g_stream)) != hipSuccess) return bail(e);
    if ((e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess) return bail(e);
    DRenderParams rp = user;
    rp.counters = nullptr; rp.flags &= ~(uint32_t)PTAMD_FLAG_COUNT;
    rp.accum = accum; rp.samples = samples;
    rp.x0 = 0; rp.y0 = 0; rp.x1 = W; rp.y1 = H; rp.tiles_x = (W + 7) / 8; rp.tiles_y = (H + 7) / 8;
    rp.spp_begin = 0; rp.spp_count = 64;
    rp.tile_size = kTile; rp.tile_phase = stride / 2; rp.tile_stride = stride; rp.tile_cols = cols;
    rp.tile_side = kTile / 8; rp.tile_waves = rp.tile_side * rp.tile_side;
    rp.tiles_owned = (int)((total - rp.tile_phase + stride - 1) / stride);
    float best[3] = { 3.0e38f, 3.0e38f, 3.0e38f };
    for (int rep = 0; rep < 4; ++rep) // the first pass warms the caches and is not timed
        for (int v = 0; v < (rep ? ncand : 1); ++v) {
            rp.walk_budget = cand[v];
            if ((e = hipEventRecord(e0, g_stream)) != hipSuccess) return bail(e);
            if ((e = launch_render(s->dscene, rp, g_stream)) != hipSuccess) return bail(e);
            if ((e = hipEventRecord(e1, g_stream)) != hipSuccess || (e = hipEventSynchronize(e1)) != hipSuccess) return bail(e);
            float ms = 0.0f;
            if ((e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) return bail(e);
            if (rep > 0) best[v] = std::min(best[v], ms);
        }
    cleanup();
    int pick = 0;
    for (int v = 1; v < ncand; ++v) if (best[v] < best[pick]) pick = v;
    s->walk_cut = best[pick] < 0.985f * best[0] ? (int)cand[pick] : 0;
    if (std::getenv("PTAMD_DEBUG")) {
        std::fprintf(stderr, "ptamd: %s measured on %d tiles:", defer ? "deferred walk" : "walk cut", rp.tiles_owned);
        for (int v = 0; v < ncand; ++v) std::fprintf(stderr, " %u: %.3f ms", cand[v], best[v]);
        std::fprintf(stderr, " -> %d\n", s->walk_cut);
    }
    return PTAMD_OK;
}
// The schedule for this launch (DRenderParams::walk_budget). may_measure: the caller may allocate (the scratch film).
static int walk_cut_for(ptamd_scene* s, const DRenderParams& rp, bool may_measure, uint32_t* out)
{
    *out = 0u;
    const bool cut = s->dscene.lds_resident && s->dscene.tail_mesh;
    const bool defer = !s->dscene.lds_resident && (s->dscene.tail_mesh & 2u);
    if (!cut && !defer) return PTAMD_OK;
    if (rp.spp_count < kSampleMajorMinSpp || rp.depth < 2 || rp.albedo_only) return PTAMD_OK; // launches that the sample-major kernel does not take
    if (const char* env = std::getenv(cut ? "PTAMD_WALK_BUDGET" : "PTAMD_WALK_DEFER")) { *out = (uint32_t)std::min(cut ? 63 : 64, std::max(0, std::atoi(env))); return PTAMD_OK; }
    if (s->walk_cut < 0 || s->walk_cut_w != rp.cam.width || s->walk_cut_h != rp.cam.height || s->walk_cut_depth != rp.depth) {
        if (!may_measure) return PTAMD_OK;
        if (tune_walk_cut(s, rp) != PTAMD_OK) { (void)hipGetLastError(); s->walk_cut = 0; } // no scratch film: render with the plain schedule
        s->walk_cut_uses = 0;
    }
    ++s->walk_cut_uses;
    *out = (uint32_t)s->walk_cut;
    return PTAMD_OK;
}

// counters: where a PTAMD_FLAG_COUNT launch tallies (kNumCounters x u64 of device memory, cleared here first); null = the
// scene's own buffer
static int render_impl(ptamd_scene* s, const ptamd_camera* cam, ptamd_film* f, uint32_t spp_begin, uint32_t spp_count,
                       int depth, int albedo_only, uint32_t seed, const ptamd_rect* rect, uint32_t flags,
                       int tile_size, int tile_phase, int tile_stride, unsigned long long* counters)
{
    if (!s || !cam || !f) return fail(PTAMD_ERR_ARG, "null argument");
    if (int rc = need_init()) return rc;
    if (!s->committed) return fail(PTAMD_ERR_STATE, "scene has uncommitted edits: call ptamd_scene_commit first");
    if (f->w <= 0 || f->h <= 0) return fail(PTAMD_ERR_STATE, "film has no pixels");
    if (s->dscene.sky_size > 0 && !s->dscene.sky) return fail(PTAMD_ERR_STATE, "sky map has no device buffer");
    if ((uint64_t)spp_begin + spp_count > 0xFFFFFFFFull) return fail(PTAMD_ERR_ARG, "sample index range overflows 32 bits");
    DRenderParams rp; std::memset(&rp, 0, sizeof rp);
    rp.cam = make_camera(*cam, f->w, f->h);
    rp.accum = f->accum; rp.samples = f->samples;
    rp.x0 = rect ? std::max(0, rect->x0) : 0; rp.y0 = rect ? std::max(0, rect->y0) : 0;
    rp.x1 = rect ? std::min(f->w, rect->x1) : f->w; rp.y1 = rect ? std::min(f->h, rect->y1) : f->h;
    if (flags & PTAMD_FLAG_COUNT) {
        if (!counters) { // the scene's own counters: two scenes' counting renders do not share tallies
            if (!s->d_counters) PT_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_counters), kNumCounters * sizeof(unsigned long long)));
            counters = s->d_counters;
            g_last_counters = counters;
        }
        PT_HIP(hipMemsetAsync(counters, 0, kNumCounters * sizeof(unsigned long long), g_stream));
        rp.counters = counters;
    }
    if (rp.x1 <= rp.x0 || rp.y1 <= rp.y0 || spp_count == 0) return PTAMD_OK;
    rp.spp_begin = spp_begin; rp.spp_count = spp_count;
    rp.depth = depth; rp.albedo_only = albedo_only ? 1 : 0; rp.seed = seed; rp.flags = flags;
    rp.tiles_x = (rp.x1 - rp.x0 + 7) / 8; rp.tiles_y = (rp.y1 - rp.y0 + 7) / 8;
    if (tile_stride > 0) {
        const int cols = (f->w + tile_size - 1) / tile_size, rows = (f->h + tile_size - 1) / tile_size;
        const long long total = (long long)cols * rows;
        if (tile_phase >= total) return PTAMD_OK; // this rank owns no tile
        rp.tile_size = tile_size; rp.tile_phase = tile_phase; rp.tile_stride = tile_stride; rp.tile_cols = cols;
        rp.tile_side = tile_size / 8; rp.tile_waves = rp.tile_side * rp.tile_side;
        rp.tiles_owned = (int)((total - tile_phase + tile_stride - 1) / tile_stride);
    }
    if (depth <= 0) {
        // TraceRay returns 0 at once (PathTracer.cpp:35-36): every sample adds black
        rp.depth = 0;
    }
    if (int rc = walk_cut_for(s, rp, counters != g_static_counters, &rp.walk_budget)) return rc; // (the entry point that allocates nothing brings those counters)
    PT_HIP(launch_render(s->dscene, rp, g_stream));
    return PTAMD_OK;
}
int ptamd_render(ptamd_scene* s, const ptamd_camera* cam, ptamd_film* f, uint32_t spp_begin, uint32_t spp_count,
                 int depth, int albedo_only, uint32_t seed, const ptamd_rect* rect, uint32_t flags)
{
    return render_impl(s, cam, f, spp_begin, spp_count, depth, albedo_only, seed, rect, flags, 0, 0, 0, nullptr);
}
int ptamd_render_counted_noalloc(ptamd_scene* s, const ptamd_camera* cam, ptamd_film* f, uint32_t spp_begin, uint32_t spp_count,
                                 int depth, int albedo_only, uint32_t seed, const ptamd_rect* rect, uint32_t flags, ptamd_stats* d_stats)
{
    if (!d_stats) return fail(PTAMD_ERR_ARG, "null device pointer for the statistics");
    if (int rc = need_init()) return rc;
    static_assert(sizeof(ptamd_stats) == kNumCounters * sizeof(unsigned long long), "ptamd_stats is the counter array");
    if (int rc = render_impl(s, cam, f, spp_begin, spp_count, depth, albedo_only, seed, rect, flags | PTAMD_FLAG_COUNT, 0, 0, 0, g_static_counters)) return rc;
    PT_HIP(hipMemcpyAsync(d_stats, g_static_counters, sizeof(ptamd_stats), hipMemcpyDeviceToDevice, g_stream));
    return PTAMD_OK;
}
int ptamd_render_tiles(ptamd_scene* s, const ptamd_camera* cam, ptamd_film* f, uint32_t spp_begin, uint32_t spp_count,
                       int depth, int albedo_only, uint32_t seed, int tile_size, int tile_phase, int tile_stride, uint32_t flags)
{
    if (tile_size < 8 || tile_size % 8 != 0 || tile_size > 4096) return fail(PTAMD_ERR_ARG, "tile_size must be a multiple of 8 in [8, 4096]");
    if (tile_stride < 1 || tile_phase < 0 || tile_phase >= tile_stride) return fail(PTAMD_ERR_ARG, "need 0 <= tile_phase < tile_stride");
    return render_impl(s, cam, f, spp_begin, spp_count, depth, albedo_only, seed, nullptr, flags, tile_size, tile_phase, tile_stride, nullptr);
}
int ptamd_render_stats(ptamd_stats* out)
{
    if (!out) return fail(PTAMD_ERR_ARG, "null argument");
    if (int rc = need_init()) return rc;
    unsigned long long c[kNumCounters];
    if (!g_last_counters) return fail(PTAMD_ERR_STATE, "no render with PTAMD_FLAG_COUNT has run (or its scene is gone)");
    PT_HIP(hipMemcpyAsync(c, g_last_counters, sizeof c, hipMemcpyDeviceToHost, g_stream));
    PT_HIP(hipStreamSynchronize(g_stream));
    out->samples = c[0]; out->rays = c[1]; out->nodes = c[2]; out->tris = c[3];
    out->ray_trips = c[4]; out->node_trips = c[5]; out->tri_trips = c[6]; out->shade_lanes = c[7]; out->shade_trips = c[8]; out->steal_rounds = c[9];
    return PTAMD_OK;
}

// ------------------------------------------------------------------------------- probes
int ptamd_camera_rays(const ptamd_camera* cam, int w, int h, float* out)
{
    if (!cam || !out || w <= 0 || h <= 0) return fail(PTAMD_ERR_ARG, "bad argument");
    if (int rc = need_init()) return rc;
    DCamera dc = make_camera(*cam, w, h);
    float* d = nullptr;
    size_t bytes = (size_t)w * h * 6 * sizeof(float);
    PT_HIP(hipMalloc(reinterpret_cast<void**>(&d), bytes));
    hipError_t e = launch_camera_rays(dc, d, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    (void)hipFree(d);
    PT_HIP(e);
    return PTAMD_OK;
}
static int closest_hits_impl(ptamd_scene* s, const float* rays, uint32_t n, float* rec, uint32_t* counts)
{
    if (!s || !rays || !rec) return fail(PTAMD_ERR_ARG, "null argument");
    if (int rc = need_init()) return rc;
    if (!s->committed) return fail(PTAMD_ERR_STATE, "scene has uncommitted edits: call ptamd_scene_commit first");
    if (n == 0) return PTAMD_OK;
    float *d_rays = nullptr, *d_rec = nullptr; uint32_t* d_cnt = nullptr;
    PT_HIP(hipMalloc(reinterpret_cast<void**>(&d_rays), (size_t)n * 24));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_rec), (size_t)n * 80);
    if (e == hipSuccess && counts) e = hipMalloc(reinterpret_cast<void**>(&d_cnt), (size_t)n * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rays, rays, (size_t)n * 24, hipMemcpyHostToDevice, g_stream);
    if (e == hipSuccess) e = launch_closest_hits(s->dscene, d_rays, n, d_rec, d_cnt, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(rec, d_rec, (size_t)n * 80, hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess && counts) e = hipMemcpyAsync(counts, d_cnt, (size_t)n * 8, hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    (void)hipFree(d_rays); (void)hipFree(d_rec); if (d_cnt) (void)hipFree(d_cnt);
    PT_HIP(e);
    return PTAMD_OK;
}
int ptamd_closest_hits(ptamd_scene* s, const float* rays, uint32_t n, float* rec) { return closest_hits_impl(s, rays, n, rec, nullptr); }
int ptamd_traversal_counts(ptamd_scene* s, const float* rays, uint32_t n, float* rec, uint32_t* counts)
{
    if (!counts) return fail(PTAMD_ERR_ARG, "null argument");
    return closest_hits_impl(s, rays, n, rec, counts);
}

} // extern "C"
