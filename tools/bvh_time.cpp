// Host-side timing of the BVH builder on a model file: build_bvh (binary SAH tree) and build_wide (eight-wide collapse).
//   g++ -O2 -std=c++17 tools/bvh_time.cpp path-tracer_amd/csrc/bvh_build.cpp path-tracer_amd/csrc/gltf_loader.cpp path-tracer_amd/csrc/jpeg_decode.cpp \
//       -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -lz -lpthread -o /tmp/bvh_time && /tmp/bvh_time model.glb
#include <chrono>
#include <cstdio>
#include "../path-tracer_amd/csrc/bvh_build.h"
#include "../path-tracer_amd/csrc/gltf_loader.h"
using namespace ptamd;
extern "C" void ptamd_pbr_default(ptamd_pbr* p)
{
    p->base_color_tex = p->normal_tex = p->metallic_roughness_tex = p->emissive_tex = p->transmission_tex = -1;
    for (int k = 0; k < 4; ++k) p->base_color_factor[k] = 1.0f;
    p->metallic_factor = 0.0f; p->roughness_factor = 1.0f; p->alpha_mode = 0; p->alpha_cutoff = 0.5f; p->normal_scale = 1.0f;
    p->emissive_factor[0] = p->emissive_factor[1] = p->emissive_factor[2] = 0.0f;
    p->transmission_factor = 0.0f; p->ior = 1.5f;
}
int main(int argc, char** argv)
{
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
    Model m;
    auto t0 = now(); load_gltf(argv[1], m); auto t1 = now();
    BvhResult r = build_bvh(m.positions.data(), m.group.data(), m.num_faces(), 32); auto t2 = now();
    build_wide(r); auto t3 = now();
    double sah = 0; // surface-area cost of the binary tree, to compare builders
    printf("faces %u: load %.3f s, build_bvh %.3f s (%zu nodes, depth %d, %u quads), build_wide %.3f s (%zu nodes, depth %d)\n", m.num_faces(), secs(t0, t1),
           secs(t1, t2), r.nodes.size(), r.depth, r.num_quads, secs(t2, t3), r.wide.size(), r.wide_depth);
    unsigned long long h = 1469598103934665603ull;
    for (uint32_t o : r.order) h = (h ^ o) * 1099511628211ull;
    for (const DNode& n : r.nodes) { h = (h ^ (unsigned)n.left) * 1099511628211ull; h = (h ^ (unsigned)n.right) * 1099511628211ull; }
    printf("tree hash %016llx\n", h); (void)sah;
    return 0;
}
