// Mutation fuzzing of the model loader under the host sanitizers (no GPU code involved):
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all tools/loader_fuzz.cpp path-tracer_amd/csrc/gltf_loader.cpp \
//       path-tracer_amd/csrc/jpeg_decode.cpp path-tracer_amd/csrc/bvh_build.cpp -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -lz -lpthread -o /tmp/loader_fuzz
//   /tmp/loader_fuzz <trials per file> <seed> file.glb [more files]
// Each trial writes a mutated copy (flipped bytes anywhere / in the JSON chunk, truncation, an overwritten run, spliced
// bytes of another place of the file), loads it and, when the load succeeds on a small model, builds its trees. A run
// that ends with exit code 0 met no memory error and no undefined behaviour.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>
#include <stdexcept>
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
    if (argc < 4) { fprintf(stderr, "usage: loader_fuzz trials seed file...\n"); return 2; }
    const int trials = atoi(argv[1]);
    std::mt19937 rng((unsigned)atoi(argv[2]));
    auto below = [&](size_t n) { return (size_t)(rng() % (n ? n : 1)); };
    size_t loaded = 0, refused = 0;
    for (int f = 3; f < argc; ++f) {
        std::ifstream in(argv[f], std::ios::binary);
        std::vector<unsigned char> raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (raw.size() < 64) { fprintf(stderr, "%s: too small\n", argv[f]); return 2; }
        const bool obj = strstr(argv[f], ".obj") != nullptr;
        for (int t = 0; t < trials; ++t) {
            std::vector<unsigned char> m = raw;
            switch (t % 5) {
            case 0: for (size_t k = 0, n = 1 + below(8); k < n; ++k) m[below(m.size())] = (unsigned char)rng(); break;
            case 1: {
                uint32_t n = 0; memcpy(&n, &m[12], 4);
                if (n > m.size() - 20) n = (uint32_t)m.size() - 20;
                for (size_t k = 0, c = 1 + below(4); k < c; ++k) m[20 + below(n)] = (unsigned char)(32 + below(95));
                break;
            }
            case 2: m.resize(1 + below(m.size() - 1)); break;
            case 3: { const size_t a = below(m.size() - 64); memset(&m[a], (int)below(256), 4 + below(60)); break; }
            default: { const size_t a = below(m.size() - 64), b = below(m.size() - 64), n = 4 + below(60); memmove(&m[a], &m[b], n); break; }
            }
            const char* path = obj ? "/tmp/loader_fuzz_case.obj" : "/tmp/loader_fuzz_case.glb";
            { std::ofstream out(path, std::ios::binary); out.write((const char*)m.data(), (std::streamsize)m.size()); }
            try {
                Model md;
                if (obj) load_obj(path, md); else load_gltf(path, md, (t & 8) ? kLoadNodeTransforms : 0u);
                ++loaded;
                if (md.num_faces() > 0 && md.num_faces() < 20000) {
                    BvhResult r = build_bvh(md.positions.data(), md.group.data(), md.num_faces(), 32);
                    build_wide(r);
                }
            } catch (const std::exception&) { ++refused; }
        }
    }
    printf("loaded %zu, refused %zu\n", loaded, refused);
    return 0;
}
