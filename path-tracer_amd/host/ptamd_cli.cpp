// Headless renderer: what the reference's main() does (main.cpp:50-600) without the window and the UI.
// Reads the plain-text scene format of this repository (the one scenes/presets.py writes), renders on the GPU
// through the host classes of ptamd.hpp and saves the image.
//
//   ptamd_cli <scene.txt> --out image.png [--spp 64] [--depth 10] [--albedo] [--seed N] [--device D]
//             [--pfm hdr.pfm] [--raw prefix] [--linear] [--no-tonemap] [--frames] [--rr]
//
// Scene records (whitespace separated, '#' comments):
//   film W H | camera px py pz rx ry rz fov [near far] | background r g b
//   sphere px py pz r <mat> | box px py pz rx ry rz sx sy sz <mat> | mesh path px py pz rx ry rz sx sy sz
//   <mat> = albedo(3) roughness metallic emissionColour(3) emissionStrength IOR transmission
//   skybox file.rgba8 N intensity   (6 faces of N x N RGBA8, +x -x +y -y +z -z; not a feature of the reference)
// --rr turns Russian roulette on (also not in the reference; off by default). --node-transforms applies the glTF
// scene graph of mesh files and --obj accepts Wavefront .obj meshes (the reference does neither).
// --frames renders one sample per pixel per launch, as the reference's main loop does (main.cpp:565-569);
// the default renders all samples in one launch. Both give the same image bit for bit.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <vector>

#include "image_io.hpp"
#include "ptamd.hpp"

using namespace ptamd_host;

static Material ReadMaterial(std::istream& in)
{
    Material m;
    in >> m.albedo.x >> m.albedo.y >> m.albedo.z >> m.roughness >> m.metallic >> m.emissionColour.x >> m.emissionColour.y >>
        m.emissionColour.z >> m.emissionStrength >> m.IOR >> m.transmission;
    return m;
}

int main(int argc, char** argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: ptamd_cli <scene.txt> --out image.png [--spp N] [--depth D] [--albedo] [--seed S] [--device I] [--pfm f] [--raw prefix] [--linear] [--no-tonemap] [--frames] [--rr] [--node-transforms] [--obj]\n");
        return 2;
    }
    std::string scenePath = argv[1], outPng, outPfm, outRaw;
    int spp = 64, depth = 10, device = 0;
    uint32_t seed = 0x5EED, loadFlags = 0;
    bool albedo = false, linear = false, noTonemap = false, frames = false, roulette = false;
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--out") outPng = next();
        else if (a == "--pfm") outPfm = next();
        else if (a == "--raw") outRaw = next();
        else if (a == "--spp") spp = std::atoi(next());
        else if (a == "--depth") depth = std::atoi(next());
        else if (a == "--seed") seed = (uint32_t)std::strtoul(next(), nullptr, 0);
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--albedo") albedo = true;
        else if (a == "--linear") linear = true;
        else if (a == "--no-tonemap") noTonemap = true;
        else if (a == "--frames") frames = true;
        else if (a == "--rr") roulette = true;
        else if (a == "--node-transforms") loadFlags |= PTAMD_LOAD_NODE_TRANSFORMS;
        else if (a == "--obj") loadFlags |= PTAMD_LOAD_OBJ;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    try {
        Init(device);
        int W = 512, H = 512; // main.cpp:52-53
        PathTracer pathTracer;
        Camera camera(ivec2(W, H));
        std::ifstream f(scenePath);
        if (!f) throw std::runtime_error("cannot open scene " + scenePath);
        std::string line;
        while (std::getline(f, line)) {
            size_t hash = line.find('#');
            if (hash != std::string::npos) line.erase(hash);
            std::istringstream in(line);
            std::string kind;
            if (!(in >> kind)) continue;
            if (kind == "film") in >> W >> H;
            else if (kind == "camera") {
                vec3 p, r; float fov, n, fr;
                in >> p.x >> p.y >> p.z >> r.x >> r.y >> r.z >> fov;
                camera.SetFov(fov); camera.SetPosition(p); camera.SetRotation(r);
                if (in >> n >> fr) { camera.SetNearPlane(n); camera.SetFarPlane(fr); }
            } else if (kind == "background") {
                vec3 c; in >> c.x >> c.y >> c.z; pathTracer.SetBackgroundColour(c);
            } else if (kind == "skybox") {
                std::string sp; int n = 0; float intensity = 1.0f; in >> sp >> n >> intensity;
                if (n <= 0) throw std::runtime_error("skybox: bad face size");
                std::ifstream sf(sp, std::ios::binary);
                std::vector<uint8_t> texels(size_t(6) * n * n * 4);
                if (!sf || !sf.read(reinterpret_cast<char*>(texels.data()), (std::streamsize)texels.size())) throw std::runtime_error("skybox: cannot read " + sp);
                const uint8_t* faces[6];
                for (int k = 0; k < 6; ++k) faces[k] = texels.data() + size_t(k) * n * n * 4;
                pathTracer.SetSkybox(n, faces, intensity);
            } else if (kind == "sphere") {
                vec3 p; float r; in >> p.x >> p.y >> p.z >> r;
                auto o = std::make_shared<Sphere>("Sphere", p, r);
                o->SetMaterial(ReadMaterial(in));
                pathTracer.AddRayObject(o);
            } else if (kind == "box") {
                vec3 p, r, s; in >> p.x >> p.y >> p.z >> r.x >> r.y >> r.z >> s.x >> s.y >> s.z;
                auto o = std::make_shared<Box>("Box");
                o->SetPosition(p); o->SetRotation(r); o->SetSize(s); o->SetMaterial(ReadMaterial(in));
                pathTracer.AddRayObject(o);
            } else if (kind == "mesh") {
                std::string mp; vec3 p, r, s; in >> mp >> p.x >> p.y >> p.z >> r.x >> r.y >> r.z >> s.x >> s.y >> s.z;
                auto o = std::make_shared<Mesh>(mp, "Mesh", loadFlags);
                o->SetPosition(p); o->SetRotation(r); o->SetScale(s);
                pathTracer.AddRayObject(o);
            } else throw std::runtime_error("unknown scene record '" + kind + "'");
        }
        auto t0 = std::chrono::steady_clock::now();
        pathTracer.Commit();
        double commitS = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        Film film(W, H);
        if (linear) film.SetColourSpace(ColourSpace::Linear);
        if (noTonemap) film.SetToneMap(ToneMap::None);
        t0 = std::chrono::steady_clock::now();
        const uint32_t flags = roulette ? PTAMD_FLAG_RUSSIAN_ROULETTE : 0u;
        if (frames) for (int i = 0; i < spp; ++i) Render(camera, pathTracer, film, (uint32_t)i, 1, depth, albedo, seed, nullptr, flags); // = RayTraceParallel per frame
        else Render(camera, pathTracer, film, 0, (uint32_t)spp, depth, albedo, seed, nullptr, flags);
        check(ptamd_synchronize());
        double renderS = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (!outPng.empty() && !SaveImagePNG(outPng, W, H, film.ResolveToRGBA8())) throw std::runtime_error("cannot write " + outPng);
        if (!outPfm.empty() && !SaveImagePFM(outPfm, W, H, film.Average())) throw std::runtime_error("cannot write " + outPfm);
        if (!outRaw.empty()) {
            auto a = film.Accum(); auto s = film.Samples();
            if (!SaveRaw(outRaw + ".accum.f32", a.data(), a.size() * 4) || !SaveRaw(outRaw + ".samples.u32", s.data(), s.size() * 4))
                throw std::runtime_error("cannot write " + outRaw);
        }
        std::printf("{\"width\": %d, \"height\": %d, \"spp\": %d, \"depth\": %d, \"objects\": %d, \"commit_s\": %.4f, \"render_s\": %.4f, \"msamples_per_s\": %.2f}\n",
                    W, H, spp, depth, pathTracer.GetSizeOfRayObjects(), commitS, renderS, double(W) * H * spp / renderS * 1e-6);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ptamd_cli: %s\n", e.what());
        return 1;
    }
    return 0;
}
