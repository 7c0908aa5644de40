// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product path.
//
// Headless driver around the reference renderer's own translation units
// (compiled where they lie under /root/reference by oracle/Makefile; the
// resulting binary lives in oracle/_ref/ and is git-ignored).  It restates the
// 31 lines of the reference frame driver (main.cpp:17-47: TracePixels /
// RayTraceParallel) because main.cpp itself needs SDL2/GLEW Windows headers,
// reads a plain-text scene description, renders with the reference classes
// and dumps raw little-endian buffers that tests/ and the fixture generator
// compare against.
//
// Modes (argv[1]):
//   render  <scene> <out_prefix> <spp> <depth> <albedoOnly 0|1> <threads> <tasks>
//           -> <out_prefix>.accum.f32 (W*H*3), <out_prefix>.samples.u32 (W*H),
//              <out_prefix>.rgba8 (Film::ResolveToRGBA8 with the film defaults)
//   rays    <scene> <out>        -> per pixel 6 f32: Camera::GetRay origin, direction
//   hits    <scene> <out>        -> per pixel 20 f32: closest-hit record of the primary ray
//                                   (hit flag, t, p, n, frontFace, 11 material floats + pad)
//   film    <in.f32> <n> <out_prefix> -> Film KAT: n RGB triples are added as samples
//                                   to an n x 1 film (2 samples each: value, value*0.5),
//                                   dumps accum + the 4 resolve variants
//
// Scene file (whitespace separated, '#' comments):
//   film W H
//   camera px py pz rx ry rz fov [near far]
//   sphere px py pz r  <mat>
//   box    px py pz rx ry rz sx sy sz <mat>
//   mesh   path px py pz rx ry rz sx sy sz
//   <mat> = albedo(3) roughness metallic emissionColour(3) emissionStrength IOR transmission
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "PathTracer/Film.h"
#include "PathTracer/Sphere.h"
#include "PathTracer/Box.h"
#include "PathTracer/Mesh.h"
#include "PathTracer/PathTracer.h"
#include "PathTracer/Camera.h"
#include "PathTracer/ThreadPool.h"

namespace {

struct SceneFile {
    int W = 64, H = 64;
    std::shared_ptr<Camera> camera;
    std::shared_ptr<PathTracer> tracer = std::make_shared<PathTracer>();
};

Material ReadMaterial(std::istream& in)
{
    Material m;
    in >> m.albedo.x >> m.albedo.y >> m.albedo.z >> m.roughness >> m.metallic
       >> m.emissionColour.x >> m.emissionColour.y >> m.emissionColour.z
       >> m.emissionStrength >> m.IOR >> m.transmission;
    return m;
}

SceneFile LoadScene(const std::string& path)
{
    SceneFile s;
    std::ifstream f(path);
    if (!f) { std::fprintf(stderr, "cannot open scene %s\n", path.c_str()); std::exit(2); }
    glm::vec3 cpos(0.f), crot(0.f);
    float fov = 40.f, nearP = 0.1f, farP = 100.f;
    std::string line;
    while (std::getline(f, line)) {
        size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        std::istringstream in(line);
        std::string kind;
        if (!(in >> kind)) continue;
        if (kind == "film") {
            in >> s.W >> s.H;
        } else if (kind == "camera") {
            in >> cpos.x >> cpos.y >> cpos.z >> crot.x >> crot.y >> crot.z >> fov;
            float a, b;
            if (in >> a >> b) { nearP = a; farP = b; }
        } else if (kind == "sphere") {
            glm::vec3 p; float r;
            in >> p.x >> p.y >> p.z >> r;
            auto o = std::make_shared<Sphere>("Sphere", p, r);
            o->SetMaterial(ReadMaterial(in));
            s.tracer->AddRayObject(o);
        } else if (kind == "box") {
            glm::vec3 p, r, sz;
            in >> p.x >> p.y >> p.z >> r.x >> r.y >> r.z >> sz.x >> sz.y >> sz.z;
            auto o = std::make_shared<Box>("Box");
            o->SetPosition(p); o->SetRotation(r); o->SetSize(sz);
            o->SetMaterial(ReadMaterial(in));
            s.tracer->AddRayObject(o);
        } else if (kind == "mesh") {
            std::string mp; glm::vec3 p, r, sc;
            in >> mp >> p.x >> p.y >> p.z >> r.x >> r.y >> r.z >> sc.x >> sc.y >> sc.z;
            auto o = std::make_shared<Mesh>(mp, "Mesh");
            o->SetPosition(p); o->SetRotation(r); o->SetScale(sc);
            s.tracer->AddRayObject(o);
        } else {
            std::fprintf(stderr, "unknown scene record '%s'\n", kind.c_str());
            std::exit(2);
        }
    }
    // Same call order as main.cpp:60-63 (each setter recomputes the matrices).
    s.camera = std::make_shared<Camera>(glm::ivec2(s.W, s.H));
    s.camera->SetFov(fov);
    s.camera->SetNearPlane(nearP);
    s.camera->SetFarPlane(farP);
    s.camera->SetPosition(cpos);
    s.camera->SetRotation(crot);
    return s;
}

// main.cpp:17-28
void TracePixelsRef(int fromy, int toy, glm::ivec2 win, std::shared_ptr<Camera> cam,
                    std::shared_ptr<PathTracer> pt, std::shared_ptr<Film> film, int depth, bool albedoOnly)
{
    for (int y = fromy; y <= toy && y < win.y; ++y)
        for (int x = 0; x < win.x; ++x) {
            Ray ray = cam->GetRay({ x, y }, win);
            glm::vec3 c = pt->TraceRay(ray, depth, albedoOnly);
            film->AddSample(x, y, c);
        }
}

// main.cpp:30-47
void RayTraceParallelRef(ThreadPool& pool, int numTasks, glm::ivec2 win, std::shared_ptr<Camera> cam,
                         std::shared_ptr<PathTracer> pt, std::shared_ptr<Film> film, int depth, bool albedoOnly)
{
    int rowsPerThread = (int)std::ceil(win.y / static_cast<float>(numTasks));
    for (int i = 0; i < numTasks; ++i) {
        int startY = i * rowsPerThread;
        int endY = std::min(startY + rowsPerThread, win.y);
        pool.EnqueueTask([=] { TracePixelsRef(startY, endY - 1, win, cam, pt, film, depth, albedoOnly); });
    }
    pool.WaitForCompletion();
}

void WriteFile(const std::string& path, const void* data, size_t bytes)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); std::exit(2); }
    std::fwrite(data, 1, bytes, f);
    std::fclose(f);
}

int ModeRender(int argc, char** argv)
{
    if (argc < 9) return 2;
    SceneFile s = LoadScene(argv[2]);
    std::string out = argv[3];
    int spp = std::atoi(argv[4]), depth = std::atoi(argv[5]);
    bool albedoOnly = std::atoi(argv[6]) != 0;
    int threads = std::atoi(argv[7]), tasks = std::atoi(argv[8]);
    auto film = std::make_shared<Film>(s.W, s.H);
    ThreadPool pool((size_t)threads);
    glm::ivec2 win(s.W, s.H);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < spp; ++i)
        RayTraceParallelRef(pool, tasks, win, s.camera, s.tracer, film, depth, albedoOnly);
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const auto& acc = film->Accum();
    const auto& smp = film->Samples();
    WriteFile(out + ".accum.f32", acc.data(), acc.size() * sizeof(glm::vec3));
    WriteFile(out + ".samples.u32", smp.data(), smp.size() * sizeof(uint32_t));
    const auto& rgba = film->ResolveToRGBA8();
    WriteFile(out + ".rgba8", rgba.data(), rgba.size());
    double samples = double(s.W) * s.H * spp;
    std::printf("{\"seconds\": %.6f, \"samples\": %.0f, \"msamples_per_s\": %.6f, \"threads\": %d, \"tasks\": %d}\n",
                sec, samples, samples / sec * 1e-6, threads, tasks);
    return 0;
}

int ModeRays(int argc, char** argv)
{
    if (argc < 4) return 2;
    SceneFile s = LoadScene(argv[2]);
    std::vector<float> out((size_t)s.W * s.H * 6);
    for (int y = 0; y < s.H; ++y)
        for (int x = 0; x < s.W; ++x) {
            Ray r = s.camera->GetRay({ x, y }, { s.W, s.H });
            float* o = &out[((size_t)y * s.W + x) * 6];
            o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
            o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
        }
    WriteFile(argv[3], out.data(), out.size() * sizeof(float));
    return 0;
}

int ModeHits(int argc, char** argv)
{
    if (argc < 4) return 2;
    SceneFile s = LoadScene(argv[2]);
    const int REC = 20;
    std::vector<float> out((size_t)s.W * s.H * REC, 0.f);
    for (int y = 0; y < s.H; ++y)
        for (int x = 0; x < s.W; ++x) {
            Ray ray = s.camera->GetRay({ x, y }, { s.W, s.H });
            // closest-hit loop of PathTracer.cpp:38-57
            const float kTMin = 1e-4f, kTMax = 1e30f;
            Hit best{}; bool hitSomething = false; float closestT = kTMax;
            Material bestMat;
            for (const auto& obj : s.tracer->GetRayObjects()) {
                Hit h{};
                if (obj->RayIntersect(ray, kTMin, closestT, h)) {
                    if (h.t >= closestT) continue;
                    hitSomething = true; closestT = h.t; best = h;
                    bestMat = *h.mat; // Mesh hands out a thread-local scratch: copy now
                }
            }
            float* o = &out[((size_t)y * s.W + x) * REC];
            o[0] = hitSomething ? 1.f : 0.f;
            if (!hitSomething) continue;
            o[1] = best.t;
            o[2] = best.p.x; o[3] = best.p.y; o[4] = best.p.z;
            o[5] = best.n.x; o[6] = best.n.y; o[7] = best.n.z;
            o[8] = best.frontFace ? 1.f : 0.f;
            o[9] = bestMat.albedo.x; o[10] = bestMat.albedo.y; o[11] = bestMat.albedo.z;
            o[12] = bestMat.roughness; o[13] = bestMat.metallic;
            o[14] = bestMat.emissionColour.x; o[15] = bestMat.emissionColour.y; o[16] = bestMat.emissionColour.z;
            o[17] = bestMat.emissionStrength; o[18] = bestMat.IOR; o[19] = bestMat.transmission;
        }
    WriteFile(argv[3], out.data(), out.size() * sizeof(float));
    return 0;
}

int ModeFilm(int argc, char** argv)
{
    if (argc < 5) return 2;
    int n = std::atoi(argv[3]);
    std::vector<float> in((size_t)n * 3);
    FILE* f = std::fopen(argv[2], "rb");
    if (!f || std::fread(in.data(), sizeof(float), in.size(), f) != in.size()) return 2;
    std::fclose(f);
    std::string out = argv[4];
    Film film(n, 1);
    for (int i = 0; i < n; ++i) {
        glm::vec3 c(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
        film.AddSample(i, 0, c);
        film.AddSample(i, 0, c * 0.5f);
    }
    WriteFile(out + ".accum.f32", film.Accum().data(), (size_t)n * 12);
    WriteFile(out + ".samples.u32", film.Samples().data(), (size_t)n * 4);
    std::vector<float> avg((size_t)n * 3);
    for (int i = 0; i < n; ++i) { glm::vec3 a = film.AverageAt(i, 0); avg[3*i]=a.x; avg[3*i+1]=a.y; avg[3*i+2]=a.z; }
    WriteFile(out + ".avg.f32", avg.data(), avg.size() * 4);
    const ColourSpace cs[2] = { ColourSpace::Linear, ColourSpace::sRGB };
    const ToneMap tm[2] = { ToneMap::None, ToneMap::Reinhard };
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            film.SetColourSpace(cs[a]); film.SetToneMap(tm[b]);
            const auto& px = film.ResolveToRGBA8();
            WriteFile(out + ".cs" + std::to_string(a) + ".tm" + std::to_string(b) + ".rgba8", px.data(), px.size());
        }
    return 0;
}

int ModeImages(int argc, char** argv)
{
    if (argc < 4) return 2;
    ModelLoader loader{ std::string(argv[2]) };
    std::string out = argv[3];
    std::ofstream list(out + ".txt");
    int i = 0;
    for (const auto& im : loader.GetEmbeddedImages()) {
        list << im.width << " " << im.height << " " << im.channels << "\n";
        WriteFile((out + "." + std::to_string(i++) + ".u8").c_str(), im.data.data(), im.data.size());
    }
    return 0;
}

} // namespace

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: ref_oracle render|rays|hits|film|images ...\n"); return 2; }
    std::string mode = argv[1];
    int rc = 2;
    try {
        if (mode == "render") rc = ModeRender(argc, argv);
        else if (mode == "rays") rc = ModeRays(argc, argv);
        else if (mode == "hits") rc = ModeHits(argc, argv);
        else if (mode == "film") rc = ModeFilm(argc, argv);
        else if (mode == "images") rc = ModeImages(argc, argv);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ref_oracle: exception: %s\n", e.what());
        return 3;
    }
    if (rc == 2) std::fprintf(stderr, "ref_oracle: bad arguments for mode %s\n", mode.c_str());
    return rc;
}
