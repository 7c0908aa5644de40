// Build-container check of integration/GpuBackend.cpp (`make -C oracle binding`): builds the reference's own scene
// objects (the literals of main.cpp:60-63, 229-305 in miniature), pushes them through the binding, and checks what can
// be checked on the host. With a GPU it also renders 24 frames; without one it expects the library's "no device" error.
// Links the reference's objects and libptamd.so; test infrastructure only.
#include <cstdio>
#include <cstring>
#include <thread>

#include "GpuBackend.h"

static int fails = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAILED: %s (line %d)\n", #cond, __LINE__); ++fails; } } while (0)

int main(int argc, char** argv)
{
    const std::string model = argc > 1 ? argv[1] : "tests/golden/cornell_mesh.glb";
    const int W = 64, H = 36;
    GpuBackend gpu(W, H, 0);

    PathTracer tracer;
    Material red; red.albedo = glm::vec3(0.651f, 0.051f, 0.051f);
    auto sphere = std::make_shared<Sphere>("Sphere", glm::vec3(1.8f, 1.0f, 3.5f), 1.0f);
    sphere->SetMaterial(red);
    auto box = std::make_shared<Box>("Light");
    box->SetPosition(glm::vec3(2.78f, 4.5f, 2.8f)); box->SetSize(glm::vec3(1.3f, 0.006f, 1.05f));
    Material light; light.albedo = glm::vec3(0.0f); light.emissionColour = glm::vec3(1.0f, 0.84f, 0.59f); light.emissionStrength = 15.0f;
    box->SetMaterial(light);
    auto mesh = std::make_shared<Mesh>(model, "Room");
    tracer.AddRayObject(sphere); tracer.AddRayObject(box); tracer.AddRayObject(mesh);

    const ptamd_material m = GpuBackend::ToC(sphere->GetMaterial());
    EXPECT(m.albedo[0] == 0.651f && m.roughness == 1.0f && m.ior == 1.5f && m.transmission == 0.0f);

    Camera camera(glm::ivec2(W, H));
    camera.SetFov(45.0f); camera.SetPosition(glm::vec3(2.78f, 2.73f, -8.0f)); camera.SetRotation(glm::vec3(0.0f, 180.0f, 0.0f));
    const ptamd_camera c = GpuBackend::ToC(camera);
    EXPECT(c.fov_deg == 45.0f && c.position[2] == -8.0f && c.rotation_deg[1] == 180.0f && c.near_plane == 0.1f && c.far_plane == 100.0f);

    bool threw = false;
    try { gpu.Upload(tracer); } catch (const std::exception& e) { threw = true; EXPECT(std::strstr(e.what(), "not registered") != nullptr); }
    EXPECT(threw); // a Mesh without its path is refused by the binding itself

    gpu.RegisterMesh(mesh, model);
    const bool haveGpu = ptamd_init(0) == PTAMD_OK;
    try {
        gpu.Upload(tracer);
        EXPECT(haveGpu);
        std::vector<uint8_t> rgba;
        for (int f = 0; f < 24; ++f) gpu.RenderFrame(camera, 5, false, true, true, rgba);
        EXPECT(gpu.Frames() == 24 && rgba.size() == (size_t)W * H * 4);
        gpu.SetSamplesPerFrame(64); // one call, 64 passes: the sample indices go on where the single passes stopped
        gpu.RenderFrame(camera, 5, false, true, true, rgba);
        EXPECT(gpu.Frames() == 88);
        size_t lit = 0;
        for (size_t p = 0; p < (size_t)W * H; ++p) lit += (rgba[4 * p] | rgba[4 * p + 1] | rgba[4 * p + 2]) != 0;
        EXPECT(lit > (size_t)W * H / 4);
        std::printf("rendered 24 frames on the GPU, %zu of %d pixels lit\n", lit, W * H);
    } catch (const std::exception& e) {
        EXPECT(!haveGpu); // without a device the commit reports it; the three objects were described before that
        std::printf("no GPU here: %s\n", e.what());
    }
    EXPECT(gpu.ObjectCount() == 3);

    // the reference's own loader output as triangle arrays (no file on the ptamd side)
    ModelLoader loaded(model);
    const int index = gpu.AddLoadedModel(loaded, glm::vec3(0.0f), glm::vec3(0.0f), glm::vec3(1.0f));
    EXPECT(index == 3 && gpu.ObjectCount() == 4);
    std::printf("binding check: %d failure(s), %zu faces from the reference's ModelLoader\n", fails, loaded.GetFaces().size());
    return fails ? 1 : 0;
}
