// Reference-side binding of the ptamd C ABI: the one translation unit a maintainer of jtothec67/Path-Tracer adds next to
// main.cpp to send the frame render to the GPU. It keeps the reference's own classes as the scene description
// (PathTracer.h:13-18, RayObject.h:41-51, Sphere.h:20-21, Box.h:17-18 / :41-42, Mesh.h:14-22, Camera.h:17-30) and uses
// only accessors those headers already have. Compiles against /root/reference/src + include/ptamd.h
// (`make -C oracle binding`); nothing of it is part of libptamd.so.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "PathTracer/Box.h"
#include "PathTracer/Camera.h"
#include "PathTracer/Mesh.h"
#include "PathTracer/PathTracer.h"
#include "PathTracer/Sphere.h"
#include "ptamd.h"

class GpuBackend
{
public:
    // Film(w, h) of main.cpp:55 lives on the GPU. device: HIP device index.
    GpuBackend(int _width, int _height, int _device = 0);
    ~GpuBackend();
    GpuBackend(const GpuBackend&) = delete;
    GpuBackend& operator=(const GpuBackend&) = delete;

    // Mesh keeps its file path and its ModelLoader private (Mesh.h:24-27), so the application names the file of each
    // Mesh it constructs (main.cpp:307-311 has the literal at hand): RegisterMesh(mesh, "../assets/models/car.glb").
    void RegisterMesh(const std::shared_ptr<Mesh>& _mesh, const std::string& _filePath);

    // After building or editing the scene (main.cpp:218-311, the ImGui editors): flattens PathTracer::GetRayObjects()
    // in its order, builds the trees, uploads. Resets the film, as the reference's "Reset" does (main.cpp:458-465).
    void Upload(PathTracer& _pathTracer);

    // The same data path without a file: a ModelLoader the application already holds (ModelLoader.h:95-128) becomes
    // the triangle arrays of ptamd_scene_add_mesh. Returns the object index.
    int AddLoadedModel(const ModelLoader& _model, const glm::vec3& _position, const glm::vec3& _rotation, const glm::vec3& _scale);

    // Replaces RayTraceParallel(...) at main.cpp:568 and film->ResolveToRGBA8() at main.cpp:573: one more sample per
    // pixel, then the display buffer (feeds Window::DrawScreen unchanged).
    void RenderFrame(Camera& _camera, int _depth, bool _albedoOnly, bool _sRGB, bool _reinhard, std::vector<uint8_t>& _rgba8);

    void ResetAccumulation();             // Film::Reset + frame counter (main.cpp:361-362)
    void Resize(int _width, int _height); // Film::Resize (main.cpp:395-404)
    uint32_t Frames() const { return mFrame; }
    // Samples per pixel that one RenderFrame call adds (default 1, the reference's pass per UI frame, main.cpp:17-28). The
    // library's fast kernel takes launches of 16 and more: at 1080p a frame of 16 takes 9 to 40 ms.
    void SetSamplesPerFrame(uint32_t _n) { mSamplesPerFrame = _n ? _n : 1u; }
    int ObjectCount() const;

    static ptamd_material ToC(const Material& _m);
    static ptamd_camera ToC(Camera& _camera);

private:
    ptamd_scene* mScene = nullptr;
    ptamd_film* mFilm = nullptr;
    uint32_t mFrame = 0;           // samples per pixel rendered so far = index of the next one
    uint32_t mSamplesPerFrame = 1;
    int mWidth = 0, mHeight = 0;
    std::map<const RayObject*, std::string> mMeshFiles;
    static void Check(int _rc);
};
