// C++ host mirror of the reference's scene / camera / film classes over the ptamd C ABI (include/ptamd.h).
//
// Same names, argument meaning and error behaviour as the reference's own classes, so code written against
// them moves over by switching the include and the namespace:
//   Material (RayObject.h:10-20), RayObject (RayObject.h:32-63), Sphere (Sphere.h), Box (Box.h), Mesh (Mesh.h),
//   Camera (Camera.h), Film (Film.h), PathTracer (PathTracer.h), RayTraceParallel (main.cpp:30-47).
// Differences that come with the GPU: the scene is uploaded by PathTracer::Commit() (the reference reads
// its objects in place), errors of the C ABI surface as std::runtime_error exactly where the reference
// throws (loader, empty mesh), and a frame pass takes a frame number instead of a thread pool.
// Header-only; link with libptamd.so. No glm dependency: vec3 is a plain struct.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ptamd.h"

namespace ptamd_host {

struct vec3 { float x = 0.f, y = 0.f, z = 0.f; vec3() = default; vec3(float a, float b, float c) : x(a), y(b), z(c) {} explicit vec3(float s) : x(s), y(s), z(s) {} };
struct ivec2 { int x = 0, y = 0; ivec2() = default; ivec2(int a, int b) : x(a), y(b) {} };

inline void check(int rc)
{
    if (rc < 0) throw std::runtime_error(std::string("ptamd: ") + ptamd_last_error());
}
inline void Init(int device = 0) { check(ptamd_init(device)); }

struct Material {
    vec3 albedo = vec3(1.0f);
    float roughness = 1.f;
    float metallic = 0.0f;
    vec3 emissionColour = vec3(1.0f);
    float emissionStrength = 0.0f;
    float IOR = 1.5f;
    float transmission = 0.f;
    ptamd_material c() const
    {
        ptamd_material m;
        m.albedo[0] = albedo.x; m.albedo[1] = albedo.y; m.albedo[2] = albedo.z;
        m.roughness = roughness; m.metallic = metallic;
        m.emission_colour[0] = emissionColour.x; m.emission_colour[1] = emissionColour.y; m.emission_colour[2] = emissionColour.z;
        m.emission_strength = emissionStrength; m.ior = IOR; m.transmission = transmission;
        return m;
    }
};

class PathTracer;

class RayObject {
public:
    virtual ~RayObject() = default;
    void SetName(const std::string& n) { mName = n; }
    const std::string& GetName() { return mName; }
    void SetPosition(const vec3& p) { mPosition = p; Push(); }
    vec3 GetPosition() { return mPosition; }
    void SetRotation(const vec3& r) { mRotation = r; Push(); }
    vec3 GetRotation() { return mRotation; }
    virtual void SetMaterial(const Material& m);
    Material GetMaterial() { return mMaterial; }

protected:
    friend class PathTracer;
    virtual int Add(ptamd_scene* s) = 0;
    virtual bool Extent(float out[3]) const { (void)out; return false; }
    void Push();
    std::string mName = "Object";
    vec3 mPosition, mRotation;
    Material mMaterial;
    ptamd_scene* mScene = nullptr;
    int mIndex = -1;
};

class Sphere : public RayObject {
public:
    explicit Sphere(std::string name) { mName = std::move(name); }
    Sphere(std::string name, vec3 position) { mName = std::move(name); mPosition = position; }
    Sphere(std::string name, vec3 position, float radius) { mName = std::move(name); mPosition = position; mRadius = radius; }
    Sphere(std::string name, vec3 position, float radius, vec3 albedo) { mName = std::move(name); mPosition = position; mRadius = radius; mMaterial.albedo = albedo; }
    void SetRadius(float r) { mRadius = r; if (mScene) check(ptamd_scene_set_radius(mScene, mIndex, r)); }
    float GetRadius() { return mRadius; }

private:
    int Add(ptamd_scene* s) override { ptamd_material m = mMaterial.c(); return ptamd_scene_add_sphere(s, &mPosition.x, mRadius, &m); }
    float mRadius = 1.f;
};

class Box : public RayObject {
public:
    explicit Box(std::string name) { mName = std::move(name); }
    void SetSize(const vec3& s) { mSize = s; Push(); }
    vec3 GetSize() { return mSize; }

private:
    int Add(ptamd_scene* s) override { ptamd_material m = mMaterial.c(); return ptamd_scene_add_box(s, &mPosition.x, &mRotation.x, &mSize.x, &m); }
    bool Extent(float out[3]) const override { out[0] = mSize.x; out[1] = mSize.y; out[2] = mSize.z; return true; }
    vec3 mSize = vec3(1.0f);
};

class Mesh : public RayObject {
public:
    // Mesh.cpp:7-13: loads the model at construction and throws what the loader throws. loadFlags (PTAMD_LOAD_*)
    // turn on loader options the reference does not have; 0 is the reference's loader.
    Mesh(const std::string& filePath, std::string name, uint32_t loadFlags = 0) : mPath(filePath)
    {
        mName = std::move(name);
        ptamd_model* m = nullptr;
        check(ptamd_model_load_ex(filePath.c_str(), loadFlags, &m));
        mModel.reset(m, ptamd_model_free);
        ptamd_mesh_desc d;
        check(ptamd_model_desc(m, &d));
        if (d.num_faces == 0) throw std::runtime_error("Mesh: no faces in model");
    }
    void SetScale(const vec3& s) { mScale = s; Push(); }
    vec3 GetScale() { return mScale; }
    void SetMaterial(const Material&) override { throw std::runtime_error("Mesh materials come from the model's material groups"); }
    uint32_t GetMaterialGroupCount() const { ptamd_mesh_desc d; check(ptamd_model_desc(mModel.get(), &d)); return d.num_groups; }
    // the per-group editor of Mesh.cpp:267-284
    void SetMaterialGroup(uint32_t group, const ptamd_pbr& pbr) { if (mScene) check(ptamd_scene_set_mesh_group(mScene, mIndex, group, &pbr)); }

private:
    int Add(ptamd_scene* s) override
    {
        ptamd_mesh_desc d;
        check(ptamd_model_desc(mModel.get(), &d));
        return ptamd_scene_add_mesh(s, &d, &mPosition.x, &mRotation.x, &mScale.x);
    }
    bool Extent(float out[3]) const override { out[0] = mScale.x; out[1] = mScale.y; out[2] = mScale.z; return true; }
    std::string mPath;
    std::shared_ptr<ptamd_model> mModel;
    vec3 mScale = vec3(1.0f);
};

class PathTracer {
public:
    PathTracer() { check(ptamd_scene_create(&mScene)); }
    ~PathTracer() { ptamd_scene_destroy(mScene); }
    PathTracer(const PathTracer&) = delete;
    PathTracer& operator=(const PathTracer&) = delete;
    const std::vector<std::shared_ptr<RayObject>>& GetRayObjects() { return rayObjects; }
    void AddRayObject(std::shared_ptr<RayObject> o)
    {
        int idx = o->Add(mScene);
        check(idx);
        o->mScene = mScene; o->mIndex = idx;
        rayObjects.push_back(std::move(o));
    }
    int GetSizeOfRayObjects() { return (int)rayObjects.size(); }
    void ClearScene()
    {
        check(ptamd_scene_clear(mScene));
        for (auto& o : rayObjects) { o->mScene = nullptr; o->mIndex = -1; }
        rayObjects.clear();
    }
    void SetBackgroundColour(const vec3& c) { check(ptamd_scene_set_background(mScene, &c.x)); }
    // Cubemap miss radiance (not in the reference): six faceSize x faceSize RGBA8 faces, +x -x +y -y +z -z; null removes it.
    void SetSkybox(int faceSize, const uint8_t* const faces[6], float intensity = 1.0f) { check(ptamd_scene_set_skybox(mScene, faceSize, faces, intensity)); }
    void SetSkyboxFloat(int faceSize, const float* const faces[6]) { check(ptamd_scene_set_skybox_f32(mScene, faceSize, faces)); } // float rgb radiance per texel
    // Flatten + BVH + upload; call after building or editing the scene, before the next frame.
    void Commit() { check(ptamd_scene_commit(mScene)); }
    ptamd_scene* Handle() { return mScene; }

private:
    ptamd_scene* mScene = nullptr;
    std::vector<std::shared_ptr<RayObject>> rayObjects;
};

inline void RayObject::Push()
{
    if (!mScene) return;
    float e[3];
    bool has = Extent(e);
    check(ptamd_scene_set_transform(mScene, mIndex, &mPosition.x, &mRotation.x, has ? e : nullptr));
}
inline void RayObject::SetMaterial(const Material& m)
{
    mMaterial = m;
    if (mScene) { ptamd_material c = m.c(); check(ptamd_scene_set_material(mScene, mIndex, &c)); }
}

class Camera {
public:
    explicit Camera(ivec2 winSize) : mLastWinSize(winSize) {}
    Camera(vec3 position, vec3 rotation, ivec2 winSize) : mPosition(position), mRotation(rotation), mLastWinSize(winSize) {}
    void SetPosition(vec3 p) { mPosition = p; }
    vec3 GetPosition() { return mPosition; }
    void SetRotation(vec3 r) { mRotation = r; }
    vec3 GetRotation() { return mRotation; }
    void SetFov(float f) { mFov = f; }
    float GetFov() { return mFov; }
    void SetNearPlane(float n) { mNearPlane = n; }
    float GetNearPlane() { return mNearPlane; }
    void SetFarPlane(float f) { mFarPlane = f; }
    float GetFarPlane() { return mFarPlane; }
    ptamd_camera c() const
    {
        ptamd_camera k;
        k.position[0] = mPosition.x; k.position[1] = mPosition.y; k.position[2] = mPosition.z;
        k.rotation_deg[0] = mRotation.x; k.rotation_deg[1] = mRotation.y; k.rotation_deg[2] = mRotation.z;
        k.fov_deg = mFov; k.near_plane = mNearPlane; k.far_plane = mFarPlane;
        return k;
    }

private:
    vec3 mPosition, mRotation;
    float mFov = 40.f, mNearPlane = 0.1f, mFarPlane = 100.f;
    ivec2 mLastWinSize;
};

enum class ColourSpace { Linear, sRGB };
enum class ToneMap { None, Reinhard };

class Film {
public:
    Film(int w, int h) { check(ptamd_film_create(w, h, &mFilm)); }
    ~Film() { ptamd_film_destroy(mFilm); }
    Film(const Film&) = delete;
    Film& operator=(const Film&) = delete;
    void Resize(int w, int h) { check(ptamd_film_resize(mFilm, w, h)); }
    void Reset() { check(ptamd_film_reset(mFilm)); }
    int Width() const { int w, h; check(ptamd_film_size(mFilm, &w, &h)); return w; }
    int Height() const { int w, h; check(ptamd_film_size(mFilm, &w, &h)); return h; }
    int PixelCount() const { return Width() * Height(); }
    void SetColourSpace(ColourSpace c) { mColourSpace = c; }
    ColourSpace GetColourSpace() const { return mColourSpace; }
    void SetToneMap(ToneMap t) { mToneMap = t; }
    ToneMap GetToneMap() const { return mToneMap; }
    // Film::Accum / Film::Samples: copies of the device buffers (3 floats / 1 count per pixel, row 0 = bottom)
    std::vector<float> Accum() { std::vector<float> a((size_t)PixelCount() * 3); check(ptamd_film_read(mFilm, a.data(), nullptr)); return a; }
    std::vector<uint32_t> Samples() { std::vector<uint32_t> s((size_t)PixelCount()); check(ptamd_film_read(mFilm, nullptr, s.data())); return s; }
    // Film::AverageAt for every pixel
    std::vector<float> Average() { std::vector<float> a((size_t)PixelCount() * 3); check(ptamd_film_average(mFilm, a.data())); return a; }
    const std::vector<std::uint8_t>& ResolveToRGBA8()
    {
        mDisplay8.resize((size_t)PixelCount() * 4);
        check(ptamd_film_resolve_rgba8(mFilm, mToneMap == ToneMap::Reinhard, mColourSpace == ColourSpace::sRGB, mDisplay8.data()));
        return mDisplay8;
    }
    ptamd_film* Handle() { return mFilm; }

private:
    ptamd_film* mFilm = nullptr;
    std::vector<std::uint8_t> mDisplay8;
    ColourSpace mColourSpace = ColourSpace::sRGB;
    ToneMap mToneMap = ToneMap::Reinhard;
};

// main.cpp:30-47: one sample per pixel over the whole frame; `frame` is the pass number (the sample index).
inline void RayTraceParallel(ivec2 /*winSize*/, Camera& camera, PathTracer& pathTracer, Film& film, int depth, bool albedoOnly,
                             uint32_t frame = 0, uint32_t seed = 0x5EED)
{
    ptamd_camera c = camera.c();
    check(ptamd_render(pathTracer.Handle(), &c, film.Handle(), frame, 1, depth, albedoOnly ? 1 : 0, seed, nullptr, 0));
}
// spp passes in one launch.
inline void Render(Camera& camera, PathTracer& pathTracer, Film& film, uint32_t sppBegin, uint32_t sppCount, int depth,
                   bool albedoOnly = false, uint32_t seed = 0x5EED, const ptamd_rect* rect = nullptr, uint32_t flags = 0)
{
    ptamd_camera c = camera.c();
    check(ptamd_render(pathTracer.Handle(), &c, film.Handle(), sppBegin, sppCount, depth, albedoOnly ? 1 : 0, seed, rect, flags));
}
// One rank's interleaved square tiles (tile index % tileStride == tilePhase) in one launch.
inline void RenderTiles(Camera& camera, PathTracer& pathTracer, Film& film, uint32_t sppBegin, uint32_t sppCount, int depth,
                        int tileSize, int tilePhase, int tileStride, bool albedoOnly = false, uint32_t seed = 0x5EED, uint32_t flags = 0)
{
    ptamd_camera c = camera.c();
    check(ptamd_render_tiles(pathTracer.Handle(), &c, film.Handle(), sppBegin, sppCount, depth, albedoOnly ? 1 : 0, seed,
                             tileSize, tilePhase, tileStride, flags));
}

} // namespace ptamd_host
