// See GpuBackend.h. Errors of the C ABI become the std::runtime_error the reference already throws from its loader
// (ModelLoader.h:137-142, Mesh.cpp:293).
#include "GpuBackend.h"

#include <stdexcept>

void GpuBackend::Check(int _rc)
{
    if (_rc < 0) throw std::runtime_error(std::string("ptamd: ") + ptamd_last_error());
}

ptamd_material GpuBackend::ToC(const Material& _m)
{
    ptamd_material m;
    m.albedo[0] = _m.albedo.r; m.albedo[1] = _m.albedo.g; m.albedo[2] = _m.albedo.b;
    m.roughness = _m.roughness;
    m.metallic = _m.metallic;
    m.emission_colour[0] = _m.emissionColour.r; m.emission_colour[1] = _m.emissionColour.g; m.emission_colour[2] = _m.emissionColour.b;
    m.emission_strength = _m.emissionStrength;
    m.ior = _m.IOR;
    m.transmission = _m.transmission;
    return m;
}

ptamd_camera GpuBackend::ToC(Camera& _camera)
{
    const glm::vec3 p = _camera.GetPosition(), r = _camera.GetRotation();
    ptamd_camera c;
    c.position[0] = p.x; c.position[1] = p.y; c.position[2] = p.z;
    c.rotation_deg[0] = r.x; c.rotation_deg[1] = r.y; c.rotation_deg[2] = r.z;
    c.fov_deg = _camera.GetFov();
    c.near_plane = _camera.GetNearPlane();
    c.far_plane = _camera.GetFarPlane();
    return c;
}

GpuBackend::GpuBackend(int _width, int _height, int _device) : mWidth(_width), mHeight(_height)
{
    Check(ptamd_scene_create(&mScene));
    // Without a usable GPU ptamd_init fails (there is no CPU fallback); the scene handle above is host-side state and
    // stays usable for describing the scene, so the error surfaces at Upload / RenderFrame with the library's message.
    if (ptamd_init(_device) == PTAMD_OK) Check(ptamd_film_create(_width, _height, &mFilm));
}

GpuBackend::~GpuBackend()
{
    if (mFilm) ptamd_film_destroy(mFilm);
    if (mScene) ptamd_scene_destroy(mScene);
}

void GpuBackend::RegisterMesh(const std::shared_ptr<Mesh>& _mesh, const std::string& _filePath)
{
    mMeshFiles[_mesh.get()] = _filePath;
}

int GpuBackend::ObjectCount() const { return ptamd_scene_object_count(mScene); }

void GpuBackend::Upload(PathTracer& _pathTracer)
{
    Check(ptamd_scene_clear(mScene));
    for (const std::shared_ptr<RayObject>& o : _pathTracer.GetRayObjects()) {
        const glm::vec3 p = o->GetPosition(), r = o->GetRotation();
        const ptamd_material m = ToC(o->GetMaterial());
        if (Sphere* s = dynamic_cast<Sphere*>(o.get())) {
            Check(ptamd_scene_add_sphere(mScene, &p.x, s->GetRadius(), &m));
        } else if (Box* b = dynamic_cast<Box*>(o.get())) {
            const glm::vec3 size = b->GetSize();
            Check(ptamd_scene_add_box(mScene, &p.x, &r.x, &size.x, &m));
        } else if (Mesh* me = dynamic_cast<Mesh*>(o.get())) {
            const auto it = mMeshFiles.find(me);
            if (it == mMeshFiles.end()) throw std::runtime_error("GpuBackend: Mesh '" + me->GetName() + "' was not registered with its file path");
            const glm::vec3 scale = me->GetScale();
            Check(ptamd_scene_add_mesh_file(mScene, it->second.c_str(), &p.x, &r.x, &scale.x));
        } else {
            throw std::runtime_error("GpuBackend: unknown RayObject type '" + o->GetName() + "'");
        }
    }
    Check(ptamd_scene_commit(mScene));
    ResetAccumulation();
}

int GpuBackend::AddLoadedModel(const ModelLoader& _model, const glm::vec3& _position, const glm::vec3& _rotation, const glm::vec3& _scale)
{
    const std::vector<ModelLoader::Face>& faces = _model.GetFaces();
    std::vector<float> pos, nrm, uv;
    std::vector<int32_t> group;
    pos.reserve(faces.size() * 9); nrm.reserve(faces.size() * 9); uv.reserve(faces.size() * 6); group.reserve(faces.size());
    for (const ModelLoader::Face& f : faces) {
        for (const ModelLoader::Vertex* v : { &f.a, &f.b, &f.c }) {
            pos.insert(pos.end(), { v->position.x, v->position.y, v->position.z });
            nrm.insert(nrm.end(), { v->normal.x, v->normal.y, v->normal.z });
            uv.insert(uv.end(), { v->texcoord.x, v->texcoord.y });
        }
        group.push_back(f.materialGroup);
    }
    std::vector<ptamd_pbr> groups;
    for (const ModelLoader::MaterialGroup& g : _model.GetMaterialGroups()) {
        const ModelLoader::PBRMaterial& s = g.pbr;
        ptamd_pbr d;
        ptamd_pbr_default(&d);
        d.base_color_tex = s.baseColorTexIndex; d.normal_tex = s.normalTexIndex; d.metallic_roughness_tex = s.metallicRoughnessTexIndex;
        d.emissive_tex = s.emissiveTexIndex; d.transmission_tex = s.transmissionTexIndex;
        for (int k = 0; k < 4; ++k) d.base_color_factor[k] = s.baseColorFactor[k];
        d.metallic_factor = s.metallicFactor; d.roughness_factor = s.roughnessFactor;
        d.alpha_mode = (int32_t)s.alphaMode; d.alpha_cutoff = s.alphaCutoff; d.normal_scale = s.normalScale;
        for (int k = 0; k < 3; ++k) d.emissive_factor[k] = s.emissiveFactor[k];
        d.transmission_factor = s.transmissionFactor; d.ior = s.ior;
        groups.push_back(d);
    }
    std::vector<ptamd_image> images;
    for (const ModelLoader::EmbeddedImage& im : _model.GetEmbeddedImages()) {
        ptamd_image d;
        d.width = im.width; d.height = im.height; d.channels = im.channels; d.data = im.data.data();
        images.push_back(d);
    }
    ptamd_mesh_desc desc;
    desc.num_faces = (uint32_t)faces.size();
    desc.positions = pos.data(); desc.normals = nrm.data(); desc.texcoords = uv.data(); desc.group = group.data();
    desc.num_groups = (uint32_t)groups.size(); desc.groups = groups.data();
    desc.num_images = (uint32_t)images.size(); desc.images = images.data();
    const int index = ptamd_scene_add_mesh(mScene, &desc, &_position.x, &_rotation.x, &_scale.x);
    Check(index);
    return index;
}

void GpuBackend::ResetAccumulation()
{
    if (mFilm) Check(ptamd_film_reset(mFilm));
    mFrame = 0;
}

void GpuBackend::Resize(int _width, int _height)
{
    mWidth = _width; mHeight = _height;
    if (mFilm) Check(ptamd_film_resize(mFilm, _width, _height));
    mFrame = 0;
}

void GpuBackend::RenderFrame(Camera& _camera, int _depth, bool _albedoOnly, bool _sRGB, bool _reinhard, std::vector<uint8_t>& _rgba8)
{
    if (!mFilm) throw std::runtime_error(std::string("ptamd: no film: ") + ptamd_last_error());
    const ptamd_camera c = ToC(_camera);
    // sample indices continue from call to call: the same (seed, index) always gives the same sample, on any number of GPUs
    Check(ptamd_render(mScene, &c, mFilm, mFrame, mSamplesPerFrame, _depth, _albedoOnly ? 1 : 0, 0x5EEDu, nullptr, 0u));
    mFrame += mSamplesPerFrame;
    _rgba8.resize((size_t)mWidth * mHeight * 4);
    Check(ptamd_film_resolve_rgba8(mFilm, _reinhard ? 1 : 0, _sRGB ? 1 : 0, _rgba8.data()));
}
