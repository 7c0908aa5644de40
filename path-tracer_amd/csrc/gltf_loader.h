// Minimal glTF 2.0 / GLB ingest. Replaces ModelLoader::LoadGLTF (ModelLoader.h:171-420), which sits on
// tinygltf + nlohmann json + stb_image: same accepted inputs for the fields the tracer uses (.glb/.gltf,
// TRIANGLES, float VEC3 POSITION, indexed; optional float NORMAL / TEXCOORD_0; PBR factors, alpha mode,
// KHR_materials_transmission / _ior; images as 8-bit RGBA), same face order (file order of meshes and
// primitives, node transforms ignored), same merging of material groups by material name.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/ptamd.h"

namespace ptamd {

struct Model {
    std::vector<float> positions, normals, texcoords; // 9 / 9 / 6 floats per face
    std::vector<int32_t> group;
    std::vector<ptamd_pbr> groups;
    std::vector<std::string> group_names;
    std::vector<std::vector<uint8_t>> image_data;   // RGBA8
    std::vector<ptamd_image> images;                // data pointers into image_data
    uint32_t num_faces() const { return (uint32_t)group.size(); }
};

enum : uint32_t {
    kLoadNodeTransforms = 1u, // apply the glTF scene graph (the reference ignores it, ModelLoader.h:204-206)
    kLoadObj = 2u,            // accept Wavefront .obj files (the reference rejects them, ModelLoader.h:139-142)
};

// Throws std::runtime_error with the reference's messages where it has them. flags = 0 is the reference's loader.
void load_gltf(const std::string& path, Model& out, uint32_t flags = 0);
void load_obj(const std::string& path, Model& out);
// One encoded image (PNG or JPEG, told apart by signature) as RGBA8, row 0 first: what tinygltf's LoadImageData gets from
// stb_image with four requested channels (contrib/include/tiny_gltf.h:2617-2722). Throws std::runtime_error.
void decode_image(const uint8_t* data, size_t size, int& w, int& h, std::vector<uint8_t>& rgba);

} // namespace ptamd
