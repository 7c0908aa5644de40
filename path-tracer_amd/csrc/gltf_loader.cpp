#include "gltf_loader.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>

namespace ptamd {
namespace {

// ------------------------------------------------------------------ tiny JSON value + parser
struct JValue;
using JPtr = std::shared_ptr<JValue>;
struct JValue {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0.0;
    bool is_int = false;
    std::string str;
    std::vector<JPtr> arr;
    std::map<std::string, JPtr> obj;

    const JValue* get(const char* key) const
    {
        if (type != Object) return nullptr;
        auto it = obj.find(key);
        return it == obj.end() ? nullptr : it->second.get();
    }
    bool is_number() const { return type == Number; }
    double number_or(double d) const { return type == Number ? num : d; }
};

struct JParser {
    const char* p; const char* end;
    [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("glTF JSON parse error: ") + what); }
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    JPtr parse()
    {
        ws();
        if (p >= end) fail("unexpected end");
        auto v = std::make_shared<JValue>();
        char c = *p;
        if (c == '{') {
            v->type = JValue::Object; ++p; ws();
            if (p < end && *p == '}') { ++p; return v; }
            for (;;) {
                ws();
                if (p >= end || *p != '"') fail("expected key");
                std::string k = parse_string();
                ws();
                if (p >= end || *p != ':') fail("expected ':'");
                ++p;
                v->obj[k] = parse();
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v->type = JValue::Array; ++p; ws();
            if (p < end && *p == ']') { ++p; return v; }
            for (;;) {
                v->arr.push_back(parse());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v->type = JValue::String; v->str = parse_string();
        } else if (c == 't' && end - p >= 4 && !std::strncmp(p, "true", 4)) { v->type = JValue::Bool; v->b = true; p += 4; }
        else if (c == 'f' && end - p >= 5 && !std::strncmp(p, "false", 5)) { v->type = JValue::Bool; v->b = false; p += 5; }
        else if (c == 'n' && end - p >= 4 && !std::strncmp(p, "null", 4)) { p += 4; }
        else {
            const char* s = p;
            bool integral = true;
            if (p < end && (*p == '-' || *p == '+')) ++p;
            while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '-' || *p == '+')) {
                if (*p == '.' || *p == 'e' || *p == 'E') integral = false;
                ++p;
            }
            if (p == s) fail("unexpected character");
            v->type = JValue::Number; v->num = std::strtod(std::string(s, p).c_str(), nullptr); v->is_int = integral;
        }
        return v;
    }
    std::string parse_string()
    {
        ++p; // opening quote
        std::string out;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                ++p;
                if (p >= end) fail("bad escape");
                switch (*p) {
                case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                case 'u': {
                    if (end - p < 5) fail("bad \\u escape");
                    unsigned cp = (unsigned)std::strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                    p += 4;
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: out += *p; break;
                }
                ++p;
            } else out += *p++;
        }
        if (p >= end) fail("unterminated string");
        ++p;
        return out;
    }
};

// ------------------------------------------------------------------ helpers
std::vector<uint8_t> read_file(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Failed to load GLTF model: " + path);
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    f.seekg(0);
    std::vector<uint8_t> d((size_t)n);
    if (n) f.read(reinterpret_cast<char*>(d.data()), n);
    return d;
}
std::vector<uint8_t> base64_decode(const std::string& s, size_t from)
{
    static int8_t T[256]; static bool init = false;
    if (!init) {
        std::memset(T, -1, sizeof T);
        const char* a = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; ++i) T[(uint8_t)a[i]] = (int8_t)i;
        init = true;
    }
    std::vector<uint8_t> out;
    uint32_t acc = 0; int bits = 0;
    for (size_t i = from; i < s.size(); ++i) {
        int8_t v = T[(uint8_t)s[i]];
        if (v < 0) continue;
        acc = (acc << 6) | (uint32_t)v; bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)((acc >> bits) & 0xFF)); }
    }
    return out;
}
std::string dir_of(const std::string& path)
{
    size_t s = path.find_last_of("/\\");
    return s == std::string::npos ? std::string(".") : path.substr(0, s);
}
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// 8-bit non-interlaced PNG -> RGBA8 (what stb_image hands tinygltf with req_comp = 4, tiny_gltf.h:2653).
void decode_png_rgba(const uint8_t* data, size_t size, int& w, int& h, std::vector<uint8_t>& rgba)
{
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (size < 8 || std::memcmp(data, sig, 8)) throw std::runtime_error("image: only PNG images are supported");
    size_t pos = 8;
    std::vector<uint8_t> idat, plte, trns;
    int depth = 0, ctype = 0, interlace = 0;
    w = h = 0;
    while (pos + 8 <= size) {
        uint32_t n = be32(data + pos);
        const uint8_t* tag = data + pos + 4;
        const uint8_t* body = data + pos + 8;
        if (pos + 12 + n > size) throw std::runtime_error("image: truncated PNG");
        if (!std::memcmp(tag, "IHDR", 4)) { w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (!std::memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + n);
        else if (!std::memcmp(tag, "PLTE", 4)) plte.assign(body, body + n);
        else if (!std::memcmp(tag, "tRNS", 4)) trns.assign(body, body + n);
        else if (!std::memcmp(tag, "IEND", 4)) break;
        pos += 12 + n;
    }
    if (depth != 8 || interlace != 0) throw std::runtime_error("image: only 8-bit non-interlaced PNG is supported");
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch || w <= 0 || h <= 0) throw std::runtime_error("image: unsupported PNG colour type");
    size_t stride = (size_t)w * ch;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size())
        throw std::runtime_error("image: PNG inflate failed");
    std::vector<uint8_t> img(stride * (size_t)h);
    for (int y = 0; y < h; ++y) {
        const uint8_t* in = &raw[(stride + 1) * (size_t)y];
        uint8_t ft = in[0];
        ++in;
        uint8_t* cur = &img[stride * (size_t)y];
        const uint8_t* prev = y ? &img[stride * (size_t)(y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= (size_t)ch ? cur[i - ch] : 0, b = prev ? prev[i] : 0, c = (prev && i >= (size_t)ch) ? prev[i - ch] : 0;
            int pred = 0;
            switch (ft) {
            case 0: pred = 0; break; case 1: pred = a; break; case 2: pred = b; break; case 3: pred = (a + b) >> 1; break;
            case 4: { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                      pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
            default: throw std::runtime_error("image: bad PNG filter");
            }
            cur[i] = (uint8_t)(in[i] + pred);
        }
    }
    rgba.assign((size_t)w * h * 4, 255);
    for (size_t p = 0; p < (size_t)w * h; ++p) {
        uint8_t* o = &rgba[4 * p];
        const uint8_t* s = &img[p * ch];
        switch (ctype) {
        case 0: o[0] = o[1] = o[2] = s[0]; break;
        case 2: o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; break;
        case 3: { size_t k = s[0]; if (3 * k + 2 < plte.size()) { o[0] = plte[3 * k]; o[1] = plte[3 * k + 1]; o[2] = plte[3 * k + 2]; }
                  if (k < trns.size()) o[3] = trns[k]; break; }
        case 4: o[0] = o[1] = o[2] = s[0]; o[3] = s[1]; break;
        case 6: o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3]; break;
        }
    }
}

struct Doc {
    JPtr root;
    std::vector<std::vector<uint8_t>> buffers;
    std::string base;
};

struct View { const uint8_t* ptr; size_t stride; size_t count; int comp; std::string type; };

int comp_size(int ct)
{
    switch (ct) { case 5120: case 5121: return 1; case 5122: case 5123: return 2; case 5125: case 5126: return 4; default: return 0; }
}
int type_count(const std::string& t)
{
    if (t == "SCALAR") return 1; if (t == "VEC2") return 2; if (t == "VEC3") return 3; if (t == "VEC4") return 4;
    if (t == "MAT2") return 4; if (t == "MAT3") return 9; if (t == "MAT4") return 16;
    return 0;
}
const JValue& at(const JValue* arr, size_t i, const char* what)
{
    if (!arr || arr->type != JValue::Array || i >= arr->arr.size()) throw std::runtime_error(std::string("glTF: bad index into ") + what);
    return *arr->arr[i];
}
View accessor_view(const Doc& d, int index)
{
    const JValue& acc = at(d.root->get("accessors"), (size_t)index, "accessors");
    const JValue* bvIdx = acc.get("bufferView");
    if (!bvIdx) throw std::runtime_error("glTF: accessor without bufferView is not supported");
    const JValue& bv = at(d.root->get("bufferViews"), (size_t)bvIdx->num, "bufferViews");
    size_t buf = (size_t)(bv.get("buffer") ? bv.get("buffer")->num : 0);
    if (buf >= d.buffers.size()) throw std::runtime_error("glTF: bad buffer index");
    View v;
    v.comp = (int)(acc.get("componentType") ? acc.get("componentType")->num : 0);
    v.type = acc.get("type") ? acc.get("type")->str : "";
    v.count = (size_t)(acc.get("count") ? acc.get("count")->num : 0);
    size_t off = (size_t)(bv.get("byteOffset") ? bv.get("byteOffset")->num : 0) + (size_t)(acc.get("byteOffset") ? acc.get("byteOffset")->num : 0);
    size_t elem = (size_t)comp_size(v.comp) * (size_t)type_count(v.type);
    size_t bs = (size_t)(bv.get("byteStride") ? bv.get("byteStride")->num : 0);
    v.stride = bs ? bs : elem;
    if (v.count && off + v.stride * (v.count - 1) + elem > d.buffers[buf].size()) throw std::runtime_error("glTF: accessor out of buffer range");
    v.ptr = d.buffers[buf].data() + off;
    return v;
}

int texture_source(const Doc& d, const JValue* texInfo)
{
    if (!texInfo) return -1;
    const JValue* idx = texInfo->get("index");
    if (!idx || !idx->is_number() || idx->num < 0) return -1;
    const JValue& t = at(d.root->get("textures"), (size_t)idx->num, "textures");
    const JValue* src = t.get("source");
    return src && src->is_number() ? (int)src->num : -1;
}

} // namespace

void load_gltf(const std::string& path, Model& out)
{
    // ModelLoader.h:129-142
    std::string ext;
    size_t dot = path.find_last_of('.');
    if (dot != std::string::npos) ext = path.substr(dot + 1);
    if (ext != "glb" && ext != "gltf") throw std::runtime_error("Model only supports .glb/.gltf: " + path);

    Doc d;
    d.base = dir_of(path);
    std::vector<uint8_t> file = read_file(path);
    std::vector<uint8_t> bin;
    std::string json;
    if (file.size() >= 12 && !std::memcmp(file.data(), "glTF", 4)) {
        size_t pos = 12;
        bool gotBin = false;
        while (pos + 8 <= file.size()) {
            uint32_t n, tag;
            std::memcpy(&n, &file[pos], 4); std::memcpy(&tag, &file[pos + 4], 4);
            if (pos + 8 + n > file.size()) throw std::runtime_error("Failed to load GLTF model: " + path);
            if (tag == 0x4E4F534Au) json.assign(reinterpret_cast<const char*>(&file[pos + 8]), n);
            else if (tag == 0x004E4942u && !gotBin) { bin.assign(file.begin() + (long)pos + 8, file.begin() + (long)pos + 8 + n); gotBin = true; }
            pos += 8 + n;
        }
    } else {
        json.assign(reinterpret_cast<const char*>(file.data()), file.size());
    }
    JParser jp{ json.data(), json.data() + json.size() };
    d.root = jp.parse();
    if (d.root->type != JValue::Object) throw std::runtime_error("Failed to load GLTF model: " + path);

    if (const JValue* bufs = d.root->get("buffers")) {
        for (const auto& b : bufs->arr) {
            const JValue* uri = b->get("uri");
            if (!uri) d.buffers.push_back(bin);
            else if (uri->str.rfind("data:", 0) == 0) d.buffers.push_back(base64_decode(uri->str, uri->str.find(',') + 1));
            else d.buffers.push_back(read_file(d.base + "/" + uri->str));
        }
    }

    // Embedded images (ModelLoader.h:187-198): always expanded to 4 channels.
    out = Model();
    if (const JValue* imgs = d.root->get("images")) {
        for (const auto& im : imgs->arr) {
            std::vector<uint8_t> raw;
            if (const JValue* bvI = im->get("bufferView")) {
                const JValue& bv = at(d.root->get("bufferViews"), (size_t)bvI->num, "bufferViews");
                size_t buf = (size_t)(bv.get("buffer") ? bv.get("buffer")->num : 0);
                size_t off = (size_t)(bv.get("byteOffset") ? bv.get("byteOffset")->num : 0);
                size_t len = (size_t)(bv.get("byteLength") ? bv.get("byteLength")->num : 0);
                if (buf >= d.buffers.size() || off + len > d.buffers[buf].size()) throw std::runtime_error("glTF: image bufferView out of range");
                raw.assign(d.buffers[buf].begin() + (long)off, d.buffers[buf].begin() + (long)(off + len));
            } else if (const JValue* uri = im->get("uri")) {
                if (uri->str.rfind("data:", 0) == 0) raw = base64_decode(uri->str, uri->str.find(',') + 1);
                else raw = read_file(d.base + "/" + uri->str);
            }
            int w = 0, h = 0;
            std::vector<uint8_t> rgba;
            decode_png_rgba(raw.data(), raw.size(), w, h, rgba);
            out.image_data.push_back(std::move(rgba));
            ptamd_image pi; pi.width = w; pi.height = h; pi.channels = 4; pi.data = nullptr;
            out.images.push_back(pi);
        }
        for (size_t i = 0; i < out.images.size(); ++i) out.images[i].data = out.image_data[i].data();
    }

    const JValue* mats = d.root->get("materials");
    const JValue* meshes = d.root->get("meshes");
    if (!meshes) return;
    for (const auto& mesh : meshes->arr) {
        const JValue* prims = mesh->get("primitives");
        if (!prims) continue;
        for (const auto& prim : prims->arr) {
            const JValue* mode = prim->get("mode");
            if (mode && (int)mode->num != 4) throw std::runtime_error("Only TRIANGLES are supported");
            const JValue* attrs = prim->get("attributes");
            if (!attrs || !attrs->get("POSITION")) throw std::runtime_error("Missing POSITION");
            View pos = accessor_view(d, (int)attrs->get("POSITION")->num);
            if (pos.type != "VEC3" || pos.comp != 5126) throw std::runtime_error("POSITION must be float vec3");
            View nrm{}; bool hasN = false;
            if (const JValue* a = attrs->get("NORMAL")) { nrm = accessor_view(d, (int)a->num); hasN = nrm.type == "VEC3" && nrm.comp == 5126; }
            View uv{}; bool hasT = false;
            if (const JValue* a = attrs->get("TEXCOORD_0")) { uv = accessor_view(d, (int)a->num); hasT = uv.type == "VEC2" && uv.comp == 5126; }
            const JValue* indices = prim->get("indices");
            if (!indices || indices->num < 0) throw std::runtime_error("Indexed geometry required");
            View iv = accessor_view(d, (int)indices->num);
            if (iv.type != "SCALAR") throw std::runtime_error("indices accessor must be SCALAR");
            if (iv.comp != 5121 && iv.comp != 5123 && iv.comp != 5125) throw std::runtime_error("Unsupported index type");

            int groupIndex = -1;
            const JValue* matI = prim->get("material");
            if (matI && matI->is_number() && matI->num >= 0) { // ModelLoader.h:263-362
                const JValue& mat = at(mats, (size_t)matI->num, "materials");
                std::string name = mat.get("name") ? mat.get("name")->str : std::string();
                for (size_t g = 0; g < out.group_names.size(); ++g) if (out.group_names[g] == name) { groupIndex = (int)g; break; }
                if (groupIndex < 0) {
                    ptamd_pbr def; ptamd_pbr_default(&def);
                    out.groups.push_back(def); out.group_names.push_back(name);
                    groupIndex = (int)out.groups.size() - 1;
                }
                ptamd_pbr& g = out.groups[(size_t)groupIndex];
                const JValue* pbr = mat.get("pbrMetallicRoughness");
                int s;
                if ((s = texture_source(d, pbr ? pbr->get("baseColorTexture") : nullptr)) >= 0) g.base_color_tex = s;
                const JValue* bcf = pbr ? pbr->get("baseColorFactor") : nullptr;
                for (int k = 0; k < 4; ++k) g.base_color_factor[k] = (bcf && bcf->arr.size() == 4) ? (float)bcf->arr[(size_t)k]->num : 1.0f;
                if ((s = texture_source(d, pbr ? pbr->get("metallicRoughnessTexture") : nullptr)) >= 0) g.metallic_roughness_tex = s;
                // The parsed glTF defaults (1, 1) overwrite the group defaults (ModelLoader.h:294-295).
                g.metallic_factor = (float)((pbr && pbr->get("metallicFactor")) ? pbr->get("metallicFactor")->number_or(1.0) : 1.0);
                g.roughness_factor = (float)((pbr && pbr->get("roughnessFactor")) ? pbr->get("roughnessFactor")->number_or(1.0) : 1.0);
                const JValue* nt = mat.get("normalTexture");
                if ((s = texture_source(d, nt)) >= 0) { g.normal_tex = s; g.normal_scale = (float)(nt->get("scale") ? nt->get("scale")->number_or(1.0) : 1.0); }
                if ((s = texture_source(d, mat.get("emissiveTexture"))) >= 0) g.emissive_tex = s;
                std::string am = mat.get("alphaMode") ? mat.get("alphaMode")->str : "OPAQUE";
                if (am == "BLEND") g.alpha_mode = 2;
                else if (am == "MASK") { g.alpha_mode = 1; g.alpha_cutoff = (float)(mat.get("alphaCutoff") ? mat.get("alphaCutoff")->number_or(0.5) : 0.5); }
                else g.alpha_mode = 0;
                const JValue* ef = mat.get("emissiveFactor");
                for (int k = 0; k < 3; ++k) g.emissive_factor[k] = (ef && ef->arr.size() == 3) ? (float)ef->arr[(size_t)k]->num : 0.0f;
                if (const JValue* exts = mat.get("extensions")) {
                    if (const JValue* tr = exts->get("KHR_materials_transmission")) {
                        const JValue* tf = tr->get("transmissionFactor");
                        if (tf && tf->is_number()) g.transmission_factor = (float)tf->num;
                        const JValue* tt = tr->get("transmissionTexture");
                        if (tt && tt->type == JValue::Object) {
                            const JValue* ti = tt->get("index");
                            const JValue* texs = d.root->get("textures");
                            if (ti && ti->is_number() && ti->is_int && ti->num >= 0 && texs && ti->num < (double)texs->arr.size()) {
                                const JValue* src = texs->arr[(size_t)ti->num]->get("source");
                                int img = src ? (int)src->num : -1;
                                if (img >= 0 && img < (int)out.images.size()) g.transmission_tex = img;
                            }
                        }
                    }
                    if (const JValue* io = exts->get("KHR_materials_ior")) {
                        const JValue* v = io->get("ior");
                        if (v && v->is_number()) g.ior = (float)v->num;
                    }
                }
            }

            auto fetch_index = [&](size_t k) -> uint32_t {
                const uint8_t* p = iv.ptr + k * iv.stride;
                if (iv.comp == 5121) return *p;
                if (iv.comp == 5123) { uint16_t v; std::memcpy(&v, p, 2); return v; }
                uint32_t v; std::memcpy(&v, p, 4); return v;
            };
            for (size_t f = 0; f + 2 < iv.count; f += 3) { // ModelLoader.h:382-415
                for (int k = 0; k < 3; ++k) {
                    uint32_t vi = fetch_index(f + (size_t)k);
                    if (vi >= pos.count) throw std::runtime_error("glTF: vertex index out of range");
                    float p[3]; std::memcpy(p, pos.ptr + (size_t)vi * pos.stride, 12);
                    out.positions.insert(out.positions.end(), p, p + 3);
                    float n[3] = { 0.f, 0.f, 0.f };
                    if (hasN && vi < nrm.count) std::memcpy(n, nrm.ptr + (size_t)vi * nrm.stride, 12);
                    out.normals.insert(out.normals.end(), n, n + 3);
                    float t[2] = { 0.f, 0.f };
                    if (hasT && vi < uv.count) std::memcpy(t, uv.ptr + (size_t)vi * uv.stride, 8);
                    out.texcoords.insert(out.texcoords.end(), t, t + 2);
                }
                out.group.push_back(groupIndex);
            }
        }
    }
}

} // namespace ptamd
