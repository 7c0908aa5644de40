#include "gltf_loader.h"

#include <fstream>
#include <functional>
#include <sstream>

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
    int depth = 0; // nesting of the value being parsed: bounded, so that a file of brackets cannot exhaust the stack
    [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("glTF JSON parse error: ") + what); }
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    JPtr parse()
    {
        struct Nest { int& d; explicit Nest(int& x) : d(x) { ++d; } ~Nest() { --d; } } nest(depth);
        if (depth > 256) fail("nesting deeper than 256 levels");
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

// PNG -> the RGBA bytes the reference's Mesh samples (Mesh.cpp:15-37). tinygltf asks stb_image for four channels
// (tiny_gltf.h:2653) and the decode below follows what stb_image does with every flavour it accepts: bit depths 1, 2, 4
// (grey scaled to 8 bits by 255, 85, 17; palette indices as they are), 8 and 16, colour types 0, 2, 3, 4, 6, Adam7
// interlacing, palette transparency and colour-key transparency (the key of a low-depth grey image scaled like its pixels).
// A 16-bit file is special: tinygltf loads it as 16-bit samples (tiny_gltf.h:2645-2661), ModelLoader copies that buffer
// as it is (ModelLoader.h:190-197) and the sampler indexes it as if it held bytes, so a texel is made of the low and high
// bytes of two 16-bit samples of the image's first half. The bytes returned here are that first half.
void decode_png_rgba(const uint8_t* data, size_t size, int& w, int& h, std::vector<uint8_t>& rgba)
{
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (size < 8 || std::memcmp(data, sig, 8)) throw std::runtime_error("image: only PNG and JPEG images are supported");
    size_t pos = 8;
    std::vector<uint8_t> idat, plte, trns;
    int depth = 0, ctype = -1, interlace = 0;
    bool haveTrns = false;
    w = h = 0;
    while (pos + 8 <= size) {
        uint32_t n = be32(data + pos);
        const uint8_t* tag = data + pos + 4;
        const uint8_t* body = data + pos + 8;
        if ((uint64_t)pos + 12 + n > size) throw std::runtime_error("image: truncated PNG");
        if (!std::memcmp(tag, "IHDR", 4)) { if (n != 13) throw std::runtime_error("image: bad PNG header");
                                           w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
                                           if (body[10] || body[11]) throw std::runtime_error("image: bad PNG compression or filter method"); }
        else if (!std::memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + n);
        else if (!std::memcmp(tag, "PLTE", 4)) plte.assign(body, body + n);
        else if (!std::memcmp(tag, "tRNS", 4)) { trns.assign(body, body + n); haveTrns = true; }
        else if (!std::memcmp(tag, "IEND", 4)) break;
        pos += 12 + (size_t)n;
    }
    if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) throw std::runtime_error("image: PNG bit depth must be 1, 2, 4, 8 or 16");
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch || w <= 0 || h <= 0 || interlace > 1) throw std::runtime_error("image: unsupported PNG colour type");
    if (ctype == 3 && depth == 16) throw std::runtime_error("image: a PNG palette image cannot be 16-bit");
    if (ctype == 3 && plte.empty()) throw std::runtime_error("image: PNG palette image without a palette");
    if ((uint64_t)w * (uint64_t)h > (1ull << 28)) throw std::runtime_error("image: PNG too large");
    if (haveTrns && ctype != 3 && ((ctype & 4) || trns.size() != (size_t)ch * 2)) throw std::runtime_error("image: bad PNG tRNS chunk");

    // pass geometry: one pass, or the seven of Adam7
    struct Pass { int x0, y0, dx, dy; };
    static const Pass adam7[7] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 }, { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
    static const Pass whole = { 0, 0, 1, 1 };
    const int numPasses = interlace ? 7 : 1;
    const size_t bpp = (size_t)std::max(1, ch * depth / 8); // filter distance in bytes
    auto row_bytes = [&](int pw) { return ((size_t)pw * ch * depth + 7) / 8; };
    size_t rawSize = 0;
    for (int k = 0; k < numPasses; ++k) {
        const Pass& ps = interlace ? adam7[k] : whole;
        const int pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        if (pw > 0 && ph > 0) rawSize += (row_bytes(pw) + 1) * (size_t)ph;
    }
    std::vector<uint8_t> raw(rawSize);
    uLongf rawLen = (uLongf)raw.size();
    const int zrc = uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size());
    if ((zrc != Z_OK && zrc != Z_BUF_ERROR) || rawLen != raw.size()) throw std::runtime_error("image: PNG inflate failed");

    // unfilter every pass and scatter its samples (as stored: 1..16 bits each) into the full image
    std::vector<uint16_t> sample((size_t)w * h * ch);
    size_t at = 0;
    for (int k = 0; k < numPasses; ++k) {
        const Pass& ps = interlace ? adam7[k] : whole;
        const int pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        if (pw <= 0 || ph <= 0) continue;
        const size_t stride = row_bytes(pw);
        std::vector<uint8_t> prev(stride, 0), cur(stride);
        for (int j = 0; j < ph; ++j) {
            const uint8_t ft = raw[at];
            const uint8_t* in = &raw[at + 1];
            at += stride + 1;
            if (ft > 4) throw std::runtime_error("image: bad PNG filter");
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
                int pred = 0;
                switch (ft) {
                case 1: pred = a; break; case 2: pred = b; break; case 3: pred = (a + b) >> 1; break;
                case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: break;
                }
                cur[i] = (uint8_t)(in[i] + pred);
            }
            const size_t y = (size_t)ps.y0 + (size_t)j * ps.dy;
            for (int i = 0; i < pw; ++i) {
                uint16_t* out = &sample[(y * w + (size_t)ps.x0 + (size_t)i * ps.dx) * ch];
                for (int c2 = 0; c2 < ch; ++c2) {
                    const size_t s = (size_t)i * ch + c2;
                    if (depth == 16) out[c2] = (uint16_t)((cur[2 * s] << 8) | cur[2 * s + 1]);
                    else if (depth == 8) out[c2] = cur[s];
                    else { const size_t bit = s * depth; out[c2] = (uint16_t)((cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1)); }
                }
            }
            prev.swap(cur);
        }
    }

    const size_t npx = (size_t)w * h;
    if (depth == 16) {
        // 16-bit RGBA in the machine's byte order (low byte first), of which the tracer's sampler sees the first half as bytes
        const uint16_t key[3] = { haveTrns ? (uint16_t)((trns[0] << 8) | trns[1]) : (uint16_t)0,
                                  haveTrns && ch == 3 ? (uint16_t)((trns[2] << 8) | trns[3]) : (uint16_t)0,
                                  haveTrns && ch == 3 ? (uint16_t)((trns[4] << 8) | trns[5]) : (uint16_t)0 };
        rgba.assign(npx * 4, 0);
        for (size_t p = 0; p < (npx + 1) / 2; ++p) { // pixels whose 8 bytes lie in the first half
            const uint16_t* s = &sample[p * ch];
            uint16_t px[4] = { 0, 0, 0, 0xFFFF };
            switch (ctype) {
            case 0: px[0] = px[1] = px[2] = s[0]; if (haveTrns && s[0] == key[0]) px[3] = 0; break;
            case 2: px[0] = s[0]; px[1] = s[1]; px[2] = s[2]; if (haveTrns && s[0] == key[0] && s[1] == key[1] && s[2] == key[2]) px[3] = 0; break;
            case 4: px[0] = px[1] = px[2] = s[0]; px[3] = s[1]; break;
            default: px[0] = s[0]; px[1] = s[1]; px[2] = s[2]; px[3] = s[3]; break;
            }
            for (int k = 0; k < 4; ++k) {
                const size_t byte = p * 8 + (size_t)k * 2;
                if (byte < npx * 4) rgba[byte] = (uint8_t)(px[k] & 255);
                if (byte + 1 < npx * 4) rgba[byte + 1] = (uint8_t)(px[k] >> 8);
            }
        }
        return;
    }
    const int scale = (ctype == 0 && depth < 8) ? (depth == 1 ? 255 : depth == 2 ? 85 : 17) : 1;
    uint8_t key[3] = { 0, 0, 0 };
    if (haveTrns && ctype != 3) for (int k = 0; k < ch; ++k) key[k] = (uint8_t)(trns[2 * k + 1] * scale); // the low byte of the key, scaled like the pixels
    rgba.assign(npx * 4, 255);
    for (size_t p = 0; p < npx; ++p) {
        uint8_t* o = &rgba[4 * p];
        const uint16_t* s = &sample[p * ch];
        switch (ctype) {
        case 0: o[0] = o[1] = o[2] = (uint8_t)(s[0] * scale); if (haveTrns && o[0] == key[0]) o[3] = 0; break;
        case 2: o[0] = (uint8_t)s[0]; o[1] = (uint8_t)s[1]; o[2] = (uint8_t)s[2];
                if (haveTrns && o[0] == key[0] && o[1] == key[1] && o[2] == key[2]) o[3] = 0; break;
        case 3: { const size_t k = s[0]; if (3 * k + 2 < plte.size()) { o[0] = plte[3 * k]; o[1] = plte[3 * k + 1]; o[2] = plte[3 * k + 2]; }
                  if (k < trns.size()) o[3] = trns[k]; break; }
        case 4: o[0] = o[1] = o[2] = (uint8_t)s[0]; o[3] = (uint8_t)s[1]; break;
        default: o[0] = (uint8_t)s[0]; o[1] = (uint8_t)s[1]; o[2] = (uint8_t)s[2]; o[3] = (uint8_t)s[3]; break;
        }
    }
}

} // namespace
bool looks_like_jpeg(const uint8_t* data, size_t size);
void decode_jpeg_rgba(const uint8_t* data, size_t size, int& w, int& h, std::vector<uint8_t>& rgba); // jpeg_decode.cpp
namespace {
// An image of a model file as RGBA8: PNG or JPEG by signature (stb_image, which decodes for the reference, sniffs the
// same way; its other formats -- BMP, GIF, PSD, PIC, PNM, HDR, TGA -- are not glTF image formats and are refused).
void decode_image_rgba(const uint8_t* data, size_t size, int& w, int& h, std::vector<uint8_t>& rgba)
{
    if (looks_like_jpeg(data, size)) decode_jpeg_rgba(data, size, w, h, rgba);
    else decode_png_rgba(data, size, w, h, rgba);
}

} // namespace
void decode_image(const uint8_t* data, size_t size, int& w, int& h, std::vector<uint8_t>& rgba) { decode_image_rgba(data, size, w, h, rgba); }
namespace {

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
// Every index, count, offset and stride of the file goes through here before it takes part in pointer arithmetic: a
// finite, non-negative whole number of at most 2^48 (absent: the default); anything else is refused, as tinygltf
// refuses it for the reference.
size_t file_uint(const JValue* v, const char* what, size_t dflt = 0)
{
    if (!v) return dflt;
    if (v->type != JValue::Number || !(v->num >= 0.0) || !(v->num <= 281474976710656.0) || v->num != std::floor(v->num))
        throw std::runtime_error(std::string("glTF: ") + what + " is not a valid unsigned integer");
    return (size_t)v->num;
}
View accessor_view(const Doc& d, const JValue* indexValue)
{
    const JValue& acc = at(d.root->get("accessors"), file_uint(indexValue, "accessor index"), "accessors");
    const JValue* bvIdx = acc.get("bufferView");
    if (!bvIdx) throw std::runtime_error("glTF: accessor without bufferView is not supported");
    const JValue& bv = at(d.root->get("bufferViews"), file_uint(bvIdx, "bufferView index"), "bufferViews");
    const size_t buf = file_uint(bv.get("buffer"), "buffer index");
    if (buf >= d.buffers.size()) throw std::runtime_error("glTF: bad buffer index");
    View v;
    v.comp = (int)file_uint(acc.get("componentType"), "componentType");
    v.type = (acc.get("type") && acc.get("type")->type == JValue::String) ? acc.get("type")->str : "";
    v.count = file_uint(acc.get("count"), "accessor count");
    const size_t off = file_uint(bv.get("byteOffset"), "byteOffset") + file_uint(acc.get("byteOffset"), "byteOffset");
    const size_t elem = (size_t)comp_size(v.comp) * (size_t)type_count(v.type);
    const size_t bs = file_uint(bv.get("byteStride"), "byteStride");
    v.stride = bs ? bs : elem;
    // range check without overflow: the first element fits, and count - 1 further strides fit behind it
    const size_t size = d.buffers[buf].size();
    if (v.count) {
        if (elem == 0 || off > size || elem > size - off) throw std::runtime_error("glTF: accessor out of buffer range");
        if (v.count - 1 > 0 && (v.stride == 0 ? false : (v.count - 1) > (size - off - elem) / v.stride)) throw std::runtime_error("glTF: accessor out of buffer range");
    }
    v.ptr = d.buffers[buf].data() + off;
    return v;
}

int texture_source(const Doc& d, const JValue* texInfo)
{
    if (!texInfo) return -1;
    const JValue* idx = texInfo->get("index");
    if (!idx || !idx->is_number() || idx->num < 0) return -1;
    const JValue& t = at(d.root->get("textures"), file_uint(idx, "texture index"), "textures");
    const JValue* src = t.get("source");
    if (!src || !src->is_number() || src->num < 0) return -1;
    const size_t i = file_uint(src, "texture source");
    if (i > 0x7fffffffu) throw std::runtime_error("glTF: texture source is not a valid image index");
    return (int)i;
}

} // namespace

static void load_any(const std::string& path, Model& out, uint32_t flags);

// Every index the kernels will follow is checked here once: a file whose material points at an image or whose face
// points at a group that does not exist is refused instead of read out of bounds on the device.
void load_gltf(const std::string& path, Model& out, uint32_t flags)
{
    load_any(path, out, flags);
    for (const ptamd_pbr& g : out.groups) {
        const int32_t tex[5] = { g.base_color_tex, g.normal_tex, g.metallic_roughness_tex, g.emissive_tex, g.transmission_tex };
        for (int32_t t : tex) if (t >= (int32_t)out.images.size()) throw std::runtime_error("model material refers to an image that the file does not hold: " + path);
    }
    for (int32_t g : out.group) if (g >= (int32_t)out.groups.size()) throw std::runtime_error("model face refers to a missing material group: " + path);
}

static void load_any(const std::string& path, Model& out, uint32_t flags)
{
    // ModelLoader.h:129-142
    std::string ext;
    size_t dot = path.find_last_of('.');
    if (dot != std::string::npos) ext = path.substr(dot + 1);
    if ((flags & kLoadObj) && (ext == "obj" || ext == "OBJ")) { load_obj(path, out); return; }
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
                const JValue& bv = at(d.root->get("bufferViews"), file_uint(bvI, "bufferView index"), "bufferViews");
                const size_t buf = file_uint(bv.get("buffer"), "buffer index");
                const size_t off = file_uint(bv.get("byteOffset"), "byteOffset");
                const size_t len = file_uint(bv.get("byteLength"), "byteLength");
                if (buf >= d.buffers.size() || off > d.buffers[buf].size() || len > d.buffers[buf].size() - off) throw std::runtime_error("glTF: image bufferView out of range");
                raw.assign(d.buffers[buf].begin() + (long)off, d.buffers[buf].begin() + (long)(off + len));
            } else if (const JValue* uri = im->get("uri")) {
                if (uri->str.rfind("data:", 0) == 0) raw = base64_decode(uri->str, uri->str.find(',') + 1);
                else raw = read_file(d.base + "/" + uri->str);
            }
            int w = 0, h = 0;
            std::vector<uint8_t> rgba;
            decode_image_rgba(raw.data(), raw.size(), w, h, rgba);
            out.image_data.push_back(std::move(rgba));
            ptamd_image pi; pi.width = w; pi.height = h; pi.channels = 4; pi.data = nullptr;
            out.images.push_back(pi);
        }
        for (size_t i = 0; i < out.images.size(); ++i) out.images[i].data = out.image_data[i].data();
    }

    const JValue* mats = d.root->get("materials");
    const JValue* meshes = d.root->get("meshes");
    if (!meshes) return;
    // M: node-to-world matrix (row-major 3x4) applied to the vertices, or null for the reference's behaviour of
    // ignoring the scene graph (ModelLoader.h:204-206).
    auto emit_mesh = [&](const JValue& meshRef, const double* M) {
        const JValue* mesh = &meshRef;
        double N[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }; // inverse transpose of the upper 3x3, for normals
        if (M) {
            const double a = M[0], b = M[1], c = M[2], d2 = M[4], e = M[5], f = M[6], g = M[8], h = M[9], i = M[10];
            const double det = a * (e * i - f * h) - b * (d2 * i - f * g) + c * (d2 * h - e * g);
            if (std::fabs(det) > 1e-300) {
                const double id = 1.0 / det;
                // cofactor matrix / det = inverse transpose
                N[0] = (e * i - f * h) * id; N[1] = (f * g - d2 * i) * id; N[2] = (d2 * h - e * g) * id;
                N[3] = (c * h - b * i) * id; N[4] = (a * i - c * g) * id; N[5] = (b * g - a * h) * id;
                N[6] = (b * f - c * e) * id; N[7] = (c * d2 - a * f) * id; N[8] = (a * e - b * d2) * id;
            }
        }
        const JValue* prims = mesh->get("primitives");
        if (!prims) return;
        for (const auto& prim : prims->arr) {
            const JValue* mode = prim->get("mode");
            if (mode && !(mode->is_number() && mode->num == 4.0)) throw std::runtime_error("Only TRIANGLES are supported");
            const JValue* attrs = prim->get("attributes");
            if (!attrs || !attrs->get("POSITION")) throw std::runtime_error("Missing POSITION");
            View pos = accessor_view(d, attrs->get("POSITION"));
            if (pos.type != "VEC3" || pos.comp != 5126) throw std::runtime_error("POSITION must be float vec3");
            View nrm{}; bool hasN = false;
            if (const JValue* a = attrs->get("NORMAL")) { nrm = accessor_view(d, a); hasN = nrm.type == "VEC3" && nrm.comp == 5126; }
            View uv{}; bool hasT = false;
            if (const JValue* a = attrs->get("TEXCOORD_0")) { uv = accessor_view(d, a); hasT = uv.type == "VEC2" && uv.comp == 5126; }
            const JValue* indices = prim->get("indices");
            if (!indices || indices->num < 0) throw std::runtime_error("Indexed geometry required");
            View iv = accessor_view(d, indices);
            if (iv.type != "SCALAR") throw std::runtime_error("indices accessor must be SCALAR");
            if (iv.comp != 5121 && iv.comp != 5123 && iv.comp != 5125) throw std::runtime_error("Unsupported index type");

            int groupIndex = -1;
            const JValue* matI = prim->get("material");
            if (matI && matI->is_number() && matI->num >= 0) { // ModelLoader.h:263-362
                const JValue& mat = at(mats, file_uint(matI, "material index"), "materials");
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
                                if (src && src->is_number() && src->num >= 0 && src->num < (double)out.images.size()) g.transmission_tex = (int)src->num;
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
                    float n[3] = { 0.f, 0.f, 0.f };
                    if (hasN && vi < nrm.count) std::memcpy(n, nrm.ptr + (size_t)vi * nrm.stride, 12);
                    if (M) {
                        const double x = p[0], y = p[1], z = p[2];
                        for (int r = 0; r < 3; ++r) p[r] = (float)(M[4 * r] * x + M[4 * r + 1] * y + M[4 * r + 2] * z + M[4 * r + 3]);
                        const double nx = n[0], ny = n[1], nz = n[2];
                        double t[3];
                        for (int r = 0; r < 3; ++r) t[r] = N[3 * r] * nx + N[3 * r + 1] * ny + N[3 * r + 2] * nz;
                        const double len = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
                        for (int r = 0; r < 3; ++r) n[r] = len > 0.0 ? (float)(t[r] / len) : 0.f;
                    }
                    out.positions.insert(out.positions.end(), p, p + 3);
                    out.normals.insert(out.normals.end(), n, n + 3);
                    float t[2] = { 0.f, 0.f };
                    if (hasT && vi < uv.count) std::memcpy(t, uv.ptr + (size_t)vi * uv.stride, 8);
                    out.texcoords.insert(out.texcoords.end(), t, t + 2);
                }
                out.group.push_back(groupIndex);
            }
        }
    };

    if (!(flags & kLoadNodeTransforms)) {
        for (const auto& mesh : meshes->arr) emit_mesh(*mesh, nullptr);
        return;
    }
    // Scene-graph mode (not in the reference): every node of the default scene that refers to a mesh emits that mesh
    // under the product of the local matrices from its root (glTF 2.0: `matrix`, or T * R * S).
    const JValue* nodes = d.root->get("nodes");
    if (!nodes) return;
    auto local = [&](const JValue& n, double L[12]) {
        const double I[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
        std::memcpy(L, I, sizeof I);
        if (const JValue* m = n.get("matrix")) {
            if (m->arr.size() == 16) for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) L[4 * r + c] = m->arr[(size_t)(4 * c + r)]->num; // column-major
            return;
        }
        double t[3] = { 0, 0, 0 }, q[4] = { 0, 0, 0, 1 }, sc[3] = { 1, 1, 1 };
        if (const JValue* v = n.get("translation")) if (v->arr.size() == 3) for (int k = 0; k < 3; ++k) t[k] = v->arr[(size_t)k]->num;
        if (const JValue* v = n.get("rotation")) if (v->arr.size() == 4) for (int k = 0; k < 4; ++k) q[k] = v->arr[(size_t)k]->num;
        if (const JValue* v = n.get("scale")) if (v->arr.size() == 3) for (int k = 0; k < 3; ++k) sc[k] = v->arr[(size_t)k]->num;
        const double x = q[0], y = q[1], z = q[2], w = q[3];
        const double R[9] = { 1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                              2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                              2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y) };
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) L[4 * r + c] = R[3 * r + c] * sc[c]; L[4 * r + 3] = t[r]; }
    };
    std::vector<int> onPath;
    std::function<void(size_t, const double*)> visit = [&](size_t ni, const double* P) {
        if (ni >= nodes->arr.size()) throw std::runtime_error("glTF: node index out of range");
        for (int seen : onPath) if ((size_t)seen == ni) throw std::runtime_error("glTF: cycle in the node hierarchy");
        onPath.push_back((int)ni);
        const JValue& n = *nodes->arr[ni];
        double L[12], W[12];
        local(n, L);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c)
                W[4 * r + c] = P[4 * r] * L[c] + P[4 * r + 1] * L[4 + c] + P[4 * r + 2] * L[8 + c] + (c == 3 ? P[4 * r + 3] : 0.0);
        if (const JValue* m = n.get("mesh")) if (m->is_number() && m->num >= 0) emit_mesh(at(meshes, file_uint(m, "mesh index"), "meshes"), W);
        if (const JValue* ch = n.get("children")) for (const auto& c : ch->arr) visit(file_uint(c.get(), "node index"), W);
        onPath.pop_back();
    };
    const double I[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    const JValue* scenes = d.root->get("scenes");
    if (scenes && !scenes->arr.empty()) {
        size_t si = (d.root->get("scene") && d.root->get("scene")->is_number()) ? file_uint(d.root->get("scene"), "scene index") : 0;
        const JValue& sc = at(scenes, si, "scenes");
        if (const JValue* roots = sc.get("nodes")) for (const auto& r : roots->arr) visit(file_uint(r.get(), "node index"), I);
    } else { // no scene: every node that is nobody's child is a root
        std::vector<char> isChild(nodes->arr.size(), 0);
        for (const auto& n : nodes->arr) if (const JValue* ch = n->get("children")) for (const auto& c : ch->arr) { const size_t ci = file_uint(c.get(), "node index"); if (ci < isChild.size()) isChild[ci] = 1; }
        for (size_t i = 0; i < nodes->arr.size(); ++i) if (!isChild[i]) visit(i, I);
    }
}

// ------------------------------------------------------------------------------------------ Wavefront OBJ
// Not in the reference (ModelLoader accepts only glTF, ModelLoader.h:139-142). Reads v / vn / vt / f (polygons as
// triangle fans, negative indices), groups faces by `usemtl`, and takes base colour, emission, index of refraction,
// dissolve and the PBR extension keys (Pr, Pm) from the `mtllib` file; `map_Kd` PNG files become base-colour
// textures. Faces without normals get their geometric normal.
namespace {
struct ObjMaterial { std::string name; ptamd_pbr pbr; };

std::vector<ObjMaterial> load_mtl(const std::string& path, Model& out)
{
    std::vector<ObjMaterial> mats;
    std::ifstream f(path);
    if (!f) return mats;
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream in(line);
        std::string key;
        if (!(in >> key) || key[0] == '#') continue;
        if (key == "newmtl") {
            ObjMaterial m; in >> m.name; ptamd_pbr_default(&m.pbr);
            m.pbr.metallic_factor = 0.0f; m.pbr.roughness_factor = 1.0f;
            mats.push_back(m);
            continue;
        }
        if (mats.empty()) continue;
        ptamd_pbr& g = mats.back().pbr;
        if (key == "Kd") in >> g.base_color_factor[0] >> g.base_color_factor[1] >> g.base_color_factor[2];
        else if (key == "Ke") in >> g.emissive_factor[0] >> g.emissive_factor[1] >> g.emissive_factor[2];
        else if (key == "Ni") in >> g.ior;
        else if (key == "d") { float d = 1.0f; in >> d; g.transmission_factor = 1.0f - d; }
        else if (key == "Tr") in >> g.transmission_factor;
        else if (key == "Pr") in >> g.roughness_factor;
        else if (key == "Pm") in >> g.metallic_factor;
        else if (key == "Ns") { float ns = 0.f; in >> ns; g.roughness_factor = std::sqrt(2.0f / (std::max(ns, 0.0f) + 2.0f)); } // Blinn exponent -> roughness
        else if (key == "map_Kd") {
            std::string file; in >> file;
            std::vector<uint8_t> raw = read_file(dir_of(path) + "/" + file);
            int w = 0, h = 0; std::vector<uint8_t> rgba;
            decode_image_rgba(raw.data(), raw.size(), w, h, rgba);
            out.image_data.push_back(std::move(rgba));
            ptamd_image pi; pi.width = w; pi.height = h; pi.channels = 4; pi.data = nullptr;
            out.images.push_back(pi);
            g.base_color_tex = (int32_t)out.images.size() - 1;
        }
    }
    return mats;
}
} // namespace

void load_obj(const std::string& path, Model& out)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("Failed to load OBJ model: " + path);
    out = Model();
    std::vector<float> V, VN, VT;
    std::vector<ObjMaterial> mats;
    int group = -1;
    std::string line;
    auto resolve = [](long i, size_t n) -> long { return i > 0 ? i - 1 : (i < 0 ? (long)n + i : -1); };
    while (std::getline(f, line)) {
        std::istringstream in(line);
        std::string key;
        if (!(in >> key) || key[0] == '#') continue;
        if (key == "v") { float x, y, z; in >> x >> y >> z; V.insert(V.end(), { x, y, z }); }
        else if (key == "vn") { float x, y, z; in >> x >> y >> z; VN.insert(VN.end(), { x, y, z }); }
        else if (key == "vt") { float u = 0, v = 0; in >> u >> v; VT.insert(VT.end(), { u, v }); }
        else if (key == "mtllib") { std::string file; in >> file; mats = load_mtl(dir_of(path) + "/" + file, out); }
        else if (key == "usemtl") {
            std::string name; in >> name;
            group = -1;
            for (size_t g = 0; g < out.group_names.size(); ++g) if (out.group_names[g] == name) group = (int)g;
            if (group < 0) {
                ptamd_pbr def; ptamd_pbr_default(&def); def.metallic_factor = 0.0f; def.roughness_factor = 1.0f;
                for (const ObjMaterial& m : mats) if (m.name == name) def = m.pbr;
                out.groups.push_back(def); out.group_names.push_back(name);
                group = (int)out.groups.size() - 1;
            }
        } else if (key == "f") {
            struct Corner { long v, t, n; };
            std::vector<Corner> cs;
            std::string tok;
            while (in >> tok) {
                Corner c{ 0, 0, 0 };
                size_t s1 = tok.find('/');
                c.v = std::atol(tok.substr(0, s1).c_str());
                if (s1 != std::string::npos) {
                    size_t s2 = tok.find('/', s1 + 1);
                    std::string t = tok.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
                    if (!t.empty()) c.t = std::atol(t.c_str());
                    if (s2 != std::string::npos) c.n = std::atol(tok.substr(s2 + 1).c_str());
                }
                cs.push_back(c);
            }
            for (size_t k = 1; k + 1 < cs.size(); ++k) {
                const Corner tri[3] = { cs[0], cs[k], cs[k + 1] };
                float P[9], Nn[9] = { 0 }, T[6] = { 0 };
                bool haveN = true;
                for (int j = 0; j < 3; ++j) {
                    long vi = resolve(tri[j].v, V.size() / 3);
                    if (vi < 0 || (size_t)vi >= V.size() / 3) throw std::runtime_error("OBJ: vertex index out of range");
                    std::memcpy(P + 3 * j, &V[3 * (size_t)vi], 12);
                    long ni = resolve(tri[j].n, VN.size() / 3);
                    if (ni >= 0 && (size_t)ni < VN.size() / 3) std::memcpy(Nn + 3 * j, &VN[3 * (size_t)ni], 12); else haveN = false;
                    long ti = resolve(tri[j].t, VT.size() / 2);
                    if (ti >= 0 && (size_t)ti < VT.size() / 2) std::memcpy(T + 2 * j, &VT[2 * (size_t)ti], 8);
                }
                if (!haveN) {
                    float e1[3] = { P[3] - P[0], P[4] - P[1], P[5] - P[2] }, e2[3] = { P[6] - P[0], P[7] - P[1], P[8] - P[2] };
                    float n[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
                    float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                    for (int j = 0; j < 3; ++j) for (int c = 0; c < 3; ++c) Nn[3 * j + c] = len > 0.f ? n[c] / len : 0.f;
                }
                out.positions.insert(out.positions.end(), P, P + 9);
                out.normals.insert(out.normals.end(), Nn, Nn + 9);
                out.texcoords.insert(out.texcoords.end(), T, T + 6);
                out.group.push_back(group);
            }
        }
    }
    for (size_t i = 0; i < out.images.size(); ++i) out.images[i].data = out.image_data[i].data();
}

} // namespace ptamd
