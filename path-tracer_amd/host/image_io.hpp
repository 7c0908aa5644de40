// Headless image output: PNG of the resolved RGBA8 film and PFM / raw dumps of the linear HDR buffer.
// SaveImagePNG follows Window::SaveImagePNG (Window.cpp:164-189): creates the parent directory, flips the
// rows (film row 0 is the bottom of the image) and writes 8-bit RGBA. The reference encodes with
// stb_image_write; this writer emits a standard zlib-compressed PNG, so files differ as byte streams but
// decode to the same pixels.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <string>
#include <vector>

namespace ptamd_host {

inline void png_chunk(std::vector<uint8_t>& out, const char tag[4], const uint8_t* data, size_t n)
{
    auto be32 = [&](uint32_t v) { for (int s = 24; s >= 0; s -= 8) out.push_back((uint8_t)(v >> s)); };
    be32((uint32_t)n);
    size_t start = out.size();
    out.insert(out.end(), tag, tag + 4);
    if (n) out.insert(out.end(), data, data + n);
    be32((uint32_t)crc32(0L, out.data() + start, (uInt)(n + 4)));
}

inline bool SaveImagePNG(const std::string& filename, int width, int height, const std::vector<uint8_t>& rgba8)
{
    if (width <= 0 || height <= 0 || rgba8.size() < (size_t)width * height * 4) return false;
    std::error_code ec;
    std::filesystem::path p(filename);
    if (p.has_parent_path()) std::filesystem::create_directories(p.parent_path(), ec);
    const size_t stride = (size_t)width * 4;
    std::vector<uint8_t> raw((stride + 1) * (size_t)height);
    for (int y = 0; y < height; ++y) { // vertical flip + filter byte 0 per row
        uint8_t* dst = raw.data() + (size_t)(height - 1 - y) * (stride + 1);
        dst[0] = 0;
        std::memcpy(dst + 1, rgba8.data() + (size_t)y * stride, stride);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    std::vector<uint8_t> out = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    uint8_t ihdr[13];
    for (int i = 0; i < 4; ++i) { ihdr[i] = (uint8_t)((uint32_t)width >> (24 - 8 * i)); ihdr[4 + i] = (uint8_t)((uint32_t)height >> (24 - 8 * i)); }
    ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    png_chunk(out, "IHDR", ihdr, 13);
    png_chunk(out, "IDAT", z.data(), zlen);
    png_chunk(out, "IEND", nullptr, 0);
    FILE* f = std::fopen(filename.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok;
}

// Linear HDR buffer (Film::AverageAt per pixel, 3 floats) as a little-endian PFM; PFM rows run bottom-to-top,
// which is the film's own order.
inline bool SaveImagePFM(const std::string& filename, int width, int height, const std::vector<float>& rgb)
{
    if (rgb.size() < (size_t)width * height * 3) return false;
    FILE* f = std::fopen(filename.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);
    bool ok = std::fwrite(rgb.data(), sizeof(float), (size_t)width * height * 3, f) == (size_t)width * height * 3;
    std::fclose(f);
    return ok;
}

inline bool SaveRaw(const std::string& filename, const void* data, size_t bytes)
{
    FILE* f = std::fopen(filename.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(data, 1, bytes, f) == bytes;
    std::fclose(f);
    return ok;
}

} // namespace ptamd_host
