// JPEG -> RGBA8 for glTF images. The reference decodes images through tinygltf -> stb_image v2.28 with four output
// channels requested (contrib/include/tiny_gltf.h:2653, contrib/include/stb_image.h), and a path tracer that samples
// nearest texels (Mesh.cpp:15-37) sees every bit of the decoder's output, so this decoder follows stb_image's published
// arithmetic where the JPEG standard leaves freedom: the 12-bit fixed-point "islow" inverse DCT with two extra bits
// between the passes, triangle-filter chroma upsampling for 2x1 / 1x2 / 2x2 subsampling and pixel replication for
// every other ratio, and the 20-bit fixed-point YCbCr -> RGB conversion whose green channel drops the low 16 bits of
// the Cb term. Baseline and progressive Huffman coding, 8 bits per sample, 1 / 3 / 4 components (grey, YCbCr or RGB,
// CMYK / YCCK with an Adobe marker), restart intervals, 8- and 16-bit quantisation tables. Arithmetic coding, lossless
// and 12-bit JPEG are refused, as stb_image refuses them.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <stdexcept>
#include <vector>

namespace ptamd {
namespace {

const uint8_t kZigZag[64 + 15] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    // a corrupt run may step up to 15 places past the end: those land on the last coefficient
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63
};

[[noreturn]] void bad(const char* what) { throw std::runtime_error(std::string("image: corrupt or unsupported JPEG (") + what + ")"); }

// Canonical Huffman table (JPEG Annex C): codes of each length are consecutive; decode by comparing the next 16 bits
// with the first code beyond each length.
struct Huffman {
    uint8_t size[257];
    uint8_t values[256];
    uint32_t maxcode[18]; // first 16-bit-aligned code that is longer than j bits
    int delta[17];        // index of the first symbol of length j minus its code
    bool defined = false;
    void build(const int count[16])
    {
        int k = 0;
        for (int len = 1; len <= 16; ++len)
            for (int j = 0; j < count[len - 1]; ++j) {
                size[k++] = (uint8_t)len;
                if (k >= 257) bad("Huffman size list");
            }
        size[k] = 0;
        uint32_t code = 0;
        k = 0;
        for (int len = 1; len <= 16; ++len) {
            delta[len] = k - (int)code;
            if (size[k] == len) {
                while (size[k] == len) { ++k; ++code; }
                if (code - 1 >= (1u << len)) bad("Huffman code lengths");
            }
            maxcode[len] = code << (16 - len);
            code <<= 1;
        }
        maxcode[17] = 0xffffffffu;
        defined = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0; // sampling factors, quantisation table
    int hd = 0, ha = 0;               // Huffman tables of the current scan
    int dc_pred = 0;
    int x = 0, y = 0;                 // size in samples
    int w2 = 0, h2 = 0;               // size of the plane, whole MCUs
    std::vector<uint8_t> plane;       // w2 * h2 decoded samples
    std::vector<int16_t> coeff;       // progressive: 64 coefficients per block, row-major blocks
    int coeff_w = 0;
};

class Decoder {
public:
    Decoder(const uint8_t* data, size_t size) : p_(data), end_(data + size) {}
    void decode(int& w, int& h, std::vector<uint8_t>& rgba)
    {
        read_headers();
        int m = next_marker();
        while (m != 0xD9) { // EOI
            if (m == 0xDA) { // SOS
                scan_header();
                entropy_coded_data();
                if (marker_ == kNone) marker_ = skip_to_marker();
                m = next_marker();
                if (m >= 0xD0 && m <= 0xD7) m = next_marker();
            } else if (m == 0xDC) { // DNL
                const int len = get16(); const int lines = get16();
                if (len != 4 || lines != height_) bad("DNL");
                m = next_marker();
            } else {
                // a segment that cannot be read after the frame header ends the decoding; what has been decoded stands
                try { table_or_app_marker(m); } catch (const std::runtime_error&) { break; }
                m = next_marker();
            }
        }
        if (progressive_) finish_progressive();
        assemble(rgba);
        w = width_; h = height_;
    }

private:
    static constexpr int kNone = 0xFF;
    const uint8_t* p_; const uint8_t* end_;
    int width_ = 0, height_ = 0, ncomp_ = 0;
    bool progressive_ = false, jfif_ = false;
    int adobe_transform_ = -1, rgb_ids_ = 0;
    int hmax_ = 1, vmax_ = 1, mcux_ = 0, mcuy_ = 0;
    int restart_interval_ = 0, todo_ = 0;
    Component comp_[4];
    Huffman dc_[4], ac_[4];
    uint16_t dequant_[4][64] = {};
    // entropy decoder state: the next bits sit at the top of buffer_
    uint32_t buffer_ = 0; int bits_ = 0; int marker_ = kNone; bool nomore_ = false;
    int scan_n_ = 0, order_[4] = {}, spec_start_ = 0, spec_end_ = 63, succ_high_ = 0, succ_low_ = 0, eob_run_ = 0;

    int get8() { return p_ < end_ ? *p_++ : 0; }
    int get16() { const int hi = get8(); return (hi << 8) | get8(); }
    bool at_end() const { return p_ >= end_; }
    void skip(int n) { if (n < 0 || (size_t)n > (size_t)(end_ - p_)) p_ = end_; else p_ += n; }

    int next_marker()
    {
        if (marker_ != kNone) { const int m = marker_; marker_ = kNone; return m; }
        int x = get8();
        if (x != 0xFF) return kNone;
        while (x == 0xFF) { if (at_end()) return kNone; x = get8(); }
        return x;
    }
    int skip_to_marker() // bytes between the end of a scan's data and the next marker are skipped
    {
        while (!at_end()) {
            int x = get8();
            while (x == 0xFF) {
                if (at_end()) return kNone;
                x = get8();
                if (x != 0x00 && x != 0xFF) return x;
            }
        }
        return kNone;
    }

    // ---- headers and tables
    void read_headers()
    {
        if (next_marker() != 0xD8) bad("no SOI");
        int m = next_marker();
        while (m != 0xC0 && m != 0xC1 && m != 0xC2) {
            table_or_app_marker(m);
            m = next_marker();
            while (m == kNone) {
                if (at_end()) bad("no frame header");
                m = next_marker();
            }
        }
        progressive_ = m == 0xC2;
        frame_header();
    }
    void table_or_app_marker(int m)
    {
        if (m == kNone) bad("expected a marker");
        if (m == 0xDD) {
            if (get16() != 4) bad("DRI length");
            restart_interval_ = get16();
            return;
        }
        if (m == 0xDB) {
            int len = get16() - 2;
            while (len > 0) {
                const int q = get8(), wide = q >> 4, t = q & 15;
                if (wide > 1) bad("quantisation table precision");
                if (t > 3) bad("quantisation table index");
                for (int i = 0; i < 64; ++i) dequant_[t][kZigZag[i]] = (uint16_t)(wide ? get16() : get8());
                len -= wide ? 129 : 65;
            }
            if (len != 0) bad("DQT length");
            return;
        }
        if (m == 0xC4) {
            int len = get16() - 2;
            while (len > 0) {
                const int q = get8(), cls = q >> 4, t = q & 15;
                if (cls > 1 || t > 3) bad("Huffman table header");
                int count[16], n = 0;
                for (int i = 0; i < 16; ++i) { count[i] = get8(); n += count[i]; }
                if (n > 256) bad("Huffman table size");
                Huffman& h = cls ? ac_[t] : dc_[t];
                h.build(count);
                for (int i = 0; i < n; ++i) h.values[i] = (uint8_t)get8();
                len -= 17 + n;
            }
            if (len != 0) bad("DHT length");
            return;
        }
        if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
            int len = get16();
            if (len < 2) bad("APP / COM length");
            len -= 2;
            if (m == 0xE0 && len >= 5) {
                static const uint8_t tag[5] = { 'J', 'F', 'I', 'F', 0 };
                bool ok = true;
                for (int i = 0; i < 5; ++i) if (get8() != tag[i]) ok = false;
                len -= 5;
                if (ok) jfif_ = true;
            } else if (m == 0xEE && len >= 12) {
                static const uint8_t tag[6] = { 'A', 'd', 'o', 'b', 'e', 0 };
                bool ok = true;
                for (int i = 0; i < 6; ++i) if (get8() != tag[i]) ok = false;
                len -= 6;
                if (ok) { get8(); get16(); get16(); adobe_transform_ = get8(); len -= 6; }
            }
            skip(len);
            return;
        }
        bad("unknown marker");
    }
    void frame_header()
    {
        const int len = get16();
        if (len < 11) bad("SOF length");
        if (get8() != 8) bad("only 8 bits per sample");
        height_ = get16(); width_ = get16();
        if (height_ == 0 || width_ == 0) bad("empty frame");
        ncomp_ = get8();
        if (ncomp_ != 1 && ncomp_ != 3 && ncomp_ != 4) bad("component count");
        if (len != 8 + 3 * ncomp_) bad("SOF length");
        for (int i = 0; i < ncomp_; ++i) {
            Component& c = comp_[i];
            c.id = get8();
            if (ncomp_ == 3 && c.id == "RGB"[i]) ++rgb_ids_;
            const int q = get8();
            c.h = q >> 4; c.v = q & 15;
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) bad("sampling factor");
            c.tq = get8();
            if (c.tq > 3) bad("quantisation table of a component");
        }
        if ((uint64_t)width_ * (uint64_t)height_ * 4u > (1ull << 31)) bad("image too large");
        for (int i = 0; i < ncomp_; ++i) { hmax_ = std::max(hmax_, comp_[i].h); vmax_ = std::max(vmax_, comp_[i].v); }
        for (int i = 0; i < ncomp_; ++i) if (hmax_ % comp_[i].h || vmax_ % comp_[i].v) bad("sampling factors do not divide");
        mcux_ = (width_ + hmax_ * 8 - 1) / (hmax_ * 8);
        mcuy_ = (height_ + vmax_ * 8 - 1) / (vmax_ * 8);
        // The planes below are sized by the header alone. Entropy-coded data spends at least one bit on every 8x8 block
        // (its DC code), so a file cannot describe more than 8 blocks per remaining byte; a header that claims 64 times
        // what its bytes could hold (truncated files still decode, as in stb_image) is refused before anything is allocated.
        {
            uint64_t blocks = 0;
            for (int i = 0; i < ncomp_; ++i) blocks += (uint64_t)mcux_ * comp_[i].h * (uint64_t)mcuy_ * comp_[i].v;
            if (blocks > 64ull * (uint64_t)(end_ - p_) + 4096ull) bad("frame far larger than the file's data");
        }
        for (int i = 0; i < ncomp_; ++i) {
            Component& c = comp_[i];
            c.x = (width_ * c.h + hmax_ - 1) / hmax_;
            c.y = (height_ * c.v + vmax_ - 1) / vmax_;
            c.w2 = mcux_ * c.h * 8; c.h2 = mcuy_ * c.v * 8;
            c.plane.assign((size_t)c.w2 * c.h2, 0);
            if (progressive_) { c.coeff_w = c.w2 / 8; c.coeff.assign((size_t)c.w2 * c.h2, 0); }
        }
    }
    void scan_header()
    {
        const int len = get16();
        scan_n_ = get8();
        if (scan_n_ < 1 || scan_n_ > 4 || scan_n_ > ncomp_) bad("component count of a scan");
        if (len != 6 + 2 * scan_n_) bad("SOS length");
        for (int i = 0; i < scan_n_; ++i) {
            const int id = get8(), q = get8();
            int which = 0;
            while (which < ncomp_ && comp_[which].id != id) ++which;
            if (which == ncomp_) bad("scan names an unknown component");
            comp_[which].hd = q >> 4; comp_[which].ha = q & 15;
            if (comp_[which].hd > 3 || comp_[which].ha > 3) bad("Huffman table of a scan");
            order_[i] = which;
        }
        spec_start_ = get8(); spec_end_ = get8();
        const int a = get8();
        succ_high_ = a >> 4; succ_low_ = a & 15;
        if (progressive_) {
            if (spec_start_ > 63 || spec_end_ > 63 || spec_start_ > spec_end_ || succ_high_ > 13 || succ_low_ > 13) bad("spectral selection");
        } else {
            if (spec_start_ != 0 || succ_high_ != 0 || succ_low_ != 0) bad("spectral selection in a sequential scan");
            spec_end_ = 63;
        }
    }

    // ---- bit reader (T.81 F.2.2.5: a zero byte after 0xFF is stuffing, anything else is a marker and ends the data)
    void fill()
    {
        do {
            const int b = nomore_ ? 0 : get8();
            if (b == 0xFF) {
                int c = get8();
                while (c == 0xFF) c = get8();
                if (c != 0) { marker_ = c; nomore_ = true; return; }
            }
            buffer_ |= (uint32_t)b << (24 - bits_);
            bits_ += 8;
        } while (bits_ <= 24);
    }
    int decode_symbol(const Huffman& h)
    {
        if (bits_ < 16) fill();
        const uint32_t top = buffer_ >> 16;
        int k = 1;
        while (top >= h.maxcode[k]) ++k;
        if (k == 17) { bits_ -= 16; return -1; }
        if (k > bits_) return -1;
        const int c = (int)((buffer_ >> (32 - k)) & ((1u << k) - 1u)) + h.delta[k];
        if (c < 0 || c >= 256) return -1;
        bits_ -= k; buffer_ <<= k;
        return h.values[c];
    }
    int receive_extend(int n) // T.81 F.2.2.1: n bits as a signed value
    {
        if (bits_ < n) fill();
        if (bits_ < n) return 0; // the stream ran out: zeros from here on
        const bool positive = (buffer_ >> 31) != 0;
        const uint32_t v = buffer_ >> (32 - n);
        buffer_ <<= n; bits_ -= n;
        return positive ? (int)v : (int)v - (1 << n) + 1;
    }
    int get_bits(int n)
    {
        if (bits_ < n) fill();
        if (bits_ < n) return 0;
        const uint32_t v = buffer_ >> (32 - n);
        buffer_ <<= n; bits_ -= n;
        return (int)v;
    }
    int get_bit()
    {
        if (bits_ < 1) fill();
        if (bits_ < 1) return 0;
        const uint32_t v = buffer_ & 0x80000000u;
        buffer_ <<= 1; --bits_;
        return v != 0;
    }
    void reset_entropy()
    {
        bits_ = 0; buffer_ = 0; nomore_ = false; marker_ = kNone; eob_run_ = 0;
        for (Component& c : comp_) c.dc_pred = 0;
        todo_ = restart_interval_ ? restart_interval_ : 0x7fffffff;
    }
    bool restart_or_stop() // after each MCU; true: the scan's data ended
    {
        if (--todo_ > 0) return false;
        if (bits_ < 24) fill();
        if (!(marker_ >= 0xD0 && marker_ <= 0xD7)) return true;
        reset_entropy();
        return false;
    }

    // ---- block decoding
    static bool fits_short_product(int a, int b) { const long long v = (long long)a * b; return v >= -32768 && v <= 32767; }
    void baseline_block(int16_t data[64], Component& c)
    {
        const Huffman& hd = dc_[c.hd]; const Huffman& ha = ac_[c.ha];
        const uint16_t* dq = dequant_[c.tq];
        const int t = decode_symbol(hd);
        if (t < 0 || t > 15) bad("DC code");
        std::memset(data, 0, 64 * sizeof(int16_t));
        const int diff = t ? receive_extend(t) : 0;
        const long long dcl = (long long)c.dc_pred + diff;
        if (dcl > 0x7fffffffLL || dcl < -0x80000000LL) bad("DC delta");
        const int dc = (int)dcl;
        c.dc_pred = dc;
        if (dc > 32767 || dc < -32768 || dq[0] > 32767 || !fits_short_product(dc, dq[0])) bad("DC range");
        data[0] = (int16_t)(dc * dq[0]);
        int k = 1;
        do {
            const int rs = decode_symbol(ha);
            if (rs < 0) bad("AC code");
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xF0) break;
                k += 16;
            } else {
                k += r;
                const int z = kZigZag[k++];
                data[z] = (int16_t)(receive_extend(s) * dq[z]);
            }
        } while (k < 64);
    }
    void progressive_dc(int16_t* data, Component& c)
    {
        if (spec_end_ != 0) bad("DC scan with AC coefficients");
        if (succ_high_ == 0) {
            std::memset(data, 0, 64 * sizeof(int16_t));
            const int t = decode_symbol(dc_[c.hd]);
            if (t < 0 || t > 15) bad("DC code");
            const int diff = t ? receive_extend(t) : 0;
            const long long dcl = (long long)c.dc_pred + diff;
            if (dcl > 0x7fffffffLL || dcl < -0x80000000LL) bad("DC delta");
            const int dc = (int)dcl;
            c.dc_pred = dc;
            if (dc > 32767 || dc < -32768 || !fits_short_product(dc, 1 << succ_low_)) bad("DC range");
            data[0] = (int16_t)(dc * (1 << succ_low_));
        } else if (get_bit()) {
            data[0] = (int16_t)(data[0] + (int16_t)(1 << succ_low_));
        }
    }
    void refine(int16_t& coef, int16_t bit)
    {
        if (get_bit() && (coef & bit) == 0) coef = (int16_t)(coef > 0 ? coef + bit : coef - bit);
    }
    void progressive_ac(int16_t* data, Component& c)
    {
        if (spec_start_ == 0) bad("AC scan that starts at the DC coefficient");
        const Huffman& ha = ac_[c.ha];
        if (succ_high_ == 0) { // first pass over this band (T.81 G.1.2.2)
            if (eob_run_) { --eob_run_; return; }
            int k = spec_start_;
            do {
                const int rs = decode_symbol(ha);
                if (rs < 0) bad("AC code");
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eob_run_ = 1 << r;
                        if (r) eob_run_ += get_bits(r);
                        --eob_run_;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    const int z = kZigZag[k++];
                    data[z] = (int16_t)(receive_extend(s) * (1 << succ_low_));
                }
            } while (k <= spec_end_);
            return;
        }
        // refinement pass (T.81 G.1.2.3)
        const int16_t bit = (int16_t)(1 << succ_low_);
        if (eob_run_) {
            --eob_run_;
            for (int k = spec_start_; k <= spec_end_; ++k) {
                int16_t& v = data[kZigZag[k]];
                if (v != 0) refine(v, bit);
            }
            return;
        }
        int k = spec_start_;
        do {
            const int rs = decode_symbol(ha);
            if (rs < 0) bad("AC code");
            int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (r < 15) {
                    eob_run_ = (1 << r) - 1;
                    if (r) eob_run_ += get_bits(r);
                    r = 64; // run to the end of the band
                }
            } else {
                if (s != 1) bad("AC refinement code");
                s = get_bit() ? bit : -bit;
            }
            while (k <= spec_end_) {
                int16_t& v = data[kZigZag[k++]];
                if (v != 0) refine(v, bit);
                else {
                    if (r == 0) { v = (int16_t)s; break; }
                    --r;
                }
            }
        } while (k <= spec_end_);
    }

    // ---- inverse DCT: one pass of the 12-bit fixed-point "islow" butterfly; returns the four sums and four differences
    static void idct_1d(const int s[8], int x[4], int t[4])
    {
        auto fx = [](float v) { return (int)(v * 4096 + 0.5); }; // single-precision constants, rounded in double, truncated
        static const int k0541 = fx(0.5411961f), kn1847 = fx(-1.847759065f), k0765 = fx(0.765366865f), k1175 = fx(1.175875602f),
                         k0298 = fx(0.298631336f), k2053 = fx(2.053119869f), k3072 = fx(3.072711026f), k1501 = fx(1.501321110f),
                         kn0899 = fx(-0.899976223f), kn2562 = fx(-2.562915447f), kn1961 = fx(-1.961570560f), kn0390 = fx(-0.390180644f);
        int p1 = (s[2] + s[6]) * k0541;
        const int e2 = p1 + s[6] * kn1847, e3 = p1 + s[2] * k0765;
        const int e0 = (s[0] + s[4]) * 4096, e1 = (s[0] - s[4]) * 4096;
        x[0] = e0 + e3; x[3] = e0 - e3; x[1] = e1 + e2; x[2] = e1 - e2;
        int t0 = s[7], t1 = s[5], t2 = s[3], t3 = s[1];
        int p3 = t0 + t2, p4 = t1 + t3;
        p1 = t0 + t3; int p2 = t1 + t2;
        const int p5 = (p3 + p4) * k1175;
        t0 *= k0298; t1 *= k2053; t2 *= k3072; t3 *= k1501;
        p1 = p5 + p1 * kn0899; p2 = p5 + p2 * kn2562; p3 *= kn1961; p4 *= kn0390;
        t[3] = t3 + p1 + p4; t[2] = t2 + p2 + p3; t[1] = t1 + p2 + p4; t[0] = t0 + p1 + p3;
    }
    static uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
    static void idct_block(uint8_t* out, int stride, const int16_t data[64])
    {
        int mid[64];
        for (int c = 0; c < 8; ++c) { // columns, keeping two extra bits
            int s[8], x[4], t[4];
            for (int r = 0; r < 8; ++r) s[r] = data[8 * r + c];
            idct_1d(s, x, t);
            for (int k = 0; k < 4; ++k) {
                mid[8 * k + c] = (x[k] + 512 + t[3 - k]) >> 10;
                mid[8 * (7 - k) + c] = (x[k] + 512 - t[3 - k]) >> 10;
            }
        }
        for (int r = 0; r < 8; ++r) { // rows: 17 bits of scale to remove, +128 level shift folded into the rounding term
            int x[4], t[4];
            idct_1d(mid + 8 * r, x, t);
            uint8_t* o = out + (size_t)r * stride;
            const int bias = 65536 + (128 << 17);
            for (int k = 0; k < 4; ++k) {
                o[k] = clamp8((x[k] + bias + t[3 - k]) >> 17);
                o[7 - k] = clamp8((x[k] + bias - t[3 - k]) >> 17);
            }
        }
    }

    // ---- scans
    void entropy_coded_data()
    {
        reset_entropy();
        for (int i = 0; i < scan_n_; ++i) {
            const Component& c = comp_[order_[i]];
            const bool needDc = !progressive_ || spec_start_ == 0, needAc = !progressive_ || spec_start_ != 0;
            if (needDc && !(progressive_ && succ_high_ != 0) && !dc_[c.hd].defined) bad("scan uses an undefined DC table");
            if (needAc && !ac_[c.ha].defined) bad("scan uses an undefined AC table");
        }
        int16_t block[64];
        if (scan_n_ == 1) { // one component: its blocks in raster order, only those that hold image samples
            Component& c = comp_[order_[0]];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; ++j)
                for (int i = 0; i < bw; ++i) {
                    if (!progressive_) {
                        baseline_block(block, c);
                        idct_block(&c.plane[(size_t)c.w2 * j * 8 + i * 8], c.w2, block);
                    } else {
                        int16_t* data = &c.coeff[64 * ((size_t)i + (size_t)j * c.coeff_w)];
                        if (spec_start_ == 0) progressive_dc(data, c); else progressive_ac(data, c);
                    }
                    if (restart_or_stop()) return;
                }
            return;
        }
        for (int j = 0; j < mcuy_; ++j) // interleaved: MCU by MCU, each component's h x v blocks
            for (int i = 0; i < mcux_; ++i) {
                for (int k = 0; k < scan_n_; ++k) {
                    Component& c = comp_[order_[k]];
                    for (int y = 0; y < c.v; ++y)
                        for (int x = 0; x < c.h; ++x) {
                            const int bx = i * c.h + x, by = j * c.v + y;
                            if (!progressive_) {
                                baseline_block(block, c);
                                idct_block(&c.plane[(size_t)c.w2 * by * 8 + bx * 8], c.w2, block);
                            } else {
                                progressive_dc(&c.coeff[64 * ((size_t)bx + (size_t)by * c.coeff_w)], c);
                            }
                        }
                }
                if (restart_or_stop()) return;
            }
    }
    void finish_progressive()
    {
        for (int n = 0; n < ncomp_; ++n) {
            Component& c = comp_[n];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            const uint16_t* dq = dequant_[c.tq];
            for (int j = 0; j < bh; ++j)
                for (int i = 0; i < bw; ++i) {
                    int16_t* data = &c.coeff[64 * ((size_t)i + (size_t)j * c.coeff_w)];
                    for (int k = 0; k < 64; ++k) data[k] = (int16_t)(data[k] * dq[k]);
                    idct_block(&c.plane[(size_t)c.w2 * j * 8 + i * 8], c.w2, data);
                }
        }
    }

    // ---- output: upsample each component to the frame, then convert
    static const uint8_t* upsample_row(std::vector<uint8_t>& line, const uint8_t* nearRow, const uint8_t* farRow, int w, int hs, int vs)
    {
        uint8_t* out = line.data();
        if (hs == 1 && vs == 1) return nearRow;
        if (hs == 1 && vs == 2) {
            for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * nearRow[i] + farRow[i] + 2) >> 2);
            return out;
        }
        if (hs == 2 && vs == 1) {
            if (w == 1) { out[0] = out[1] = nearRow[0]; return out; }
            out[0] = nearRow[0];
            out[1] = (uint8_t)((nearRow[0] * 3 + nearRow[1] + 2) >> 2);
            for (int i = 1; i < w - 1; ++i) {
                const int n = 3 * nearRow[i] + 2;
                out[2 * i] = (uint8_t)((n + nearRow[i - 1]) >> 2);
                out[2 * i + 1] = (uint8_t)((n + nearRow[i + 1]) >> 2);
            }
            out[2 * (w - 1)] = (uint8_t)((nearRow[w - 2] * 3 + nearRow[w - 1] + 2) >> 2);
            out[2 * (w - 1) + 1] = nearRow[w - 1];
            return out;
        }
        if (hs == 2 && vs == 2) {
            if (w == 1) { out[0] = out[1] = (uint8_t)((3 * nearRow[0] + farRow[0] + 2) >> 2); return out; }
            int t1 = 3 * nearRow[0] + farRow[0];
            out[0] = (uint8_t)((t1 + 2) >> 2);
            for (int i = 1; i < w; ++i) {
                const int t0 = t1;
                t1 = 3 * nearRow[i] + farRow[i];
                out[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
                out[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
            }
            out[2 * w - 1] = (uint8_t)((t1 + 2) >> 2);
            return out;
        }
        for (int i = 0; i < w; ++i) for (int j = 0; j < hs; ++j) out[i * hs + j] = nearRow[i];
        return out;
    }
    static uint8_t mul8(int x, int y) { const unsigned t = (unsigned)(x * y + 128); return (uint8_t)((t + (t >> 8)) >> 8); }
    static void ycc_to_rgba(uint8_t* out, const uint8_t* y, const uint8_t* cb, const uint8_t* cr, int count)
    {
        auto fx = [](float v) { return ((int)(v * 4096.0f + 0.5f)) << 8; };
        static const int kCrR = fx(1.40200f), kCrG = -fx(0.71414f), kCbG = -fx(0.34414f), kCbB = fx(1.77200f);
        for (int i = 0; i < count; ++i, out += 4) {
            const int yf = (y[i] << 20) + (1 << 19);
            const int r8 = cr[i] - 128, b8 = cb[i] - 128;
            int r = yf + r8 * kCrR;
            int g = yf + r8 * kCrG + (int)((unsigned)(b8 * kCbG) & 0xffff0000u);
            int b = yf + b8 * kCbB;
            r >>= 20; g >>= 20; b >>= 20;
            out[0] = clamp8(r); out[1] = clamp8(g); out[2] = clamp8(b); out[3] = 255;
        }
    }
    void assemble(std::vector<uint8_t>& rgba)
    {
        const bool isRgb = ncomp_ == 3 && (rgb_ids_ == 3 || (adobe_transform_ == 0 && !jfif_));
        struct Up { int hs, vs, ystep, ypos, wLow; const uint8_t* line0; const uint8_t* line1; std::vector<uint8_t> buf; } up[4];
        for (int k = 0; k < ncomp_; ++k) {
            up[k].hs = hmax_ / comp_[k].h; up[k].vs = vmax_ / comp_[k].v;
            up[k].ystep = up[k].vs >> 1; up[k].ypos = 0;
            up[k].wLow = (width_ + up[k].hs - 1) / up[k].hs;
            up[k].line0 = up[k].line1 = comp_[k].plane.data();
            up[k].buf.assign((size_t)up[k].wLow * up[k].hs + 8, 0);
        }
        rgba.assign((size_t)width_ * height_ * 4, 255);
        for (int j = 0; j < height_; ++j) {
            const uint8_t* row[4] = { nullptr, nullptr, nullptr, nullptr };
            for (int k = 0; k < ncomp_; ++k) {
                Up& u = up[k];
                const bool bottom = u.ystep >= (u.vs >> 1);
                row[k] = upsample_row(u.buf, bottom ? u.line1 : u.line0, bottom ? u.line0 : u.line1, u.wLow, u.hs, u.vs);
                if (++u.ystep >= u.vs) {
                    u.ystep = 0;
                    u.line0 = u.line1;
                    if (++u.ypos < comp_[k].y) u.line1 += comp_[k].w2;
                }
            }
            uint8_t* out = &rgba[(size_t)width_ * j * 4];
            if (ncomp_ == 3 && isRgb) {
                for (int i = 0; i < width_; ++i) { out[4 * i] = row[0][i]; out[4 * i + 1] = row[1][i]; out[4 * i + 2] = row[2][i]; out[4 * i + 3] = 255; }
            } else if (ncomp_ == 3) {
                ycc_to_rgba(out, row[0], row[1], row[2], width_);
            } else if (ncomp_ == 4 && adobe_transform_ == 0) { // CMYK
                for (int i = 0; i < width_; ++i) {
                    const int m = row[3][i];
                    out[4 * i] = mul8(row[0][i], m); out[4 * i + 1] = mul8(row[1][i], m); out[4 * i + 2] = mul8(row[2][i], m); out[4 * i + 3] = 255;
                }
            } else if (ncomp_ == 4) {
                ycc_to_rgba(out, row[0], row[1], row[2], width_);
                if (adobe_transform_ == 2) // YCCK
                    for (int i = 0; i < width_; ++i) {
                        const int m = row[3][i];
                        out[4 * i] = mul8(255 - out[4 * i], m); out[4 * i + 1] = mul8(255 - out[4 * i + 1], m); out[4 * i + 2] = mul8(255 - out[4 * i + 2], m);
                    }
            } else {
                for (int i = 0; i < width_; ++i) { out[4 * i] = out[4 * i + 1] = out[4 * i + 2] = row[0][i]; out[4 * i + 3] = 255; }
            }
        }
    }
};

} // namespace

bool looks_like_jpeg(const uint8_t* data, size_t size) { return size >= 3 && data[0] == 0xFF && data[1] == 0xD8 && data[2] == 0xFF; }

void decode_jpeg_rgba(const uint8_t* data, size_t size, int& w, int& h, std::vector<uint8_t>& rgba)
{
    Decoder d(data, size);
    d.decode(w, h, rgba);
}

} // namespace ptamd
