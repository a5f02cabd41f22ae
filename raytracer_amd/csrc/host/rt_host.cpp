// rt_host.cpp -- scene-building API, BVH construction, .obj loading, flattening, camera set-up and PNG output.
// See include/rt_host.hpp.  Built with -ffp-contract=off: the constants derived here (boxes, plane normals, w
// vectors, camera frame) must carry the reference's rounding, since the device compares against them exactly.
#include "rt_host.hpp"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
#include <iterator>
#include <limits>
#include <map>
#include <sstream>
#include <stdexcept>

namespace rtamd {

static const double INF = std::numeric_limits<double>::infinity();
static const double PI = 3.14159265358979323846264338327950288;

// ---- src/point3/mod.rs operators (only what construction needs) ---------------------------------------------
static inline Point3 operator-(Point3 a) { return Point3(-a.x, -a.y, -a.z); }
static inline Point3 operator+(Point3 a, Point3 b) { return Point3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline Point3 operator-(Point3 a, Point3 b) { return Point3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline Point3 operator*(Point3 a, double s) { return Point3(a.x * s, a.y * s, a.z * s); }
static inline Point3 operator*(double s, Point3 a) { return Point3(s * a.x, s * a.y, s * a.z); }
static inline Point3 operator*(Point3 a, Point3 b) { return Point3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline Point3 operator/(Point3 a, double s) { return a * (1.0 / s); }  // point3/mod.rs:137-144
static inline double dot(Point3 u, Point3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
static inline double length(Point3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline Point3 cross(Point3 u, Point3 v) {
    return Point3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
static inline Point3 unit_vector(Point3 u) { return u / length(u); }

// ---- src/aabb.rs --------------------------------------------------------------------------------------------
static Range pad_to_minimums(Range r) {  // aabb.rs:71-85
    const double DELTA = 0.0001;
    if (r.end < r.start) throw std::runtime_error("Tried to pad reversed (and therefore empty) range");
    if (r.end - r.start < DELTA) return Range{r.start + DELTA, r.end + DELTA};
    return r;
}
AABB AABB::from_points(Point3 a, Point3 b) {  // aabb.rs:16-26
    AABB o;
    o.x = pad_to_minimums(a.x <= b.x ? Range{a.x, b.x} : Range{b.x, a.x});
    o.y = pad_to_minimums(a.y <= b.y ? Range{a.y, b.y} : Range{b.y, a.y});
    o.z = pad_to_minimums(a.z <= b.z ? Range{a.z, b.z} : Range{b.z, a.z});
    return o;
}
AABB::AABB() {
    Range r = pad_to_minimums(Range{0.0, 0.0});
    x = y = z = r;
}
int AABB::longest_axis() const {  // aabb.rs:251-264
    double xs = x.end - x.start, ys = y.end - y.start, zs = z.end - z.start;
    if (xs > ys && xs > zs) return 0;
    if (ys > xs && ys > zs) return 1;
    return 2;
}
AABB AABB::shifted(Point3 by) const {
    AABB o;
    o.x = Range{x.start + by.x, x.end + by.x};
    o.y = Range{y.start + by.y, y.end + by.y};
    o.z = Range{z.start + by.z, z.end + by.z};
    return o;
}
static Range unite(const Range& a, const Range& b) { return Range{std::fmin(a.start, b.start), std::fmax(a.end, b.end)}; }
AABB join_aabbs(const AABB& a, const AABB& b) {  // aabb.rs:93-103
    AABB o;
    o.x = pad_to_minimums(unite(a.x, b.x));
    o.y = pad_to_minimums(unite(a.y, b.y));
    o.z = pad_to_minimums(unite(a.z, b.z));
    return o;
}

// ---- textures / materials -----------------------------------------------------------------------------------
std::shared_ptr<SolidColor> SolidColor::create(Color c) {
    auto t = std::make_shared<SolidColor>();
    t->albedo = c;
    return t;
}
std::shared_ptr<CheckerTexture> CheckerTexture::from_pointers(double scale, TexturePtr even, TexturePtr odd) {
    auto t = std::make_shared<CheckerTexture>();
    t->inverse_scale = 1.0 / scale;
    t->even = even;
    t->odd = odd;
    return t;
}
std::shared_ptr<CheckerTexture> CheckerTexture::from_colors(double scale, Color even, Color odd) {
    return from_pointers(scale, SolidColor::create(even), SolidColor::create(odd));
}
std::shared_ptr<ImageTexture> ImageTexture::from_rgb8(const uint8_t* rgb, uint32_t w, uint32_t h) {
    auto t = std::make_shared<ImageTexture>();
    t->rgb.assign(rgb, rgb + (size_t)3 * w * h);
    t->width = w;
    t->height = h;
    return t;
}
static bool read_ppm(const std::string& path, std::vector<uint8_t>& rgb, uint32_t& w, uint32_t& h) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string magic;
    f >> magic;
    if (magic != "P6") return false;
    auto next_int = [&](long& v) -> bool {
        for (;;) {
            int c = f.peek();
            if (c == '#') { std::string skip; std::getline(f, skip); continue; }
            if (c == ' ' || c == '\n' || c == '\r' || c == '\t') { f.get(); continue; }
            break;
        }
        return (bool)(f >> v);
    };
    long W, H, M;
    if (!next_int(W) || !next_int(H) || !next_int(M)) return false;
    if (W <= 0 || H <= 0 || M != 255 || W > 65536 || H > 65536) return false;
    f.get();  // the single whitespace after maxval
    rgb.resize((size_t)3 * W * H);
    f.read((char*)rgb.data(), (std::streamsize)rgb.size());
    if ((size_t)f.gcount() != rgb.size()) return false;
    w = (uint32_t)W;
    h = (uint32_t)H;
    return true;
}
// PNG (the reference decodes through the `image` crate, texture.rs:90-93): 8- and 16-bit grey, grey + alpha, RGB,
// RGBA and palette images, filters 0..4; interlaced files are refused.  Alpha is dropped and 16-bit samples keep their
// high byte, where `image`'s to_rgb8() rounds (v * 255 + 32767) / 65535: the two differ by one level for some samples
// (the reference holds no 16-bit image, so that case is unpinned).
static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
static bool read_png(const std::string& path, std::vector<uint8_t>& rgb, uint32_t& w, uint32_t& h) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) return false;
    std::vector<uint8_t> idat, palette;
    uint32_t W = 0, H = 0, depth = 0, ctype = 0;
    bool have_ihdr = false, done = false;
    for (size_t at = 8; at + 12 <= file.size() && !done;) {
        const uint32_t len = rd32(&file[at]);
        if (len > file.size() || at + 12 + (size_t)len > file.size()) return false;
        const uint8_t* tag = &file[at + 4];
        const uint8_t* data = &file[at + 8];
        if ((uint32_t)crc32(0L, tag, (uInt)(len + 4)) != rd32(data + len)) return false;
        if (!std::memcmp(tag, "IHDR", 4)) {
            if (len != 13) return false;
            W = rd32(data); H = rd32(data + 4); depth = data[8]; ctype = data[9];
            if (data[10] != 0 || data[11] != 0 || data[12] != 0) return false;  // compression, filter method, interlace
            have_ihdr = true;
        } else if (!std::memcmp(tag, "PLTE", 4)) {
            palette.assign(data, data + len);
        } else if (!std::memcmp(tag, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(tag, "IEND", 4)) {
            done = true;
        }
        at += 12 + (size_t)len;
    }
    if (!have_ihdr || !done || W == 0 || H == 0 || W > 65536 || H > 65536) return false;
    uint32_t channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: return false;
    }
    if (ctype == 3 ? (depth != 1 && depth != 2 && depth != 4 && depth != 8)
                   : (depth != 8 && depth != 16 && !(ctype == 0 && (depth == 1 || depth == 2 || depth == 4)))) return false;
    if (ctype == 3 && (palette.empty() || palette.size() % 3 != 0)) return false;
    const size_t bits_per_pixel = (size_t)channels * depth;
    const size_t bpp = std::max<size_t>(1, bits_per_pixel / 8);      // the filters' byte distance
    const size_t stride = ((size_t)W * bits_per_pixel + 7) / 8;
    // deflate cannot expand beyond 1032 : 1, so a header that promises more than the IDAT data can hold is refused
    // before anything of that size is allocated (and nothing above 1 GiB decoded is taken at all)
    const size_t raw_size = (stride + 1) * (size_t)H;
    if (raw_size > ((size_t)1 << 30) || raw_size / 1032 > idat.size() + 1) return false;
    std::vector<uint8_t> raw(raw_size);
    uLongf got = (uLongf)raw.size();
    if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size()) return false;
    std::vector<uint8_t> prev(stride, 0), line(stride);
    rgb.assign((size_t)3 * W * H, 0);
    for (uint32_t y = 0; y < H; ++y) {
        const uint8_t* in = &raw[(stride + 1) * (size_t)y];
        const uint8_t ft = in[0];
        if (ft > 4) return false;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? line[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) / 2;
            else if (ft == 4) {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            line[i] = (uint8_t)(in[1 + i] + pred);
        }
        uint8_t* out = &rgb[(size_t)3 * W * y];
        for (uint32_t x = 0; x < W; ++x) {
            auto sample = [&](uint32_t ch) -> uint32_t {  // channel ch of pixel x, as stored
                if (depth == 8) return line[(size_t)x * channels + ch];
                if (depth == 16) return line[((size_t)x * channels + ch) * 2];  // high byte
                const size_t bit = (size_t)x * depth;                           // 1, 2, 4 bits: one channel
                return (line[bit / 8] >> (8 - depth - bit % 8)) & ((1u << depth) - 1u);
            };
            if (ctype == 3) {
                const uint32_t idx = sample(0);
                if ((size_t)3 * idx + 2 >= palette.size()) return false;
                out[3 * x] = palette[3 * idx]; out[3 * x + 1] = palette[3 * idx + 1]; out[3 * x + 2] = palette[3 * idx + 2];
            } else if (ctype == 0 || ctype == 4) {
                uint32_t v = sample(0);
                if (depth < 8) v = v * 255u / ((1u << depth) - 1u);
                out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = (uint8_t)v;
            } else {
                out[3 * x] = (uint8_t)sample(0); out[3 * x + 1] = (uint8_t)sample(1); out[3 * x + 2] = (uint8_t)sample(2);
            }
        }
        prev.swap(line);
    }
    w = W;
    h = H;
    return true;
}
// JPEG (the reference decodes textures/earthmap.jpg through the `image` crate, texture.rs:90-93, which hands it to
// zune-jpeg): baseline and extended sequential DCT, Huffman coded, 8 bits per sample, one (grey) or three (YCbCr)
// components with sampling factors 1 or 2, restart intervals.  Progressive, arithmetic-coded, lossless, 12-bit and
// four-component files are refused (ImageTexture then falls back as it does for any unreadable file).  The inverse
// transform is the 13-bit fixed-point one of the IJG's jidctint, chroma is brought to full size with the IJG's
// triangle filter ("fancy upsampling"), the colour conversion is JFIF's; decoders agree on such an image to within a
// level or two per sample, not bit for bit, so texel values are "as decoded here" -- feed PNG or PPM where they matter.
namespace jpeg {
struct Huff { uint8_t bits[17] = {0}; uint8_t vals[256] = {0}; int mincode[17], maxcode[18], valptr[17]; bool set = false; };
static void huff_build(Huff& h) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        h.valptr[l] = k; h.mincode[l] = code;
        code += h.bits[l]; k += h.bits[l];
        h.maxcode[l] = h.bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    h.maxcode[17] = 0x7FFFFFFF;
    h.set = true;
}
struct Bits {
    const uint8_t* p; const uint8_t* end; uint32_t acc = 0; int n = 0; bool marker = false;
    int bit() {
        if (n == 0) {
            uint8_t b = 0;
            if (!marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    if (p < end && *p == 0x00) ++p;           // a stuffed zero
                    else { marker = true; --p; b = 0; }       // a marker: the scan's data end here (zeros from now on)
                }
            }
            acc = b; n = 8;
        }
        --n;
        return (int)((acc >> n) & 1u);
    }
    int get(int count) { int v = 0; while (count-- > 0) v = (v << 1) | bit(); return v; }
    void reset() { n = 0; acc = 0; }
};
static int decode(Bits& b, const Huff& h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | b.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}
static int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }
static const uint8_t zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
// jidctint.c's accurate integer inverse DCT (CONST_BITS 13, PASS1_BITS 2) on dequantised coefficients.
static void idct(const int* in, uint8_t* out, size_t stride) {
    const int C0298 = 2446, C0390 = 3196, C0541 = 4433, C0765 = 6270, C0899 = 7373, C1175 = 9633, C1501 = 12299, C1847 = 15137,
              C1961 = 16069, C2053 = 16819, C2562 = 20995, C3072 = 25172;
    long ws[64];
    for (int c = 0; c < 8; ++c) {
        const int* p = in + c;
        long z2 = p[16], z3 = p[48];
        long z1 = (z2 + z3) * C0541;
        long tmp2 = z1 + z3 * (-C1847), tmp3 = z1 + z2 * C0765;
        z2 = p[0]; z3 = p[32];
        long tmp0 = (z2 + z3) << 13, tmp1 = (z2 - z3) << 13;
        long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = p[56]; tmp1 = p[40]; tmp2 = p[24]; tmp3 = p[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
        long z5 = (z3 + z4) * C1175;
        tmp0 *= C0298; tmp1 *= C2053; tmp2 *= C3072; tmp3 *= C1501;
        z1 *= -C0899; z2 *= -C2562; z3 *= -C1961; z4 *= -C0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        auto ds = [](long x) { return (x + (1L << 10)) >> 11; };
        ws[c] = ds(tmp10 + tmp3); ws[56 + c] = ds(tmp10 - tmp3); ws[8 + c] = ds(tmp11 + tmp2); ws[48 + c] = ds(tmp11 - tmp2);
        ws[16 + c] = ds(tmp12 + tmp1); ws[40 + c] = ds(tmp12 - tmp1); ws[24 + c] = ds(tmp13 + tmp0); ws[32 + c] = ds(tmp13 - tmp0);
    }
    for (int r = 0; r < 8; ++r) {
        const long* p = ws + 8 * r;
        long z2 = p[2], z3 = p[6];
        long z1 = (z2 + z3) * C0541;
        long tmp2 = z1 + z3 * (-C1847), tmp3 = z1 + z2 * C0765;
        long tmp0 = (p[0] + p[4]) << 13, tmp1 = (p[0] - p[4]) << 13;
        long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = p[7]; tmp1 = p[5]; tmp2 = p[3]; tmp3 = p[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
        long z5 = (z3 + z4) * C1175;
        tmp0 *= C0298; tmp1 *= C2053; tmp2 *= C3072; tmp3 *= C1501;
        z1 *= -C0899; z2 *= -C2562; z3 *= -C1961; z4 *= -C0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        auto px = [](long x) { long v = ((x + (1L << 17)) >> 18) + 128; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
        uint8_t* o = out + stride * r;
        o[0] = px(tmp10 + tmp3); o[7] = px(tmp10 - tmp3); o[1] = px(tmp11 + tmp2); o[6] = px(tmp11 - tmp2);
        o[2] = px(tmp12 + tmp1); o[5] = px(tmp12 - tmp1); o[3] = px(tmp13 + tmp0); o[4] = px(tmp13 - tmp0);
    }
}
struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; size_t bw = 0, bh = 0; std::vector<uint8_t> px; };
}  // namespace jpeg
static bool read_jpeg(const std::string& path, std::vector<uint8_t>& rgb, uint32_t& w, uint32_t& h) {
    using namespace jpeg;
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (file.size() < 4 || file[0] != 0xFF || file[1] != 0xD8) return false;
    uint16_t qt[4][64];
    bool have_q[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Comp> comps;
    uint32_t W = 0, H = 0;
    int restart = 0, hmax = 1, vmax = 1;
    bool have_frame = false;
    size_t at = 2;
    auto rd16 = [&](size_t i) { return (uint32_t)file[i] << 8 | file[i + 1]; };
    while (at + 4 <= file.size()) {
        if (file[at] != 0xFF) return false;
        const uint8_t m = file[at + 1];
        if (m == 0xFF) { ++at; continue; }            // fill bytes
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) { at += 2; continue; }
        if (m == 0xD9) return false;                  // the image ends without a scan
        const size_t len = rd16(at + 2);
        if (len < 2 || at + 2 + len > file.size()) return false;
        const uint8_t* d = &file[at + 4];
        const size_t n = len - 2;
        if (m == 0xDB) {                              // quantisation tables
            for (size_t i = 0; i < n;) {
                const int pq = d[i] >> 4, tq = d[i] & 15;
                if (tq > 3 || pq > 1 || i + 1 + 64u * (pq + 1) > n) return false;
                for (int k = 0; k < 64; ++k) qt[tq][zigzag[k]] = pq ? (uint16_t)(d[i + 1 + 2 * k] << 8 | d[i + 2 + 2 * k]) : d[i + 1 + k];
                have_q[tq] = true;
                i += 1 + 64u * (pq + 1);
            }
        } else if (m == 0xC4) {                       // Huffman tables
            for (size_t i = 0; i < n;) {
                if (i + 17 > n) return false;
                const int tc = d[i] >> 4, th = d[i] & 15;
                if (tc > 1 || th > 3) return false;
                Huff& t = tc ? ac[th] : dc[th];
                size_t total = 0;
                for (int l = 1; l <= 16; ++l) { t.bits[l] = d[i + l]; total += d[i + l]; }
                if (total > 256 || i + 17 + total > n) return false;
                std::memcpy(t.vals, d + i + 17, total);
                huff_build(t);
                i += 17 + total;
            }
        } else if (m == 0xC0 || m == 0xC1) {          // baseline / extended sequential frame
            if (have_frame || n < 6 || d[0] != 8) return false;
            H = rd16(at + 5); W = rd16(at + 7);
            const int nc = d[5];
            if (W == 0 || H == 0 || (nc != 1 && nc != 3) || n < 6u + 3u * nc) return false;
            comps.resize((size_t)nc);
            for (int c = 0; c < nc; ++c) {
                comps[c].id = d[6 + 3 * c]; comps[c].h = d[7 + 3 * c] >> 4; comps[c].v = d[7 + 3 * c] & 15; comps[c].tq = d[8 + 3 * c];
                if (comps[c].h < 1 || comps[c].h > 2 || comps[c].v < 1 || comps[c].v > 2 || comps[c].tq > 3) return false;
                hmax = std::max(hmax, comps[c].h); vmax = std::max(vmax, comps[c].v);
            }
            if (nc == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }
            have_frame = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return false;                             // progressive, lossless, arithmetic: not read here
        } else if (m == 0xDD) {
            if (n < 2) return false;
            restart = (int)rd16(at + 4);
        } else if (m == 0xDA) {                       // the scan (one, holding every component)
            if (!have_frame || n < 1 || d[0] != comps.size() || n < 1u + 2u * comps.size() + 3u) return false;
            for (size_t c = 0; c < comps.size(); ++c) {
                if (d[1 + 2 * c] != comps[c].id) return false;
                comps[c].td = d[2 + 2 * c] >> 4; comps[c].ta = d[2 + 2 * c] & 15;
                if (comps[c].td > 3 || comps[c].ta > 3 || !dc[comps[c].td].set || !ac[comps[c].ta].set || !have_q[comps[c].tq]) return false;
            }
            const size_t mcux = ((size_t)W + 8u * hmax - 1) / (8u * hmax), mcuy = ((size_t)H + 8u * vmax - 1) / (8u * vmax);
            if (mcux * mcuy > (1u << 22)) return false;
            for (Comp& c : comps) { c.bw = mcux * c.h * 8; c.bh = mcuy * c.v * 8; c.px.assign(c.bw * c.bh, 0); c.pred = 0; }
            Bits b; b.p = d + n; b.end = file.data() + file.size();
            int coef[64], left = restart;
            for (size_t my = 0; my < mcuy; ++my)
                for (size_t mx = 0; mx < mcux; ++mx) {
                    if (restart && left == 0) {       // a restart marker: byte-align, skip it, reset the predictions
                        b.reset(); b.marker = false;
                        while (b.p + 1 < b.end && !(b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7)) ++b.p;
                        if (b.p + 1 < b.end) b.p += 2;
                        for (Comp& c : comps) c.pred = 0;
                        left = restart;
                    }
                    for (Comp& c : comps)
                        for (int by = 0; by < c.v; ++by)
                            for (int bx = 0; bx < c.h; ++bx) {
                                std::memset(coef, 0, sizeof coef);
                                const int t = decode(b, dc[c.td]);
                                if (t < 0 || t > 11) return false;
                                c.pred += extend(b.get(t), t);
                                coef[0] = c.pred * qt[c.tq][0];
                                for (int k = 1; k < 64;) {
                                    const int rs = decode(b, ac[c.ta]);
                                    if (rs < 0) return false;
                                    const int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                                    k += r;
                                    if (k > 63) return false;
                                    coef[zigzag[k]] = extend(b.get(sz), sz) * qt[c.tq][zigzag[k]];
                                    ++k;
                                }
                                idct(coef, &c.px[(my * c.v + by) * 8 * c.bw + (mx * c.h + bx) * 8], c.bw);
                            }
                    if (restart) --left;
                }
            // full-size planes: components sampled at half rate go through the triangle filter, 3/4 nearer + 1/4 farther
            // sample, columns first (h2v1), then rows (rounding as the IJG's jdsample.c)
            std::vector<std::vector<uint8_t>> full(comps.size());
            for (size_t ci = 0; ci < comps.size(); ++ci) {
                Comp& c = comps[ci];
                const size_t cw = ((size_t)W * c.h + hmax - 1) / hmax, ch = ((size_t)H * c.v + vmax - 1) / vmax;  // samples that count
                std::vector<uint8_t> wide;
                size_t ww = cw;
                if (c.h == hmax) { wide.resize(cw * ch); for (size_t y = 0; y < ch; ++y) std::memcpy(&wide[y * cw], &c.px[y * c.bw], cw); }
                else {
                    ww = 2 * cw; wide.resize(ww * ch);
                    for (size_t y = 0; y < ch; ++y) {
                        const uint8_t* in = &c.px[y * c.bw];
                        uint8_t* o = &wide[y * ww];
                        for (size_t x = 0; x < cw; ++x) {
                            const int cur = in[x], prev = in[x ? x - 1 : 0], next = in[x + 1 < cw ? x + 1 : x];
                            o[2 * x] = (uint8_t)(x == 0 ? cur : (3 * cur + prev + 1) >> 2);
                            o[2 * x + 1] = (uint8_t)(x + 1 == cw ? cur : (3 * cur + next + 2) >> 2);
                        }
                    }
                }
                if (c.v == vmax) { full[ci].swap(wide); full[ci].resize(ww * ch); }
                else {
                    full[ci].resize(ww * 2 * ch);
                    for (size_t y = 0; y < ch; ++y)
                        for (int half = 0; half < 2; ++half) {
                            const uint8_t* near = &wide[y * ww];
                            const uint8_t* far = &wide[(half == 0 ? (y ? y - 1 : 0) : (y + 1 < ch ? y + 1 : y)) * ww];
                            uint8_t* o = &full[ci][(2 * y + half) * ww];
                            for (size_t x = 0; x < ww; ++x) o[x] = (uint8_t)((3 * near[x] + far[x] + (half == 0 ? 1 : 2)) >> 2);
                        }
                }
                c.bw = ww;  // (row length of the full-size plane)
            }
            rgb.assign((size_t)3 * W * H, 0);
            for (size_t y = 0; y < H; ++y)
                for (size_t x = 0; x < W; ++x) {
                    uint8_t* o = &rgb[3 * (y * W + x)];
                    const int Y = full[0][y * comps[0].bw + x];
                    if (comps.size() == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; continue; }
                    const int cb = full[1][y * comps[1].bw + x] - 128, cr = full[2][y * comps[2].bw + x] - 128;
                    auto clamp = [](int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
                    // JFIF: R = Y + 1.402 Cr, G = Y - 0.344136 Cb - 0.714136 Cr, B = Y + 1.772 Cb (16-bit fixed point, rounded)
                    o[0] = clamp(Y + ((91881 * cr + 32768) >> 16));
                    o[1] = clamp(Y + ((-22554 * cb - 46802 * cr + 32768) >> 16));
                    o[2] = clamp(Y + ((116130 * cb + 32768) >> 16));
                }
            w = W; h = H;
            return true;
        }
        at += 2 + len;
    }
    return false;
}
bool read_image_rgb8(const std::string& path, std::vector<uint8_t>& rgb, uint32_t& w, uint32_t& h) {
    try {
        return read_ppm(path, rgb, w, h) || read_png(path, rgb, w, h) || read_jpeg(path, rgb, w, h);
    } catch (const std::bad_alloc&) {  // a file too large for this host is a file that cannot be read
        rgb.clear();
        return false;
    }
}
TexturePtr ImageTexture::new_or_fallback(const std::string& path) {
    std::vector<uint8_t> rgb;
    uint32_t w = 0, h = 0;
    if (read_image_rgb8(path, rgb, w, h)) return from_rgb8(rgb.data(), w, h);
    std::fprintf(stderr, "Could not load the image texture %s. Falling back to default.\n", path.c_str());
    return CheckerTexture::from_colors(2.0, Color(1.0, 0.0, 0.862745098039), Color(0.00392156862745, 0.0, 0.00392156862745));
}
PerlinNoise create_perlin_noise(const double* d) {
    PerlinNoise n;
    for (int i = 0; i < 256; ++i) {
        n.random_vectors[i][0] = d[3 * i];
        n.random_vectors[i][1] = d[3 * i + 1];
        n.random_vectors[i][2] = d[3 * i + 2];
        n.x_perm[i] = n.y_perm[i] = n.z_perm[i] = (uint32_t)i;
    }
    return n;
}
std::shared_ptr<PerlinNoiseTexture> PerlinNoiseTexture::create(const PerlinNoise& n, double scale) {
    auto t = std::make_shared<PerlinNoiseTexture>();
    t->perlin_noise = n;
    t->scale = scale;
    return t;
}
std::shared_ptr<Lambertian> Lambertian::from_color(Color c) { return from_texture(SolidColor::create(c)); }
std::shared_ptr<Lambertian> Lambertian::from_texture(TexturePtr t) {
    auto m = std::make_shared<Lambertian>();
    m->texture = t;
    return m;
}
std::shared_ptr<Metal> metal(Color albedo, double fuzz) {
    auto m = std::make_shared<Metal>();
    m->albedo = albedo;
    m->fuzz = fuzz;
    return m;
}
std::shared_ptr<Dielectric> dielectric(double ri) {
    auto m = std::make_shared<Dielectric>();
    m->refraction_index = ri;
    return m;
}
std::shared_ptr<DiffuseLight> DiffuseLight::from_color(Color c) { return from_texture(SolidColor::create(c)); }
std::shared_ptr<DiffuseLight> DiffuseLight::from_texture(TexturePtr t) {
    auto m = std::make_shared<DiffuseLight>();
    m->texture = t;
    return m;
}

// ---- primitives ---------------------------------------------------------------------------------------------
Sphere::Sphere(Point3 c, double r, MaterialPtr m) : center(c), radius(std::fmax(r, 0.0)), material(m) {
    Point3 rv(r, r, r);
    bbox_ = AABB::from_points(c - rv, c + rv);  // sphere.rs:24-29
}
static void derive_plane(Point3 q, Vector3 u, Vector3 v, Vector3& normal, double& d, Vector3& w) {
    Vector3 n = cross(u, v);  // parallelogram.rs:43-46, triangle.rs:42-45
    normal = unit_vector(n);
    d = dot(normal, q);
    w = n / dot(n, n);
}
Parallelogram::Parallelogram(Point3 q_, Vector3 u_, Vector3 v_, MaterialPtr m) : q(q_), u(u_), v(v_), material(m) {
    AABB b0 = AABB::from_points(q, q + u + v);  // parallelogram.rs:31-37
    AABB b1 = AABB::from_points(q + u, q + v);
    bbox_ = join_aabbs(b0, b1);
    derive_plane(q, u, v, normal, d, w);
}
Triangle::Triangle(Point3 q_, Vector3 u_, Vector3 v_, MaterialPtr m) : q(q_), u(u_), v(v_), material(m) {
    AABB b0 = AABB::from_points(q, q + u);  // triangle.rs:30-36
    AABB b1 = AABB::from_points(q, q + v);
    bbox_ = join_aabbs(b0, b1);
    derive_plane(q, u, v, normal, d, w);
}
std::shared_ptr<Triangle> Triangle::from_vertex_locations(Point3 v1, Point3 v2, Point3 v3, MaterialPtr m) {
    return std::make_shared<Triangle>(v1, v2 - v1, v3 - v1, m);  // triangle.rs:58-87
}
std::shared_ptr<Quadric> make_quadric(Point3 p1, Point3 p2, Point3 p3, double j, MaterialPtr m, AABB box) {
    auto q = std::make_shared<Quadric>();
    q->p1 = p1; q->p2 = p2; q->p3 = p3; q->j = j; q->material = m; q->bbox_ = box;
    return q;
}
std::shared_ptr<Quadric> y_cylinder(Point3 c, double r, MaterialPtr m) {
    Point3 rv(r, r, r);
    return make_quadric(Point3(1.0, 0.0, 1.0), Point3(), c * Point3(-2.0, 0.0, -2.0), -r * r + c.x * c.x + c.z * c.z, m,
                        AABB::from_points(c - rv, c + rv));
}
std::shared_ptr<Quadric> quadric_sphere(Point3 c, double r, MaterialPtr m) {
    Point3 rv(r, r, r);
    return make_quadric(Point3(1.0, 1.0, 1.0), Point3(), -2.0 * c, -r * r + dot(c, c), m, AABB::from_points(c - rv, c + rv));
}
std::shared_ptr<Quadric> y_cone(Point3 c, Point3 off, MaterialPtr m) {
    Point3 ind(1.0, -1.0, 1.0);
    return make_quadric(ind, Point3(), (-2.0 * ind) * c, dot(c * ind, c), m, AABB::from_points(c - off, c + off));
}

// ---- collections --------------------------------------------------------------------------------------------
void HittableList::add_pointer(HittablePtr o) {  // hittable_list.rs:32-40
    bounding_box = join_aabbs(bounding_box, o->bounding_box());
    objects.push_back(o);
}
std::shared_ptr<HittableSlice> HittableList::to_hittable_slice() const { return HittableSlice::from_hittable_list(*this); }
std::shared_ptr<BVHNode> HittableList::to_bvh_node() const { return BVHNode::from_hittable_list(*this); }

namespace {
struct SliceAccess : HittableSlice {
    void set_box(const AABB& b) { bbox_ = b; }
};
}  // namespace
std::shared_ptr<HittableSlice> HittableSlice::from_hittable_list(const HittableList& l) {
    auto s = std::make_shared<SliceAccess>();
    s->objects = l.objects;
    s->set_box(l.bounding_box);
    return s;
}
HittableList create_box(Point3 a, Point3 b, MaterialPtr m) {  // parallelogram.rs:146-195
    HittableList sides;
    Point3 mn(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z));
    Point3 mx(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z));
    Point3 dx(mx.x - mn.x, 0.0, 0.0), dy(0.0, mx.y - mn.y, 0.0), dz(0.0, 0.0, mx.z - mn.z);
    sides.add(std::make_shared<Parallelogram>(Point3(mn.x, mn.y, mx.z), dx, dy, m));
    sides.add(std::make_shared<Parallelogram>(Point3(mx.x, mn.y, mx.z), -dz, dy, m));
    sides.add(std::make_shared<Parallelogram>(Point3(mx.x, mn.y, mn.z), -dx, dy, m));
    sides.add(std::make_shared<Parallelogram>(Point3(mn.x, mn.y, mn.z), dz, dy, m));
    sides.add(std::make_shared<Parallelogram>(Point3(mn.x, mx.y, mx.z), dx, -dz, m));
    sides.add(std::make_shared<Parallelogram>(Point3(mn.x, mn.y, mn.z), dx, dz, m));
    return sides;
}

static inline int64_t total_order_key(double x) {  // f64::total_cmp
    int64_t b;
    std::memcpy(&b, &x, 8);
    b ^= (int64_t)((uint64_t)(b >> 63) >> 1);
    return b;
}
BVHNode::BVHNode(std::vector<HittablePtr> objs) {  // bvh.rs:87-138
    AABB box;
    for (auto& o : objs) box = join_aabbs(box, o->bounding_box());
    int axis = box.longest_axis();
    const size_t THRESHOLD = 16;
    if (objs.size() <= THRESHOLD) {
        HittableList l;
        for (auto& o : objs) l.add_pointer(o);
        leaf = true;
        objects = HittableSlice::from_hittable_list(l);
        bbox_ = l.bounding_box;
    } else {
        // Vec::sort_by is stable; the upper half (split_off(mid)) becomes `left` and is visited first.
        std::stable_sort(objs.begin(), objs.end(), [axis](const HittablePtr& a, const HittablePtr& b) {
            return total_order_key(a->bounding_box().axis_interval(axis).start) <
                   total_order_key(b->bounding_box().axis_interval(axis).start);
        });
        size_t mid = objs.size() / 2;
        std::vector<HittablePtr> upper(objs.begin() + (std::ptrdiff_t)mid, objs.end());
        objs.resize(mid);
        left = std::make_shared<BVHNode>(std::move(upper));
        right = std::make_shared<BVHNode>(std::move(objs));
        bbox_ = box;
    }
}
std::shared_ptr<BVHNode> BVHNode::from_hittable_list(const HittableList& l) { return std::make_shared<BVHNode>(l.objects); }
std::shared_ptr<BVHNode> BVHNode::from_vec(std::vector<HittablePtr> o) { return std::make_shared<BVHNode>(std::move(o)); }

Translate::Translate(HittablePtr o, Vector3 off) : object(o), offset(off) { bbox_ = o->bounding_box().shifted(off); }
RotateY::RotateY(HittablePtr o, double deg) : object(o) {  // hittable/mod.rs:143-188
    double radians = deg * (PI / 180.0);
    sin_theta = std::sin(radians);
    cos_theta = std::cos(radians);
    const AABB& bb = o->bounding_box();
    double mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF};
    for (int ii = 0; ii < 2; ++ii)
        for (int jj = 0; jj < 2; ++jj)
            for (int kk = 0; kk < 2; ++kk) {
                double i = ii, j = jj, k = kk;
                double x = i * bb.x.end + (1.0 - i) * bb.x.start;
                double y = j * bb.y.end + (1.0 - j) * bb.y.start;
                double z = k * bb.z.end + (1.0 - k) * bb.z.start;
                double t[3] = {cos_theta * x + sin_theta * z, y, -sin_theta * x + cos_theta * z};
                // As the reference computes it (:174-177): only x and y, and the maximum also takes `min`.  The box
                // is never tested in the reference's scenes (rotated instances sit in top-level slices); it is kept
                // so that enclosing boxes come out the same.
                for (int c = 0; c < 2; ++c) {
                    mn[c] = std::fmin(mn[c], t[c]);
                    mx[c] = std::fmin(mx[c], t[c]);
                }
            }
    bbox_ = AABB::from_points(Point3(mn[0], mn[1], mn[2]), Point3(mx[0], mx[1], mx[2]));
}
ConstantMedium::ConstantMedium(HittablePtr b, double density, TexturePtr tex) : boundary(b), neg_inv_density(-1.0 / density) {
    auto iso = std::make_shared<Isotropic>();
    iso->texture = tex;
    phase_function = iso;
    bbox_ = b->bounding_box();
}
std::shared_ptr<ConstantMedium> ConstantMedium::from_color(HittablePtr b, double density, Color c) {
    return std::make_shared<ConstantMedium>(b, density, SolidColor::create(c));
}

// ---- load_obj.rs -------------------------------------------------------------------------------------------
std::shared_ptr<BVHNode> load_model(const std::string& path, double scale, MaterialPtr material) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("load_model: cannot open " + path);
    std::vector<Point3> verts;
    std::vector<std::vector<size_t>> faces;
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ls(line);
        std::string first;
        if (!(ls >> first)) continue;
        if (first == "v") {
            std::vector<double> c;
            std::string tok;
            while (ls >> tok) {
                char* endp = nullptr;
                double v = std::strtod(tok.c_str(), &endp);
                if (endp == tok.c_str() || *endp) throw std::runtime_error("load_model: bad coordinate '" + tok + "'");
                c.push_back(v);
            }
            if (c.size() != 3) throw std::runtime_error("unable to parse non 3d coordinates");
            verts.push_back(Point3(c[0], c[1], c[2]));
        } else if (first == "f") {
            std::vector<size_t> vs;
            std::string tok;
            while (ls >> tok) {
                std::string head = tok.substr(0, tok.find('/'));
                char* endp = nullptr;
                long long i = std::strtoll(head.c_str(), &endp, 10);
                if (endp == head.c_str() || *endp || i < 1) throw std::runtime_error("load_model: bad face index '" + tok + "'");
                vs.push_back((size_t)i - 1);
            }
            faces.push_back(vs);
        } else if (first == "vn" || first == "vt" || first == "#" || first == "s" || first == "o" || first == "vp" ||
                   first == "usemtl" || first == "mtllib" || first == "g") {
            // ignored (load_obj.rs:55-73)
        } else {
            throw std::runtime_error("Unexpected line in `.obj` file. Line is: " + line);
        }
    }
    std::vector<HittablePtr> triangles;
    for (auto& face : faces) {
        if (face.empty()) continue;
        for (size_t ix : face)
            if (ix >= verts.size()) throw std::runtime_error("load_model: face index out of range");
        auto scaled = [&](size_t ix) { return Point3(scale * verts[ix].x, scale * verts[ix].y, scale * verts[ix].z); };
        Point3 v1 = scaled(face[0]);
        for (size_t k = 1; k + 1 < face.size(); ++k)  // fan triangulation, load_obj.rs:88-106
            triangles.push_back(Triangle::from_vertex_locations(v1, scaled(face[k]), scaled(face[k + 1]), material));
    }
    return BVHNode::from_vec(std::move(triangles));
}

// ---- flatten -----------------------------------------------------------------------------------------------
class Flattener {
public:
    FlatScene out;
    std::map<const Material*, uint32_t> mat_ids;
    std::map<const Texture*, uint32_t> tex_ids;
    std::map<const Hittable*, uint32_t> prim_ids;
    std::vector<rt_xform_op> chain_stack;
    std::map<std::vector<std::pair<uint32_t, std::vector<double>>>, uint32_t> chain_ids;
    uint32_t level = 0;
    bool in_medium = false;
    uint32_t slice_depth = 0;   // HittableSlice nesting of the object being flattened
    size_t merge_barrier = 0;   // runs below this index are closed: a new primitive must not extend them

    Flattener() {
        rt_chain c0 = {0, 0};
        out.chains.push_back(c0);
        chain_ids[{}] = 0;
    }
    uint32_t prim_id(const Hittable* h) {  // first visit wins, in depth-first order
        auto it = prim_ids.find(h);
        if (it != prim_ids.end()) return it->second;
        uint32_t id = out.n_prim_ids++;
        prim_ids[h] = id;
        return id;
    }
    uint32_t texture(const Texture* t) {
        auto it = tex_ids.find(t);
        if (it != tex_ids.end()) return it->second;
        rt_texture r;
        std::memset(&r, 0, sizeof r);
        r.kind = t->kind();
        switch (t->kind()) {
            case RT_TEX_SOLID: {
                auto* s = static_cast<const SolidColor*>(t);
                r.color[0] = s->albedo.x; r.color[1] = s->albedo.y; r.color[2] = s->albedo.z;
                break;
            }
            case RT_TEX_CHECKER: {
                auto* c = static_cast<const CheckerTexture*>(t);
                r.a = texture(c->even.get());  // children first: a parent's index is above its children's
                r.b = texture(c->odd.get());
                r.scale = c->inverse_scale;
                break;
            }
            case RT_TEX_IMAGE: {
                auto* im = static_cast<const ImageTexture*>(t);
                FlatScene::Image fi;
                fi.rgb = im->rgb; fi.w = im->width; fi.h = im->height;
                r.a = (uint32_t)out.images.size();
                out.images.push_back(std::move(fi));
                break;
            }
            case RT_TEX_PERLIN: {
                auto* p = static_cast<const PerlinNoiseTexture*>(t);
                FlatScene::Perlin fp;
                for (int i = 0; i < 256; ++i)
                    for (int k = 0; k < 3; ++k) fp.vec.push_back(p->perlin_noise.random_vectors[i][k]);
                fp.perm.insert(fp.perm.end(), p->perlin_noise.x_perm, p->perlin_noise.x_perm + 256);
                fp.perm.insert(fp.perm.end(), p->perlin_noise.y_perm, p->perlin_noise.y_perm + 256);
                fp.perm.insert(fp.perm.end(), p->perlin_noise.z_perm, p->perlin_noise.z_perm + 256);
                r.a = (uint32_t)out.perlins.size();
                r.scale = p->scale;
                out.perlins.push_back(std::move(fp));
                break;
            }
        }
        uint32_t id = (uint32_t)out.textures.size();
        out.textures.push_back(r);
        tex_ids[t] = id;
        return id;
    }
    uint32_t material(const Material* m) {
        auto it = mat_ids.find(m);
        if (it != mat_ids.end()) return it->second;
        rt_material r;
        std::memset(&r, 0, sizeof r);
        r.kind = m->kind();
        switch (m->kind()) {
            case RT_MAT_LAMBERTIAN: r.texture = texture(static_cast<const Lambertian*>(m)->texture.get()); break;
            case RT_MAT_METAL: {
                auto* mt = static_cast<const Metal*>(m);
                r.albedo[0] = mt->albedo.x; r.albedo[1] = mt->albedo.y; r.albedo[2] = mt->albedo.z;
                r.param = mt->fuzz;
                break;
            }
            case RT_MAT_DIELECTRIC: r.param = static_cast<const Dielectric*>(m)->refraction_index; break;
            case RT_MAT_DIFFUSE_LIGHT: r.texture = texture(static_cast<const DiffuseLight*>(m)->texture.get()); break;
            case RT_MAT_ISOTROPIC: r.texture = texture(static_cast<const Isotropic*>(m)->texture.get()); break;
        }
        uint32_t id = (uint32_t)out.materials.size();
        out.materials.push_back(r);
        mat_ids[m] = id;
        return id;
    }
    uint32_t chain() {
        std::vector<std::pair<uint32_t, std::vector<double>>> key;
        for (auto& op : chain_stack) key.push_back({op.kind, {op.v[0], op.v[1], op.v[2]}});
        auto it = chain_ids.find(key);
        if (it != chain_ids.end()) return it->second;
        if (out.chains.size() >= 65535) throw std::runtime_error("flatten: too many distinct instance chains");
        rt_chain c;
        c.first_op = (uint32_t)out.ops.size();
        c.n_ops = (uint32_t)chain_stack.size();
        out.ops.insert(out.ops.end(), chain_stack.begin(), chain_stack.end());
        uint32_t id = (uint32_t)out.chains.size();
        out.chains.push_back(c);
        chain_ids[key] = id;
        return id;
    }
    void emit_run(rt_node_kind kind, uint32_t index) {
        uint32_t ch = chain();
        if (out.nodes.size() > merge_barrier) {
            rt_node& last = out.nodes.back();
            if (last.kind == kind && last.level == level && last.chain == ch && last.first + last.count == index) {
                last.count++;
                return;
            }
        }
        rt_node n;
        n.kind = (uint8_t)kind;
        n.level = (uint8_t)level;
        n.chain = (uint16_t)ch;
        n.first = index;
        n.count = 1;
        n.skip = (uint32_t)out.nodes.size() + 1u;
        out.nodes.push_back(n);
    }
    size_t open(rt_node_kind kind, uint32_t first) {
        if (level + 2u > RT_MAX_LEVELS) throw std::runtime_error("flatten: BVH/medium nesting deeper than RT_MAX_LEVELS");
        rt_node n;
        n.kind = (uint8_t)kind;
        n.level = (uint8_t)level;
        n.chain = (uint16_t)chain();
        n.first = first;
        n.count = 0;
        n.skip = 0;
        out.nodes.push_back(n);
        level++;
        return out.nodes.size() - 1;
    }
    void close(size_t at) {
        level--;
        out.nodes[at].skip = (uint32_t)out.nodes.size();
    }
    // A CULL entry brackets the entries up to close_cull() without opening an interval level.
    size_t open_cull(uint32_t box) {
        rt_node n;
        n.kind = (uint8_t)RT_NODE_CULL;
        n.level = (uint8_t)level;
        n.chain = (uint16_t)chain();
        n.first = box;
        n.count = 0;
        n.skip = 0;
        out.nodes.push_back(n);
        return out.nodes.size() - 1;
    }
    void close_cull(size_t at) {
        out.nodes[at].skip = (uint32_t)out.nodes.size();
        merge_barrier = out.nodes.size();
    }
    // Appends a box and its cull box; returns the box index.
    uint32_t add_box(const double ref[6], const double cull[6]) {
        uint32_t k = (uint32_t)(out.boxes.size() / 6);
        out.boxes.insert(out.boxes.end(), ref, ref + 6);
        out.cull_boxes.insert(out.cull_boxes.end(), cull, cull + 6);
        return k;
    }
};

// ---- extents and cull boxes ----------------------------------------------------------------------------------
static void extent_empty(double lo[3], double hi[3]) {
    for (int c = 0; c < 3; ++c) { lo[c] = INF; hi[c] = -INF; }
}
static void extent_add(double lo[3], double hi[3], Point3 p) {
    const double v[3] = {p.x, p.y, p.z};
    for (int c = 0; c < 3; ++c) { lo[c] = std::fmin(lo[c], v[c]); hi[c] = std::fmax(hi[c], v[c]); }
}
static void extent_everything(double lo[3], double hi[3]) {
    for (int c = 0; c < 3; ++c) { lo[c] = -INF; hi[c] = INF; }
}
// The extent of `h` widened by a margin that dwarfs the rounding of the intersection routines and of the box test
// itself (their errors are of the order 1e-16 relative to the coordinates involved), as x0 x1 y0 y1 z0 z1.
// Returns false when the extent is unbounded or empty: such a box never culls.
static bool cull_box_of(const Hittable& h, double out[6]) {
    double lo[3], hi[3];
    extent_empty(lo, hi);
    h.extent(lo, hi);
    bool bounded = true;
    double big = 0.0;
    for (int c = 0; c < 3; ++c) {
        if (!(lo[c] <= hi[c]) || std::isinf(lo[c]) || std::isinf(hi[c])) bounded = false;
        big = std::fmax(big, std::fmax(std::fabs(lo[c]), std::fabs(hi[c])));
    }
    if (!bounded) {
        for (int c = 0; c < 3; ++c) { out[2 * c] = -INF; out[2 * c + 1] = INF; }
        return false;
    }
    const double margin = 1e-5 * (1.0 + big);
    for (int c = 0; c < 3; ++c) { out[2 * c] = lo[c] - margin; out[2 * c + 1] = hi[c] + margin; }
    return true;
}
void Sphere::extent(double lo[3], double hi[3]) const {
    Point3 rv(radius, radius, radius);
    extent_add(lo, hi, center - rv);
    extent_add(lo, hi, center + rv);
}
void Parallelogram::extent(double lo[3], double hi[3]) const {
    extent_add(lo, hi, q); extent_add(lo, hi, q + u); extent_add(lo, hi, q + v); extent_add(lo, hi, q + u + v);
}
void Triangle::extent(double lo[3], double hi[3]) const {
    extent_add(lo, hi, q); extent_add(lo, hi, q + u); extent_add(lo, hi, q + v);
}
void Quadric::extent(double lo[3], double hi[3]) const { extent_everything(lo, hi); }  // the surfaces are unbounded
void HittableSlice::extent(double lo[3], double hi[3]) const {
    for (auto& o : objects) o->extent(lo, hi);
}
void BVHNode::extent(double lo[3], double hi[3]) const {
    if (leaf) { objects->extent(lo, hi); return; }
    left->extent(lo, hi);
    right->extent(lo, hi);
}
void Translate::extent(double lo[3], double hi[3]) const {
    double a[3], b[3];
    extent_empty(a, b);
    object->extent(a, b);
    if (!(a[0] <= b[0])) return;  // empty
    extent_add(lo, hi, Point3(a[0] + offset.x, a[1] + offset.y, a[2] + offset.z));
    extent_add(lo, hi, Point3(b[0] + offset.x, b[1] + offset.y, b[2] + offset.z));
}
void RotateY::extent(double lo[3], double hi[3]) const {
    double a[3], b[3];
    extent_empty(a, b);
    object->extent(a, b);
    if (!(a[0] <= b[0])) return;  // empty
    if (std::isinf(a[0]) || std::isinf(b[0]) || std::isinf(a[2]) || std::isinf(b[2])) {
        extent_add(lo, hi, Point3(-INF, a[1], -INF));
        extent_add(lo, hi, Point3(INF, b[1], INF));
        return;
    }
    // object space -> parent space is the rotation hit points get (hittable/mod.rs:121-137): all 8 corners
    for (int i = 0; i < 2; ++i)
        for (int k = 0; k < 2; ++k) {
            double x = i ? b[0] : a[0], z = k ? b[2] : a[2];
            double px = cos_theta * x + sin_theta * z, pz = -sin_theta * x + cos_theta * z;
            extent_add(lo, hi, Point3(px, a[1], pz));
            extent_add(lo, hi, Point3(px, b[1], pz));
        }
}
// A medium decides from the whole line, not from the ray's interval, whether it draws a random number
// (constant_medium.rs:55-63), so nothing that contains one may be skipped on an interval test.
void ConstantMedium::extent(double lo[3], double hi[3]) const { extent_everything(lo, hi); }

void Sphere::flatten(Flattener& f) const {
    uint32_t k = (uint32_t)f.out.sphere_mat.size();
    double rec[4] = {center.x, center.y, center.z, radius};
    f.out.spheres.insert(f.out.spheres.end(), rec, rec + 4);
    f.out.sphere_mat.push_back(f.material(material.get()));
    f.out.sphere_id.push_back(f.prim_id(this));
    f.emit_run(RT_NODE_SPHERES, k);
}
static void planar_record(std::vector<double>& dst, const Vector3& n, double d, const Vector3& q, const Vector3& u,
                          const Vector3& v, const Vector3& w) {
    double rec[16] = {n.x, n.y, n.z, d, q.x, q.y, q.z, u.x, u.y, u.z, v.x, v.y, v.z, w.x, w.y, w.z};
    dst.insert(dst.end(), rec, rec + 16);
}
void Parallelogram::flatten(Flattener& f) const {
    uint32_t k = (uint32_t)f.out.quad_mat.size();
    planar_record(f.out.quads, normal, d, q, u, v, w);
    f.out.quad_mat.push_back(f.material(material.get()));
    f.out.quad_id.push_back(f.prim_id(this));
    f.emit_run(RT_NODE_QUADS, k);
}
void Triangle::flatten(Flattener& f) const {
    uint32_t k = (uint32_t)f.out.tri_mat.size();
    planar_record(f.out.triangles, normal, d, q, u, v, w);
    f.out.tri_mat.push_back(f.material(material.get()));
    f.out.tri_id.push_back(f.prim_id(this));
    f.emit_run(RT_NODE_TRIANGLES, k);
}
void Quadric::flatten(Flattener& f) const {
    uint32_t k = (uint32_t)f.out.quadric_mat.size();
    double rec[10] = {p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, p3.x, p3.y, p3.z, j};
    f.out.quadrics.insert(f.out.quadrics.end(), rec, rec + 10);
    f.out.quadric_mat.push_back(f.material(material.get()));
    f.out.quadric_id.push_back(f.prim_id(this));
    f.emit_run(RT_NODE_QUADRICS, k);
}
void HittableSlice::flatten_items(Flattener& f) const {
    f.slice_depth++;
    for (auto& o : objects) o->flatten(f);  // list order = visit order (hittable_list.rs:93-99)
    f.slice_depth--;
}
void HittableSlice::flatten(Flattener& f) const {
    // A list inside something else (a box of six sides, a group under an instance) gets a cull box of its own:
    // the reference tests every child of a list against every ray that reaches the list.
    double cull[6];
    if (f.slice_depth > 0 && objects.size() >= 3 && cull_box_of(*this, cull)) {
        size_t at = f.open_cull(f.add_box(cull, cull));
        flatten_items(f);
        f.close_cull(at);
    } else {
        flatten_items(f);
    }
}
void BVHNode::flatten(Flattener& f) const {
    const AABB& b = bbox_;
    double rec[6] = {b.x.start, b.x.end, b.y.start, b.y.end, b.z.start, b.z.end};
    double cull[6];
    cull_box_of(*this, cull);
    size_t at = f.open(RT_NODE_BVH, f.add_box(rec, cull));
    if (leaf) {
        objects->flatten_items(f);  // the leaf's own cull box already bounds them
    } else {
        left->flatten(f);  // bvh.rs:50-57: left first, then right
        right->flatten(f);
    }
    f.close(at);
}
void Translate::flatten(Flattener& f) const {
    rt_xform_op op;
    std::memset(&op, 0, sizeof op);
    op.kind = RT_XF_TRANSLATE;
    op.v[0] = offset.x; op.v[1] = offset.y; op.v[2] = offset.z;
    f.chain_stack.push_back(op);
    object->flatten(f);
    f.chain_stack.pop_back();
}
void RotateY::flatten(Flattener& f) const {
    rt_xform_op op;
    std::memset(&op, 0, sizeof op);
    op.kind = RT_XF_ROTATE_Y;
    op.v[0] = sin_theta; op.v[1] = cos_theta;
    f.chain_stack.push_back(op);
    object->flatten(f);
    f.chain_stack.pop_back();
}
void ConstantMedium::flatten(Flattener& f) const {
    if (f.in_medium) throw std::runtime_error("flatten: a ConstantMedium inside a medium's boundary is not supported");
    rt_medium m;
    m.neg_inv_density = neg_inv_density;
    m.material = f.material(phase_function.get());
    m.prim_id = f.prim_id(this);
    uint32_t k = (uint32_t)f.out.media.size();
    f.out.media.push_back(m);
    size_t at = f.open(RT_NODE_MEDIUM, k);
    f.in_medium = true;
    boundary->flatten(f);
    f.in_medium = false;
    f.close(at);
}

FlatScene flatten(const Hittable& world) {
    Flattener f;
    world.flatten(f);
    return std::move(f.out);
}

rt_scene* FlatScene::commit(int device) const {
    rt_scene* s = rt_scene_create();
    if (!s) return nullptr;
    bool ok = rt_scene_set_materials(s, materials.data(), (uint32_t)materials.size()) == RT_OK &&
              rt_scene_set_textures(s, textures.data(), (uint32_t)textures.size()) == RT_OK;
    for (size_t i = 0; ok && i < images.size(); ++i)
        ok = rt_scene_set_image(s, (uint32_t)i, images[i].rgb.data(), images[i].w, images[i].h) == RT_OK;
    for (size_t i = 0; ok && i < perlins.size(); ++i)
        ok = rt_scene_set_perlin(s, (uint32_t)i, perlins[i].vec.data(), perlins[i].perm.data()) == RT_OK;
    ok = ok && rt_scene_set_spheres(s, spheres.data(), sphere_mat.data(), sphere_id.data(), (uint32_t)sphere_mat.size()) == RT_OK;
    ok = ok && rt_scene_set_quads(s, quads.data(), quad_mat.data(), quad_id.data(), (uint32_t)quad_mat.size()) == RT_OK;
    ok = ok && rt_scene_set_triangles(s, triangles.data(), tri_mat.data(), tri_id.data(), (uint32_t)tri_mat.size()) == RT_OK;
    ok = ok && rt_scene_set_quadrics(s, quadrics.data(), quadric_mat.data(), quadric_id.data(), (uint32_t)quadric_mat.size()) == RT_OK;
    ok = ok && rt_scene_set_media(s, media.data(), (uint32_t)media.size()) == RT_OK;
    ok = ok && rt_scene_set_xforms(s, ops.data(), (uint32_t)ops.size(), chains.data(), (uint32_t)chains.size()) == RT_OK;
    ok = ok && rt_scene_set_nodes(s, nodes.data(), (uint32_t)nodes.size(), boxes.data(), (uint32_t)(boxes.size() / 6)) == RT_OK;
    ok = ok && rt_scene_set_cull_boxes(s, cull_boxes.data(), (uint32_t)(cull_boxes.size() / 6)) == RT_OK;
    ok = ok && rt_scene_commit(s, device) == RT_OK;
    if (!ok) {
        rt_scene_destroy(s);
        return nullptr;
    }
    return s;
}

// ---- camera ------------------------------------------------------------------------------------------------
static void st3(double* o, Point3 p) { o[0] = p.x; o[1] = p.y; o[2] = p.z; }
Camera Camera::create(double aspect_ratio, uint32_t image_width, ImageQuality q, double vfov, ThinLens lens,
                      CameraPosition pos, Color background) {  // camera.rs:80-143
    Camera c;
    std::memset(&c.cam, 0, sizeof c.cam);
    uint32_t hh = (uint32_t)((double)image_width / aspect_ratio);
    uint32_t image_height = hh < 1 ? 1 : hh;
    double theta = vfov * (PI / 180.0);
    double h = std::tan(theta * 0.5);
    double viewport_height = 2.0 * h * lens.focus_distance;
    double viewport_width = viewport_height * ((double)image_width / (double)image_height);
    Vector3 w = unit_vector(pos.look_from - pos.look_at);
    Vector3 u = unit_vector(cross(pos.view_up, w));
    Vector3 v = unit_vector(cross(w, u));
    Vector3 viewport_u = u * viewport_width;
    Vector3 viewport_v = v * (-viewport_height);
    Vector3 du = viewport_u / (double)image_width;
    Vector3 dv = viewport_v / (double)image_height;
    Point3 upper_left = pos.look_from - w * lens.focus_distance - viewport_u * 0.5 - viewport_v * 0.5;
    Point3 pixel00 = upper_left + (du + dv) * 0.5;
    double defocus_radius = lens.focus_distance * std::tan((lens.defocus_angle * 0.5) * (PI / 180.0));
    c.cam.width = image_width;
    c.cam.height = image_height;
    st3(c.cam.pixel00, pixel00);
    st3(c.cam.delta_u, du);
    st3(c.cam.delta_v, dv);
    st3(c.cam.center, pos.look_from);
    c.cam.defocus_angle = lens.defocus_angle;
    st3(c.cam.disk_u, u * defocus_radius);
    st3(c.cam.disk_v, v * defocus_radius);
    c.samples_per_pixel = q.samples_per_pixel;
    c.max_depth = q.max_depth;
    c.background_color = background;
    return c;
}

int Camera::render(const Hittable& world, rt_stats* stats) const {
    FlatScene flat;
    try {
        flat = flatten(world);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "flatten failed: %s\n", e.what());
        return RT_ERR_INVALID;
    }
    rt_scene* s = flat.commit(device);
    if (!s) {
        std::fprintf(stderr, "scene commit failed: %s\n", rt_last_error());
        return RT_ERR_DEVICE;
    }
    rt_render_opts o;
    std::memset(&o, 0, sizeof o);
    o.spp_begin = 0;
    o.spp_count = samples_per_pixel;
    o.max_depth = max_depth;
    o.seed = seed;
    o.background[0] = background_color.x; o.background[1] = background_color.y; o.background[2] = background_color.z;
    size_t npix = (size_t)cam.width * cam.height;
    std::vector<double> film(3 * npix);
    std::vector<uint8_t> rgb(3 * npix);
    rt_stats local;
    rt_stats* stats_or_local = stats ? stats : &local;
    int rc = rt_render(s, &cam, &o, film.data(), stats_or_local);
    if (rc == RT_OK) rc = rt_tonemap_rgb8(s, film.data(), cam.width, cam.height, samples_per_pixel, rgb.data());
    if (rc != RT_OK) std::fprintf(stderr, "render failed: %s\n", rt_last_error());
    rt_scene_destroy(s);
    if (rc == RT_OK && metrics && stats_or_local) {
        // one JSON line where the reference prints its banners and progress bar (camera.rs:252,256,269)
        const rt_stats& t = *stats_or_local;
        std::fprintf(metrics, "{\"event\": \"render_done\", \"width\": %u, \"height\": %u, \"spp\": %u, \"max_depth\": %u, "
                              "\"paths\": %llu, \"rays\": %llu, \"ms\": %.3f, \"mrays_per_s\": %.3f, \"output\": \"%s\"}\n",
                     cam.width, cam.height, samples_per_pixel, max_depth, (unsigned long long)t.paths, (unsigned long long)t.rays,
                     t.ms_total, t.ms_total > 0 ? (double)t.rays / t.ms_total * 1e-3 : 0.0, output_path.c_str());
        std::fflush(metrics);
    }
    if (rc == RT_OK && !write_png(output_path, rgb.data(), cam.width, cam.height)) {
        std::fprintf(stderr, "could not write %s\n", output_path.c_str());
        return RT_ERR_INVALID;
    }
    return rc;
}

// ---- PNG ---------------------------------------------------------------------------------------------------
static void be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
static void chunk(std::vector<uint8_t>& out, const char* tag, const std::vector<uint8_t>& data) {
    be32(out, (uint32_t)data.size());
    std::vector<uint8_t> body(tag, tag + 4);
    body.insert(body.end(), data.begin(), data.end());
    out.insert(out.end(), body.begin(), body.end());
    be32(out, (uint32_t)crc32(0L, body.data(), (uInt)body.size()));
}
bool write_png(const std::string& path, const uint8_t* rgb, uint32_t w, uint32_t h) {
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (3 * w + 1));
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back(0);  // filter: none
        raw.insert(raw.end(), rgb + (size_t)3 * w * y, rgb + (size_t)3 * w * (y + 1));
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    comp.resize(clen);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    be32(ihdr, w);
    be32(ihdr, h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", comp);
    chunk(out, "IEND", {});
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) return false;
    bool ok = std::fwrite(out.data(), 1, out.size(), fp) == out.size();
    ok = (std::fclose(fp) == 0) && ok;
    return ok;
}

}  // namespace rtamd
