// scene_builder.cpp -- host-side scene construction (include/gfxexp_host.h).
//
// Mirrors the asset path of the reference host program without assimp / textures:
//   OBJ + MTL reader          createTriangleMeshes, common/common_host.cpp:2178-2429
//   immediate material values createImmTexture + sRGB sampler, common_host.cpp:1045-1073, 1602-1659
//   rectangle light           createRectangleLight, common_host.cpp:2431-2476
//   instances                 createInstance, common_host.cpp:2582-2656
// plus a procedural "street" scene standing in for Bistro Exterior (not present offline).
#include <new>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cctype>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <random>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>
#include "../../../include/gfxexp_host.h"

namespace {

thread_local std::string g_hostError;

struct Geom { std::vector<gfx_vertex> v; std::vector<uint32_t> t; uint32_t mat; };
struct Inst { uint32_t group; float xfm[12]; };
struct Tex { uint32_t width = 0, height = 0, format = 0; std::vector<uint8_t> texels; };

} // namespace

struct gfxh_scene {
    std::vector<gfx_material> materials;
    std::vector<Geom> geoms;
    std::vector<std::vector<uint32_t>> groups;
    std::vector<Inst> insts;
    std::vector<Tex> textures;                       // textures[k] is texture slot k + 1
    std::map<std::string, uint32_t> textureCache;    // file path + format -> slot (TextureCacheKey, common_host.cpp:1163-1182)
};

namespace {

struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 operator*(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 normalize(V3 a) { const float l = std::sqrt(dot(a, a)); const float r = 1 / l; return { a.x * r, a.y * r, a.z * r }; }

// makeCoordinateSystem, common/common_host.cpp:2349-2356
inline V3 tangent_from_normal(V3 n) {
    const float sign = n.z >= 0 ? 1.0f : -1.0f;
    const float a = -1 / (sign + n.z);
    const float b = n.x * n.y * a;
    return { 1 + sign * n.x * n.x * a, sign * b, -sign * n.x };
}

inline gfx_vertex make_vertex(V3 p, V3 n, V3 t, float u, float v) {
    gfx_vertex o;
    o.position[0] = p.x; o.position[1] = p.y; o.position[2] = p.z;
    o.normal[0] = n.x; o.normal[1] = n.y; o.normal[2] = n.z;
    o.texCoord0Dir[0] = t.x; o.texCoord0Dir[1] = t.y; o.texCoord0Dir[2] = t.z;
    o.texCoord[0] = u; o.texCoord[1] = v;
    return o;
}

// 8-bit immediate texture value (common_host.cpp:1045-1073) ...
// The reference converts the float straight to uint32_t -- undefined for a negative or non-finite material constant (an .mtl file is
// untrusted input); what its x86-64 build does is cvttss2si to 64 bits and keep the low word, which is spelled out here.
inline uint32_t float_to_u32_like_x86_64(float f) {
    if (!(f > -9.2e18f && f < 9.2e18f)) return 0u;                 // NaN / outside int64: the "integer indefinite" 0x8000...0, low word 0
    return static_cast<uint32_t>(static_cast<uint64_t>(static_cast<int64_t>(f)));
}
inline float quantize8(float v) { const uint32_t q = std::min(float_to_u32_like_x86_64(255 * v), 255u); return q / 255.0f; }
// ... read through an sRGB-decoding sampler (basic_types.h:5396-5402 states the formula)
inline float srgb_degamma(float v) {
    if (v <= 0.04045f) return v / 12.92f;
    return std::pow((v + 0.055f) / 1.055f, 2.4f);
}

void mat3_mul(const double a[9], const double b[9], double o[9]) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o[r * 3 + c] = a[r * 3 + 0] * b[0 * 3 + c] + a[r * 3 + 1] * b[1 * 3 + c] + a[r * 3 + 2] * b[2 * 3 + c];
}
// qFromEulerAngles = Rz(roll) * Ry(yaw) * Rx(pitch), common/basic_types.h:5120-5123
void euler_matrix(double roll, double pitch, double yaw, double o[9]) {
    const double cz = std::cos(roll), sz = std::sin(roll), cy = std::cos(yaw), sy = std::sin(yaw), cx = std::cos(pitch), sx = std::sin(pitch);
    const double Rz[9] = { cz, -sz, 0, sz, cz, 0, 0, 0, 1 };
    const double Ry[9] = { cy, 0, sy, 0, 1, 0, -sy, 0, cy };
    const double Rx[9] = { 1, 0, 0, 0, cx, -sx, 0, sx, cx };
    double t[9];
    mat3_mul(Rz, Ry, t);
    mat3_mul(t, Rx, o);
}

void xfm_point(const float m[12], const float p[3], float o[3]) {
    for (int r = 0; r < 3; ++r) o[r] = m[r * 4 + 0] * p[0] + m[r * 4 + 1] * p[1] + m[r * 4 + 2] * p[2] + m[r * 4 + 3];
}

// ---- primitive meshes for the procedural scene
void add_quad(Geom& g, V3 p0, V3 p1, V3 p2, V3 p3) { // CCW p0..p3
    const V3 n = normalize(cross(p1 - p0, p3 - p0));
    const V3 t = normalize(p1 - p0);
    const uint32_t b = static_cast<uint32_t>(g.v.size());
    g.v.push_back(make_vertex(p0, n, t, 0, 0));
    g.v.push_back(make_vertex(p1, n, t, 1, 0));
    g.v.push_back(make_vertex(p2, n, t, 1, 1));
    g.v.push_back(make_vertex(p3, n, t, 0, 1));
    const uint32_t idx[6] = { b, b + 1, b + 2, b, b + 2, b + 3 };
    g.t.insert(g.t.end(), idx, idx + 6);
}
void add_box(Geom& g, V3 lo, V3 hi) {
    const V3 c[8] = { { lo.x, lo.y, lo.z }, { hi.x, lo.y, lo.z }, { hi.x, hi.y, lo.z }, { lo.x, hi.y, lo.z },
                      { lo.x, lo.y, hi.z }, { hi.x, lo.y, hi.z }, { hi.x, hi.y, hi.z }, { lo.x, hi.y, hi.z } };
    add_quad(g, c[1], c[0], c[3], c[2]);  // -z
    add_quad(g, c[4], c[5], c[6], c[7]);  // +z
    add_quad(g, c[0], c[4], c[7], c[3]);  // -x
    add_quad(g, c[5], c[1], c[2], c[6]);  // +x
    add_quad(g, c[3], c[7], c[6], c[2]);  // +y
    add_quad(g, c[0], c[1], c[5], c[4]);  // -y
}
// grid of quads on the plane spanned by (ex, ey) from origin o, displaced along the normal by h(i,j)
template <typename H>
void add_grid(Geom& g, V3 o, V3 ex, V3 ey, uint32_t nx, uint32_t ny, H height) {
    const V3 n = normalize(cross(ex, ey));
    const V3 t = normalize(ex);
    const uint32_t b = static_cast<uint32_t>(g.v.size());
    for (uint32_t j = 0; j <= ny; ++j)
        for (uint32_t i = 0; i <= nx; ++i) {
            const float u = static_cast<float>(i) / nx, v = static_cast<float>(j) / ny;
            const V3 p = o + ex * u + ey * v + n * height(i, j);
            g.v.push_back(make_vertex(p, n, t, u, v));
        }
    for (uint32_t j = 0; j < ny; ++j)
        for (uint32_t i = 0; i < nx; ++i) {
            const uint32_t a = b + j * (nx + 1) + i, c = a + 1, d = a + nx + 1, e = d + 1;
            const uint32_t idx[6] = { a, c, e, a, e, d };
            g.t.insert(g.t.end(), idx, idx + 6);
        }
}
void make_icosphere(Geom& g, uint32_t subdiv, float radius) {
    const float t = (1.0f + std::sqrt(5.0f)) / 2.0f;
    std::vector<V3> p = { { -1, t, 0 }, { 1, t, 0 }, { -1, -t, 0 }, { 1, -t, 0 }, { 0, -1, t }, { 0, 1, t },
                          { 0, -1, -t }, { 0, 1, -t }, { t, 0, -1 }, { t, 0, 1 }, { -t, 0, -1 }, { -t, 0, 1 } };
    for (V3& v : p) v = normalize(v);
    std::vector<uint32_t> f = { 0, 11, 5, 0, 5, 1, 0, 1, 7, 0, 7, 10, 0, 10, 11, 1, 5, 9, 5, 11, 4, 11, 10, 2, 10, 7, 6, 7, 1, 8,
                                3, 9, 4, 3, 4, 2, 3, 2, 6, 3, 6, 8, 3, 8, 9, 4, 9, 5, 2, 4, 11, 6, 2, 10, 8, 6, 7, 9, 8, 1 };
    for (uint32_t s = 0; s < subdiv; ++s) {
        std::map<std::pair<uint32_t, uint32_t>, uint32_t> mid;
        auto midpoint = [&](uint32_t a, uint32_t b) {
            const auto key = std::make_pair(std::min(a, b), std::max(a, b));
            auto it = mid.find(key);
            if (it != mid.end()) return it->second;
            p.push_back(normalize((p[a] + p[b]) * 0.5f));
            const uint32_t idx = static_cast<uint32_t>(p.size() - 1);
            mid[key] = idx;
            return idx;
        };
        std::vector<uint32_t> nf;
        for (size_t i = 0; i < f.size(); i += 3) {
            const uint32_t a = f[i], b = f[i + 1], c = f[i + 2];
            const uint32_t ab = midpoint(a, b), bc = midpoint(b, c), ca = midpoint(c, a);
            const uint32_t tri[12] = { a, ab, ca, b, bc, ab, c, ca, bc, ab, bc, ca };
            nf.insert(nf.end(), tri, tri + 12);
        }
        f.swap(nf);
    }
    const uint32_t b = static_cast<uint32_t>(g.v.size());
    for (const V3& n : p) {
        const V3 tg = normalize(tangent_from_normal(n));
        const float u = 0.5f + std::atan2(n.z, n.x) / (2 * 3.14159265f), v = 0.5f - std::asin(std::min(1.0f, std::max(-1.0f, n.y))) / 3.14159265f;
        g.v.push_back(make_vertex(n * radius, n, tg, u, v));
    }
    for (uint32_t idx : f) g.t.push_back(b + idx);
}

struct Rng {
    std::mt19937 gen;
    explicit Rng(uint32_t seed) : gen(seed) {}
    float uni() { return (gen() >> 8) * (1.0f / 16777216.0f); }
    float range(float a, float b) { return a + (b - a) * uni(); }
};

} // namespace

extern "C" {

const char* gfxh_last_error(void) { return g_hostError.c_str(); }
gfxh_scene* gfxh_scene_create(void) { return new gfxh_scene(); }
void gfxh_scene_destroy(gfxh_scene* s) { delete s; }

uint32_t gfxh_scene_add_material(gfxh_scene* s, const gfx_material* m) {
    s->materials.push_back(*m);
    return static_cast<uint32_t>(s->materials.size() - 1);
}

uint32_t gfxh_scene_add_material_traditional(gfxh_scene* s, const float diffuse[3], const float specular[3],
                                             float smoothness, const float emittance[3]) {
    gfx_material m;
    std::memset(&m, 0, sizeof(m));
    m.bsdfType = GFX_BSDF_DIFFUSE_AND_SPECULAR;
    for (int i = 0; i < 3; ++i) {
        m.a[i] = srgb_degamma(quantize8(diffuse[i]));
        m.b[i] = srgb_degamma(quantize8(specular[i]));
        m.emittance[i] = emittance ? emittance[i] : 0.0f;
    }
    m.smoothness = quantize8(smoothness);
    m.hasEmittance = (m.emittance[0] != 0.0f || m.emittance[1] != 0.0f || m.emittance[2] != 0.0f) ? 1u : 0u;
    return gfxh_scene_add_material(s, &m);
}

// ---------------------------------------------------------------- textures
static size_t tex_bytes_per_texel(uint32_t format) {
    switch (format) {
    case GFX_TEX_RGBA8_SRGB: case GFX_TEX_RGBA8_UNORM: return 4;
    case GFX_TEX_R8_UNORM: return 1;
    case GFX_TEX_RG8_UNORM: return 2;
    case GFX_TEX_RGBA32F: return 16;
    default: return 0;
    }
}
constexpr uint32_t kMaxTextureDim = 16384;   // TexDimInfo packs 14 bits per dimension (gfx_texture_set rejects more at upload)
uint32_t gfxh_scene_add_texture(gfxh_scene* s, uint32_t width, uint32_t height, uint32_t format, const void* texels) {
    const size_t bpp = tex_bytes_per_texel(format);
    if (!bpp || !width || !height || !texels) { g_hostError = "gfxh_scene_add_texture: bad arguments"; return 0; }
    if (width > kMaxTextureDim || height > kMaxTextureDim) { g_hostError = "gfxh_scene_add_texture: texture larger than 16384 x 16384"; return 0; }
    try {   // nothing may unwind through the C boundary (a bad_alloc from the copy)
        Tex t;
        t.width = width; t.height = height; t.format = format;
        t.texels.assign(static_cast<const uint8_t*>(texels), static_cast<const uint8_t*>(texels) + bpp * width * height);
        s->textures.push_back(std::move(t));
    }
    catch (const std::exception& e) { g_hostError = std::string("gfxh_scene_add_texture: ") + e.what(); return 0; }
    return static_cast<uint32_t>(s->textures.size());   // 1-based slot
}
uint32_t gfxh_scene_num_textures(gfxh_scene* s) { return static_cast<uint32_t>(s->textures.size()); }
int gfxh_scene_get_texture(gfxh_scene* s, uint32_t slot, uint32_t* width, uint32_t* height, uint32_t* format, const void** texels) {
    if (slot == 0 || slot > s->textures.size()) { g_hostError = "gfxh_scene_get_texture: bad slot"; return 1; }
    const Tex& t = s->textures[slot - 1];
    *width = t.width; *height = t.height; *format = t.format; *texels = t.texels.data();
    return 0;
}

namespace {
// Decoded image: 8-bit RGBA (stbi_load(..., 4) in the reference, common_host.cpp:1211-1226) or float RGBA (.pfm).
struct Image { uint32_t w = 0, h = 0; bool isFloat = false; std::vector<uint8_t> rgba8; std::vector<float> rgba32f; };

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}
// next whitespace-separated token of a Netpbm header ('#' comments skipped)
bool pnm_token(const std::vector<uint8_t>& d, size_t& at, std::string& tok) {
    tok.clear();
    while (at < d.size()) {
        if (d[at] == '#') { while (at < d.size() && d[at] != '\n') ++at; }
        else if (std::isspace(d[at])) ++at;
        else break;
    }
    while (at < d.size() && !std::isspace(d[at])) tok.push_back(static_cast<char>(d[at++]));
    return !tok.empty();
}
// ---- OpenEXR (the format the reference reads "-env-texture" from: loadEnvTexture -> tinyexr LoadEXR, common_host.cpp:2674): single-part
// scanline files, channels R G B A (or Y) stored as HALF / FLOAT / UINT, compression NONE, RLE, ZIPS, ZIP -- what OpenEXR's own tools and
// most exporters write by default besides PIZ, which this reader names and refuses.  The file layout follows the OpenEXR file-layout
// document (magic, version, attribute list, chunk offset table, chunks of 1 / 16 scanlines each stored channel by channel in
// alphabetical order); ZIP / RLE chunks are a zlib stream (RFC 1950 / 1951, inflated below) or run lengths over the chunk's bytes
// after a byte-delta predictor and an even / odd byte split.
struct BitReader {
    const uint8_t* p; size_t n, at = 0; uint32_t acc = 0; int have = 0; bool bad = false;
    uint32_t bits(int k) {
        while (have < k) { if (at >= n) { bad = true; return 0; } acc |= static_cast<uint32_t>(p[at++]) << have; have += 8; }
        const uint32_t v = acc & ((k == 32) ? 0xFFFFFFFFu : ((1u << k) - 1u));
        acc = k >= 32 ? 0 : acc >> k; have -= k;
        return v;
    }
};
struct Huffman { uint16_t count[16]; uint16_t symbol[288]; };
void build_huffman(Huffman& h, const uint8_t* lengths, int n) {
    std::memset(h.count, 0, sizeof(h.count));
    for (int i = 0; i < n; ++i) ++h.count[lengths[i]];
    h.count[0] = 0;
    uint16_t offs[16]; offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = static_cast<uint16_t>(offs[l] + h.count[l]);
    for (int i = 0; i < n; ++i) if (lengths[i]) h.symbol[offs[lengths[i]]++] = static_cast<uint16_t>(i);
}
int decode_symbol(BitReader& br, const Huffman& h) {       // canonical code, one bit at a time (RFC 1951 3.2.2)
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; ++len) {
        code |= static_cast<int>(br.bits(1));
        if (br.bad) return -1;
        const int count = h.count[len];
        if (code - count < first) return h.symbol[index + (code - first)];
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return -1;
}
// zlib stream -> exactly `want` bytes; false on any malformed input
bool inflate_zlib(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t want) {
    static const uint16_t lenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
    static const uint16_t lenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
    static const uint16_t distBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
    static const uint16_t distExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
    if (n < 2 || (src[0] & 0x0F) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) return false;
    BitReader br{ src + 2, n - 2 };
    out.clear(); out.reserve(want);
    for (bool last = false; !last;) {
        last = br.bits(1) != 0;
        const uint32_t type = br.bits(2);
        if (br.bad) return false;
        if (type == 0) {
            br.acc = 0; br.have = 0;                                   // to the next byte boundary
            if (br.at + 4 > br.n) return false;
            const uint32_t len = br.p[br.at] | (br.p[br.at + 1] << 8), nlen = br.p[br.at + 2] | (br.p[br.at + 3] << 8);
            br.at += 4;
            if ((len ^ 0xFFFFu) != nlen || br.at + len > br.n || out.size() + len > want) return false;
            out.insert(out.end(), br.p + br.at, br.p + br.at + len);
            br.at += len;
            continue;
        }
        if (type == 3) return false;
        Huffman lit, dist;
        uint8_t lengths[320];
        if (type == 1) {
            for (int i = 0; i < 144; ++i) lengths[i] = 8;
            for (int i = 144; i < 256; ++i) lengths[i] = 9;
            for (int i = 256; i < 280; ++i) lengths[i] = 7;
            for (int i = 280; i < 288; ++i) lengths[i] = 8;
            build_huffman(lit, lengths, 288);
            for (int i = 0; i < 30; ++i) lengths[i] = 5;
            build_huffman(dist, lengths, 30);
        }
        else {
            static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
            const int nlen = static_cast<int>(br.bits(5)) + 257, ndist = static_cast<int>(br.bits(5)) + 1, ncode = static_cast<int>(br.bits(4)) + 4;
            if (br.bad || nlen > 286 || ndist > 30) return false;
            uint8_t cl[19] = { 0 };
            for (int i = 0; i < ncode; ++i) cl[order[i]] = static_cast<uint8_t>(br.bits(3));
            Huffman lc;
            build_huffman(lc, cl, 19);
            int i = 0;
            while (i < nlen + ndist) {
                const int sym = decode_symbol(br, lc);
                if (sym < 0) return false;
                if (sym < 16) { lengths[i++] = static_cast<uint8_t>(sym); continue; }
                int rep, val = 0;
                if (sym == 16) { if (i == 0) return false; val = lengths[i - 1]; rep = 3 + static_cast<int>(br.bits(2)); }
                else if (sym == 17) rep = 3 + static_cast<int>(br.bits(3));
                else rep = 11 + static_cast<int>(br.bits(7));
                if (br.bad || i + rep > nlen + ndist) return false;
                while (rep--) lengths[i++] = static_cast<uint8_t>(val);
            }
            if (lengths[256] == 0) return false;
            build_huffman(lit, lengths, nlen);
            build_huffman(dist, lengths + nlen, ndist);
        }
        for (;;) {
            const int sym = decode_symbol(br, lit);
            if (sym < 0) return false;
            if (sym < 256) { if (out.size() >= want) return false; out.push_back(static_cast<uint8_t>(sym)); continue; }
            if (sym == 256) break;
            if (sym > 285) return false;
            const uint32_t len = lenBase[sym - 257] + br.bits(lenExtra[sym - 257]);
            const int ds = decode_symbol(br, dist);
            if (ds < 0 || ds > 29) return false;
            const uint32_t d = distBase[ds] + br.bits(distExtra[ds]);
            if (br.bad || d > out.size() || out.size() + len > want) return false;
            for (uint32_t k = 0; k < len; ++k) out.push_back(out[out.size() - d]);
        }
    }
    return out.size() == want;
}
inline float half_to_float(uint16_t h) {
    const uint32_t sign = static_cast<uint32_t>(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else { int shift = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; ++shift; } bits = sign | ((113u - shift) << 23) | ((mm & 0x3FFu) << 13); }
    }
    else if (e == 31) bits = sign | 0x7F800000u | (m << 13);
    else bits = sign | ((e + 112u) << 23) | (m << 13);
    float f; std::memcpy(&f, &bits, 4);
    return f;
}
bool decode_exr(const std::vector<uint8_t>& d, const std::string& path, Image& img, std::string& err) {
    auto fail = [&](const std::string& what) { err = "EXR: " + what + ": " + path; return false; };
    size_t at = 8;
    const uint32_t version = d[4] | (d[5] << 8) | (d[6] << 16) | (static_cast<uint32_t>(d[7]) << 24);
    if ((version & 0xFFu) != 2u) return fail("unknown file version");
    if (version & 0x1A00u) return fail("tiled, deep and multi-part files are not read (single-part scanline only)");
    struct Channel { std::string name; int type; };
    std::vector<Channel> channels;
    int compression = -1;
    int32_t win[4] = { 0, 0, -1, -1 };
    bool haveWindow = false;
    auto rd_i32 = [&](size_t o) { int32_t v; std::memcpy(&v, d.data() + o, 4); return v; };
    for (;;) {
        if (at >= d.size()) return fail("truncated header");
        if (d[at] == 0) { ++at; break; }
        std::string name, type;
        while (at < d.size() && d[at]) name.push_back(static_cast<char>(d[at++]));
        ++at;
        while (at < d.size() && d[at]) type.push_back(static_cast<char>(d[at++]));
        ++at;
        if (at + 4 > d.size()) return fail("truncated header");
        const int32_t size = rd_i32(at); at += 4;
        if (size < 0 || at + static_cast<size_t>(size) > d.size()) return fail("truncated header");
        if (name == "channels") {
            size_t c = at;
            const size_t end = at + size;
            while (c < end && d[c]) {
                Channel ch;
                while (c < end && d[c]) ch.name.push_back(static_cast<char>(d[c++]));
                ++c;
                if (c + 16 > end) return fail("truncated channel list");
                ch.type = rd_i32(c);
                if (rd_i32(c + 8) != 1 || rd_i32(c + 12) != 1) return fail("subsampled channels are not read");
                if (ch.type < 0 || ch.type > 2) return fail("unknown pixel type");
                c += 16;
                channels.push_back(ch);
            }
        }
        else if (name == "compression" && size == 1) compression = d[at];
        else if (name == "dataWindow" && size == 16) { for (int k = 0; k < 4; ++k) win[k] = rd_i32(at + 4 * k); haveWindow = true; }
        at += size;
    }
    if (channels.empty() || !haveWindow || compression < 0) return fail("header without channels / dataWindow / compression");
    static const char* names[] = { "NONE", "RLE", "ZIPS", "ZIP", "PIZ", "PXR24", "B44", "B44A", "DWAA", "DWAB" };
    if (compression > 3) return fail(std::string("compression ") + (compression < 10 ? names[compression] : "?") + " is not read (NONE, RLE, ZIPS, ZIP are; re-save the file)");
    const int64_t w64 = static_cast<int64_t>(win[2]) - win[0] + 1, h64 = static_cast<int64_t>(win[3]) - win[1] + 1;
    if (w64 <= 0 || h64 <= 0 || w64 > kMaxTextureDim || h64 > kMaxTextureDim) return fail("image larger than 16384 x 16384 or empty");
    const uint32_t w = static_cast<uint32_t>(w64), h = static_cast<uint32_t>(h64);
    const uint32_t linesPerChunk = compression == 3 ? 16u : 1u;
    const uint32_t numChunks = (h + linesPerChunk - 1) / linesPerChunk;
    if (at + 8ull * numChunks > d.size()) return fail("truncated offset table");
    size_t lineBytes = 0;
    for (const Channel& c : channels) lineBytes += (c.type == 1 ? 2ull : 4ull) * w;
    // which file channel feeds which of R G B A (a lone Y feeds R, G and B)
    int src[4] = { -1, -1, -1, -1 };
    for (size_t c = 0; c < channels.size(); ++c) {
        const std::string& nm = channels[c].name;
        if (nm == "R") src[0] = static_cast<int>(c); else if (nm == "G") src[1] = static_cast<int>(c);
        else if (nm == "B") src[2] = static_cast<int>(c); else if (nm == "A") src[3] = static_cast<int>(c);
    }
    if (src[0] < 0 && src[1] < 0 && src[2] < 0)
        for (size_t c = 0; c < channels.size(); ++c) if (channels[c].name == "Y") src[0] = src[1] = src[2] = static_cast<int>(c);
    if (src[0] < 0 && src[1] < 0 && src[2] < 0) return fail("no R, G, B or Y channel");
    // The offset table and the chunk headers are checked BEFORE the 16 w h bytes of the image are asked for: a compressed file has no
    // size bound of its own, so a crafted header must not be able to make a tiny file allocate gigabytes (and an allocation that still
    // fails is an error return, not an exception through the extern "C" loader).
    for (uint32_t k = 0; k < numChunks; ++k) {
        uint64_t off; std::memcpy(&off, d.data() + at + 8ull * k, 8);
        if (off > d.size() || d.size() - off < 8) return fail("chunk offset outside the file");
        const int32_t y0 = rd_i32(off), size = rd_i32(off + 4);
        const int64_t row0 = static_cast<int64_t>(y0) - win[1];
        if (size < 0 || static_cast<uint64_t>(size) > d.size() - off - 8 || row0 < 0 || row0 >= h) return fail("malformed chunk");
        if (row0 % linesPerChunk != 0) return fail("chunk that does not start on a multiple of its line count (it would overlap its neighbours)");
    }
    // every scan line costs the file at least a byte or two (ZIP / RLE shrink a constant line by ~1000 : 1 at best)
    if (static_cast<uint64_t>(lineBytes) * h / 4096u > d.size()) return fail("image far larger than its file can hold");
    img.w = w; img.h = h; img.isFloat = true;
    try { img.rgba32f.assign(4ull * w * h, 0.0f); }
    catch (const std::bad_alloc&) { return fail("out of memory for the image"); }
    for (size_t i = 0; i < static_cast<size_t>(w) * h; ++i) img.rgba32f[4 * i + 3] = 1.0f;
    std::vector<uint8_t> raw, tmp;
    for (uint32_t k = 0; k < numChunks; ++k) {
        uint64_t off; std::memcpy(&off, d.data() + at + 8ull * k, 8);
        if (off > d.size() || d.size() - off < 8) return fail("chunk offset outside the file");            // (no off + 8: the field is untrusted)
        const int32_t y0 = rd_i32(off), size = rd_i32(off + 4);
        const int64_t row0 = static_cast<int64_t>(y0) - win[1];
        if (size < 0 || static_cast<uint64_t>(size) > d.size() - off - 8 || row0 < 0 || row0 >= h) return fail("malformed chunk");
        const uint32_t lines = std::min<uint32_t>(linesPerChunk, h - static_cast<uint32_t>(row0));
        const size_t want = lineBytes * lines;
        const uint8_t* body = d.data() + off + 8;
        if (compression == 0 || static_cast<size_t>(size) == want) {       // a chunk that did not shrink is stored as it is
            if (static_cast<size_t>(size) != want) return fail("chunk of the wrong size");
            raw.assign(body, body + want);
        }
        else {
            if (compression == 1) {                                           // run lengths: n < 0 -> -n literal bytes, else n + 1 copies of the next
                tmp.clear();
                size_t i = 0;
                while (i < static_cast<size_t>(size)) {
                    const int n = static_cast<int8_t>(body[i++]);
                    if (n < 0) { if (i + static_cast<size_t>(-n) > static_cast<size_t>(size)) return fail("malformed RLE chunk"); tmp.insert(tmp.end(), body + i, body + i - n); i += static_cast<size_t>(-n); }
                    else { if (i >= static_cast<size_t>(size)) return fail("malformed RLE chunk"); tmp.insert(tmp.end(), static_cast<size_t>(n) + 1, body[i++]); }
                    if (tmp.size() > want) return fail("malformed RLE chunk");
                }
                if (tmp.size() != want) return fail("malformed RLE chunk");
            }
            else if (!inflate_zlib(body, static_cast<size_t>(size), tmp, want)) return fail("malformed ZIP chunk");
            for (size_t i = 1; i < want; ++i) tmp[i] = static_cast<uint8_t>(tmp[i - 1] + tmp[i] - 128);      // byte-delta predictor
            raw.resize(want);
            const size_t half = (want + 1) / 2;
            for (size_t i = 0; i < want; ++i) raw[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];              // even bytes first, then odd
        }
        for (uint32_t l = 0; l < lines; ++l) {
            const uint8_t* line = raw.data() + lineBytes * l;
            float* out = img.rgba32f.data() + 4ull * (static_cast<size_t>(row0) + l) * w;
            size_t chOff = 0;
            for (size_t c = 0; c < channels.size(); ++c) {
                const int type = channels[c].type;
                for (int k4 = 0; k4 < 4; ++k4) {
                    if (src[k4] != static_cast<int>(c)) continue;
                    for (uint32_t x = 0; x < w; ++x) {
                        float v;
                        if (type == 1) { uint16_t hv; std::memcpy(&hv, line + chOff + 2ull * x, 2); v = half_to_float(hv); }
                        else if (type == 2) std::memcpy(&v, line + chOff + 4ull * x, 4);
                        else { uint32_t u; std::memcpy(&u, line + chOff + 4ull * x, 4); v = static_cast<float>(u); }
                        out[4ull * x + k4] = v;
                    }
                }
                chOff += (type == 1 ? 2ull : 4ull) * w;
            }
        }
    }
    return true;
}

// Uncompressed formats (no third-party decoders in this build): binary PPM / PGM (8 bit), PFM, BMP 24 / 32 bit, TGA types 2 / 3
// (24 / 32 / 8 bit) -- and OpenEXR above.  BC-compressed DDS and PNG files of the original assets are decoded offline (tools/dds_convert.py).
bool decode_image(const std::string& path, Image& img, std::string& err) {
    std::vector<uint8_t> d;
    if (!read_file(path, d)) { err = "cannot read " + path; return false; }
    if (d.size() >= 8 && d[0] == 0x76 && d[1] == 0x2f && d[2] == 0x31 && d[3] == 0x01) return decode_exr(d, path, img, err);
    if (d.size() >= 2 && d[0] == 'P' && (d[1] == '6' || d[1] == '5')) {
        size_t at = 2; std::string t;
        uint32_t vals[3];
        for (int k = 0; k < 3; ++k) { if (!pnm_token(d, at, t)) { err = "truncated PNM header"; return false; } vals[k] = static_cast<uint32_t>(std::strtoul(t.c_str(), nullptr, 10)); }
        ++at;   // the single whitespace after maxval
        const uint32_t ch = d[1] == '6' ? 3 : 1;
        // dimensions are bounded BEFORE any size arithmetic: header fields are untrusted and w * h * ch must not wrap
        if (vals[0] > kMaxTextureDim || vals[1] > kMaxTextureDim) { err = "image larger than 16384 x 16384: " + path; return false; }
        if (vals[2] != 255 || vals[0] == 0 || vals[1] == 0 || d.size() < at + static_cast<size_t>(vals[0]) * vals[1] * ch) { err = "unsupported PNM (8-bit binary only)"; return false; }
        img.w = vals[0]; img.h = vals[1]; img.rgba8.resize(4ull * img.w * img.h);
        for (size_t i = 0; i < static_cast<size_t>(img.w) * img.h; ++i) {
            const uint8_t* px = d.data() + at + i * ch;
            img.rgba8[4 * i] = px[0]; img.rgba8[4 * i + 1] = ch == 3 ? px[1] : px[0]; img.rgba8[4 * i + 2] = ch == 3 ? px[2] : px[0]; img.rgba8[4 * i + 3] = 255;
        }
        return true;
    }
    if (d.size() >= 2 && d[0] == 'P' && (d[1] == 'F' || d[1] == 'f')) {
        size_t at = 2; std::string t;
        if (!pnm_token(d, at, t)) { err = "truncated PFM header"; return false; }
        const uint32_t w = static_cast<uint32_t>(std::strtoul(t.c_str(), nullptr, 10));
        if (!pnm_token(d, at, t)) { err = "truncated PFM header"; return false; }
        const uint32_t h = static_cast<uint32_t>(std::strtoul(t.c_str(), nullptr, 10));
        if (!pnm_token(d, at, t)) { err = "truncated PFM header"; return false; }
        const double scale = std::strtod(t.c_str(), nullptr);
        ++at;
        const uint32_t ch = d[1] == 'F' ? 3 : 1;
        if (w > kMaxTextureDim || h > kMaxTextureDim) { err = "image larger than 16384 x 16384: " + path; return false; }
        if (scale == 0 || !w || !h || d.size() < at + 4ull * w * h * ch) { err = "truncated or malformed PFM: " + path; return false; }
        const bool bigEndian = scale > 0;      // the sign of the scale line is the byte order of the samples
        img.w = w; img.h = h; img.isFloat = true; img.rgba32f.resize(4ull * w * h);
        for (uint32_t y = 0; y < h; ++y)   // PFM rows run bottom to top
            for (uint32_t x = 0; x < w; ++x) {
                float px[3] = { 0, 0, 0 };
                unsigned char raw[12];
                std::memcpy(raw, d.data() + at + 4ull * ch * (static_cast<size_t>(h - 1 - y) * w + x), 4ull * ch);
                if (bigEndian)
                    for (uint32_t c = 0; c < ch; ++c) { std::swap(raw[4 * c], raw[4 * c + 3]); std::swap(raw[4 * c + 1], raw[4 * c + 2]); }
                std::memcpy(px, raw, 4ull * ch);
                float* o = img.rgba32f.data() + 4ull * (static_cast<size_t>(y) * w + x);
                o[0] = px[0]; o[1] = ch == 3 ? px[1] : px[0]; o[2] = ch == 3 ? px[2] : px[0]; o[3] = 1.0f;
            }
        return true;
    }
    if (d.size() >= 54 && d[0] == 'B' && d[1] == 'M') {
        auto u32 = [&](size_t o) { uint32_t v; std::memcpy(&v, d.data() + o, 4); return v; };
        auto i32 = [&](size_t o) { int32_t v; std::memcpy(&v, d.data() + o, 4); return v; };
        const uint32_t off = u32(10); const int32_t w = i32(18), hh = i32(22);
        uint16_t bpp; std::memcpy(&bpp, d.data() + 28, 2);
        const uint32_t comp = u32(30);
        if (w <= 0 || hh == 0 || (bpp != 24 && bpp != 32) || (comp != 0 && comp != 3)) { err = "unsupported BMP (24 / 32 bit uncompressed only)"; return false; }
        // |hh| without negating INT_MIN; both dimensions bounded before off + stride * h is formed
        const int64_t h64 = hh < 0 ? -static_cast<int64_t>(hh) : static_cast<int64_t>(hh);
        if (w > static_cast<int32_t>(kMaxTextureDim) || h64 > static_cast<int64_t>(kMaxTextureDim)) { err = "image larger than 16384 x 16384: " + path; return false; }
        const uint32_t h = static_cast<uint32_t>(h64);
        const size_t stride = (static_cast<size_t>(w) * (bpp / 8) + 3) & ~size_t(3);
        if (d.size() < static_cast<size_t>(off) + stride * h) { err = "truncated BMP"; return false; }
        img.w = static_cast<uint32_t>(w); img.h = h; img.rgba8.resize(4ull * img.w * h);
        for (uint32_t y = 0; y < h; ++y) {
            const uint8_t* row = d.data() + off + stride * (hh < 0 ? y : h - 1 - y);
            for (uint32_t x = 0; x < img.w; ++x) {
                const uint8_t* px = row + static_cast<size_t>(x) * (bpp / 8);
                uint8_t* o = img.rgba8.data() + 4ull * (static_cast<size_t>(y) * img.w + x);
                o[0] = px[2]; o[1] = px[1]; o[2] = px[0]; o[3] = bpp == 32 ? px[3] : 255;
            }
        }
        return true;
    }
    if (d.size() >= 18 && (d[2] == 2 || d[2] == 3) && d[1] == 0) {   // TGA, uncompressed true colour / grey
        const uint32_t idLen = d[0];
        uint16_t w, h; std::memcpy(&w, d.data() + 12, 2); std::memcpy(&h, d.data() + 14, 2);
        const uint32_t bpp = d[16]; const bool topDown = (d[17] & 0x20) != 0;
        const uint32_t ch = bpp / 8;
        if (!w || !h || (d[2] == 2 && ch != 3 && ch != 4) || (d[2] == 3 && ch != 1) || d.size() < 18 + idLen + static_cast<size_t>(w) * h * ch) { err = "unsupported TGA (uncompressed 8 / 24 / 32 bit only)"; return false; }
        img.w = w; img.h = h; img.rgba8.resize(4ull * w * h);
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) {
                const uint8_t* px = d.data() + 18 + idLen + (static_cast<size_t>(topDown ? y : h - 1 - y) * w + x) * ch;
                uint8_t* o = img.rgba8.data() + 4ull * (static_cast<size_t>(y) * w + x);
                if (ch == 1) { o[0] = o[1] = o[2] = px[0]; o[3] = 255; }
                else { o[0] = px[2]; o[1] = px[1]; o[2] = px[0]; o[3] = ch == 4 ? px[3] : 255; }
            }
        return true;
    }
    err = "unsupported image format (PPM / PGM / PFM / BMP / TGA uncompressed, EXR): " + path;
    return false;
}
} // namespace

// loadTexture (common_host.cpp:1163-1244): cached per path; 8-bit images become RGBA8 read through `format8`
// (GFX_TEX_RGBA8_SRGB for colour maps = needsDegamma, GFX_TEX_RGBA8_UNORM for normal maps, GFX_TEX_R8_UNORM takes
// the red channel); float images become GFX_TEX_RGBA32F (isHDR).  Returns the texture slot, 0 on failure.
uint32_t gfxh_scene_load_texture(gfxh_scene* s, const char* path, uint32_t format8) {
    const std::string key = std::string(path) + "#" + std::to_string(format8);
    auto it = s->textureCache.find(key);
    if (it != s->textureCache.end()) return it->second;
    Image img; std::string err;
    try {
        if (!decode_image(path, img, err)) { g_hostError = err; return 0; }
    }
    catch (const std::exception& e) { g_hostError = std::string("gfxh_scene_load_texture: ") + e.what(); return 0; }   // a bad_alloc from the decode buffers
    uint32_t slot = 0;
    if (img.isFloat) slot = gfxh_scene_add_texture(s, img.w, img.h, GFX_TEX_RGBA32F, img.rgba32f.data());
    else if (format8 == GFX_TEX_R8_UNORM || format8 == GFX_TEX_RG8_UNORM) {
        const uint32_t ch = format8 == GFX_TEX_R8_UNORM ? 1 : 2;
        std::vector<uint8_t> packed;
        try { packed.resize(static_cast<size_t>(img.w) * img.h * ch); }
        catch (const std::exception& e) { g_hostError = std::string("gfxh_scene_load_texture: ") + e.what(); return 0; }
        for (size_t i = 0; i < static_cast<size_t>(img.w) * img.h; ++i) for (uint32_t c = 0; c < ch; ++c) packed[i * ch + c] = img.rgba8[4 * i + c];
        slot = gfxh_scene_add_texture(s, img.w, img.h, format8, packed.data());
    }
    else slot = gfxh_scene_add_texture(s, img.w, img.h, format8 == GFX_TEX_RGBA8_UNORM ? GFX_TEX_RGBA8_UNORM : GFX_TEX_RGBA8_SRGB, img.rgba8.data());
    if (slot) s->textureCache[key] = slot;
    return slot;
}

uint32_t gfxh_scene_add_geom(gfxh_scene* s, const gfx_vertex* v, uint32_t nv, const uint32_t* tris, uint32_t nt, uint32_t matSlot) {
    Geom g;
    g.v.assign(v, v + nv);
    g.t.assign(tris, tris + 3ull * nt);
    g.mat = matSlot;
    s->geoms.push_back(std::move(g));
    return static_cast<uint32_t>(s->geoms.size() - 1);
}
uint32_t gfxh_scene_add_group(gfxh_scene* s, const uint32_t* geomSlots, uint32_t n) {
    s->groups.emplace_back(geomSlots, geomSlots + n);
    return static_cast<uint32_t>(s->groups.size() - 1);
}
uint32_t gfxh_scene_add_instance(gfxh_scene* s, uint32_t group, const float xfm[12]) {
    Inst i;
    i.group = group;
    std::memcpy(i.xfm, xfm, sizeof(float) * 12);
    s->insts.push_back(i);
    return static_cast<uint32_t>(s->insts.size() - 1);
}

static uint32_t load_obj_impl(gfxh_scene* s, const char* path, int simplePbr);
uint32_t gfxh_scene_load_obj(gfxh_scene* s, const char* path) { return gfxh_scene_load_obj_conv(s, path, GFXH_MATCONV_TRADITIONAL); }
uint32_t gfxh_scene_load_obj_conv(gfxh_scene* s, const char* path, int materialConvention) {
    try { return load_obj_impl(s, path, materialConvention == GFXH_MATCONV_SIMPLE_PBR ? 1 : 0); }   // nothing may unwind through the C boundary
    catch (const std::exception& e) { g_hostError = std::string("gfxh_scene_load_obj: ") + e.what(); return 0xFFFFFFFFu; }
}
static uint32_t load_obj_impl(gfxh_scene* s, const char* path, int simplePbr) {
    std::ifstream in(path);
    if (!in) { g_hostError = std::string("cannot open ") + path; return 0xFFFFFFFFu; }
    const std::string dir = std::string(path).substr(0, std::string(path).find_last_of("/\\") + 1);
    std::vector<V3> pos, nrm;
    std::vector<std::pair<float, float>> uv;
    struct MtlDesc {
        float kd[3] = { 0, 0, 0 }, ks[3] = { 0, 0, 0 }, ke[3] = { 0, 0, 0 }; float ns = 0;
        std::string mapKd, mapKs, mapKe, mapBump, mapNormal;   // AI_MATKEY_TEXTURE_DIFFUSE / SPECULAR / EMISSIVE / HEIGHT / NORMALS
    };
    std::map<std::string, MtlDesc> mtl;
    std::vector<std::string> matOrder;
    struct Corner { int v, t, n; };
    std::map<std::string, std::vector<Corner>> facesByMat;   // triangulated corner list per material
    std::string curMat = "";
    std::string line;
    auto parse_mtl = [&](const std::string& file) {
        std::ifstream m(dir + file);
        std::string l, cur;
        while (std::getline(m, l)) {
            std::istringstream ss(l);
            std::string k; ss >> k;
            if (k == "newmtl") { ss >> cur; mtl[cur] = MtlDesc(); }
            else if (k == "Kd") ss >> mtl[cur].kd[0] >> mtl[cur].kd[1] >> mtl[cur].kd[2];
            else if (k == "Ks") ss >> mtl[cur].ks[0] >> mtl[cur].ks[1] >> mtl[cur].ks[2];
            else if (k == "Ke") ss >> mtl[cur].ke[0] >> mtl[cur].ke[1] >> mtl[cur].ke[2];
            else if (k == "Ns") ss >> mtl[cur].ns;
            else if (k == "map_Kd" || k == "map_Ks" || k == "map_Ke" || k == "map_bump" || k == "map_Bump" || k == "bump" || k == "norm" || k == "map_Kn") {
                // last token = file name (options such as "-bm 1.0" come before it)
                std::string tok, file;
                while (ss >> tok) file = tok;
                for (char& ch : file) if (ch == '\\') ch = '/';
                MtlDesc& d = mtl[cur];
                if (k == "map_Kd") d.mapKd = file;
                else if (k == "map_Ks") d.mapKs = file;
                else if (k == "map_Ke") d.mapKe = file;
                else if (k == "norm" || k == "map_Kn") d.mapNormal = file;
                else d.mapBump = file;
            }
        }
    };
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string k; ss >> k;
        if (k == "v") { V3 p; ss >> p.x >> p.y >> p.z; pos.push_back(p); }
        else if (k == "vn") { V3 p; ss >> p.x >> p.y >> p.z; nrm.push_back(p); }
        else if (k == "vt") { float a = 0, b = 0; ss >> a >> b; uv.push_back({ a, b }); }
        else if (k == "mtllib") { std::string f; while (ss >> f) parse_mtl(f); }      // "mtllib a.mtl b.mtl": every library named
        else if (k == "usemtl") { ss >> curMat; }
        else if (k == "f") {
            std::vector<Corner> cs;
            std::string tok;
            while (ss >> tok) {
                Corner c = { 0, 0, 0 };
                int idx[3] = { 0, 0, 0 };
                int which = 0; std::string num;
                for (size_t i = 0; i <= tok.size(); ++i) {
                    if (i == tok.size() || tok[i] == '/') {
                        if (!num.empty()) {
                            char* end = nullptr;
                            const long val = std::strtol(num.c_str(), &end, 10);
                            if (*end != 0 || val < -2147483647L || val > 2147483647L) { g_hostError = std::string("bad face index '") + tok + "' in " + path; return 0xFFFFFFFFu; }
                            idx[which] = static_cast<int>(val);
                        }
                        num.clear(); ++which; if (which > 2) break;
                    }
                    else num.push_back(tok[i]);
                }
                c.v = idx[0] < 0 ? static_cast<int>(pos.size()) + idx[0] : idx[0] - 1;
                c.t = idx[1] == 0 ? -1 : (idx[1] < 0 ? static_cast<int>(uv.size()) + idx[1] : idx[1] - 1);
                c.n = idx[2] == 0 ? -1 : (idx[2] < 0 ? static_cast<int>(nrm.size()) + idx[2] : idx[2] - 1);
                if (c.v < 0 || c.v >= static_cast<int>(pos.size()) || c.t >= static_cast<int>(uv.size()) || c.n >= static_cast<int>(nrm.size()) ||
                    (idx[1] != 0 && c.t < 0) || (idx[2] != 0 && c.n < 0)) {
                    g_hostError = std::string("face index out of range '") + tok + "' in " + path; return 0xFFFFFFFFu;
                }
                cs.push_back(c);
            }
            if (!facesByMat.count(curMat)) matOrder.push_back(curMat);
            std::vector<Corner>& dst = facesByMat[curMat];
            for (size_t i = 1; i + 1 < cs.size(); ++i) { dst.push_back(cs[0]); dst.push_back(cs[i]); dst.push_back(cs[i + 1]); }
        }
    }
    std::vector<uint32_t> geomSlots;
    for (const std::string& name : matOrder) {
        const MtlDesc d = mtl.count(name) ? mtl[name] : MtlDesc();
        // smoothness = sqrt(Ns) / 11 (common_host.cpp:2271-2274); four Bistro pavement materials are pinned to 0.2 (:2286-2297)
        float smoothness = std::sqrt(d.ns) / 11.0f;
        if (name == "Pavement_Cobblestone_Big_BLENDSHADER" || name == "Pavement_Cobblestone_Small_BLENDSHADER" ||
            name == "Pavement_Brick_BLENDSHADER" || name == "Pavement_Cobblestone_Wet_BLENDSHADER") smoothness = 0.2f;
        const uint32_t matSlot = gfxh_scene_add_material_traditional(s, d.kd, d.ks, smoothness, d.ke);
        {   // texture maps (createDiffuseAndSpecularMaterial, common_host.cpp:1560-1700): a map that cannot be read
            // leaves the immediate value in place
            gfx_material& m = s->materials[matSlot];
            if (!d.mapKd.empty()) m.texA = gfxh_scene_load_texture(s, (dir + d.mapKd).c_str(), GFX_TEX_RGBA8_SRGB);
            if (!d.mapKs.empty()) m.texB = gfxh_scene_load_texture(s, (dir + d.mapKs).c_str(), simplePbr ? GFX_TEX_RGBA8_UNORM : GFX_TEX_RGBA8_SRGB);
            if (simplePbr) {
                // MaterialConvention::SimplePBR (common_host.cpp:2323-2334, createSimplePBRMaterial :1689-1760): the diffuse slot
                // holds base colour (+ opacity) behind the sRGB sampler, the specular slot (occlusion, roughness, metallic) behind
                // the normalised-float sampler -- no degamma; no smoothness
                m.bsdfType = GFX_BSDF_SIMPLE_PBR;
                for (int i = 0; i < 3; ++i) m.b[i] = quantize8(d.ks[i]);
                m.smoothness = 0.0f;
            }
            const std::string& nmap = !d.mapBump.empty() ? d.mapBump : d.mapNormal;   // TEXTURE_HEIGHT first, then TEXTURE_NORMALS (:2278-2282)
            if (!nmap.empty()) { m.texNormal = gfxh_scene_load_texture(s, (dir + nmap).c_str(), GFX_TEX_RGBA8_UNORM); m.bumpMapType = GFX_BUMP_NORMAL_MAP; }
            if (!d.mapKe.empty()) {
                m.texEmittance = gfxh_scene_load_texture(s, (dir + d.mapKe).c_str(), GFX_TEX_RGBA8_SRGB);
                if (m.texEmittance) m.hasEmittance = 1u;
            }
        }
        const std::vector<Corner>& cs = facesByMat[name];
        Geom g; g.mat = matSlot;
        std::map<std::tuple<int, int, int>, uint32_t> dedup;   // aiProcess_JoinIdenticalVertices
        for (size_t f = 0; f + 2 < cs.size(); f += 3) {
            V3 fn = { 0, 0, 1 };
            bool needFaceNormal = cs[f].n < 0 || cs[f + 1].n < 0 || cs[f + 2].n < 0;
            if (needFaceNormal) fn = normalize(cross(pos[cs[f + 1].v] - pos[cs[f].v], pos[cs[f + 2].v] - pos[cs[f].v]));
            for (int k = 0; k < 3; ++k) {
                const Corner c = cs[f + k];
                const auto key = std::make_tuple(c.v, c.t, needFaceNormal ? -2 - static_cast<int>(f) : c.n);
                auto it = dedup.find(key);
                uint32_t vi;
                if (it != dedup.end()) vi = it->second;
                else {
                    const V3 n = normalize(needFaceNormal ? fn : nrm[c.n]);
                    const V3 tg = normalize(tangent_from_normal(n));
                    const float u = c.t >= 0 ? uv[c.t].first : 0.0f;
                    const float v = c.t >= 0 ? 1.0f - uv[c.t].second : 0.0f;   // aiProcess_FlipUVs
                    g.v.push_back(make_vertex(pos[c.v], n, tg, u, v));
                    vi = static_cast<uint32_t>(g.v.size() - 1);
                    dedup[key] = vi;
                }
                g.t.push_back(vi);
            }
        }
        // aiProcess_CalcTangentSpace (common_host.cpp:2163, 2346-2368: texCoord0Dir = aiMesh->mTangents when the mesh has texture
        // coordinates, the frame built from the normal otherwise): the tangent of a vertex is the direction in which u grows, dP/du of
        // its triangles -- (e1 dv2 - e2 dv1) / (du1 dv2 - du2 dv1), unchanged by the v flip above --, summed over the triangles that
        // share the vertex, made orthogonal to the normal.  Triangles without texture coordinates or with a degenerate mapping
        // contribute nothing; a vertex nothing contributed to keeps the frame built from its normal.
        {
            std::vector<V3> sum(g.v.size(), V3{ 0, 0, 0 });
            for (size_t f = 0; f + 2 < cs.size(); f += 3) {
                if (cs[f].t < 0 || cs[f + 1].t < 0 || cs[f + 2].t < 0) continue;
                const uint32_t i0 = g.t[f], i1 = g.t[f + 1], i2 = g.t[f + 2];
                const V3 e1 = pos[cs[f + 1].v] - pos[cs[f].v], e2 = pos[cs[f + 2].v] - pos[cs[f].v];
                const double du1 = static_cast<double>(uv[cs[f + 1].t].first) - uv[cs[f].t].first, du2 = static_cast<double>(uv[cs[f + 2].t].first) - uv[cs[f].t].first;
                const double dv1 = -(static_cast<double>(uv[cs[f + 1].t].second) - uv[cs[f].t].second), dv2 = -(static_cast<double>(uv[cs[f + 2].t].second) - uv[cs[f].t].second);
                const double det = du1 * dv2 - du2 * dv1;
                if (!(std::fabs(det) > 1e-20)) continue;
                V3 t = { static_cast<float>((e1.x * dv2 - e2.x * dv1) / det), static_cast<float>((e1.y * dv2 - e2.y * dv1) / det), static_cast<float>((e1.z * dv2 - e2.z * dv1) / det) };
                const float len = std::sqrt(t.x * t.x + t.y * t.y + t.z * t.z);
                if (!(len > 0.0f) || !std::isfinite(len)) continue;
                t = { t.x / len, t.y / len, t.z / len };
                for (uint32_t i : { i0, i1, i2 }) sum[i] = sum[i] + t;
            }
            for (size_t i = 0; i < g.v.size(); ++i) {
                const V3 n = { g.v[i].normal[0], g.v[i].normal[1], g.v[i].normal[2] };
                const float d = sum[i].x * n.x + sum[i].y * n.y + sum[i].z * n.z;
                const V3 t = { sum[i].x - n.x * d, sum[i].y - n.y * d, sum[i].z - n.z * d };
                const float len = std::sqrt(t.x * t.x + t.y * t.y + t.z * t.z);
                if (!(len > 1e-6f) || !std::isfinite(len)) continue;
                g.v[i].texCoord0Dir[0] = t.x / len; g.v[i].texCoord0Dir[1] = t.y / len; g.v[i].texCoord0Dir[2] = t.z / len;
            }
        }
        s->geoms.push_back(std::move(g));
        geomSlots.push_back(static_cast<uint32_t>(s->geoms.size() - 1));
    }
    if (geomSlots.empty()) { g_hostError = std::string("no faces in ") + path; return 0xFFFFFFFFu; }
    return gfxh_scene_add_group(s, geomSlots.data(), static_cast<uint32_t>(geomSlots.size()));
}

uint32_t gfxh_scene_add_rectangle(gfxh_scene* s, float width, float depth, const float emittance[3]) {
    const float refl[3] = { 0.01f, 0.01f, 0.01f }, spec[3] = { 0, 0, 0 };
    const uint32_t mat = gfxh_scene_add_material_traditional(s, refl, spec, 0.3f, emittance);
    const V3 n = { 0, -1, 0 }, t = { 1, 0, 0 };
    const gfx_vertex v[4] = {
        make_vertex({ -0.5f * width, 0.0f, -0.5f * depth }, n, t, 0.0f, 1.0f),
        make_vertex({ 0.5f * width, 0.0f, -0.5f * depth }, n, t, 1.0f, 1.0f),
        make_vertex({ 0.5f * width, 0.0f, 0.5f * depth }, n, t, 1.0f, 0.0f),
        make_vertex({ -0.5f * width, 0.0f, 0.5f * depth }, n, t, 0.0f, 0.0f) };
    const uint32_t tris[6] = { 0, 1, 2, 0, 2, 3 };
    const uint32_t g = gfxh_scene_add_geom(s, v, 4, tris, 2, mat);
    return gfxh_scene_add_group(s, &g, 1);
}

uint32_t gfxh_scene_add_rectangle_textured(gfxh_scene* s, float width, float depth, const float emittance[3], const char* emitterTexturePath) {
    const uint32_t group = gfxh_scene_add_rectangle(s, width, depth, emittance);
    if (group == 0xFFFFFFFFu || !emitterTexturePath || !emitterTexturePath[0]) return group;
    // createEmittanceTexture (common_host.cpp:1524-1531): an 8-bit image goes behind the sRGB sampler, a float image behind the
    // float sampler; a file that cannot be read leaves the immediate emittance in place
    const uint32_t tex = gfxh_scene_load_texture(s, emitterTexturePath, GFX_TEX_RGBA8_SRGB);
    if (tex) { gfx_material& m = s->materials.back(); m.texEmittance = tex; m.hasEmittance = 1u; }
    return group;
}

void gfxh_make_transform(float scale, float rollDeg, float pitchDeg, float yawDeg, const float pos[3], float out[12]) {
    const double d2r = 3.14159265358979323846 / 180.0;
    double R[9];
    euler_matrix(rollDeg * d2r, pitchDeg * d2r, yawDeg * d2r, R);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) out[r * 4 + c] = static_cast<float>(R[r * 3 + c] * scale);
        out[r * 4 + 3] = pos[r];
    }
}
void gfxh_make_orientation(float rollDeg, float pitchDeg, float yawDeg, float out[9]) {
    const double d2r = 3.14159265358979323846 / 180.0;
    double R[9];
    euler_matrix(rollDeg * d2r, pitchDeg * d2r, yawDeg * d2r, R);
    for (int i = 0; i < 9; ++i) out[i] = static_cast<float>(R[i]);
}

void gfxh_seed_rng_states(uint64_t* states, uint64_t count, uint64_t seed) {
    std::mt19937_64 gen(seed);
    for (uint64_t i = 0; i < count; ++i) states[i] = gen();
}

int gfxh_scene_make_street(gfxh_scene* s, const gfxh_street_params* p) {
    Rng rng(p->seed);
    const float E = p->extent;
    auto mat = [&](float r, float g, float b, float sr, float sm, float e0 = 0, float e1 = 0, float e2 = 0) {
        const float d[3] = { r, g, b }, sp[3] = { sr, sr, sr }, em[3] = { e0, e1, e2 };
        return gfxh_scene_add_material_traditional(s, d, sp, sm, em);
    };
    const float ident[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    uint32_t groundMat = 0, groundGeom = 0, crateMat = 0;
    std::vector<uint32_t> wallMats, signMats;
    // ---- ground: cobbled street (height noise) as one big static instance
    {
        Geom g; g.mat = mat(0.35f, 0.33f, 0.30f, 0.2f, 0.2f);
        groundMat = g.mat;
        std::mt19937 hgen(p->seed * 7919u + 1);
        std::vector<float> h((p->groundTess + 1) * (p->groundTess + 1));
        for (float& v : h) v = ((hgen() >> 8) * (1.0f / 16777216.0f)) * 0.02f;
        const uint32_t nt = p->groundTess;
        add_grid(g, { -E, 0, E }, { 2 * E, 0, 0 }, { 0, 0, -2 * E }, nt, nt, [&](uint32_t i, uint32_t j) { return h[j * (nt + 1) + i]; });
        s->geoms.push_back(std::move(g));
        const uint32_t gs = static_cast<uint32_t>(s->geoms.size() - 1);
        groundGeom = gs;
        gfxh_scene_add_instance(s, gfxh_scene_add_group(s, &gs, 1), ident);
    }
    // ---- buildings: a few facade prototypes (wall grid with recessed windows + roof box), instanced
    const uint32_t numProto = 6;
    std::vector<uint32_t> protoGroups;
    std::vector<V3> protoSize;
    for (uint32_t k = 0; k < numProto; ++k) {
        const float w = rng.range(6, 14), hgt = rng.range(8, 22), dpt = rng.range(6, 12);
        Geom wall; wall.mat = mat(rng.range(0.4f, 0.8f), rng.range(0.35f, 0.7f), rng.range(0.3f, 0.6f), 0.04f, 0.1f);
        wallMats.push_back(wall.mat);
        Geom glass; glass.mat = mat(0.05f, 0.06f, 0.08f, 0.6f, 0.85f);
        const uint32_t ft = p->facadeTess;
        // four facades: displaced grids (window recesses) facing outward
        const V3 o[4] = { { -w / 2, 0, dpt / 2 }, { w / 2, 0, dpt / 2 }, { w / 2, 0, -dpt / 2 }, { -w / 2, 0, -dpt / 2 } };
        const V3 ex[4] = { { w, 0, 0 }, { 0, 0, -dpt }, { -w, 0, 0 }, { 0, 0, dpt } };
        for (int f = 0; f < 4; ++f) {
            add_grid(wall, o[f], ex[f], { 0, hgt, 0 }, ft, ft, [&](uint32_t i, uint32_t j) {
                const bool window = (i % 4 == 1 || i % 4 == 2) && (j % 4 == 1 || j % 4 == 2) && j > 3;
                return window ? -0.25f : 0.0f;
            });
        }
        add_box(wall, { -w / 2, hgt, -dpt / 2 }, { w / 2, hgt + 0.4f, dpt / 2 });
        // glass panes inside the recesses of the front facade
        for (uint32_t j = 5; j + 2 < ft; j += 4)
            for (uint32_t i = 1; i + 2 < ft; i += 4) {
                const float x0 = -w / 2 + w * (i + 0.1f) / ft, x1 = -w / 2 + w * (i + 1.9f) / ft;
                const float y0 = hgt * (j + 0.1f) / ft, y1 = hgt * (j + 1.9f) / ft;
                add_quad(glass, { x0, y0, dpt / 2 - 0.2f }, { x1, y0, dpt / 2 - 0.2f }, { x1, y1, dpt / 2 - 0.2f }, { x0, y1, dpt / 2 - 0.2f });
            }
        s->geoms.push_back(std::move(wall));
        s->geoms.push_back(std::move(glass));
        const uint32_t gs[2] = { static_cast<uint32_t>(s->geoms.size() - 2), static_cast<uint32_t>(s->geoms.size() - 1) };
        protoGroups.push_back(gfxh_scene_add_group(s, gs, 2));
        protoSize.push_back({ w, hgt, dpt });
    }
    struct Placed { V3 pos; float yaw; uint32_t proto; };
    std::vector<Placed> placed;
    for (uint32_t b = 0; b < p->numBuildings; ++b) {
        // two rows along the street (z axis), facing the street
        const bool left = (b & 1) != 0;
        const float z = -E * 0.9f + (2 * E * 0.9f) * (static_cast<float>(b / 2) + 0.5f) / std::max(1u, (p->numBuildings + 1) / 2);
        const uint32_t proto = rng.gen() % numProto;
        const float x = (left ? -1.0f : 1.0f) * (E * 0.35f + protoSize[proto].z * 0.5f);
        const float yaw = left ? 90.0f : -90.0f;
        const float pos[3] = { x, 0, z };
        float xfm[12];
        gfxh_make_transform(1.0f, 0, 0, yaw, pos, xfm);
        gfxh_scene_add_instance(s, protoGroups[proto], xfm);
        placed.push_back({ { x, 0, z }, yaw, proto });
    }
    // ---- props: icospheres (planters / bollards) and crates, instanced with random scale
    {
        Geom sphere; sphere.mat = mat(0.55f, 0.25f, 0.2f, 0.1f, 0.5f);
        make_icosphere(sphere, p->propSubdiv, 0.5f);
        Geom crate; crate.mat = mat(0.45f, 0.32f, 0.18f, 0.03f, 0.2f);
        crateMat = crate.mat;
        add_box(crate, { -0.5f, 0, -0.5f }, { 0.5f, 1, 0.5f });
        s->geoms.push_back(std::move(sphere));
        const uint32_t gsph = static_cast<uint32_t>(s->geoms.size() - 1);
        s->geoms.push_back(std::move(crate));
        const uint32_t gcr = static_cast<uint32_t>(s->geoms.size() - 1);
        const uint32_t grpS = gfxh_scene_add_group(s, &gsph, 1), grpC = gfxh_scene_add_group(s, &gcr, 1);
        for (uint32_t k = 0; k < p->numProps; ++k) {
            const bool sph = (k % 3) != 0;
            const float sc = rng.range(0.3f, 1.2f);
            const float pos[3] = { rng.range(-E * 0.33f, E * 0.33f), sph ? sc * 0.5f : 0.0f, rng.range(-E * 0.95f, E * 0.95f) };
            float xfm[12];
            gfxh_make_transform(sc, 0, 0, rng.range(0, 360), pos, xfm);
            gfxh_scene_add_instance(s, sph ? grpS : grpC, xfm);
        }
    }
    // ---- lamps: pole + small emissive box head; a handful of colour temperatures
    {
        Geom pole; pole.mat = mat(0.1f, 0.1f, 0.1f, 0.3f, 0.6f);
        add_box(pole, { -0.05f, 0, -0.05f }, { 0.05f, 3.5f, 0.05f });
        s->geoms.push_back(std::move(pole));
        const uint32_t gpole = static_cast<uint32_t>(s->geoms.size() - 1);
        std::vector<uint32_t> lampGroups;
        const float tints[4][3] = { { 1.0f, 0.85f, 0.6f }, { 1.0f, 0.95f, 0.85f }, { 0.8f, 0.9f, 1.0f }, { 1.0f, 0.7f, 0.4f } };
        for (int k = 0; k < 4; ++k) {
            Geom head; head.mat = mat(0.01f, 0.01f, 0.01f, 0, 0.3f, p->lampEmittance * tints[k][0], p->lampEmittance * tints[k][1], p->lampEmittance * tints[k][2]);
            add_box(head, { -0.15f, 3.5f, -0.15f }, { 0.15f, 3.7f, 0.15f });
            s->geoms.push_back(std::move(head));
            const uint32_t gs[2] = { gpole, static_cast<uint32_t>(s->geoms.size() - 1) };
            lampGroups.push_back(gfxh_scene_add_group(s, gs, 2));
        }
        for (uint32_t k = 0; k < p->numLamps; ++k) {
            const float pos[3] = { rng.range(-E * 0.34f, E * 0.34f), 0, rng.range(-E * 0.95f, E * 0.95f) };
            float xfm[12];
            gfxh_make_transform(rng.range(0.8f, 1.2f), 0, 0, rng.range(0, 360), pos, xfm);
            gfxh_scene_add_instance(s, lampGroups[rng.gen() % 4], xfm);
        }
    }
    // ---- signs: emissive quads mounted on facades
    {
        std::vector<uint32_t> signGroups;
        const float cols[5][3] = { { 1, 0.2f, 0.2f }, { 0.2f, 1, 0.3f }, { 0.2f, 0.4f, 1 }, { 1, 0.9f, 0.2f }, { 1, 0.3f, 0.9f } };
        for (int k = 0; k < 5; ++k) {
            Geom sign; sign.mat = mat(0.01f, 0.01f, 0.01f, 0, 0.3f, p->signEmittance * cols[k][0], p->signEmittance * cols[k][1], p->signEmittance * cols[k][2]);
            signMats.push_back(sign.mat);
            add_grid(sign, { -0.6f, -0.2f, 0 }, { 1.2f, 0, 0 }, { 0, 0.4f, 0 }, 4, 2, [](uint32_t, uint32_t) { return 0.0f; });
            s->geoms.push_back(std::move(sign));
            const uint32_t gs = static_cast<uint32_t>(s->geoms.size() - 1);
            signGroups.push_back(gfxh_scene_add_group(s, &gs, 1));
        }
        for (uint32_t k = 0; k < p->numSigns && !placed.empty(); ++k) {
            const Placed& b = placed[rng.gen() % placed.size()];
            const V3 sz = protoSize[b.proto];
            // local position on the front facade (+z of the prototype), slightly in front of it
            const float lp[3] = { rng.range(-sz.x * 0.4f, sz.x * 0.4f), rng.range(2.5f, std::max(3.0f, sz.y * 0.8f)), sz.z * 0.5f + 0.05f };
            float bx[12];
            const float bpos[3] = { b.pos.x, b.pos.y, b.pos.z };
            gfxh_make_transform(1.0f, 0, 0, b.yaw, bpos, bx);
            float wp[3];
            xfm_point(bx, lp, wp);
            float xfm[12];
            gfxh_make_transform(rng.range(0.7f, 1.6f), 0, 0, b.yaw, wp, xfm);
            gfxh_scene_add_instance(s, signGroups[rng.gen() % 5], xfm);
        }
    }
    // ---- depth complexity (p->numTrees, p->numWires, p->numRailings): what Bistro's vegetation, cables and balcony railings
    // do to a ray tracer -- clumps of small randomly oriented leaf cards (thousands of overlapping boxes a ray grazes without
    // hitting anything), and long thin boxes whose bounding volumes cover mostly air.  Own RNG stream: scenes without these
    // parameters are unchanged.
    if (p->numTrees || p->numWires || p->numRailings) {
        Rng crng(p->seed * 747796405u + 2891336453u);
        const uint32_t leafMat = mat(0.12f, 0.32f, 0.08f, 0.04f, 0.3f), barkMat = mat(0.25f, 0.18f, 0.12f, 0.02f, 0.1f), metalMat = mat(0.3f, 0.3f, 0.32f, 0.5f, 0.7f);
        if (p->numTrees) {
            std::vector<uint32_t> treeGroups;
            for (int proto = 0; proto < 3; ++proto) {
                Geom trunk; trunk.mat = barkMat;
                const float th = crng.range(2.5f, 3.5f);
                add_box(trunk, { -0.12f, 0, -0.12f }, { 0.12f, th, 0.12f });
                Geom leaves; leaves.mat = leafMat;
                const float rx = crng.range(1.4f, 2.2f), ry = crng.range(1.2f, 2.0f), rz = crng.range(1.4f, 2.2f);
                for (uint32_t k = 0; k < p->leavesPerTree; ++k) {
                    // a point inside the crown ellipsoid (rejection), a random card orientation, 12-30 cm
                    V3 c;
                    do { c = { crng.range(-1, 1), crng.range(-1, 1), crng.range(-1, 1) }; } while (c.x * c.x + c.y * c.y + c.z * c.z > 1.0f);
                    c = { c.x * rx, th + ry * 0.8f + c.y * ry, c.z * rz };
                    V3 u = normalize({ crng.range(-1, 1), crng.range(-1, 1), crng.range(-1, 1) });
                    V3 w = normalize(cross(u, { crng.range(-1, 1), crng.range(-1, 1) + 1.5f, crng.range(-1, 1) }));
                    const float hs = crng.range(0.06f, 0.15f);
                    const V3 a = { u.x * hs, u.y * hs, u.z * hs }, b = { w.x * hs, w.y * hs, w.z * hs };
                    add_quad(leaves, c - a - b, c + a - b, c + a + b, c - a + b);
                }
                s->geoms.push_back(std::move(trunk));
                s->geoms.push_back(std::move(leaves));
                const uint32_t gs[2] = { static_cast<uint32_t>(s->geoms.size() - 2), static_cast<uint32_t>(s->geoms.size() - 1) };
                treeGroups.push_back(gfxh_scene_add_group(s, gs, 2));
            }
            for (uint32_t k = 0; k < p->numTrees; ++k) {   // two rows along the kerbs
                const float side = (k & 1u) ? 1.0f : -1.0f;
                const float pos[3] = { side * E * crng.range(0.22f, 0.3f), 0, -E * 0.92f + 2 * E * 0.92f * (static_cast<float>(k / 2) + crng.range(0.2f, 0.8f)) / std::max(1u, (p->numTrees + 1) / 2) };
                float xfm[12];
                gfxh_make_transform(crng.range(0.8f, 1.3f), 0, 0, crng.range(0, 360), pos, xfm);
                gfxh_scene_add_instance(s, treeGroups[crng.gen() % 3], xfm);
            }
        }
        if (p->numWires) {   // cables across the street between the facade rows, slightly sagging: 8 thin segments each
            Geom wires; wires.mat = metalMat;
            for (uint32_t k = 0; k < p->numWires; ++k) {
                const float z0 = crng.range(-E * 0.9f, E * 0.9f), z1 = z0 + crng.range(-6.0f, 6.0f), y = crng.range(5.0f, 9.0f);
                const float x0 = -E * 0.36f, x1 = E * 0.36f, r = 0.015f;
                for (int sgm = 0; sgm < 8; ++sgm) {
                    const float t0 = sgm / 8.0f, t1 = (sgm + 1) / 8.0f;
                    const float xa = x0 + (x1 - x0) * t0, xb = x0 + (x1 - x0) * t1, za = z0 + (z1 - z0) * t0, zb = z0 + (z1 - z0) * t1;
                    const float ya = y - 1.2f * 4 * t0 * (1 - t0), yb = y - 1.2f * 4 * t1 * (1 - t1);
                    // a box around the segment, axis-aligned in x (the sag and the skew make its BVH box mostly empty)
                    add_quad(wires, { xa, ya - r, za - r }, { xb, yb - r, zb - r }, { xb, yb + r, zb - r }, { xa, ya + r, za - r });
                    add_quad(wires, { xa, ya + r, za + r }, { xb, yb + r, zb + r }, { xb, yb - r, zb + r }, { xa, ya - r, za + r });
                    add_quad(wires, { xa, ya + r, za - r }, { xb, yb + r, zb - r }, { xb, yb + r, zb + r }, { xa, ya + r, za + r });
                    add_quad(wires, { xa, ya - r, za + r }, { xb, yb - r, zb + r }, { xb, yb - r, zb - r }, { xa, ya - r, za - r });
                }
            }
            s->geoms.push_back(std::move(wires));
            const uint32_t gs = static_cast<uint32_t>(s->geoms.size() - 1);
            gfxh_scene_add_instance(s, gfxh_scene_add_group(s, &gs, 1), ident);
        }
        if (p->numRailings) {   // a railing segment = two rails + 24 thin bars, instanced along the kerbs
            Geom rail; rail.mat = metalMat;
            add_box(rail, { -1.5f, 0.95f, -0.02f }, { 1.5f, 1.0f, 0.02f });
            add_box(rail, { -1.5f, 0.1f, -0.02f }, { 1.5f, 0.14f, 0.02f });
            for (int b = 0; b < 24; ++b) { const float x = -1.5f + 3.0f * (b + 0.5f) / 24; add_box(rail, { x - 0.008f, 0.14f, -0.008f }, { x + 0.008f, 0.95f, 0.008f }); }
            s->geoms.push_back(std::move(rail));
            const uint32_t gs = static_cast<uint32_t>(s->geoms.size() - 1);
            const uint32_t grp = gfxh_scene_add_group(s, &gs, 1);
            for (uint32_t k = 0; k < p->numRailings; ++k) {
                const float side = (k & 1u) ? 1.0f : -1.0f;
                const float pos[3] = { side * E * 0.2f, 0, -E * 0.95f + 2 * E * 0.95f * (static_cast<float>(k / 2) + 0.5f) / std::max(1u, (p->numRailings + 1) / 2) };
                float xfm[12];
                gfxh_make_transform(1.0f, 0, 0, 90.0f, pos, xfm);
                gfxh_scene_add_instance(s, grp, xfm);
            }
        }
    }
    // ---- textures (p->textured): the geometry above is unchanged, materials get maps instead of constants --
    // cobbled ground and plastered / bricked facades with albedo, smoothness and normal maps, wooden crates, and
    // signs whose emittance is a float texture (lettering-like stripes), so every texture fetch of the reference
    // path is exercised: setupBSDFBody's three reads, the normal map under bump mapping, and the emittance reads of
    // sampleLight, the shading pass and computeTriangleImportance.
    if (p->textured) {
        std::mt19937 tgen(p->seed * 2654435761u + 17u);
        auto hash01 = [](uint32_t x, uint32_t y, uint32_t k) {   // integer hash -> [0, 1)
            uint32_t h = x * 0x9E3779B1u ^ (y * 0x85EBCA77u + k * 0xC2B2AE3Du);
            h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
            return (h >> 8) * (1.0f / 16777216.0f);
        };
        auto to8 = [](float v) { return static_cast<uint8_t>(std::min(255.0f, std::max(0.0f, v * 255.0f + 0.5f))); };
        // height field of a tiling pattern: cells of (cw x ch) texels with a groove of `gap` texels, per-cell tint
        auto tile_maps = [&](uint32_t N, uint32_t cw, uint32_t ch, uint32_t gap, bool stagger, const float base[3], float tintAmp, uint32_t salt,
                             std::vector<uint8_t>& albedo, std::vector<uint8_t>& normal, std::vector<uint8_t>& smooth) {
            std::vector<float> height(static_cast<size_t>(N) * N);
            albedo.resize(4ull * N * N); normal.resize(4ull * N * N); smooth.resize(static_cast<size_t>(N) * N);
            for (uint32_t y = 0; y < N; ++y)
                for (uint32_t x = 0; x < N; ++x) {
                    const uint32_t row = y / ch;
                    const uint32_t xs = stagger && (row & 1u) ? x + cw / 2 : x;
                    const uint32_t col = (xs / cw) % (N / cw);
                    const uint32_t ix = xs % cw, iy = y % ch;
                    const bool groove = ix < gap || iy < gap;
                    const float grain = hash01(x, y, salt);
                    const float tint = 1.0f + tintAmp * (hash01(col, row, salt + 1) - 0.5f);
                    height[static_cast<size_t>(y) * N + x] = groove ? 0.0f : 0.7f + 0.3f * grain;
                    uint8_t* a = albedo.data() + 4ull * (static_cast<size_t>(y) * N + x);
                    for (int c = 0; c < 3; ++c) a[c] = to8((groove ? 0.45f : 1.0f) * base[c] * tint * (0.9f + 0.2f * grain));
                    a[3] = 255;
                    smooth[static_cast<size_t>(y) * N + x] = to8(groove ? 0.05f : 0.15f + 0.25f * hash01(col, row, salt + 2));
                }
            for (uint32_t y = 0; y < N; ++y)
                for (uint32_t x = 0; x < N; ++x) {
                    const float hx = height[static_cast<size_t>(y) * N + (x + 1) % N] - height[static_cast<size_t>(y) * N + (x + N - 1) % N];
                    const float hy = height[static_cast<size_t>((y + 1) % N) * N + x] - height[static_cast<size_t>((y + N - 1) % N) * N + x];
                    const V3 n = normalize({ -1.5f * hx, -1.5f * hy, 1.0f });
                    uint8_t* o = normal.data() + 4ull * (static_cast<size_t>(y) * N + x);
                    o[0] = to8(0.5f * n.x + 0.5f); o[1] = to8(0.5f * n.y + 0.5f); o[2] = to8(0.5f * n.z + 0.5f); o[3] = 255;
                }
        };
        auto texture_material = [&](uint32_t matSlot, uint32_t N, uint32_t cw, uint32_t ch, uint32_t gap, bool stagger, const float base[3], float tintAmp) {
            std::vector<uint8_t> albedo, normal, smooth;
            tile_maps(N, cw, ch, gap, stagger, base, tintAmp, tgen(), albedo, normal, smooth);
            gfx_material& m = s->materials[matSlot];
            m.texA = gfxh_scene_add_texture(s, N, N, GFX_TEX_RGBA8_SRGB, albedo.data());
            m.texSmoothness = gfxh_scene_add_texture(s, N, N, GFX_TEX_R8_UNORM, smooth.data());
            m.texNormal = gfxh_scene_add_texture(s, N, N, GFX_TEX_RGBA8_UNORM, normal.data());
            m.bumpMapType = GFX_BUMP_NORMAL_MAP;
        };
        const float cobble[3] = { 0.62f, 0.58f, 0.52f };
        texture_material(groundMat, 256, 32, 32, 3, true, cobble, 0.5f);
        for (gfx_vertex& v : s->geoms[groundGeom].v) { v.texCoord[0] *= 0.5f * E; v.texCoord[1] *= 0.5f * E; }   // one tile = 4 m
        for (size_t k = 0; k < wallMats.size(); ++k) {
            const gfx_material& wm = s->materials[wallMats[k]];
            // bricks for every other prototype, large plaster panels for the rest; tinted by the prototype's own colour
            const float base[3] = { std::min(1.0f, 0.35f + 1.2f * wm.a[0]), std::min(1.0f, 0.3f + 1.2f * wm.a[1]), std::min(1.0f, 0.28f + 1.2f * wm.a[2]) };
            if (k & 1u) texture_material(wallMats[k], 256, 32, 16, 2, true, base, 0.35f);
            else texture_material(wallMats[k], 128, 64, 64, 1, false, base, 0.12f);
        }
        {
            const float wood[3] = { 0.72f, 0.52f, 0.30f };
            texture_material(crateMat, 128, 128, 16, 1, false, wood, 0.4f);
        }
        for (size_t k = 0; k < signMats.size(); ++k) {   // float emittance map: bright strokes on a dim panel
            const uint32_t W = 64, H = 32;
            gfx_material& m = s->materials[signMats[k]];
            std::vector<float> e(4ull * W * H);
            const uint32_t salt = tgen();
            for (uint32_t y = 0; y < H; ++y)
                for (uint32_t x = 0; x < W; ++x) {
                    const bool border = x < 2 || y < 2 || x >= W - 2 || y >= H - 2;
                    const bool stroke = y > 8 && y < 24 && ((x / 4) % 2 == 0) && hash01(x / 4, y / 8, salt) > 0.25f;
                    const float level = border ? 1.0f : stroke ? 1.6f : 0.25f;
                    float* o = e.data() + 4ull * (static_cast<size_t>(y) * W + x);
                    for (int c = 0; c < 3; ++c) o[c] = level * m.emittance[c];
                    o[3] = 1.0f;
                }
            m.texEmittance = gfxh_scene_add_texture(s, W, H, GFX_TEX_RGBA32F, e.data());
        }
    }
    return 0;
}

int gfxh_scene_counts(gfxh_scene* s, uint32_t counts[5]) {
    counts[0] = static_cast<uint32_t>(s->materials.size());
    counts[1] = static_cast<uint32_t>(s->geoms.size());
    counts[2] = static_cast<uint32_t>(s->groups.size());
    counts[3] = static_cast<uint32_t>(s->insts.size());
    uint64_t tris = 0;
    for (const Inst& i : s->insts) for (uint32_t g : s->groups[i.group]) tris += s->geoms[g].t.size() / 3;
    counts[4] = static_cast<uint32_t>(tris);
    return 0;
}
int gfxh_scene_get_material(gfxh_scene* s, uint32_t i, gfx_material* out) { if (i >= s->materials.size()) return 1; *out = s->materials[i]; return 0; }
int gfxh_scene_get_geom(gfxh_scene* s, uint32_t i, const gfx_vertex** v, uint32_t* nv, const uint32_t** tris, uint32_t* nt, uint32_t* matSlot) {
    if (i >= s->geoms.size()) return 1;
    const Geom& g = s->geoms[i];
    *v = g.v.data(); *nv = static_cast<uint32_t>(g.v.size()); *tris = g.t.data(); *nt = static_cast<uint32_t>(g.t.size() / 3); *matSlot = g.mat;
    return 0;
}
int gfxh_scene_get_group(gfxh_scene* s, uint32_t i, const uint32_t** geomSlots, uint32_t* n) {
    if (i >= s->groups.size()) return 1;
    *geomSlots = s->groups[i].data(); *n = static_cast<uint32_t>(s->groups[i].size());
    return 0;
}
int gfxh_scene_get_instance(gfxh_scene* s, uint32_t i, uint32_t* group, float xfm[12]) {
    if (i >= s->insts.size()) return 1;
    *group = s->insts[i].group; std::memcpy(xfm, s->insts[i].xfm, sizeof(float) * 12);
    return 0;
}
int gfxh_scene_bounds(gfxh_scene* s, float bounds[6]) {
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (const Inst& i : s->insts)
        for (uint32_t g : s->groups[i.group])
            for (const gfx_vertex& v : s->geoms[g].v) {
                float w[3];
                xfm_point(i.xfm, v.position, w);
                for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], w[k]); hi[k] = std::max(hi[k], w[k]); }
            }
    for (int k = 0; k < 3; ++k) { bounds[k] = lo[k]; bounds[3 + k] = hi[k]; }
    return 0;
}

int gfxh_scene_upload(gfxh_scene* s, gfx_ctx* ctx) {
    for (uint32_t t = 0; t < s->textures.size(); ++t) {
        const Tex& tx = s->textures[t];
        if (gfx_texture_set(ctx, t + 1, tx.width, tx.height, tx.format, tx.texels.data())) { g_hostError = gfx_last_error(ctx); return 1; }
    }
    for (uint32_t i = 0; i < s->materials.size(); ++i)
        if (gfx_material_set(ctx, i, &s->materials[i])) { g_hostError = gfx_last_error(ctx); return 1; }
    for (const Geom& g : s->geoms) {
        uint32_t slot;
        if (gfx_geom_create(ctx, g.v.data(), sizeof(gfx_vertex), static_cast<uint32_t>(g.v.size()), g.t.data(),
                            static_cast<uint32_t>(g.t.size() / 3), g.mat, &slot)) { g_hostError = gfx_last_error(ctx); return 1; }
    }
    for (const auto& grp : s->groups) {
        uint32_t slot;
        if (gfx_group_create(ctx, grp.data(), static_cast<uint32_t>(grp.size()), &slot)) { g_hostError = gfx_last_error(ctx); return 1; }
    }
    for (const Inst& i : s->insts) {
        uint32_t slot;
        if (gfx_instance_create(ctx, i.group, i.xfm, &slot)) { g_hostError = gfx_last_error(ctx); return 1; }
    }
    return 0;
}

// restir_di_main.cpp:1487-1542 -- Halton(2,3) through the concentric square->disk map.  The host
// program uses <cmath> cos/sin; here the table goes through the same deterministic sincos as the
// kernels so any consumer (including a CPU checker) reproduces it bit for bit.
static void host_sincos(float x, float* s, float* c) {
    // identical algorithm to gfx::gm_sincos (gm_math.hip.h), host build
    const float q = std::rint(x * 0.6366197466850281f);
    float r = std::fma(q, -1.5703125f, x);
    r = std::fma(q, -0.0004837512969970703f, r);
    r = std::fma(q, -7.549790126404332e-08f, r);
    const int n = static_cast<int>(q);
    const float r2 = r * r;
    float ps = std::fma(-1.9515295891e-4f, r2, 8.3321608736e-3f);
    ps = std::fma(ps, r2, -1.6666654611e-1f);
    const float sr = std::fma(ps * r2, r, r);
    float pc = std::fma(2.443315711809948e-5f, r2, -1.388731625493765e-3f);
    pc = std::fma(pc, r2, 4.166664568298827e-2f);
    const float cr = std::fma(pc * r2, r2, std::fma(-0.5f, r2, 1.0f));
    const float ss = (n & 1) ? cr : sr, cc = (n & 1) ? sr : cr;
    *s = (n & 2) ? -ss : ss;
    *c = ((n + 1) & 2) ? -cc : cc;
}
// RegularConstantContinuousDistribution1D::initialize, common/common_host.cpp:292-316 (Kahan sums)
static float build_rccd1d(const float* values, uint32_t n, float* pdf, float* cdf) {
    float result = 0.0f, comp = 0.0f;   // CompensatedSum_T, common/basic_types.h:5428-5452
    for (uint32_t i = 0; i < n; ++i) {
        cdf[i] = result;
        const float input = values[i] / n - comp;
        const float t = result + input;
        comp = (t - result) - input;
        result = t;
    }
    const float integral = result;
    for (uint32_t i = 0; i < n; ++i) { pdf[i] = values[i] / integral; cdf[i] /= integral; }
    cdf[n] = 1.0f;
    return integral;
}

int gfxh_env_build_importance(float* texels, uint32_t w, uint32_t h, float* rowPDF, float* rowCDF,
                              float* rowIntegrals, float* topPDF, float* topCDF, float* topIntegral) {
    std::vector<float> importance(static_cast<size_t>(w) * h);
    for (uint32_t y = 0; y < h; ++y) {
        const float theta = 3.14159265358979323846f * (y + 0.5f) / h;
        float sinTheta, cosTheta;
        host_sincos(theta, &sinTheta, &cosTheta);
        for (uint32_t x = 0; x < w; ++x) {
            float* t = texels + 4 * (static_cast<size_t>(y) * w + x);
            for (int c = 0; c < 3; ++c) t[c] = std::min(std::max(t[c], 0.0f), 65504.0f);
            importance[static_cast<size_t>(y) * w + x] = (0.2126729f * t[0] + 0.7151522f * t[1] + 0.0721750f * t[2]) * sinTheta;
        }
    }
    for (uint32_t y = 0; y < h; ++y)
        rowIntegrals[y] = build_rccd1d(importance.data() + static_cast<size_t>(y) * w, w, rowPDF + static_cast<size_t>(y) * w,
                                       rowCDF + static_cast<size_t>(y) * (w + 1));
    *topIntegral = build_rccd1d(rowIntegrals, h, topPDF, topCDF);
    return 0;
}

// guide[k] = largest index i in [0, n) with cell(cdf[i]) <= k, cell(x) = min(n - 1, uint(x * n)): the device
// samplers (shading.hip.h, EnvMap::sample1d) bracket the search for u with guide[cell(u) - 1] .. guide[cell(u)].
static bool build_guide(const float* cdf, uint32_t n, uint16_t* guide) {
    if (n == 0 || n > 65536u) return false;
    if (!(cdf[0] == 0.0f)) return false;
    for (uint32_t i = 0; i + 1 < n; ++i) if (!(cdf[i] <= cdf[i + 1])) return false;
    auto cell = [n](float x) { return std::min<uint32_t>(n - 1u, static_cast<uint32_t>(x * static_cast<float>(n))); };
    uint32_t idx = 0;
    for (uint32_t k = 0; k < n; ++k) {
        while (idx + 1 < n && cell(cdf[idx + 1]) <= k) ++idx;
        guide[k] = static_cast<uint16_t>(idx);
    }
    return true;
}

int gfxh_env_build_guides(const float* rowCDF, const float* topCDF, uint32_t w, uint32_t h, uint16_t* rowGuide, uint16_t* topGuide) {
    if (!build_guide(topCDF, h, topGuide)) return 0;
    for (uint32_t y = 0; y < h; ++y)
        if (!build_guide(rowCDF + static_cast<size_t>(y) * (w + 1), w, rowGuide + static_cast<size_t>(y) * w)) return 0;
    return 1;
}

void gfxh_env_build_row_table(const float* texels, const float* rowPDF, const float* rowCDF, const uint16_t* rowGuide, uint32_t w, uint32_t h, void* outRecords) {
    // record (row, i) of 32 bytes: {cdf, pdf, guide, r | g, b, cdf of record i + 1, 0} (shading.hip.h EnvRowRec); i = w: the row's final CDF
    // value alone; rows GFX_ENV_ROW_STRIDE(w) records apart, so that four consecutive records from a multiple of four are one 128-byte line
    uint32_t* out = static_cast<uint32_t*>(outRecords);
    auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    const size_t stride = GFX_ENV_ROW_STRIDE(w);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t i = 0; i < stride; ++i) {
            uint32_t* rec = out + 8 * (static_cast<size_t>(y) * stride + i);
            for (int k = 0; k < 8; ++k) rec[k] = 0u;
            if (i > w) continue;
            rec[0] = bits(rowCDF[static_cast<size_t>(y) * (w + 1) + i]);
            if (i < w) {
                const float* t = texels + 4 * (static_cast<size_t>(y) * w + i);
                rec[1] = bits(rowPDF[static_cast<size_t>(y) * w + i]);
                rec[2] = rowGuide[static_cast<size_t>(y) * w + i];
                rec[3] = bits(t[0]); rec[4] = bits(t[1]); rec[5] = bits(t[2]);
                rec[6] = bits(rowCDF[static_cast<size_t>(y) * (w + 1) + i + 1]);
            }
        }
}

// The column RegularConstantContinuousDistribution1D::sample's bisection ends on: the largest index of [0, n - 1] whose CDF value is <= u.
static uint32_t env_column_of(const float* cdf, uint32_t n, float u) {
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (cdf[mid] <= u) lo = mid; else hi = mid - 1; }
    return lo;
}
// The device's interpolation between two knots (shading.hip.h EnvMap::sample1d_row_sketch: the same operations in the same order; the
// position t inside the cell is exact, the rest is one subtraction, one product, one sum).
static int env_sketch_interpolate(const float* knots, uint32_t k, float t) {
    const float d = knots[k + 1] - knots[k];
    const float p = knots[k] + t * d;
    return static_cast<int>(p);
}
// One sketch record over [uLo, uLo + 32 step): 33 knots of the row's inverse CDF and the mask of the cells whose interpolation is within
// one column of the bisection's answer for EVERY u of the cell.  `local(u, k, t)`: the cell and the position inside it the device derives
// for u at this level.  The interpolation is monotone in u inside a cell (a product and a sum of non-negative terms, correctly rounded)
// and the column is constant between the lowest and the highest u that end on it: testing both ends of every column covers the cell.
extern "C++" {
template <typename Local>
static uint32_t env_sketch_record(const float* cdf, uint32_t w, float uLo, float step, Local local, float knots[GFX_ENV_SKETCH_CELLS + 1]) {
    const uint32_t K = GFX_ENV_SKETCH_CELLS;
    for (uint32_t j = 0; j <= K; ++j) {
        const float u = uLo + static_cast<float>(j) * step;          // exact: powers of two
        float pos = static_cast<float>(w);
        if (u < 1.0f) {
            const uint32_t c = env_column_of(cdf, w, u);
            const float width = cdf[c + 1] - cdf[c];
            const float t = width > 0.0f ? (u - cdf[c]) / width : 0.0f;
            pos = static_cast<float>(c) + std::min(std::max(t, 0.0f), 1.0f);
        }
        knots[j] = pos;
    }
    for (uint32_t j = 0; j < K; ++j) if (!(knots[j] <= knots[j + 1])) return 0u;   // the prediction must not decrease inside a cell
    uint32_t mask = 0;
    for (uint32_t k = 0; k < K; ++k) {
        const float cLo = uLo + static_cast<float>(k) * step, cHi = std::nextafter(uLo + static_cast<float>(k + 1) * step, 0.0f);
        const uint32_t cFirst = env_column_of(cdf, w, cLo), cLast = env_column_of(cdf, w, cHi);
        bool ok = true;
        for (uint32_t c = cFirst; c <= cLast && ok; ++c) {
            const float a = std::max(cLo, cdf[c]);
            const float b = cdf[c + 1] <= cHi ? std::nextafter(cdf[c + 1], 0.0f) : cHi;
            if (!(a <= b)) continue;                                  // an empty column: never the bisection's answer
            for (float u : { a, b }) {
                if (env_column_of(cdf, w, u) != c) continue;          // (ties: this u belongs to a later column of equal CDF value, tested there)
                uint32_t kk; float t;
                local(u, kk, t);
                if (kk != k) { ok = false; break; }                   // (cannot happen: the cell bounds are exact)
                const int pred = env_sketch_interpolate(knots, kk, t);
                if (pred < static_cast<int>(c) - 1 || pred > static_cast<int>(c) + 1) ok = false;
            }
        }
        if (ok) mask |= 1u << k;
    }
    return mask;
}
}   // extern "C++"

uint32_t gfxh_env_build_row_sketch(const float* rowCDF, uint32_t w, uint32_t h, void* outSketch, uint32_t capacityRecords, uint32_t* numRecords) {
    const uint32_t K = GFX_ENV_SKETCH_CELLS, W = GFX_ENV_SKETCH_WORDS;
    std::vector<uint32_t> rows(static_cast<size_t>(h) * W, 0u), children;
    uint32_t good = 0, numChildren = 0;
    auto cell_of = [](float x, uint32_t& k, float& t) {
        const uint32_t K = GFX_ENV_SKETCH_CELLS;             // the device's split of a position in [0, 1) into cell and remainder
        const float xk = x * static_cast<float>(K);
        k = static_cast<uint32_t>(xk);
        if (k > K - 1u) k = K - 1u;
        t = xk - static_cast<float>(k);
    };
    for (uint32_t y = 0; y < h; ++y) {
        const float* cdf = rowCDF + static_cast<size_t>(y) * (w + 1);
        bool monotone = cdf[0] == 0.0f;
        for (uint32_t i = 0; i < w && monotone; ++i) monotone = cdf[i] <= cdf[i + 1];
        float knots[GFX_ENV_SKETCH_CELLS + 1];
        uint32_t mask = 0;
        if (monotone) mask = env_sketch_record(cdf, w, 0.0f, 1.0f / K, [&](float u, uint32_t& k, float& t) { cell_of(u, k, t); }, knots);
        else for (uint32_t j = 0; j <= K; ++j) knots[j] = 0.0f;
        uint32_t* row = rows.data() + static_cast<size_t>(y) * W;
        std::memcpy(row, knots, 4 * (K + 1));
        for (uint32_t k = 0; k < K; ++k) if (!((mask >> k) & 1u)) row[k] |= 0x80000000u;   // the sign bit of knot k repeats mask bit k (cleared = verified): one sector per sample
        row[K + 1] = mask; row[K + 2] = numChildren;
        for (uint32_t k = 0; k < K; ++k) {
            if ((mask >> k) & 1u) { ++good; continue; }
            // a child record for the failing cell: the same at 1/32 of the step (a row whose CDF is not monotone gets empty children: the guide)
            float sub[GFX_ENV_SKETCH_CELLS + 1];
            uint32_t subMask = 0;
            if (monotone) {
                const uint32_t k1 = k;
                subMask = env_sketch_record(cdf, w, static_cast<float>(k1) / K, 1.0f / (K * K), [&](float u, uint32_t& kk, float& t) {
                    uint32_t ka; float ta;
                    cell_of(u, ka, ta);                              // level 1: ka == k1 for every u of this cell
                    cell_of(ta, kk, t);                              // level 2: the remainder is the position inside the cell
                    if (ka != k1) kk = K;                            // (reported as a mismatch)
                }, sub);
            }
            else for (uint32_t j = 0; j <= K; ++j) sub[j] = 0.0f;
            children.resize(children.size() + W, 0u);
            uint32_t* rec = children.data() + static_cast<size_t>(numChildren) * W;
            std::memcpy(rec, sub, 4 * (K + 1));
            for (uint32_t j = 0; j < K; ++j) if (!((subMask >> j) & 1u)) rec[j] |= 0x80000000u;
            rec[K + 1] = subMask;
            ++numChildren;
        }
    }
    if (numRecords) *numRecords = h + numChildren;
    if (outSketch && capacityRecords >= h + numChildren) {
        std::memcpy(outSketch, rows.data(), 4 * rows.size());
        if (!children.empty()) std::memcpy(static_cast<uint32_t*>(outSketch) + rows.size(), children.data(), 4 * children.size());
    }
    return good;
}

void gfxh_env_make_sky(uint32_t w, uint32_t h, float sunElevationDeg, float sunAzimuthDeg, float sunRadiance, float* texels) {
    const float d2r = 3.14159265358979323846f / 180.0f;
    const float se = sunElevationDeg * d2r, sa = sunAzimuthDeg * d2r;
    const V3 sun = { -std::sin(sa) * std::cos(se), std::sin(se), std::cos(sa) * std::cos(se) };
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const float theta = 3.14159265358979323846f * (y + 0.5f) / h, phi = 2 * 3.14159265358979323846f * (x + 0.5f) / w;
            const V3 d = { -std::sin(phi) * std::sin(theta), std::cos(theta), std::cos(phi) * std::sin(theta) };   // fromPolarYUp
            const float up = std::max(d.y, 0.0f);
            float r = 0.25f + 0.5f * (1 - up), g = 0.35f + 0.45f * (1 - up), b = 0.7f + 0.2f * (1 - up);
            if (d.y < 0) { r = g = b = 0.05f; }
            const float c = dot(d, sun);
            if (c > 0.9995f) { r += sunRadiance; g += sunRadiance * 0.95f; b += sunRadiance * 0.85f; }
            else if (c > 0.99f) { const float k = (c - 0.99f) / 0.0095f; r += 4 * k; g += 3.6f * k; b += 3 * k; }
            float* t = texels + 4 * (static_cast<size_t>(y) * w + x);
            t[0] = r; t[1] = g; t[2] = b; t[3] = 1.0f;
        }
}

void gfxh_spatial_neighbor_deltas(float* out) {
    auto halton = [](uint32_t base, uint32_t idx) {
        const float recBase = 1.0f / base;
        float ret = 0.0f, scale = 1.0f;
        while (idx) { scale *= recBase; ret += (idx % base) * scale; idx /= base; }
        return ret;
    };
    for (uint32_t i = 0; i < 1024; ++i) {
        const float u0 = halton(2, i), u1 = halton(3, i);
        float dx = 0, dy = 0;
        const float sx = 2 * u0 - 1, sy = 2 * u1 - 1;
        if (!(sx == 0 && sy == 0)) {
            float r, theta;
            if (sx >= -sy) {
                if (sx > sy) { r = sx; theta = sy / sx; }
                else { r = sy; theta = 2 - sx / sy; }
            }
            else {
                if (sx > sy) { r = -sy; theta = 6 + sx / sy; }
                else { r = -sx; theta = 4 + sy / sx; }
            }
            theta *= 3.14159265358979323846f / 4;
            float s, c;
            host_sincos(theta, &s, &c);
            dx = r * c; dy = r * s;
        }
        out[2 * i] = dx; out[2 * i + 1] = dy;
    }
}

} // extern "C"

// ---------------------------------------------------------------- output chain
// saveImage(float4 -> 8-bit) of common/common_host.cpp:2859-2897: tone map on the luminance, sRGB gamma, quantise.
extern "C" void gfxh_tonemap_sdr(uint32_t width, uint32_t height, const float* rgba, const gfxh_sdr_config* cfg, uint32_t* out) {
    for (uint32_t y = 0; y < height; ++y) {
        const uint32_t sy = cfg->flipY ? (height - 1 - y) : y;
        for (uint32_t x = 0; x < width; ++x) {
            const float* src = rgba + 4 * (static_cast<size_t>(sy) * width + x);
            float r = src[0], g = src[1], b = src[2], a = src[3];
            if (cfg->alphaForOverride >= 0.0f) a = cfg->alphaForOverride;
            if (cfg->applyToneMap) {
                if (!(std::isfinite(r) && std::isfinite(g) && std::isfinite(b))) { r = 0.0f; g = 0.0f; b = 0.0f; }
                const float lum = 0.2126729f * r + 0.7151522f * g + 0.0721750f * b;      // sRGB_calcLuminance
                const float lumT = 1 - std::exp(-(cfg->brightnessScale * lum));           // simpleToneMap_s
                const float s = lum > 0.0f ? lumT / lum : 0.0f;
                r *= s; g *= s; b *= s;
            }
            if (cfg->apply_sRGB_gammaCorrection) {                                        // sRGB_gamma_s
                auto gamma = [](float v) { return v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 1 / 2.4f) - 0.055f; };
                r = gamma(r); g = gamma(g); b = gamma(b);
            }
            auto q = [](float v) { return v > 0.0f ? std::min<uint32_t>(static_cast<uint32_t>(std::min(v * 255, 4.0e9f)), 255u) : 0u; };
            out[static_cast<size_t>(y) * width + x] = q(r) | (q(g) << 8) | (q(b) << 16) | (q(a) << 24);
        }
    }
}

static bool has_ext(const char* path, const char* ext) {
    const size_t n = std::strlen(path), m = std::strlen(ext);
    return n >= m && std::strcmp(path + n - m, ext) == 0;
}

extern "C" int gfxh_save_image_sdr(const char* path, uint32_t width, uint32_t height, const float* rgba, const gfxh_sdr_config* cfg) {
    std::vector<uint32_t> px(static_cast<size_t>(width) * height);
    gfxh_tonemap_sdr(width, height, rgba, cfg, px.data());
    FILE* f = std::fopen(path, "wb");
    if (!f) { g_hostError = std::string("cannot open ") + path; return 1; }
    if (has_ext(path, ".ppm")) {
        std::fprintf(f, "P6\n%u %u\n255\n", width, height);
        for (uint32_t p : px) { const unsigned char c[3] = { static_cast<unsigned char>(p), static_cast<unsigned char>(p >> 8), static_cast<unsigned char>(p >> 16) }; std::fwrite(c, 1, 3, f); }
    }
    else if (has_ext(path, ".bmp")) {
        const uint32_t rowBytes = (3 * width + 3) & ~3u, dataBytes = rowBytes * height;
        unsigned char hdr[54] = { 'B', 'M' };
        auto put32 = [&](int o, uint32_t v) { hdr[o] = v & 255; hdr[o + 1] = (v >> 8) & 255; hdr[o + 2] = (v >> 16) & 255; hdr[o + 3] = (v >> 24) & 255; };
        put32(2, 54 + dataBytes); put32(10, 54); put32(14, 40); put32(18, width); put32(22, height);
        hdr[26] = 1; hdr[28] = 24; put32(34, dataBytes);
        std::fwrite(hdr, 1, 54, f);
        std::vector<unsigned char> row(rowBytes, 0);
        for (uint32_t y = 0; y < height; ++y) {                       // bottom-up, BGR
            const uint32_t* src = px.data() + static_cast<size_t>(height - 1 - y) * width;
            for (uint32_t x = 0; x < width; ++x) { row[3 * x] = (src[x] >> 16) & 255; row[3 * x + 1] = (src[x] >> 8) & 255; row[3 * x + 2] = src[x] & 255; }
            std::fwrite(row.data(), 1, rowBytes, f);
        }
    }
    else { std::fclose(f); g_hostError = "gfxh_save_image_sdr: .bmp or .ppm"; return 1; }
    std::fclose(f);
    return 0;
}

// fp32 -> fp16, round to nearest even (what tinyexr's float_to_half_full does for the requested HALF pixel type)
static uint16_t float_to_half(float v) {
    uint32_t x; std::memcpy(&x, &v, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t mag = x & 0x7FFFFFFFu;
    if (mag >= 0x7F800000u) return static_cast<uint16_t>(sign | 0x7C00u | (mag > 0x7F800000u ? 0x200u : 0u));   // inf / NaN
    if (mag >= 0x477FF000u) return static_cast<uint16_t>(sign | 0x7C00u);                                        // rounds to >= 65520 -> inf
    if (mag < 0x33000001u) return static_cast<uint16_t>(sign);                                                   // below half of the smallest subnormal
    const int32_t e = static_cast<int32_t>(mag >> 23) - 127;
    uint32_t m = (mag & 0x7FFFFFu) | 0x800000u;
    uint32_t half;
    uint32_t shift;
    if (e < -14) { shift = static_cast<uint32_t>(13 + (-14 - e)); half = 0; }        // subnormal half
    else { shift = 13; half = static_cast<uint32_t>(e + 15) << 10; m &= 0x7FFFFFu; }
    const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    half += q;
    if (rem > halfway || (rem == halfway && (half & 1u))) ++half;                      // carries into the exponent correctly
    return static_cast<uint16_t>(sign | half);
}

// saveImageHDR (common_host.cpp:2762-2857): OpenEXR scanline file, channels A B G R stored as HALF.  The reference goes
// through tinyexr (ZIP-compressed); this writer emits the same pixels uncompressed (compression = NO_COMPRESSION).
static int save_exr(const char* path, uint32_t width, uint32_t height, float brightnessScale, const float* rgba, int flipY) {
    FILE* f = std::fopen(path, "wb");
    if (!f) { g_hostError = std::string("cannot open ") + path; return 1; }
    std::vector<uint8_t> hdr;
    auto put = [&](const void* p, size_t n) { hdr.insert(hdr.end(), static_cast<const uint8_t*>(p), static_cast<const uint8_t*>(p) + n); };
    auto put_str = [&](const char* t) { put(t, std::strlen(t) + 1); };
    auto put_i32 = [&](int32_t v) { put(&v, 4); };
    auto put_f32 = [&](float v) { put(&v, 4); };
    const uint32_t magic = 20000630u, version = 2u;
    put(&magic, 4); put(&version, 4);
    put_str("channels"); put_str("chlist"); put_i32(4 * 18 + 1);
    for (const char* name : { "A", "B", "G", "R" }) { put_str(name); put_i32(1 /* HALF */); const uint8_t lin[4] = { 0, 0, 0, 0 }; put(lin, 4); put_i32(1); put_i32(1); }
    { const uint8_t z = 0; put(&z, 1); }
    put_str("compression"); put_str("compression"); put_i32(1); { const uint8_t c = 0; put(&c, 1); }
    put_str("dataWindow"); put_str("box2i"); put_i32(16); put_i32(0); put_i32(0); put_i32(static_cast<int32_t>(width) - 1); put_i32(static_cast<int32_t>(height) - 1);
    put_str("displayWindow"); put_str("box2i"); put_i32(16); put_i32(0); put_i32(0); put_i32(static_cast<int32_t>(width) - 1); put_i32(static_cast<int32_t>(height) - 1);
    put_str("lineOrder"); put_str("lineOrder"); put_i32(1); { const uint8_t c = 0; put(&c, 1); }
    put_str("pixelAspectRatio"); put_str("float"); put_i32(4); put_f32(1.0f);
    put_str("screenWindowCenter"); put_str("v2f"); put_i32(8); put_f32(0.0f); put_f32(0.0f);
    put_str("screenWindowWidth"); put_str("float"); put_i32(4); put_f32(1.0f);
    { const uint8_t z = 0; put(&z, 1); }
    std::fwrite(hdr.data(), 1, hdr.size(), f);
    const uint64_t rowBytes = 8ull + 4ull * 2ull * width;   // y, size, then A B G R planes of half
    uint64_t offset = hdr.size() + 8ull * height;
    for (uint32_t y = 0; y < height; ++y) { std::fwrite(&offset, 8, 1, f); offset += rowBytes; }
    std::vector<uint16_t> row(4ull * width);
    for (uint32_t y = 0; y < height; ++y) {
        const uint32_t sy = flipY ? (height - 1 - y) : y;
        const float* src = rgba + 4ull * static_cast<size_t>(sy) * width;
        for (uint32_t x = 0; x < width; ++x)
            for (int c = 0; c < 4; ++c) row[static_cast<size_t>(c) * width + x] = float_to_half(brightnessScale * src[4 * x + (3 - c)]);   // A, B, G, R planes
        const int32_t yy = static_cast<int32_t>(y), size = static_cast<int32_t>(8ull * width);
        std::fwrite(&yy, 4, 1, f); std::fwrite(&size, 4, 1, f);
        std::fwrite(row.data(), 2, row.size(), f);
    }
    std::fclose(f);
    return 0;
}

extern "C" int gfxh_save_image_hdr(const char* path, uint32_t width, uint32_t height, float brightnessScale, const float* rgba, int flipY) {
    if (has_ext(path, ".exr")) return save_exr(path, width, height, brightnessScale, rgba, flipY);
    if (!has_ext(path, ".pfm")) { g_hostError = "gfxh_save_image_hdr: .exr or .pfm"; return 1; }
    FILE* f = std::fopen(path, "wb");
    if (!f) { g_hostError = std::string("cannot open ") + path; return 1; }
    std::fprintf(f, "PF\n%u %u\n-1.0\n", width, height);              // little endian, rows bottom to top
    std::vector<float> row(3 * static_cast<size_t>(width));
    for (uint32_t y = 0; y < height; ++y) {
        const uint32_t top = height - 1 - y;                            // the image row this file row holds
        const uint32_t sy = flipY ? (height - 1 - top) : top;
        for (uint32_t x = 0; x < width; ++x)
            for (int c = 0; c < 3; ++c) row[3 * x + c] = brightnessScale * rgba[4 * (static_cast<size_t>(sy) * width + x) + c];
        std::fwrite(row.data(), sizeof(float), row.size(), f);
    }
    std::fclose(f);
    return 0;
}
