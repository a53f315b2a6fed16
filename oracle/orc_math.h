// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into or called from the product path
// (gfxexp_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// orc_math.h: scalar fp32 math used by the CPU restatement of the GfxExp hot path.
//  * vector/point/RGB helpers restate common/basic_types.h (reference) operation by operation:
//      v / s  ==  v * (1 / s)            basic_types.h:2564-2570, 5203-5209
//      dot    ==  a.x*b.x + a.y*b.y + a.z*b.z   basic_types.h:2745-2748
//      cross                                    basic_types.h:2751-2757
//      lerp   ==  (1 - t) * v0 + t * v1          basic_types.h:261-263
//  * "gm_" transcendental functions are the build's DETERMINISTIC MATH CONTRACT (SURVEY.md
//    section 7 step 0): the reference calls sincosf/acos/atan2/tan of the CUDA math library
//    (common/common_device.cuh:14-25), whose results are neither specified bit-for-bit nor
//    reproducible on another vendor.  The contract fixes one algorithm (Cody-Waite reduction +
//    cephes-style minimax kernels, explicit fmaf, no contraction); the HIP kernels implement the
//    same algorithm independently in gfxexp_amd/csrc/gm_math.hip.h.
//  Compile with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

static inline uint32_t f2bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float bits2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

constexpr float kPi = 3.14159265358979323846f;      // pi_v<float>
constexpr float kTwoPi = 2 * 3.14159265358979323846f;

static inline float fmin2(float a, float b) { return a < b ? a : b; }   // min(a,b) of CUDA for non-NaN
static inline float fmax2(float a, float b) { return a > b ? a : b; }
static inline float pow2(float x) { return x * x; }
static inline float pow4(float x) { return pow2(pow2(x)); }
static inline float pow5(float x) { return x * pow4(x); }                 // basic_types.h:256-258
static inline float lerpf(float v0, float v1, float t) { return (1 - t) * v0 + t * v1; }
static inline bool finitef(float x) { return (f2bits(x) & 0x7F800000u) != 0x7F800000u; }

// saturating float -> integer conversions (the reference relies on in-range values;
// the contract makes the out-of-range behaviour explicit so CPU and GPU agree).
static inline uint32_t f2u(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return static_cast<uint32_t>(x);
}
static inline int32_t f2i(float x) {
    if (!(x == x)) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return -2147483647 - 1;
    return static_cast<int32_t>(x);
}

// ---------------------------------------------------------------- deterministic transcendental
static inline void gm_sincos(float x, float* s, float* c) {
    // q = nearest integer to x * 2/pi; r = x - q*pi/2 in three Cody-Waite steps.
    const float q = std::rint(x * 0.6366197466850281f);
    float r = std::fma(q, -1.5703125f, x);
    r = std::fma(q, -0.0004837512969970703f, r);
    r = std::fma(q, -7.549790126404332e-08f, r);
    const int32_t n = f2i(q);
    const float r2 = r * r;
    // sin kernel on [-pi/4, pi/4]
    float ps = std::fma(-1.9515295891e-4f, r2, 8.3321608736e-3f);
    ps = std::fma(ps, r2, -1.6666654611e-1f);
    const float sr = std::fma(ps * r2, r, r);
    // cos kernel
    float pc = std::fma(2.443315711809948e-5f, r2, -1.388731625493765e-3f);
    pc = std::fma(pc, r2, 4.166664568298827e-2f);
    const float cr = std::fma(pc * r2, r2, std::fma(-0.5f, r2, 1.0f));
    const float ss = (n & 1) ? cr : sr;
    const float cc = (n & 1) ? sr : cr;
    *s = (n & 2) ? -ss : ss;
    *c = ((n + 1) & 2) ? -cc : cc;
}
static inline float gm_sin(float x) { float s, c; gm_sincos(x, &s, &c); return s; }
static inline float gm_cos(float x) { float s, c; gm_sincos(x, &s, &c); return c; }
static inline float gm_tan(float x) { float s, c; gm_sincos(x, &s, &c); return s / c; }

// exp(x) for |x| < 80: n = round(x / ln2), Cody-Waite r = x - n ln2 (hi + lo), degree-6 kernel, 2^n by bits
static inline float gm_exp(float x) {
    const float n = std::floor(x * 1.44269504088896341f + 0.5f);
    float r = std::fma(n, -0.693359375f, x);
    r = std::fma(n, 2.12194440e-4f, r);
    float p = std::fma(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = std::fma(p, r, 8.3334519073e-3f);
    p = std::fma(p, r, 4.1665795894e-2f);
    p = std::fma(p, r, 1.6666665459e-1f);
    p = std::fma(p, r, 5.0000001201e-1f);
    const float y = std::fma(p, r * r, r) + 1.0f;
    const int32_t e = static_cast<int32_t>(n);
    return y * bits2f(static_cast<uint32_t>(e + 127) << 23);
}

static inline float gm_asin_kernel(float x) { // |x| <= 0.5
    const float z = x * x;
    float p = std::fma(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = std::fma(p, z, 4.5470025998e-2f);
    p = std::fma(p, z, 7.4953002686e-2f);
    p = std::fma(p, z, 1.6666752422e-1f);
    return std::fma(p * z, x, x);
}
static inline float gm_acos(float x) { // x in [-1, 1]
    if (x > 0.5f) {
        const float t = std::sqrt(0.5f * (1.0f - x));
        return 2.0f * gm_asin_kernel(t);
    }
    if (x < -0.5f) {
        const float t = std::sqrt(0.5f * (1.0f + x));
        return kPi - 2.0f * gm_asin_kernel(t);
    }
    return 1.5707963705062866f - gm_asin_kernel(x);
}
static inline float gm_atan_pos(float t) { // t >= 0
    float y0 = 0.0f;
    if (t > 2.414213562373095f) { y0 = 1.5707963705062866f; t = -1.0f / t; }
    else if (t > 0.4142135623730950f) { y0 = 0.7853981852531433f; t = (t - 1.0f) / (t + 1.0f); }
    const float z = t * t;
    float p = std::fma(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = std::fma(p, z, 1.99777106478e-1f);
    p = std::fma(p, z, -3.33329491539e-1f);
    return y0 + std::fma(p * z, t, t);
}
static inline float gm_atan(float t) { return t < 0.0f ? -gm_atan_pos(-t) : gm_atan_pos(t); }
static inline float gm_atan2(float y, float x) {
    if (x == 0.0f && y == 0.0f) return 0.0f;
    const float ax = std::fabs(x), ay = std::fabs(y);
    float a;                              // angle of (|x|, |y|) in [0, pi/2]
    if (ax == 0.0f) a = 1.5707963705062866f;
    else a = gm_atan_pos(ay / ax);
    if (x < 0.0f) a = kPi - a;
    return y < 0.0f ? -a : a;
}

// ---------------------------------------------------------------- vectors
struct V2 { float x, y; };
struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float v) : x(v), y(v), z(v) {}
    V3(float xx, float yy, float zz) : x(xx), y(yy), z(zz) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
static inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(float s, V3 a) { return V3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator/(V3 a, float s) { const float r = 1 / s; return V3(a.x * r, a.y * r, a.z * r); }
static inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
static inline V3& operator*=(V3& a, float s) { a = a * s; return a; }
static inline V3& operator*=(V3& a, V3 b) { a = a * b; return a; }
static inline V3& operator/=(V3& a, float s) { a = a / s; return a; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float sqLength(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static inline float length(V3 a) { return std::sqrt(sqLength(a)); }
static inline V3 normalize(V3 a) { return a / length(a); }
static inline bool allFinite(V3 a) { return finitef(a.x) && finitef(a.y) && finitef(a.z); }
static inline V3 vmin(V3 a, V3 b) { return V3(fmin2(a.x, b.x), fmin2(a.y, b.y), fmin2(a.z, b.z)); }
static inline V3 vmax(V3 a, V3 b) { return V3(fmax2(a.x, b.x), fmax2(a.y, b.y), fmax2(a.z, b.z)); }
static inline V3 lerp3(V3 v0, V3 v1, float t) { return (1 - t) * v0 + t * v1; }
using RGB = V3; // r=x, g=y, b=z

// 3x3 row-major; rows r0,r1,r2.  Matrix3x3 * v = (dot(row0,v), ...)  basic_types.h:4263-4270
struct M3 {
    V3 r0, r1, r2;
};
static inline V3 mul(const M3& m, V3 v) { return V3(dot(m.r0, v), dot(m.r1, v), dot(m.r2, v)); }
// Matrix3x3::invert: det by the rule of Sarrus, adjugate, m /= det (scalar divide == *1/det)
// basic_types.h:4118-4157, 4087
static inline M3 invert(const M3& m) {
    const float m00 = m.r0.x, m01 = m.r0.y, m02 = m.r0.z;
    const float m10 = m.r1.x, m11 = m.r1.y, m12 = m.r1.z;
    const float m20 = m.r2.x, m21 = m.r2.y, m22 = m.r2.z;
    const float det = m00 * m11 * m22 + m01 * m12 * m20 + m02 * m10 * m21
        - m02 * m11 * m20 - m01 * m10 * m22 - m00 * m12 * m21;
    M3 a;
    a.r0 = V3((m11 * m22 - m12 * m21), -(m01 * m22 - m02 * m21), (m01 * m12 - m02 * m11));
    a.r1 = V3(-(m10 * m22 - m12 * m20), (m00 * m22 - m02 * m20), -(m00 * m12 - m02 * m10));
    a.r2 = V3((m10 * m21 - m11 * m20), -(m00 * m21 - m01 * m20), (m00 * m11 - m01 * m10));
    const float r = 1 / det;
    a.r0 = a.r0 * r; a.r1 = a.r1 * r; a.r2 = a.r2 * r;
    return a;
}
static inline M3 transpose(const M3& m) {
    M3 t;
    t.r0 = V3(m.r0.x, m.r1.x, m.r2.x);
    t.r1 = V3(m.r0.y, m.r1.y, m.r2.y);
    t.r2 = V3(m.r0.z, m.r1.z, m.r2.z);
    return t;
}
// 3x4 affine, row-major rows (x y z w).  Matrix4x4 * Point3D = dot4(row, (p,1)),
// Matrix4x4 * Vector3D = dot4(row, (v,0))     basic_types.h:4746-4766
struct M34 {
    float m[12];
};
static inline V3 xfmPoint(const M34& a, V3 p) {
    return V3(a.m[0] * p.x + a.m[1] * p.y + a.m[2] * p.z + a.m[3] * 1.0f,
              a.m[4] * p.x + a.m[5] * p.y + a.m[6] * p.z + a.m[7] * 1.0f,
              a.m[8] * p.x + a.m[9] * p.y + a.m[10] * p.z + a.m[11] * 1.0f);
}
static inline V3 xfmVector(const M34& a, V3 v) {
    return V3(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z + a.m[3] * 0.0f,
              a.m[4] * v.x + a.m[5] * v.y + a.m[6] * v.z + a.m[7] * 0.0f,
              a.m[8] * v.x + a.m[9] * v.y + a.m[10] * v.z + a.m[11] * 0.0f);
}
// prevTransform * invert(matM2W) of InstanceController::update (common/common_host.h:851), rows 0..2.
// Matrix4x4::invert (basic_types.h:4597-4626): element (row i, column j) of the inverse is the signed 3x3
// minor of the matrix without row j and column i, expanded in the reference's term order
//   m[r0][c0] m[r1][c1] m[r2][c2] - m[r2][c0] m[r1][c1] m[r0][c2] + m[r1][c0] m[r2][c1] m[r0][c2]
// - m[r0][c0] m[r2][c1] m[r1][c2] + m[r2][c0] m[r0][c1] m[r1][c2] - m[r1][c0] m[r0][c1] m[r2][c2],
// all sixteen scaled by recDet = 1 / (m00 inv00 + m10 inv01 + m20 inv02 + m30 inv03);
// the product is Matrix4x4::operator*= (:4552-4559): dot4(row of the left, column of the right).
static inline void invert44(const float m[4][4], float inv[4][4]) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            int r[3], c[3], nr = 0, nc = 0;
            for (int k = 0; k < 4; ++k) { if (k != j) r[nr++] = k; if (k != i) c[nc++] = k; }
            const float v = (m[r[0]][c[0]] * m[r[1]][c[1]] * m[r[2]][c[2]]) - (m[r[2]][c[0]] * m[r[1]][c[1]] * m[r[0]][c[2]]) +
                            (m[r[1]][c[0]] * m[r[2]][c[1]] * m[r[0]][c[2]]) - (m[r[0]][c[0]] * m[r[2]][c[1]] * m[r[1]][c[2]]) +
                            (m[r[2]][c[0]] * m[r[0]][c[1]] * m[r[1]][c[2]]) - (m[r[1]][c[0]] * m[r[0]][c[1]] * m[r[2]][c[2]]);
            inv[i][j] = ((i + j) & 1) ? -v : v;
        }
    const float recDet = 1.0f / (m[0][0] * inv[0][0] + m[1][0] * inv[0][1] + m[2][0] * inv[0][2] + m[3][0] * inv[0][3]);
    for (int j = 0; j < 4; ++j)          // the reference scales its column-major array in order
        for (int i = 0; i < 4; ++i) inv[i][j] *= recDet;
}
static inline void toM44(const float a[12], float m[4][4]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) m[i][j] = a[4 * i + j];
    m[3][0] = 0.0f; m[3][1] = 0.0f; m[3][2] = 0.0f; m[3][3] = 1.0f;
}
static inline M34 curToPrev(const M34& prev, const M34& cur) {
    float p[4][4], c[4][4], inv[4][4];
    toM44(prev.m, p); toM44(cur.m, c);
    invert44(c, inv);
    M34 out;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j)
            out.m[4 * i + j] = p[i][0] * inv[0][j] + p[i][1] * inv[1][j] + p[i][2] * inv[2][j] + p[i][3] * inv[3][j];
    return out;
}
static inline M3 upperLeft(const M34& a) {
    M3 m;
    m.r0 = V3(a.m[0], a.m[1], a.m[2]);
    m.r1 = V3(a.m[4], a.m[5], a.m[6]);
    m.r2 = V3(a.m[8], a.m[9], a.m[10]);
    return m;
}

static inline float sRGB_calcLuminance(RGB v) { // basic_types.h:5420-5423
    return 0.2126729f * v.x + 0.7151522f * v.y + 0.0721750f * v.z;
}

} // namespace orc
