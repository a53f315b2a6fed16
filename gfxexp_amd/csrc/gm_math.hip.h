// gm_math.hip.h -- device-side fp32 vector math and the deterministic transcendental contract
// ("gm" = gfx math, contract version 1).
//
// The reference evaluates sincosf / acos / atan2 / tan of the CUDA math library inside its kernels
// (common/common_device.cuh:14-25).  Those are not bit-specified, and reservoir acceptance
// (restir_di/restir_di_shared.h:118-125) and the 16-bit polar quantisers
// (common/common_device.cuh:27-79) are discontinuous in their inputs, so this build fixes ONE
// algorithm: Cody-Waite reduction by pi/2 in three fmaf steps + cephes-style minimax kernels,
// every multiply-add spelled as fmaf, everything else plain IEEE fp32 with contraction off
// (-ffp-contract=off) and correctly rounded division / sqrt.  Same inputs -> same bits on any
// IEEE machine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gfx {

#define GFX_DEV __device__ __forceinline__

// GFX_LANE_PROFILE (an experiment build: `python gfxexp_amd/build.py --variant laneprof GFX_LANE_PROFILE`, tools/lane_profile.py):
// per code section, how often a wave enters it and with how many lanes.  g_laneProfile[2 k] += active lanes, [2 k + 1] += 1.
// Expands to nothing in the product build.
#ifdef GFX_LANE_PROFILE
static __device__ unsigned long long g_laneProfile[64];
#define GFX_PROF(k) do { const unsigned long long m__ = __ballot(1); if ((threadIdx.x & 63) == __builtin_ctzll(m__)) { \
    atomicAdd(&g_laneProfile[2 * (k)], static_cast<unsigned long long>(__popcll(m__))); atomicAdd(&g_laneProfile[2 * (k) + 1], 1ull); } } while (0)
// ... and where a wave's clock cycles go: GFX_CYC(k) closes the running section and opens section k (s_memtime; wave-level, so a
// section entered by any lane is charged to the whole wave); GFX_CYC_BEGIN / GFX_CYC_END bracket a kernel.  Totals in
// g_laneProfile[32 + k].
#define GFX_CYC_BEGIN unsigned long long cyc__[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; unsigned long long cycT__ = __builtin_amdgcn_s_memtime(); int cycK__ = 7;
#define GFX_CYC(k) do { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); cyc__[cycK__] += t__ - cycT__; cycT__ = t__; cycK__ = (k); } while (0)
#define GFX_CYC_END do { GFX_CYC(7); if ((threadIdx.x & 63) == 0) for (int k__ = 0; k__ < 8; ++k__) atomicAdd(&g_laneProfile[32 + k__], cyc__[k__]); } while (0)
#else
#define GFX_PROF(k) do { } while (0)
#define GFX_CYC_BEGIN
#define GFX_CYC(k) do { } while (0)
#define GFX_CYC_END do { } while (0)
#endif

constexpr float kPi = 3.14159265358979323846f;
constexpr float kTwoPi = 2 * 3.14159265358979323846f;
constexpr float kHalfPi = 1.5707963705062866f;
constexpr float kQuarterPi = 0.7853981852531433f;

GFX_DEV uint32_t f2bits(float f) { return __float_as_uint(f); }
GFX_DEV float bits2f(uint32_t u) { return __uint_as_float(u); }
GFX_DEV float fmin2(float a, float b) { return a < b ? a : b; }
GFX_DEV float fmax2(float a, float b) { return a > b ? a : b; }
GFX_DEV float sq(float x) { return x * x; }
GFX_DEV float pow5(float x) { const float x2 = x * x; return x * (x2 * x2); }
GFX_DEV float mixf(float v0, float v1, float t) { return (1 - t) * v0 + t * v1; }
GFX_DEV bool is_finite(float x) { return (f2bits(x) & 0x7F800000u) != 0x7F800000u; }

// Saturating conversions: defined for every input so host and device agree.
GFX_DEV uint32_t f2u_sat(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return static_cast<uint32_t>(x);
}
GFX_DEV int32_t f2i_sat(float x) {
    if (!(x == x)) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return -2147483647 - 1;
    return static_cast<int32_t>(x);
}

GFX_DEV void gm_sincos(float x, float& s, float& c) {
    const float q = rintf(x * 0.6366197466850281f);
    float r = fmaf(q, -1.5703125f, x);
    r = fmaf(q, -0.0004837512969970703f, r);
    r = fmaf(q, -7.549790126404332e-08f, r);
    const int32_t n = f2i_sat(q);
    const float r2 = r * r;
    float ps = fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f);
    ps = fmaf(ps, r2, -1.6666654611e-1f);
    const float sr = fmaf(ps * r2, r, r);
    float pc = fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f);
    pc = fmaf(pc, r2, 4.166664568298827e-2f);
    const float cr = fmaf(pc * r2, r2, fmaf(-0.5f, r2, 1.0f));
    const float ss = (n & 1) ? cr : sr;
    const float cc = (n & 1) ? sr : cr;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}
GFX_DEV float gm_sin(float x) { float s, c; gm_sincos(x, s, c); return s; }
GFX_DEV float gm_cos(float x) { float s, c; gm_sincos(x, s, c); return c; }
GFX_DEV float gm_tan(float x) { float s, c; gm_sincos(x, s, c); return s / c; }

// exp(x), |x| < 80: Cody-Waite ln2 split + degree-6 polynomial + exponent insertion
GFX_DEV float gm_exp(float x) {
    const float n = floorf(x * 1.44269504088896341f + 0.5f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float y = fmaf(p, r * r, r) + 1.0f;
    return y * bits2f(static_cast<uint32_t>(static_cast<int32_t>(n) + 127) << 23);
}

GFX_DEV float gm_asin_core(float x) {
    const float z = x * x;
    float p = fmaf(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = fmaf(p, z, 4.5470025998e-2f);
    p = fmaf(p, z, 7.4953002686e-2f);
    p = fmaf(p, z, 1.6666752422e-1f);
    return fmaf(p * z, x, x);
}
GFX_DEV float gm_acos(float x) {
    if (x > 0.5f) return 2.0f * gm_asin_core(sqrtf(0.5f * (1.0f - x)));
    if (x < -0.5f) return kPi - 2.0f * gm_asin_core(sqrtf(0.5f * (1.0f + x)));
    return kHalfPi - gm_asin_core(x);
}
GFX_DEV float gm_atan_nonneg(float t) {
    float y0 = 0.0f;
    if (t > 2.414213562373095f) { y0 = kHalfPi; t = -1.0f / t; }
    else if (t > 0.4142135623730950f) { y0 = kQuarterPi; t = (t - 1.0f) / (t + 1.0f); }
    const float z = t * t;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    return y0 + fmaf(p * z, t, t);
}
GFX_DEV float gm_atan(float t) { return t < 0.0f ? -gm_atan_nonneg(-t) : gm_atan_nonneg(t); }
GFX_DEV float gm_atan2(float y, float x) {
    if (x == 0.0f && y == 0.0f) return 0.0f;
    const float ax = fabsf(x), ay = fabsf(y);
    float a = (ax == 0.0f) ? kHalfPi : gm_atan_nonneg(ay / ax);
    if (x < 0.0f) a = kPi - a;
    return y < 0.0f ? -a : a;
}

// ---------------------------------------------------------------- float3 with the reference's
// operator semantics: v / s multiplies by the reciprocal (common/basic_types.h:2564-2570).
struct f3 {
    float x, y, z;
    GFX_DEV f3() {}
    GFX_DEV explicit f3(float v) : x(v), y(v), z(v) {}
    GFX_DEV f3(float a, float b, float c) : x(a), y(b), z(c) {}
};
GFX_DEV f3 operator+(f3 a, f3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
GFX_DEV f3 operator-(f3 a, f3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
GFX_DEV f3 operator-(f3 a) { return f3(-a.x, -a.y, -a.z); }
GFX_DEV f3 operator*(f3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
GFX_DEV f3 operator*(float s, f3 a) { return f3(a.x * s, a.y * s, a.z * s); }
GFX_DEV f3 operator*(f3 a, f3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
GFX_DEV f3 operator/(f3 a, float s) { const float r = 1 / s; return f3(a.x * r, a.y * r, a.z * r); }
GFX_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
GFX_DEV f3 cross(f3 a, f3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
GFX_DEV float len2(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
GFX_DEV float len(f3 a) { return sqrtf(len2(a)); }
GFX_DEV f3 unit(f3 a) { return a / len(a); }
GFX_DEV bool all_finite(f3 a) { return is_finite(a.x) && is_finite(a.y) && is_finite(a.z); }
GFX_DEV f3 mix3(f3 v0, f3 v1, float t) { return (1 - t) * v0 + t * v1; }
GFX_DEV f3 min3(f3 a, f3 b) { return f3(fmin2(a.x, b.x), fmin2(a.y, b.y), fmin2(a.z, b.z)); }
GFX_DEV f3 max3(f3 a, f3 b) { return f3(fmax2(a.x, b.x), fmax2(a.y, b.y), fmax2(a.z, b.z)); }
GFX_DEV float luminance_srgb(f3 c) { return 0.2126729f * c.x + 0.7151522f * c.y + 0.0721750f * c.z; }

// row-major 3x3 and 3x4 (rows of an affine Matrix4x4; w row implied)
struct m33 { f3 r0, r1, r2; };
GFX_DEV f3 mul(const m33& m, f3 v) { return f3(dot(m.r0, v), dot(m.r1, v), dot(m.r2, v)); }
GFX_DEV m33 inverse(const m33& m) { // Matrix3x3::invert, common/basic_types.h:4150-4157
    const float a = m.r0.x, b = m.r0.y, c = m.r0.z, d = m.r1.x, e = m.r1.y, f = m.r1.z, g = m.r2.x, h = m.r2.y, i = m.r2.z;
    const float det = a * e * i + b * f * g + c * d * h - c * e * g - b * d * i - a * f * h;
    const float r = 1 / det;
    m33 o;
    o.r0 = f3((e * i - f * h), -(b * i - c * h), (b * f - c * e)) * r;
    o.r1 = f3(-(d * i - f * g), (a * i - c * g), -(a * f - c * d)) * r;
    o.r2 = f3((d * h - e * g), -(a * h - b * g), (a * e - b * d)) * r;
    return o;
}
struct m34 { float m[12]; };
GFX_DEV f3 xfm_point(const m34& a, f3 p) {
    return f3(a.m[0] * p.x + a.m[1] * p.y + a.m[2] * p.z + a.m[3] * 1.0f,
              a.m[4] * p.x + a.m[5] * p.y + a.m[6] * p.z + a.m[7] * 1.0f,
              a.m[8] * p.x + a.m[9] * p.y + a.m[10] * p.z + a.m[11] * 1.0f);
}
GFX_DEV f3 xfm_vector(const m34& a, f3 v) {
    return f3(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z + a.m[3] * 0.0f,
              a.m[4] * v.x + a.m[5] * v.y + a.m[6] * v.z + a.m[7] * 0.0f,
              a.m[8] * v.x + a.m[9] * v.y + a.m[10] * v.z + a.m[11] * 0.0f);
}

} // namespace gfx
