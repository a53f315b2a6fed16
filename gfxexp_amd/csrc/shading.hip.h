// shading.hip.h -- device shading library of the ReSTIR / path-tracing kernels.
//
// What each block computes and where the reference defines it:
//   Pcg32                     common/common_shared.h:116-138 (PCG32RNG; LCG increment is literally 1)
//   discrete_sample           common/common_shared.h:209-247 (DiscreteDistribution1D::sample)
//   polar quantisers          common/common_device.cuh:14-79
//   offset_ray_origin         common/common_device.cuh:112-140
//   Frame                     common/common_device.cuh:149-174 (ReferenceFrame)
//   Bsdf                      common/common_device.cuh:335-374 (Lambert), :443-765 (DiffuseAndSpecular),
//                             :767-776 (SimplePBR), setup :376-385,778-826
//   sample_light              restir_di/restir_di_shared.h:320-516 (sampleLight<false>)
//   direct_lighting           restir_di/restir_di_shared.h:518-557 (performDirectLighting, unshadowed part)
// Arithmetic follows the reference operation by operation (fp32, no contraction); rng() calls
// passed as two function arguments in the reference are drawn left to right here.
#pragma once
#include "device_types.h"
#include "gm_math.hip.h"
#include "texture.hip.h"

namespace gfx {

struct Pcg32 {
    uint64_t state;
    GFX_DEV uint32_t next() {
        const uint64_t old = state;
        state = old * 6364136223846793005ULL + 1;
        const uint32_t xs = static_cast<uint32_t>(((old >> 18u) ^ old) >> 27u);
        const uint32_t rot = static_cast<uint32_t>(old >> 59u);
        return (xs >> rot) | (xs << ((0u - rot) & 31u));
    }
    GFX_DEV float uniform() { return bits2f((next() >> 9) | 0x3f800000u) - 1.0f; }
};

GFX_DEV uint32_t next_pow2(uint32_t x) { return x <= 1 ? x : 1u << (32 - __clz(x - 1)); }

// Branch-light binary search over an exclusive-prefix CDF.  probs[i] = weight[i] / integral, the quotient
// the reference forms per call, is tabulated by the light-distribution build (lights.hip); prob may be null.
GFX_DEV uint32_t discrete_sample(const float* __restrict__ probs, const float* __restrict__ cdf,
                                 float integral, uint32_t n, float u, float* prob, float* remapped) {
    u *= integral;
    int idx = 0;
    for (int d = static_cast<int>(next_pow2(n) >> 1); d >= 1; d >>= 1) {
        if (idx + d >= static_cast<int>(n)) continue;
        if (cdf[idx + d] <= u) idx += d;
    }
    if (remapped) {
        const float lo = cdf[idx];
        float hi = integral;
        if (idx < static_cast<int>(n) - 1) hi = cdf[idx + 1];
        *remapped = (u - lo) / (hi - lo);
    }
    if (prob) *prob = probs[idx];
    return static_cast<uint32_t>(idx);
}

// Instance-level (level 0) distribution plus its guide table (lights.hip, k_inst_guide).  guide == nullptr
// selects the plain search.
struct InstDist {
    const float* probs;
    const float* cdf;
    const uint16_t* guide;
    float guideScale;
    uint32_t guideCells;
};

GFX_DEV uint32_t guide_cell(float x, float scale, uint32_t cells) {
    return min(cells - 1u, static_cast<uint32_t>(x * scale));
}

// Same result as discrete_sample (largest idx with cdf[idx] <= u) for a monotone CDF, found inside the
// bracket the guide table gives instead of by log2(n) dependent loads.
GFX_DEV uint32_t discrete_sample_guided(const InstDist& d, float integral, uint32_t n, float u, float& prob, float* remapped) {
    if (!d.guide) return discrete_sample(d.probs, d.cdf, integral, n, u, &prob, remapped);
    u *= integral;
    const uint32_t k = guide_cell(u, d.guideScale, d.guideCells);
    int hi = d.guide[k];
    int lo = k ? d.guide[k - 1] : 0;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (d.cdf[mid] <= u) lo = mid;
        else hi = mid - 1;
    }
    const int idx = lo;
    if (remapped) {
        const float l = d.cdf[idx];
        float h = integral;
        if (idx < static_cast<int>(n) - 1) h = d.cdf[idx + 1];
        *remapped = (u - l) / (h - l);
    }
    prob = d.probs[idx];
    return static_cast<uint32_t>(idx);
}

// ---------------------------------------------------------------- quantisers
GFX_DEV f3 from_polar_yup(float phi, float theta) {
    float sp, cp, st, ct;
    gm_sincos(phi, sp, cp);
    gm_sincos(theta, st, ct);
    return f3(-sp * st, ct, cp * st);
}
GFX_DEV void to_polar_yup(f3 v, float& phi, float& theta) {
    theta = gm_acos(fmin2(fmax2(v.y, -1.0f), 1.0f));
    const float a = gm_atan2(-v.x, v.z) + kTwoPi;   // in [pi, 3pi]: fmod(a, 2pi) is one exact subtraction
    phi = a >= kTwoPi ? a - kTwoPi : a;
}
GFX_DEV uint32_t q16(float x01) { const uint32_t q = f2u_sat(x01 * 65535u); return q > 65535u ? 65535u : q; }
GFX_DEV uint32_t encode_dir(f3 v) {
    float phi, theta;
    to_polar_yup(v, phi, theta);
    return (q16(theta / kPi) << 16) | q16(phi / kTwoPi);
}
GFX_DEV f3 decode_dir(uint32_t q) {
    const float phi = kTwoPi * ((q & 0xFFFF) / 65535.0f);
    const float theta = kPi * ((q >> 16) / 65535.0f);
    return from_polar_yup(phi, theta);
}
GFX_DEV uint32_t encode_bc(float bc) { return q16(bc); }
GFX_DEV float decode_bc(uint32_t q) { return q / 65535.0f; }
GFX_DEV uint32_t encode_uv(float u, float v) { return (q16(v - floorf(v)) << 16) | q16(u - floorf(u)); }

GFX_DEV f3 offset_ray_origin(f3 p, f3 ng) {
    constexpr float kOrigin = 1.0f / 32.0f, kFloatScale = 1.0f / 65536.0f, kIntScale = 256.0f;
    const int32_t ox = f2i_sat(kIntScale * ng.x), oy = f2i_sat(kIntScale * ng.y), oz = f2i_sat(kIntScale * ng.z);
    const f3 pi(__int_as_float(__float_as_int(p.x) + (p.x < 0 ? -1 : 1) * ox),
                __int_as_float(__float_as_int(p.y) + (p.y < 0 ? -1 : 1) * oy),
                __int_as_float(__float_as_int(p.z) + (p.z < 0 ? -1 : 1) * oz));
    const f3 pf = p + kFloatScale * ng;
    return f3(fabsf(p.x) < kOrigin ? pf.x : pi.x, fabsf(p.y) < kOrigin ? pf.y : pi.y, fabsf(p.z) < kOrigin ? pf.z : pi.z);
}

GFX_DEV void make_coordinate_system(f3 normal, f3& tangent, f3& bitangent) { // common_shared.h:92-100
    const float sign = normal.z >= 0 ? 1.0f : -1.0f;
    const float a = -1 / (sign + normal.z);
    const float b = normal.x * normal.y * a;
    tangent = f3(1 + sign * normal.x * normal.x * a, sign * b, -sign * normal.x);
    bitangent = f3(b, sign + normal.y * normal.y * a, -normal.y);
}

struct Frame {
    f3 t, b, n;
    GFX_DEV Frame() {}
    GFX_DEV Frame(f3 normal, f3 tangent) : t(tangent), n(normal) { b = cross(n, t); }
    GFX_DEV f3 to_local(f3 v) const { return f3(dot(t, v), dot(b, v), dot(n, v)); }
    GFX_DEV f3 from_local(f3 v) const {
        return f3(dot(f3(t.x, b.x, n.x), v), dot(f3(t.y, b.y, n.y), v), dot(f3(t.z, b.z, n.z), v));
    }
};

GFX_DEV void decode_uv(uint32_t q, float& u, float& v) {   // decodeTexCoords, common_device.cuh:72-78
    u = (q & 0xFFFFu) / 65535.0f;
    v = (q >> 16) / 65535.0f;
}

// mat.emittance read: RGB of tex2DLod(mat.emittance, texCoord, 0) (optix_restir_di_kernels.cu:595-599)
GFX_DEV f3 material_emittance(const DevScene& sc, const gfx_material& m, float u, float v) {
    if (!m.hasEmittance) return f3(0.0f);
    if (m.texEmittance) { const float4 t = tex2d(sc, m.texEmittance, u, v); return f3(t.x, t.y, t.z); }
    return f3(m.emittance[0], m.emittance[1], m.emittance[2]);
}

// readModifiedNormalFromNormalMap / ...2ch / ...FromHeightMap (common/common_device.cuh:205-240).
// No normal texture = the reference's 1x1 (0.5, 0.5, 1) texture (common_host.cpp:1399-1403) read as a normal map.
GFX_DEV f3 read_modified_normal(const DevScene& sc, const gfx_material& m, float u, float v) {
    const uint32_t kind = m.bumpMapType & 0xFFu;
    f3 n;
    if (!m.texNormal) n = 2.0f * f3(0.5f, 0.5f, 1.0f) - f3(1.0f);
    else if (kind == GFX_BUMP_NORMAL_MAP) {
        const float4 t = tex2d(sc, m.texNormal, u, v);
        n = 2.0f * f3(t.x, t.y, t.z) - f3(1.0f);
    }
    else if (kind == GFX_BUMP_NORMAL_MAP_2CH) {
        const float4 t = tex2d(sc, m.texNormal, u, v);
        const float x = 2.0f * t.x - 1.0f, y = 2.0f * t.y - 1.0f;
        n = f3(x, y, sqrtf(1.0f - sq(x) - sq(y)));
    }
    else {
        const float4 h = tex2d_gather_r(sc, m.texNormal, u, v);
        const DevTexture t = sc.textures[m.texNormal];
        constexpr float coeff = 5.0f / 1024;
        const float dhdu = (coeff * t.width) * (h.y - h.x);
        const float dhdv = (coeff * t.height) * (h.x - h.w);
        return unit(f3(-dhdu, dhdv, 1));     // height maps ignore isLeftHanded in the reference
    }
    if (m.bumpMapType & GFX_BUMP_LEFT_HANDED) n.y *= -1;
    return n;
}

// applyBumpMapping, common/common_device.cuh:176-203
GFX_DEV void apply_bump_mapping(f3 modNormalInTF, Frame& frame) {
    const float projLength = sqrtf(modNormalInTF.x * modNormalInTF.x + modNormalInTF.y * modNormalInTF.y);
    if (projLength < 1e-3f) return;
    const float tiltAngle = gm_atan(projLength / modNormalInTF.z);
    float qSin, qCos;
    gm_sincos(tiltAngle / 2, qSin, qCos);
    const float qX = (-modNormalInTF.y / projLength) * qSin;
    const float qY = (modNormalInTF.x / projLength) * qSin;
    const float qW = qCos;
    const f3 modTangentInTF(1 - 2 * qY * qY, 2 * qX * qY, -2 * qY * qW);
    const f3 modBitangentInTF(2 * qX * qY, 1 - 2 * qX * qX, 2 * qX * qW);
    // matTFtoW = columns (tangent, bitangent, normal); ReferenceFrame(t, b, n) keeps the three vectors as given
    Frame out;
    out.t = frame.from_local(modTangentInTF);
    out.b = frame.from_local(modBitangentInTF);
    out.n = frame.from_local(modNormalInTF);
    frame = out;
}

GFX_DEV void concentric_disk(float u0, float u1, float& dx, float& dy) { // common_device.cuh:285-318
    const float sx = 2 * u0 - 1, sy = 2 * u1 - 1;
    if (sx == 0 && sy == 0) { dx = 0; dy = 0; return; }
    float r, theta;
    if (sx >= -sy) {
        if (sx > sy) { r = sx; theta = sy / sx; }
        else { r = sy; theta = 2 - sx / sy; }
    }
    else {
        if (sx > sy) { r = -sy; theta = 6 + sx / sy; }
        else { r = -sx; theta = 4 + sy / sx; }
    }
    theta *= kPi / 4;
    float s, c;
    gm_sincos(theta, s, c);
    dx = r * c;
    dy = r * s;
}
GFX_DEV f3 cosine_hemisphere(float u0, float u1) {
    float x, y;
    concentric_disk(u0, u1, x, y);
    return f3(x, y, sqrtf(fmax2(0.0f, 1.0f - x * x - y * y)));
}

// ---------------------------------------------------------------- BSDF
struct Bsdf {
    uint32_t type;
    f3 diffuse;      // Lambert: reflectance
    f3 specularF0;
    float roughness;

    // setupBSDFBody<> (common/common_device.cuh:376-385, 778-826): every value is tex2DLod(texture, texCoord, 0);
    // texture slot 0 = the constant of the material (the reference's 1x1 immediate texture)
    GFX_DEV void setup(const DevScene& sc, const gfx_material& m, float u, float v) {
        type = m.bsdfType;
        f3 a(m.a[0], m.a[1], m.a[2]);
        if (m.texA) { const float4 t = tex2d(sc, m.texA, u, v); a = f3(t.x, t.y, t.z); }
        diffuse = a;
        specularF0 = f3(0.0f);
        roughness = 1.0f;
        if (type == GFX_BSDF_DIFFUSE_AND_SPECULAR) {
            f3 b(m.b[0], m.b[1], m.b[2]);
            if (m.texB) { const float4 t = tex2d(sc, m.texB, u, v); b = f3(t.x, t.y, t.z); }
            float smooth = m.smoothness;
            if (m.texSmoothness) smooth = tex2d(sc, m.texSmoothness, u, v).x;
            specularF0 = b;
            roughness = 1 - fmin2(smooth, 0.999f);
        }
        else if (type == GFX_BSDF_SIMPLE_PBR) {
            f3 orm(m.b[0], m.b[1], m.b[2]);
            if (m.texB) { const float4 t = tex2d(sc, m.texB, u, v); orm = f3(t.x, t.y, t.z); }
            const f3 base = a;
            const float smoothness = fmin2(1.0f - orm.y, 0.999f);
            const float metallic = orm.z;
            diffuse = base * (1 - metallic);
            specularF0 = f3(0.16f * sq(0.5f) * (1 - metallic)) + base * metallic;
            roughness = 1 - smoothness;
        }
    }

    static GFX_DEV float ggx_d(float ag, f3 m) {
        if (m.z <= 0.0f) return 0.0f;
        const float t = sq(m.x) + sq(m.y) + sq(m.z * ag);
        return sq(ag) / (kPi * sq(t));
    }
    static GFX_DEV float ggx_g1(float ag, f3 v, f3 m) {
        if (dot(v, m) * v.z <= 0) return 0.0f;
        const float t = sq(ag) * (sq(v.x) + sq(v.y)) / sq(v.z);
        return 2 / (1 + sqrtf(1 + t));
    }
    static GFX_DEV float ggx_g_height_correlated(float ag, f3 v1, f3 v2, f3 m) {
        const float a1 = sq(ag) * (sq(v1.x) + sq(v1.y)) / sq(v1.z);
        const float a2 = sq(ag) * (sq(v2.x) + sq(v2.y)) / sq(v2.z);
        const float l1 = (-1 + sqrtf(1 + a1)) / 2;
        const float l2 = (-1 + sqrtf(1 + a2)) / 2;
        // chi+(dot(v, m) / v.z) without the division: v.z is a non-zero component of a unit vector (|v.z| < 2),
        // so the quotient of a non-zero numerator cannot round to zero and its sign is the sign pair's.
        const float d1 = dot(v1, m), d2 = dot(v2, m);
        const float c1 = ((d1 > 0 && v1.z > 0) || (d1 < 0 && v1.z < 0)) ? 1.0f : 0.0f;
        const float c2 = ((d2 > 0 && v2.z > 0) || (d2 < 0 && v2.z < 0)) ? 1.0f : 0.0f;
        return c1 * c2 / (1 + l1 + l2);
    }
    static GFX_DEV float ggx_pdf(float ag, f3 v, f3 m) {
        return ggx_g1(ag, v, m) * fabsf(dot(v, m)) * ggx_d(ag, m) / fabsf(v.z);
    }
    static GFX_DEV float ggx_sample(float ag, f3 v, float u0, float u1, f3& m, float& mPdf) {
        const f3 sv = unit(f3(ag * v.x, ag * v.y, v.z));
        const float d2 = sqrtf(sv.x * sv.x + sv.y * sv.y);
        const float rd2 = 1.0f / d2;
        const f3 T1 = (sv.z < 0.9999f) ? f3(sv.y * rd2, -sv.x * rd2, 0) : f3(1, 0, 0);
        const f3 T2(T1.y * sv.z, -T1.x * sv.z, d2);
        const float a = 1.0f / (1.0f + sv.z);
        const float r = sqrtf(u0);
        const float phi = kPi * ((u1 < a) ? u1 / a : 1 + (u1 - a) / (1.0f - a));
        float sp, cp;
        gm_sincos(phi, sp, cp);
        const float P1 = r * cp;
        const float P2 = r * sp * ((u1 < a) ? 1.0f : sv.z);
        m = P1 * T1 + P2 * T2 + sqrtf(1.0f - P1 * P1 - P2 * P2) * sv;
        m = unit(f3(ag * m.x, ag * m.y, m.z));
        const float D = ggx_d(ag, m);
        mPdf = ggx_g1(ag, v, m) * fabsf(dot(v, m)) * D / fabsf(v.z);
        return D;
    }

    // shared tail of evaluate / sampleThroughput: f = diffuse lobe + specular lobe
    GFX_DEV f3 lobes(f3 dirL, f3 dirV, f3 m, float dotLH, float D, float oneMinusDotVN5) const {
        const float ag = roughness * roughness;
        const float oneMinusDotLH5 = pow5(1 - dotLH);
        const float G = ggx_g_height_correlated(ag, dirL, dirV, m);
        const f3 F = mix3(specularF0, f3(1.0f), oneMinusDotLH5);
        const float denom = 4 * dirL.z * dirV.z;
        f3 spec = F * ((D * G) / denom);
        if (G == 0) spec = f3(0.0f);
        const float F_D90 = 0.5f * roughness + 2 * roughness * dotLH * dotLH;
        const float oneMinusDotLN5 = pow5(1 - dirL.z);
        const float fOut = mixf(1.0f, F_D90, oneMinusDotVN5);
        const float fIn = mixf(1.0f, F_D90, oneMinusDotLN5);
        const f3 diff = diffuse * (fOut * fIn * mixf(1.0f, 1.0f / 1.51f, roughness) / kPi);
        return diff + spec;
    }

    GFX_DEV f3 evaluate(f3 vGiven, f3 vSampled) const {
        if (type == GFX_BSDF_LAMBERT)
            return vGiven.z * vSampled.z > 0 ? diffuse / kPi : f3(0.0f);
        if (vSampled.z * vGiven.z <= 0) return f3(0.0f);
        GFX_PROF(3);
        const bool entering = vGiven.z >= 0.0f;
        const f3 dirV = entering ? vGiven : -vGiven;
        const f3 dirL = entering ? vSampled : -vSampled;
        const f3 m = unit(dirL + dirV);
        const float dotLH = dot(dirL, m);
        const float D = ggx_d(roughness * roughness, m);
        return lobes(dirL, dirV, m, dotLH, D, pow5(1 - dirV.z));
    }

    GFX_DEV void lobe_weights(f3 vGiven, f3 dirV, float& wDiff, float& wSpec) const {
        const float eF_D90 = 0.5f * roughness + 2 * roughness * vGiven.z * vGiven.z;
        const float oneMinusDotVN5 = pow5(1 - dirV.z);
        const float eDiffFresnel = mixf(1.0f, eF_D90, oneMinusDotVN5);
        wDiff = luminance_srgb(diffuse) * sq(eDiffFresnel) * mixf(1.0f, 1.0f / 1.51f, roughness);
        const float eOneMinusDotVH5 = pow5(1 - dirV.z);
        wSpec = mixf(luminance_srgb(specularF0), 1.0f, eOneMinusDotVH5);
    }

    GFX_DEV f3 sample_throughput(f3 vGiven, float u0, float u1, f3& vSampled, float& pdf) const {
        if (type == GFX_BSDF_LAMBERT) {
            vSampled = cosine_hemisphere(u0, u1);
            pdf = vSampled.z / kPi;
            if (vGiven.z <= 0.0f) vSampled.z *= -1;
            return diffuse;
        }
        const float ag = roughness * roughness;
        const bool entering = vGiven.z >= 0.0f;
        const f3 dirV = entering ? vGiven : -vGiven;
        const float oneMinusDotVN5 = pow5(1 - dirV.z);
        float wDiff, wSpec;
        lobe_weights(vGiven, dirV, wDiff, wSpec);
        const float sumW = wDiff + wSpec;
        if (sumW == 0.0f) { pdf = 0.0f; return f3(0.0f); }
        const float uComp = u1;
        f3 dirL, m;
        float pdfDiff, pdfSpec, dotLH, D;
        if (sumW * uComp < wDiff) {
            u1 = (sumW * uComp - 0) / wDiff;
            dirL = cosine_hemisphere(u0, u1);
            pdfDiff = dirL.z / kPi;
            m = unit(dirL + dirV);
            dotLH = fmin2(dot(dirL, m), 1.0f);
            const float common = 1.0f / (4 * dotLH);
            pdfSpec = common * ggx_pdf(ag, dirV, m);
            D = ggx_d(ag, m);
        }
        else {
            u1 = (sumW * uComp - wDiff) / wSpec;
            float mPdf;
            D = ggx_sample(ag, dirV, u0, u1, m, mPdf);
            const float dotVH = fmin2(dot(dirV, m), 1.0f);
            dotLH = dotVH;
            dirL = 2 * dotVH * m - dirV;
            if (dirL.z * dirV.z <= 0) { pdf = 0.0f; return f3(0.0f); }
            const float common = 1.0f / (4 * dotLH);
            pdfSpec = common * mPdf;
            pdfDiff = dirL.z / kPi;
        }
        f3 ret = lobes(dirL, dirV, m, dotLH, D, oneMinusDotVN5);
        vSampled = entering ? dirL : -dirL;
        pdf = (pdfDiff * wDiff + pdfSpec * wSpec) / sumW;
        ret = ret * (dirL.z / pdf);
        return ret;
    }

    GFX_DEV float evaluate_pdf(f3 vGiven, f3 vSampled) const {
        if (type == GFX_BSDF_LAMBERT)
            return vGiven.z * vSampled.z > 0 ? fabsf(vSampled.z) / kPi : 0.0f;
        const float ag = roughness * roughness;
        const bool entering = vGiven.z >= 0.0f;
        const f3 dirV = entering ? vGiven : -vGiven;
        const f3 dirL = entering ? vSampled : -vSampled;
        const f3 m = unit(dirL + dirV);
        const float dotLH = dot(dirL, m);
        const float common = 1.0f / (4 * dotLH);
        float wDiff, wSpec;
        lobe_weights(vGiven, dirV, wDiff, wSpec);
        const float sumW = wDiff + wSpec;
        if (sumW == 0.0f) return 0.0f;
        const float pdfDiff = dirL.z / kPi;
        const float pdfSpec = common * ggx_pdf(ag, dirV, m);
        return (pdfDiff * wDiff + pdfSpec * wSpec) / sumW;
    }

    GFX_DEV f3 dh_reflectance_estimate(f3 vGiven) const {
        if (type == GFX_BSDF_LAMBERT) return diffuse;
        const f3 dirV = vGiven.z >= 0.0f ? vGiven : -vGiven;
        const float eF_D90 = 0.5f * roughness + 2 * roughness * sq(dirV.z);
        const float oneMinusDotVN5 = pow5(1 - dirV.z);
        const float eDiffFGiven = mixf(1.0f, eF_D90, oneMinusDotVN5);
        const f3 diffuseDHR = diffuse * eDiffFGiven * 1.0f * mixf(1.0f, 1.0f / 1.51f, roughness);
        const float eOneMinusDotVH5 = pow5(1 - dirV.z) * (1 - roughness);
        const f3 specularDHR = mix3(specularF0, f3(1.0f), eOneMinusDotVH5);
        return min3(diffuseDHR + specularDHR, f3(1.0f));
    }
};

// ---------------------------------------------------------------- light sampling
struct LightSample {
    f3 emittance;
    f3 position;
    f3 normal;
    uint32_t atInfinity;
};
// An emittance-texture read that has not happened yet (k_initial_candidates fetches it only for candidates whose
// geometric / BSDF term is non-zero: the emittance of a zero-weight candidate is never observed).
// Holds the record and the barycentric coordinates; the record's EmitterTexRef (texture coordinates + descriptor, its own 32-byte
// gather) is read with the texels, not before.
struct PendingEmittance { uint32_t tex; uint32_t rec; float bcA, bcB, bcC; };

// One texel of the environment map as the row searches want it (gfx_restir_static_params::envRowTable, gfxh_env_build_row_table): the
// conditional CDF / PDF entry of the texel's column, the row guide's entry and the texel itself in ONE 32-byte record, w + 1 records per
// row (the last holds the row's final CDF value).  A light sample on the map is a guided search in a row picked at random out of h: with
// five separate arrays (texels 32 MB, PDFs 8 MB, CDFs 8 MB, guide 4 MB for the 2048 x 1024 map of configs[4]) every step of it was a
// 64-byte sector of its own from HBM -- 8.2 GB per candidate pass, 4.9 TB/s: that pass ran at the memory system's rate
// (profiles/r05_experiments.txt 5).  Interleaved, the guide pair, the probes and the final (cdf, cdf', pdf, texel) of a sample lie in the
// two or three sectors around the texel it ends on.  Same values, same arithmetic: same samples bit for bit.
struct EnvRowRec { float cdf, pdf; uint32_t guide; float r, g, b; float cdfNext; uint32_t pad1; };
static_assert(sizeof(EnvRowRec) == 32, "two records per 64-byte sector");

// The largest index of [lo, hi] whose CDF value is <= u (cdfAt(lo) <= u is known; the CDF is monotone): what the bisection of
// DiscreteDistribution / RegularConstantContinuousDistribution1D::sample returns, found with PIVOTS probes per dependent round trip
// instead of one.  A guide bracket is 0-2 entries wide for nearly every sample, but a map with a sun in it has rows in which a few guide
// cells cover a hundred dim texels each: one lane of a wave lands there in every other iteration, and the wave waited for its
// seven to eleven dependent loads (profiles/r05_experiments.txt 5).  Narrows [lo, hi] until at most three entries are left.
template <int PIVOTS, typename CdfAt>
GFX_DEV void narrow_bracket(CdfAt cdfAt, float u, int& lo, int& hi) {
    while (hi - lo > 2) {
        const int span = hi - lo;
        int piv[PIVOTS]; float c[PIVOTS];
#pragma unroll
        for (int j = 0; j < PIVOTS; ++j) { piv[j] = lo + ((span * (j + 1)) / (PIVOTS + 1)); c[j] = cdfAt(piv[j]); }
        int nlo = lo, nhi = hi;
#pragma unroll
        for (int j = 0; j < PIVOTS; ++j) {
            if (c[j] <= u) nlo = max(nlo, piv[j]);
            else nhi = min(nhi, piv[j] - 1);
        }
        lo = nlo; hi = nhi;
    }
}

struct EnvMap { // RegularConstantContinuousDistribution2D + lat-long texture
    const float4* texels;
    const float* rowPDF; const float* rowCDF; const float* topPDF; const float* topCDF;
    const uint16_t* rowGuide; const uint16_t* topGuide;   // optional guide tables (gfxh_env_build_guides), or null
    const EnvRowRec* rowTable;                            // optional interleaved rows (gfxh_env_build_row_table; needs the guides), or null
    const uint32_t* rowSketch;                            // optional, with rowTable: 33 inverse-CDF knots + a mask of verified cells per row (gfxh_env_build_row_sketch), or null
    int32_t w, h;
    GFX_DEV bool present() const { return texels != nullptr; }
    GFX_DEV const EnvRowRec* table_row(uint32_t row) const { return rowTable + static_cast<size_t>(row) * GFX_ENV_ROW_STRIDE(w); }
    GFX_DEV f3 fetch(float u, float v) const { // nearest texel (the build's tex2DLod contract)
        uint32_t x = f2u_sat(u * w); if (x > static_cast<uint32_t>(w - 1)) x = w - 1;
        uint32_t y = f2u_sat(v * h); if (y > static_cast<uint32_t>(h - 1)) y = h - 1;
        if (rowTable) {
            const float4 t = reinterpret_cast<const float4*>(table_row(y) + x)[1];     // (b | pad) of the record's second half; r, g sit in the first
            const float4 s = reinterpret_cast<const float4*>(table_row(y) + x)[0];
            return f3(s.w, t.x, t.y);
        }
        const float4 t = texels[static_cast<size_t>(y) * w + x];
        return f3(t.x, t.y, t.z);
    }
    static GFX_DEV float sample1d(const float* pdf, const float* cdf, uint32_t n, float u, float& p, const uint16_t* guide) {
        int idx = 0;
        if (guide) {
            // largest idx with cdf[idx] <= u, inside the bracket of u's cell: cdf[guide[k - 1]] lies in an earlier
            // cell (so below u), nothing past guide[k] can be <= u (cell() is monotone, the builder checked the CDF)
            const uint32_t k = min(n - 1u, static_cast<uint32_t>(u * static_cast<float>(n)));
            int hi = guide[k];
            int lo = k ? guide[k - 1] : 0;
            narrow_bracket<3>([&](int i) { return cdf[i]; }, u, lo, hi);     // wide brackets: three probes per round trip
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (cdf[mid] <= u) lo = mid;
                else hi = mid - 1;
            }
            idx = lo;
        }
        else
        for (int d = static_cast<int>(next_pow2(n) >> 1); d >= 1; d >>= 1) {
            if (idx + d >= static_cast<int>(n)) continue;
            if (cdf[idx + d] <= u) idx += d;
        }
        const float t = (u - cdf[idx]) / (cdf[idx + 1] - cdf[idx]);
        p = pdf[idx];
        return (idx + t) / n;
    }
    // sample1d with the guide over one row of the interleaved table: the same search on the same values.  A light sample on the map is a
    // chain of dependent loads from a table far larger than the L2s (guide pair -> probes -> the column's entries), and a wave's eight
    // environment candidates are eight such chains one after the other: when the guide's bracket is at most three columns wide -- the usual
    // case -- the columns it can end on are loaded TOGETHER right after the guide pair (they share two or three sectors) and the search
    // finishes in registers: two dependent round trips per row instead of three or four.  The column found is the one the bisection
    // finds (the largest index of the bracket whose CDF value is <= u; the builder verified that the CDF is monotone).
    static GFX_DEV float sample1d_row(const EnvRowRec* row, uint32_t n, float u, float& p) {
        const uint32_t k = min(n - 1u, static_cast<uint32_t>(u * static_cast<float>(n)));
        int hi = static_cast<int>(row[k].guide);
        int lo = k ? static_cast<int>(row[k - 1].guide) : 0;
        float2 here; float next;
        int idx;
        narrow_bracket<7>([&](int i) { return row[i].cdf; }, u, lo, hi);  // wide brackets (dim stretches of a row with a sun in it): seven probes per round trip
        {
            const int last = static_cast<int>(n);                       // record n holds the row's final CDF value
            const float2 c0 = *reinterpret_cast<const float2*>(row + lo);
            const float2 c1 = *reinterpret_cast<const float2*>(row + min(lo + 1, last));
            const float2 c2 = *reinterpret_cast<const float2*>(row + min(lo + 2, last));
            const float c3 = row[min(lo + 3, last)].cdf;
            idx = lo; here = c0; next = c1.x;
            if (hi >= lo + 1 && c1.x <= u) { idx = lo + 1; here = c1; next = c2.x; }
            if (hi >= lo + 2 && c2.x <= u) { idx = lo + 2; here = c2; next = c3; }
        }
        const float t = (u - here.x) / (next - here.x);
        p = here.y;
        return (idx + t) / n;
    }
    // sample1d_row without the guide: the row's sketch (33 knots of its inverse CDF, L2-resident) predicts the column to within one -- the
    // builder verified that for every u of this cell -- so the answer is in the 128-byte line of four records around the prediction, or in
    // the first / last record of a neighbouring line when the prediction sits on the line's edge: one line of the table per sample
    // instead of the guide's and the column's.  The column chosen is the largest one whose CDF value is <= u, as the bisection's.
    static GFX_DEV bool sample1d_row_sketch(const EnvRowRec* row, const uint32_t* sketch, uint32_t rowIndex, uint32_t numRows, uint32_t n, float u, float& p, float& d0) {
        // the row's record, then -- in a cell whose interpolation the builder could not verify -- the cell's child record at 1/32 of the step
        const uint32_t* rec = sketch + static_cast<size_t>(rowIndex) * GFX_ENV_SKETCH_WORDS;
        float x = u, pred = 0.0f;
        bool found = false;
#pragma unroll
        for (int level = 0; level < 2 && !found; ++level) {
            const float xk = x * static_cast<float>(GFX_ENV_SKETCH_CELLS);
            uint32_t k = static_cast<uint32_t>(xk);
            if (k > GFX_ENV_SKETCH_CELLS - 1u) k = GFX_ENV_SKETCH_CELLS - 1u;
            const float t = xk - static_cast<float>(k);                  // exact
            const uint32_t w0 = rec[k], w1 = rec[k + 1u];               // knot k carries the cell's verdict in its sign bit (set = not verified)
            if (!(w0 >> 31)) {
                const float k0 = bits2f(w0), k1 = bits2f(w1 & 0x7FFFFFFFu);
                const float d = k1 - k0;
                pred = k0 + t * d;
                found = true;
            }
            else if (level == 0) {
                const uint32_t mask = rec[GFX_ENV_SKETCH_CELLS + 1u];
                const uint32_t child = rec[GFX_ENV_SKETCH_CELLS + 2u] + static_cast<uint32_t>(__popc(~mask & ((1u << k) - 1u)));
                rec = sketch + static_cast<size_t>(numRows + child) * GFX_ENV_SKETCH_WORDS;
                x = t;
            }
        }
        if (!found) return false;
        int fp = static_cast<int>(pred);
        fp = fp < 0 ? 0 : (fp > static_cast<int>(n) - 1 ? static_cast<int>(n) - 1 : fp);
        const int a = fp & ~3;                                       // n is a multiple of four or the last group is padded (GFX_ENV_ROW_STRIDE)
        float2 c[4]; float nx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[j] = *reinterpret_cast<const float2*>(row + a + j); nx[j] = row[a + j].cdfNext; }
        int idx = a; float2 here = c[0]; float next = nx[0];
        const int last = static_cast<int>(n) - 1;
#pragma unroll
        for (int j = 1; j < 4; ++j) if (a + j <= last && c[j].x <= u) { idx = a + j; here = c[j]; next = nx[j]; }
        if (c[0].x > u && a > 0) {                                   // the column before the line (prediction one too high at the line's first record)
            idx = a - 1; here = *reinterpret_cast<const float2*>(row + idx); next = c[0].x;
        }
        else if (idx == a + 3 && a + 4 <= last && next <= u) {       // the column after the line
            idx = a + 4; here = *reinterpret_cast<const float2*>(row + idx); next = row[idx].cdfNext;
        }
        const float tt = (u - here.x) / (next - here.x);
        p = here.y;
        d0 = (idx + tt) / n;
        return true;
    }
    GFX_DEV float evaluate_pdf(float d0, float d1) const { // common_shared.h:344-348, 380-383
        uint32_t row = f2u_sat(d1 * h); if (row > static_cast<uint32_t>(h - 1)) row = h - 1;
        uint32_t col = f2u_sat(d0 * w); if (col > static_cast<uint32_t>(w - 1)) col = w - 1;
        if (rowTable) return topPDF[row] * table_row(row)[col].pdf;
        return topPDF[row] * rowPDF[static_cast<size_t>(row) * w + col];
    }
    GFX_DEV void sample(float u0, float u1, float& d0, float& d1, float& p) const { // common_shared.h:372-379
        float topP;
        d1 = sample1d(topPDF, topCDF, h, u1, topP, topGuide);
        uint32_t row = f2u_sat(d1 * h); if (row > static_cast<uint32_t>(h - 1)) row = h - 1;
        if (rowTable) {
            if (!(rowSketch && sample1d_row_sketch(table_row(row), rowSketch, row, static_cast<uint32_t>(h), w, u0, p, d0)))
                d0 = sample1d_row(table_row(row), w, u0, p);
        }
        else d0 = sample1d(rowPDF + static_cast<size_t>(row) * w, rowCDF + static_cast<size_t>(row) * (w + 1), w, u0, p,
                           rowGuide ? rowGuide + static_cast<size_t>(row) * w : nullptr);
        p *= topP;
    }
};

GFX_DEV DevVertex load_vertex(const DevVertex* __restrict__ v) {
    const float4* p = reinterpret_cast<const float4*>(v);
    const float4 a = p[0], b = p[1], c = p[2];
    DevVertex r;
    r.px = a.x; r.py = a.y; r.pz = a.z; r.nx = a.w;
    r.ny = b.x; r.nz = b.y; r.tx = b.z; r.ty = b.w;
    r.tz = c.x; r.u = c.y; r.v = c.z; r.pad = 0;
    return r;
}
GFX_DEV m34 load_m34(const float* __restrict__ p) {
    const float4* q = reinterpret_cast<const float4*>(p);
    const float4 a = q[0], b = q[1], c = q[2];
    m34 m;
    m.m[0] = a.x; m.m[1] = a.y; m.m[2] = a.z; m.m[3] = a.w;
    m.m[4] = b.x; m.m[5] = b.y; m.m[6] = b.z; m.m[7] = b.w;
    m.m[8] = c.x; m.m[9] = c.y; m.m[10] = c.z; m.m[11] = c.w;
    return m;
}
GFX_DEV m33 load_m33_rows(const float* __restrict__ p) {   // 3 rows padded to float4
    const float4* q = reinterpret_cast<const float4*>(p);
    const float4 a = q[0], b = q[1], c = q[2];
    m33 m;
    m.r0 = f3(a.x, a.y, a.z); m.r1 = f3(b.x, b.y, b.z); m.r2 = f3(c.x, c.y, c.z);
    return m;
}

// the instance-level distribution (level 0) in global memory
GFX_DEV InstDist inst_dist_global(const DevScene& sc) {
    InstDist d;
    d.probs = sc.lightProbs + sc.lightInstDistOffset;
    d.cdf = sc.lightCDF + sc.lightInstDistOffset;
    const bool usable = reinterpret_cast<const uint32_t*>(sc.lightInstIntegral)[2] != 0u;
    d.guide = usable ? sc.lightInstGuide : nullptr;
    d.guideScale = sc.lightInstIntegral[1];
    d.guideCells = sc.lightInstGuideCells;
    return d;
}

// the same without the guide table: the plain binary search of the reference at every level
GFX_DEV InstDist inst_dist_global_unguided(const DevScene& sc) {
    InstDist d = inst_dist_global(sc);
    d.guide = nullptr;
    return d;
}

// The reference's three nested searches (restir_di_shared.h:366-415): which emitter record does ul select?
// Returns false on the two early outs (an instance or geometry instance of probability zero).
// partialProb = (1 * instProb) * geomInstProb.  This is the definition the interval table is built from
// and verified against (lights.hip); sample_light only runs it when the build withdrew the table.
GFX_DEV bool light_locate_3level(const DevScene& sc, const InstDist& instDist, float ul,
                                 uint32_t& recIndex, uint32_t& instSlot, float& partialProb) {
    float lightProb = 1.0f;
    float instProb, uGeomInst;
    instSlot = discrete_sample_guided(instDist, *sc.lightInstIntegral, sc.numInsts, ul, instProb, &uGeomInst);
    lightProb *= instProb;
    if (instProb == 0.0f) return false;
    const DevInstance* inst = sc.insts + instSlot;
    // (distOffset, numGeomInsts, distIntegral, slotsOffset) in one 16-byte load
    const uint4 ih = *reinterpret_cast<const uint4*>(&inst->distOffset);

    float geomInstProb, uPrim;
    const uint32_t gi = discrete_sample(sc.lightProbs + ih.x, sc.lightCDF + ih.x, bits2f(ih.z), ih.y, uGeomInst, &geomInstProb, &uPrim);
    lightProb *= geomInstProb;
    if (geomInstProb == 0.0f) return false;
    const uint4 gr = *reinterpret_cast<const uint4*>(sc.lightGeomRefs + ih.x + gi);   // LightGeomRef

    const uint32_t prim = discrete_sample(nullptr, sc.lightCDF + gr.y, bits2f(gr.w), gr.z, uPrim, nullptr, nullptr);
    recIndex = gr.x + prim;
    partialProb = lightProb;
    return true;
}

// The emittance texel of a point (barycentric coordinates) on emitter record `rec`: texture coordinates and descriptor out of the
// record's EmitterTexRef, then the build's tex2DLod (restir_di_shared.h:504-514).
GFX_DEV float4 emitter_texel(const DevScene& sc, uint32_t rec, float bcA, float bcB, float bcC) {
    const float4* tp = reinterpret_cast<const float4*>(sc.emitterTexRefs + rec);
    const float4 t0 = tp[0], t1 = tp[1];
    const float tu = bcA * t0.x + bcB * t0.z + bcC * t1.x;
    const float tv = bcA * t0.y + bcB * t0.w + bcC * t1.y;
    const uint32_t dims = f2bits(t1.w);
    DevTexture desc;
    desc.offset = f2bits(t1.z); desc.width = (dims & 0x3FFFu) + 1u; desc.height = ((dims >> 14) & 0x3FFFu) + 1u; desc.format = dims >> 28;
    return tex2d_desc(sc, desc, tu, tv);
}

GFX_DEV uint32_t emitter_matrix_index(uint32_t recFlags) { return (recFlags >> kEmitterMatrixShift) & kEmitterMatrixMask; }

// Which emitter record a light-selection number ul picks, and what comes with it.
struct LightPick {
    uint32_t rec, instSlot;
    float density;        // table: the area density of a sample on the record
    float partialProb;    // search fallback: (1 * instProb) * geomInstProb
    bool ok;              // false: the reference returns early with a zero density (restir_di_shared.h:372, 391)
    bool table;
};

// Selection in one call: the table when the build verified it (wave-uniform), else the reference's searches.
GFX_DEV LightPick light_select(const DevScene& sc, float ul) {
#ifdef GFX_LIGHT_TABLE_ONLY   // experiment: what the kernels cost without the search fallback compiled in
    const bool table = true;
#else
    const bool table = sc.spanHeader[0] != 0u;
#endif
    if (table) {
        if (sc.numSpans == 0) { LightPick pk; pk.rec = 0; pk.instSlot = 0; pk.density = 0; pk.partialProb = 0; pk.ok = false; pk.table = true; return pk; }
        // guide cell -> short search on the spans' begin values -> ONE 16-byte span load (loading both bracketing
        // spans up front instead measured 15 % slower: profiles/r02_initial_candidates.txt)
        EmitterSpan span;
        const int32_t j = span_lookup(sc.spans, sc.numSpans, sc.spanGuide, sc.spanGuideCells, ul, span);
        LightPick pk; pk.rec = j < 0 ? 0u : static_cast<uint32_t>(j); pk.instSlot = span.instSlot; pk.density = span.density; pk.partialProb = 0; pk.ok = j >= 0; pk.table = true;
        return pk;
    }
    LightPick pk;
    pk.density = 0.0f; pk.table = false;
    pk.ok = light_locate_3level(sc, inst_dist_global(sc), ul, pk.rec, pk.instSlot, pk.partialProb);
    return pk;
}

// Spherical triangle of (pA, pB, pC) seen from a point (restir_di_shared.h:430-445, path_tracing_shared.h:551-561)
struct SphericalTriangle { f3 A, B, C; float cos_c, cosAlpha, alpha, sinAlpha, sphArea; };
GFX_DEV SphericalTriangle spherical_triangle(f3 pA, f3 pB, f3 pC, f3 ref) {
    SphericalTriangle t;
    t.A = unit(pA - ref); t.B = unit(pB - ref); t.C = unit(pC - ref);
    const f3 cAB = unit(cross(t.A, t.B)), cBC = unit(cross(t.B, t.C)), cCA = unit(cross(t.C, t.A));
    t.cos_c = dot(t.A, t.B);
    t.cosAlpha = -dot(cAB, cCA);
    const float cosBeta = -dot(cBC, cAB);
    const float cosGamma = -dot(cCA, cBC);
    t.alpha = gm_acos(t.cosAlpha);
    t.sinAlpha = sqrtf(1 - sq(t.cosAlpha));
    t.sphArea = t.alpha + gm_acos(cosBeta) + gm_acos(cosGamma) - kPi;
    return t;
}

// Selection by the reference's searches regardless of the table (solid-angle sampling needs the plain probability).
GFX_DEV LightPick light_select_search(const DevScene& sc, float ul) {
    LightPick pk;
    pk.density = 0.0f; pk.table = false;
    pk.ok = light_locate_3level(sc, inst_dist_global(sc), ul, pk.rec, pk.instSlot, pk.partialProb);
    return pk;
}

// The rest of sampleLight<false> for a picked record: point on the triangle, normal, emittance, area density.
// EMITTER_TEX = false compiles the emittance-texture read out (kernels instantiate both and the host picks by
// whether any emitter material has an emittance texture).
// SOLID_ANGLE = sampleLight<true> (restir_di_shared.h:419-483): the point is drawn uniformly in the solid angle the
// triangle subtends from shadingPoint.
template <bool EMITTER_TEX = true, bool SOLID_ANGLE = false>
GFX_DEV void light_from_record(const DevScene& sc, const LightPick& pk, float4 r0, float4 r1, float4 r2, float4 r3, const m33& normalMatrix,
                               float u0, float u1, LightSample& ls, float& areaPDensity, f3 shadingPoint = f3(0.0f),
                               PendingEmittance* pending = nullptr) {
    // EmitterRec (r0..r3): world-space triangle + first vertex normal + emittance in 64 bytes; the other two normals (smooth emitters
    // only), 2 / |ng| and the primitive's probability (three-search fallback, solid-angle sampling) in EmitterRecExtra
    const f3 pA(r0.x, r0.y, r0.z), pB(r0.w, r1.x, r1.y), pC(r1.z, r1.w, r2.x);
    const f3 nA(r2.y, r2.z, r2.w);
    const uint32_t flags = f2bits(r3.w);
    f3 nB = nA, nC = nA;
    float twoOverLenNg = 0.0f, primProb = 0.0f;
    if ((flags & kEmitterSmooth) || SOLID_ANGLE || !pk.table) {
        GFX_PROF(7);
        const float4* xp = reinterpret_cast<const float4*>(sc.emitterRecExtras + pk.rec);
        const float4 x0 = xp[0], x1 = xp[1];
        if (flags & kEmitterSmooth) { nB = f3(x0.x, x0.y, x0.z); nC = f3(x0.w, x1.x, x1.y); }
        twoOverLenNg = x1.z; primProb = x1.w;
    }

    float bcA, bcB, bcC;
    if (SOLID_ANGLE) {
        // the interval table tabulates lightProb * (2 / |ng|), not lightProb: solid-angle sampling always selects with the
        // reference's three searches (light_select_search), whose partial product times the primitive's probability it is
        const float lightProb = pk.partialProb * primProb;
        const SphericalTriangle st = spherical_triangle(pA, pB, pC, shadingPoint);
        const float sphAreaHat = st.sphArea * u0;
        float s, t;
        gm_sincos(sphAreaHat - st.alpha, s, t);
        const float uu = t - st.cosAlpha;
        const float vv = s + st.sinAlpha * st.cos_c;
        const float q = ((vv * t - uu * s) * st.cosAlpha - vv) / ((vv * s + uu * t) * st.sinAlpha);
        const f3 cHat = q * st.A + sqrtf(1 - sq(q)) * unit(st.C - dot(st.C, st.A) * st.A);
        const float z = 1 - u1 * (1 - dot(cHat, st.B));
        const f3 dir = z * st.B + sqrtf(1 - sq(z)) * unit(cHat - dot(cHat, st.B) * st.B);
        const f3 eAB = pB - pA, eAC = pC - pA;
        const f3 pVec = cross(dir, eAC);
        const float recDet = 1.0f / dot(eAB, pVec);
        const f3 tVec = shadingPoint - pA;
        bcB = dot(tVec, pVec) * recDet;
        const f3 qVec = cross(tVec, eAB);
        bcC = dot(dir, qVec) * recDet;
        const float dist = dot(eAC, qVec) * recDet;
        bcA = 1 - (bcB + bcC);
        const float dirPDF = 1 / st.sphArea;
        const f3 gn = unit(cross(pB - pA, pC - pA));
        const float lpCos = -dot(dir, gn);
        areaPDensity = (lpCos > 0 && is_finite(dirPDF)) ? lightProb * (dirPDF * lpCos / sq(dist)) : 0.0f;
    }
    else {
        bcA = 0.5f * u0;
        bcB = 0.5f * u1;
        const float off = bcB - bcA;
        if (off > 0) bcB += off;
        else bcA -= off;
        bcC = 1 - (bcA + bcB);
        areaPDensity = pk.table ? pk.density : (pk.partialProb * primProb) * twoOverLenNg;
    }

    ls.position = bcA * pA + bcB * pB + bcC * pC;
    ls.atInfinity = 0;
    const f3 n = bcA * nA + bcB * nB + bcC * nC;
    ls.normal = unit(mul(normalMatrix, n));
    ls.emittance = f3(r3.x, r3.y, r3.z);
    const uint32_t tex = EMITTER_TEX ? (flags & kEmitterTexMask) : 0u;   // emittance-texture slot (restir_di_shared.h:504-514)
    if (tex) {
        if (pending) { pending->tex = tex; pending->rec = pk.rec; pending->bcA = bcA; pending->bcB = bcB; pending->bcC = bcC; }
        else {
            const float4 tv4 = emitter_texel(sc, pk.rec, bcA, bcB, bcC);
            ls.emittance = f3(1.0f) * f3(tv4.x, tv4.y, tv4.z);
        }
    }
}

// The same with the record and the instance's normal-matrix rows gathered by the calling lane itself (seven 16-byte loads);
// k_initial_candidates gathers them cooperatively instead (coop_fetch.hip.h) and calls light_from_record directly.
template <bool EMITTER_TEX = true, bool SOLID_ANGLE = false>
GFX_DEV void light_fetch(const DevScene& sc, const LightPick& pk, float u0, float u1, LightSample& ls, float& areaPDensity, f3 shadingPoint = f3(0.0f),
                         PendingEmittance* pending = nullptr) {
    const float4* rp = reinterpret_cast<const float4*>(sc.emitterRecs + pk.rec);
    const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
    const m33 normalMatrix = load_m33_rows(sc.lightNormalMatrices + 16u * emitter_matrix_index(f2bits(r3.w)));
    light_from_record<EMITTER_TEX, SOLID_ANGLE>(sc, pk, r0, r1, r2, r3, normalMatrix, u0, u1, ls, areaPDensity, shadingPoint, pending);
}

GFX_DEV void sample_env_light(const EnvMap& env, float envRotation, float envPowerCoeff, float u0, float u1, LightSample& ls, float& areaPDensity) {
    float u, v, uvPDF;
    env.sample(u0, u1, u, v, uvPDF);
    const float phi = 2 * kPi * u;
    const float theta = kPi * v;
    float posPhi = phi - envRotation;
    posPhi = posPhi - floorf(posPhi / (2 * kPi)) * 2 * kPi;
    const f3 dir = from_polar_yup(posPhi, theta);
    ls.position = dir;
    ls.atInfinity = 1;
    ls.normal = -dir;
    const float sinTheta = gm_sin(theta);
    if (sinTheta == 0.0f) { areaPDensity = 0.0f; return; }
    areaPDensity = uvPDF / (2 * kPi * kPi * sinTheta);
    ls.emittance = f3(kPi * envPowerCoeff) * env.fetch(u, v);
}

// sampleLight<false>.  Returns the area density; sample left untouched past an early out exactly
// like the reference (the caller starts from a default-constructed LightSample).
template <bool EMITTER_TEX = true>
GFX_DEV void sample_light(const DevScene& sc,
                          const EnvMap& env, float envRotation, float envPowerCoeff,
                          float ul, bool sampleEnv, float u0, float u1, LightSample& ls, float& areaPDensity) {
    if (sampleEnv) { sample_env_light(env, envRotation, envPowerCoeff, u0, u1, ls, areaPDensity); return; }
    const LightPick pk = light_select(sc, ul);
    if (!pk.ok) { areaPDensity = 0.0f; return; }
    light_fetch<EMITTER_TEX>(sc, pk, u0, u1, ls, areaPDensity);
}
// sampleLight<true>: solid-angle sampling of the selected triangle from shadingPoint
GFX_DEV void sample_light_solid_angle(const DevScene& sc, const EnvMap& env, float envRotation, float envPowerCoeff, f3 shadingPoint,
                                      float ul, bool sampleEnv, float u0, float u1, LightSample& ls, float& areaPDensity) {
    if (sampleEnv) { sample_env_light(env, envRotation, envPowerCoeff, u0, u1, ls, areaPDensity); return; }
    const LightPick pk = light_select_search(sc, ul);
    if (!pk.ok) { areaPDensity = 0.0f; return; }
    light_fetch<true, true>(sc, pk, u0, u1, ls, areaPDensity, shadingPoint);
}

// Geometry of a shadow ray toward a light sample (restir_di_shared.h:524-545, 564-581).
struct ShadowRay { f3 dir; float dist2; float tmax; };
GFX_DEV ShadowRay shadow_ray(f3 shadingPoint, const LightSample& ls) {
    f3 d = ls.atInfinity ? ls.position : (ls.position - shadingPoint);
    ShadowRay r;
    r.dist2 = len2(d);
    float dist = sqrtf(r.dist2);
    r.dir = d / dist;
    if (ls.atInfinity) dist = 1e+10f;
    r.tmax = dist * 0.9999f;
    return r;
}

// Unshadowed contribution f * Le * G (performDirectLighting<.., false>); the visibility factor of
// the <.., true> form is applied by the caller from the traced occlusion bit.
// The same with the emittance texture of the sample still pending: it is read (into ls.emittance) only when
// f * G is non-zero, i.e. when the product can be non-zero at all.
GFX_DEV f3 direct_lighting_pending(const DevScene& sc, f3 shadingPoint, f3 vOutLocal, const Frame& frame, const Bsdf& bsdf, LightSample& ls,
                                   const PendingEmittance& pending) {
    const ShadowRay sr = shadow_ray(shadingPoint, ls);
    const f3 dirLocal = frame.to_local(sr.dir);
    const float lpCos = dot(-sr.dir, ls.normal);
    const float spCos = dirLocal.z;
    if (lpCos > 0) {
        GFX_PROF(2);
        const f3 fs = bsdf.evaluate(vOutLocal, dirLocal);
        const float G = lpCos * fabsf(spCos) / sr.dist2;
        const bool zero = (fs.x == 0.0f && fs.y == 0.0f && fs.z == 0.0f) || G == 0.0f;
        if (pending.tex && !zero) {
            GFX_PROF(4);
            const float4 t = emitter_texel(sc, pending.rec, pending.bcA, pending.bcB, pending.bcC);
            ls.emittance = f3(1.0f) * f3(t.x, t.y, t.z);
        }
        const f3 Le = ls.emittance / kPi;
        return fs * Le * G;
    }
    return f3(0.0f);
}

GFX_DEV f3 direct_lighting(f3 shadingPoint, f3 vOutLocal, const Frame& frame, const Bsdf& bsdf, const LightSample& ls) {
    const ShadowRay sr = shadow_ray(shadingPoint, ls);
    const f3 dirLocal = frame.to_local(sr.dir);
    const float lpCos = dot(-sr.dir, ls.normal);
    const float spCos = dirLocal.z;
    if (lpCos > 0) {
        const f3 Le = ls.emittance / kPi;
        const f3 fs = bsdf.evaluate(vOutLocal, dirLocal);
        const float G = lpCos * fabsf(spCos) / sr.dist2;
        return fs * Le * G;
    }
    return f3(0.0f);
}

GFX_DEV float target_weight(f3 c) { return (c.x + c.y + c.z) / 3; } // convertToWeight, restir_di_shared.h:82-85

} // namespace gfx
