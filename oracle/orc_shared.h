// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// orc_shared.h: CPU restatement of the host/device shared types and the device shading library
// of the reference: PCG32RNG, DiscreteDistribution1D, Reservoir, quantisers, offsetRayOrigin,
// ReferenceFrame, Lambert / DiffuseAndSpecular / SimplePBR BRDFs.
// Each function cites the reference file:line it follows (paths relative to the GfxExp tree).
//
// Evaluation-order contract: wherever the reference passes two rng() calls as function arguments
// (unspecified order in C++), this restatement draws LEFT TO RIGHT.
#pragma once
#include "orc_math.h"
#include "orc_texture.h"

namespace orc {

// ---------------------------------------------------------------- common/common_shared.h:116-138
struct PCG32RNG {
    uint64_t state;
    void setState(uint64_t s) { state = s; }
    uint32_t operator()() {
        const uint64_t oldstate = state;
        state = oldstate * 6364136223846793005ULL + 1;   // literal "+ 1" (:127)
        const uint32_t xorshifted = static_cast<uint32_t>(((oldstate >> 18u) ^ oldstate) >> 27u);
        const uint32_t rot = static_cast<uint32_t>(oldstate >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((-static_cast<int32_t>(rot)) & 31));
    }
    float getFloat0cTo1o() {
        const uint32_t fractionBits = ((*this)() >> 9) | 0x3f800000u;
        return bits2f(fractionBits) - 1.0f;
    }
};

static inline uint32_t nextPowerOf2(uint32_t x) { // common/basic_types.h:344-374
    if (x == 0) return 0;
    uint32_t e = 0;
    while ((1ull << e) < x) ++e;
    return 1u << e;
}

// common/common_shared.h:142-152
static inline uint32_t mapPrimarySampleToDiscrete(float u01, uint32_t numValues, float* uRemapped = nullptr) {
    uint32_t idx = f2u(u01 * numValues);
    if (idx > numValues - 1) idx = numValues - 1;
    if (uRemapped) *uRemapped = u01 * numValues - idx;
    return idx;
}

// ---------------------------------------------------------------- common/common_shared.h:175-276
struct DiscreteDistribution1D {
    const float* weights = nullptr;
    const float* CDF = nullptr;
    float integralValue = 0.0f;
    uint32_t numValues = 0;

    uint32_t sample(float u, float* prob, float* remapped = nullptr) const { // :209-247
        u *= integralValue;
        int idx = 0;
        for (int d = nextPowerOf2(numValues) >> 1; d >= 1; d >>= 1) {
            if (idx + d >= static_cast<int>(numValues))
                continue;
            if (CDF[idx + d] <= u)
                idx += d;
        }
        if (remapped) {
            const float lCDF = CDF[idx];
            float rCDF = integralValue;
            if (idx < static_cast<int>(numValues) - 1)
                rCDF = CDF[idx + 1];
            *remapped = (u - lCDF) / (rCDF - lCDF);
        }
        *prob = weights[idx] / integralValue;
        return static_cast<uint32_t>(idx);
    }
    float evaluatePMF(uint32_t idx) const { // :249-254
        if (!weights || integralValue == 0.0f) return 0.0f;
        return weights[idx] / integralValue;
    }
    float integral() const { return integralValue; }
};

// ---------------------------------------------------------------- common/common_shared.h:282-386
struct RegularConstantContinuousDistribution1D {
    const float* PDF = nullptr;
    const float* CDF = nullptr;
    float integralValue = 0.0f;
    uint32_t numValues = 0;
    float sample(float u, float* probDensity) const { // :316-343
        int idx = 0;
        for (int d = nextPowerOf2(numValues) >> 1; d >= 1; d >>= 1) {
            if (idx + d >= static_cast<int>(numValues))
                continue;
            if (CDF[idx + d] <= u)
                idx += d;
        }
        const float t = (u - CDF[idx]) / (CDF[idx + 1] - CDF[idx]);
        *probDensity = PDF[idx];
        return (idx + t) / numValues;
    }
    float evaluatePDF(float smp) const { // :344-348
        uint32_t idx = f2u(smp * numValues);
        if (idx > numValues - 1) idx = numValues - 1;
        return PDF[idx];
    }
};
struct RegularConstantContinuousDistribution2D {
    const float* rowPDF = nullptr;   // [h][w]
    const float* rowCDF = nullptr;   // [h][w+1]
    const float* rowIntegrals = nullptr;
    uint32_t w = 0, h = 0;
    RegularConstantContinuousDistribution1D top;
    RegularConstantContinuousDistribution1D row(uint32_t i) const {
        RegularConstantContinuousDistribution1D d;
        d.PDF = rowPDF + static_cast<size_t>(i) * w;
        d.CDF = rowCDF + static_cast<size_t>(i) * (w + 1);
        d.integralValue = rowIntegrals[i];
        d.numValues = w;
        return d;
    }
    void sample(float u0, float u1, float* d0, float* d1, float* probDensity) const { // :372-379
        float topPDF;
        *d1 = top.sample(u1, &topPDF);
        const uint32_t idx1D = mapPrimarySampleToDiscrete(*d1, top.numValues);
        *d0 = row(idx1D).sample(u0, probDensity);
        *probDensity *= topPDF;
    }
};

// ---------------------------------------------------------------- restir_di/restir_di_shared.h:82-144
static inline float convertToWeight(RGB color) { return (color.x + color.y + color.z) / 3; }

struct LightSample {
    RGB emittance;
    V3 position;
    V3 normal;
    uint32_t atInfinity = 0;
};
struct Reservoir {
    LightSample sample;
    float sumWeights = 0;
    uint32_t streamLength = 0;
    void initialize(const LightSample& s) { sample = s; sumWeights = 0; streamLength = 0; }
    bool update(const LightSample& newSample, float weight, float u) { // :118-125
        sumWeights += weight;
        const bool accepted = u < weight / sumWeights;
        if (accepted) sample = newSample;
        ++streamLength;
        return accepted;
    }
};
struct ReservoirInfo { float recPDFEstimate, targetDensity; };

// ---------------------------------------------------------------- common/common_device.cuh:14-79
static inline V3 fromPolarYUp(float phi, float theta) {
    float sinPhi, cosPhi, sinTheta, cosTheta;
    gm_sincos(phi, &sinPhi, &cosPhi);
    gm_sincos(theta, &sinTheta, &cosTheta);
    return V3(-sinPhi * sinTheta, cosTheta, cosPhi * sinTheta);
}
static inline void toPolarYUp(V3 v, float* phi, float* theta) {
    *theta = gm_acos(fmin2(fmax2(v.y, -1.0f), 1.0f));
    // fmod(atan2 + 2pi, 2pi): the argument is in [pi, 3pi], where fmod is one exact subtraction.
    const float a = gm_atan2(-v.x, v.z) + kTwoPi;
    *phi = a >= kTwoPi ? a - kTwoPi : a;
}
static inline uint16_t encodeBarycentric(float bc) {
    uint32_t q = f2u(bc * 65535u);
    if (q > 65535u) q = 65535u;
    return static_cast<uint16_t>(q);
}
static inline float decodeBarycentric(uint16_t qbc) { return qbc / 65535.0f; }
static inline uint32_t encodeVector(V3 v) {
    float phi, theta;
    toPolarYUp(v, &phi, &theta);
    uint32_t qPhi = f2u((phi / kTwoPi) * 65535u);
    if (qPhi > 65535u) qPhi = 65535u;
    uint32_t qTheta = f2u((theta / kPi) * 65535u);
    if (qTheta > 65535u) qTheta = 65535u;
    return (qTheta << 16) | qPhi;
}
static inline V3 decodeVector(uint32_t qv) {
    const uint32_t qPhi = qv & 0xFFFF;
    const uint32_t qTheta = qv >> 16;
    const float phi = kTwoPi * (qPhi / 65535.0f);
    const float theta = kPi * (qTheta / 65535.0f);
    return fromPolarYUp(phi, theta);
}
static inline uint32_t encodeNormal(V3 n) { return encodeVector(n); }   // :51-57 (same arithmetic)
static inline V3 decodeNormal(uint32_t qn) { return decodeVector(qn); } // :59-65
static inline uint32_t encodeTexCoords(V2 tc) {
    uint32_t q0 = f2u((tc.x - std::floor(tc.x)) * 65535u);
    if (q0 > 65535u) q0 = 65535u;
    uint32_t q1 = f2u((tc.y - std::floor(tc.y)) * 65535u);
    if (q1 > 65535u) q1 = 65535u;
    return (q1 << 16) | q0;
}
static inline V2 decodeTexCoords(uint32_t qtc) {
    return V2{ (qtc & 0xFFFF) / 65535.0f, (qtc >> 16) / 65535.0f };
}

static inline V3 halfVector(V3 a, V3 b) { return normalize(a + b); } // :81-83

static inline void makeCoordinateSystem(V3 normal, V3* tangent, V3* bitangent) { // :92-100
    const float sign = normal.z >= 0 ? 1.0f : -1.0f;
    const float a = -1 / (sign + normal.z);
    const float b = normal.x * normal.y * a;
    *tangent = V3(1 + sign * normal.x * normal.x * a, sign * b, -sign * normal.x);
    *bitangent = V3(b, sign + normal.y * normal.y * a, -normal.y);
}

// common/common_device.cuh:112-140 (Ray Tracing Gems ch. 6)
static inline V3 offsetRayOrigin(V3 p, V3 geometricNormal) {
    constexpr float kOrigin = 1.0f / 32.0f;
    constexpr float kFloatScale = 1.0f / 65536.0f;
    constexpr float kIntScale = 256.0f;
    const int32_t offsetInInt[3] = {
        f2i(kIntScale * geometricNormal.x),
        f2i(kIntScale * geometricNormal.y),
        f2i(kIntScale * geometricNormal.z) };
    auto addInt = [](float v, int32_t o) {
        const int32_t i = static_cast<int32_t>(f2bits(v)) + (v < 0 ? -1 : 1) * o;
        return bits2f(static_cast<uint32_t>(i));
    };
    const V3 newP1(addInt(p.x, offsetInInt[0]), addInt(p.y, offsetInInt[1]), addInt(p.z, offsetInInt[2]));
    const V3 newP2 = p + kFloatScale * geometricNormal;
    return V3(std::fabs(p.x) < kOrigin ? newP2.x : newP1.x,
              std::fabs(p.y) < kOrigin ? newP2.y : newP1.y,
              std::fabs(p.z) < kOrigin ? newP2.z : newP1.z);
}

// common/common_device.cuh:149-174
struct ReferenceFrame {
    V3 tangent, bitangent, normal;
    ReferenceFrame() {}
    explicit ReferenceFrame(V3 n) : normal(n) { makeCoordinateSystem(normal, &tangent, &bitangent); }
    ReferenceFrame(V3 n, V3 t) : tangent(t), normal(n) { bitangent = cross(normal, tangent); }
    V3 toLocal(V3 v) const { return V3(dot(tangent, v), dot(bitangent, v), dot(normal, v)); }
    V3 fromLocal(V3 v) const {
        return V3(dot(V3(tangent.x, bitangent.x, normal.x), v),
                  dot(V3(tangent.y, bitangent.y, normal.y), v),
                  dot(V3(tangent.z, bitangent.z, normal.z), v));
    }
};

// common/common_device.cuh:285-324
// applyBumpMapping, common/common_device.cuh:176-203
static inline void applyBumpMapping(V3 modNormalInTF, ReferenceFrame* frameToModify) {
    const float projLength = std::sqrt(modNormalInTF.x * modNormalInTF.x + modNormalInTF.y * modNormalInTF.y);
    if (projLength < 1e-3f) return;
    const float tiltAngle = gm_atan(projLength / modNormalInTF.z);
    float qSin, qCos;
    gm_sincos(tiltAngle / 2, &qSin, &qCos);
    const float qX = (-modNormalInTF.y / projLength) * qSin;
    const float qY = (modNormalInTF.x / projLength) * qSin;
    const float qW = qCos;
    const V3 modTangentInTF(1 - 2 * qY * qY, 2 * qX * qY, -2 * qY * qW);
    const V3 modBitangentInTF(2 * qX * qY, 1 - 2 * qX * qX, 2 * qX * qW);
    // matTFtoW = Matrix3x3(tangent, bitangent, normal) (columns); M * v = (dot(row_k, v))
    ReferenceFrame bumpShadingFrame;
    bumpShadingFrame.tangent = frameToModify->fromLocal(modTangentInTF);
    bumpShadingFrame.bitangent = frameToModify->fromLocal(modBitangentInTF);
    bumpShadingFrame.normal = frameToModify->fromLocal(modNormalInTF);
    *frameToModify = bumpShadingFrame;
}

static inline void concentricSampleDisk(float u0, float u1, float* dx, float* dy) {
    float r, theta;
    const float sx = 2 * u0 - 1;
    const float sy = 2 * u1 - 1;
    if (sx == 0 && sy == 0) { *dx = 0; *dy = 0; return; }
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
    gm_sincos(theta, &s, &c);
    *dx = r * c;
    *dy = r * s;
}
static inline V3 cosineSampleHemisphere(float u0, float u1) {
    float x, y;
    concentricSampleDisk(u0, u1, &x, &y);
    return V3(x, y, std::sqrt(fmax2(0.0f, 1.0f - x * x - y * y)));
}

// ---------------------------------------------------------------- BSDFs
// MaterialData (common/common_shared.h:1144-1177): every value is a texture slot (1-based index into the
// scene's textures) or, with slot 0, the sampled value of the reference's 1x1 immediate texture
// (common/common_host.cpp:1045-1073).
struct MaterialData {
    uint32_t bsdfType;   // 0 Lambert, 1 DiffuseAndSpecular, 2 SimplePBR
    float a[3];
    float b[3];
    float smoothness;
    float emittance[3];
    uint32_t hasEmittance;
    uint32_t texA = 0, texB = 0, texSmoothness = 0, texNormal = 0, texEmittance = 0;
    uint32_t bumpMapType = 0;   // 0 normal map, 1 two-channel normal map, 2 height map; | 0x100 left-handed
};
using TextureTable = std::vector<Texture>;

// mat.emittance read at a surface point (optix_restir_di_kernels.cu:595-599; path tracers likewise)
static inline RGB materialEmittance(const TextureTable& textures, const MaterialData& mat, V2 texCoord) {
    if (!mat.hasEmittance) return RGB(0.0f, 0.0f, 0.0f);
    if (mat.texEmittance) { const Texel4 t = textures[mat.texEmittance].sample(texCoord.x, texCoord.y); return RGB(t.x, t.y, t.z); }
    return RGB(mat.emittance[0], mat.emittance[1], mat.emittance[2]);
}

// readModifiedNormalFromNormalMap / ...2ch / ...FromHeightMap, common/common_device.cuh:205-240
static inline V3 readModifiedNormal(const TextureTable& textures, const MaterialData& mat, V2 texCoord) {
    const uint32_t kind = mat.bumpMapType & 0xFFu;
    const bool leftHanded = (mat.bumpMapType & 0x100u) != 0;
    V3 modLocalNormal;
    if (!mat.texNormal) {   // the 1x1 (0.5, 0.5, 1) normal texture of materials without a normal map (common_host.cpp:1399-1403)
        modLocalNormal = 2.0f * V3(0.5f, 0.5f, 1.0f) - V3(1.0f);
    }
    else if (kind == 0) {
        const Texel4 t = textures[mat.texNormal].sample(texCoord.x, texCoord.y);
        modLocalNormal = 2.0f * V3(t.x, t.y, t.z) - V3(1.0f);
    }
    else if (kind == 1) {
        const Texel4 t = textures[mat.texNormal].sample(texCoord.x, texCoord.y);
        const float x = 2.0f * t.x - 1.0f, y = 2.0f * t.y - 1.0f;
        const float z = std::sqrt(1.0f - pow2(x) - pow2(y));
        modLocalNormal = V3(x, y, z);
    }
    else {
        const Texture& tex = textures[mat.texNormal];
        const Texel4 h = tex.gatherR(texCoord.x, texCoord.y);
        constexpr float coeff = (5.0f / 1024);
        const float dhdu = (coeff * tex.width) * (h.y - h.x);
        const float dhdv = (coeff * tex.height) * (h.x - h.w);
        return normalize(V3(-dhdu, dhdv, 1));
    }
    if (leftHanded) modLocalNormal.y *= -1;
    return modLocalNormal;
}

// One struct covers LambertBRDF (common_device.cuh:335-374) and DiffuseAndSpecularBRDF /
// SimplePBR_BRDF (:443-776); dispatch follows BSDF::setup/evaluate (:890-963).
struct BSDF {
    uint32_t type = 0;
    RGB diffuseColor;     // Lambert: reflectance
    RGB specularF0Color;
    float roughness = 0;

    // setupBSDFBody<> :376-385, 778-826: every value is sample<>(texture, texCoord, mipLevel 0)
    void setup(const TextureTable& textures, const MaterialData& mat, V2 texCoord) {
        type = mat.bsdfType;
        RGB valA(mat.a[0], mat.a[1], mat.a[2]);
        if (mat.texA) { const Texel4 t = textures[mat.texA].sample(texCoord.x, texCoord.y); valA = RGB(t.x, t.y, t.z); }
        RGB valB(mat.b[0], mat.b[1], mat.b[2]);
        if (mat.texB && type != 0) { const Texel4 t = textures[mat.texB].sample(texCoord.x, texCoord.y); valB = RGB(t.x, t.y, t.z); }
        if (type == 0) {
            diffuseColor = valA;
        }
        else if (type == 1) {
            float smoothness = mat.smoothness;
            if (mat.texSmoothness) smoothness = textures[mat.texSmoothness].sample(texCoord.x, texCoord.y).x;
            diffuseColor = valA;
            specularF0Color = valB;
            roughness = 1 - fmin2(smoothness, 0.999f);               // :518-523, 802
        }
        else {
            const RGB baseColor = valA;
            const float smoothness = fmin2(1.0f - valB.y, 0.999f);          // :819
            const float metallic = valB.z;
            const float reflectance = 0.5f;                                 // :825
            diffuseColor = baseColor * (1 - metallic);                      // :772
            specularF0Color = RGB(0.16f * pow2(reflectance) * (1 - metallic)) + baseColor * metallic;
            roughness = 1 - smoothness;
        }
    }

    // GGXMicrofacetDistribution :444-508
    static float ggxD(float alpha_g, V3 m) {
        if (m.z <= 0.0f) return 0.0f;
        const float temp = pow2(m.x) + pow2(m.y) + pow2(m.z * alpha_g);
        return pow2(alpha_g) / (kPi * pow2(temp));
    }
    static float ggxSmithG1(float alpha_g, V3 v, V3 m) {
        if (dot(v, m) * v.z <= 0) return 0.0f;
        const float temp = pow2(alpha_g) * (pow2(v.x) + pow2(v.y)) / pow2(v.z);
        return 2 / (1 + std::sqrt(1 + temp));
    }
    static float ggxHeightCorrelatedSmithG(float alpha_g, V3 v1, V3 v2, V3 m) {
        const float a1 = pow2(alpha_g) * (pow2(v1.x) + pow2(v1.y)) / pow2(v1.z);
        const float a2 = pow2(alpha_g) * (pow2(v2.x) + pow2(v2.y)) / pow2(v2.z);
        const float Lambda1 = (-1 + std::sqrt(1 + a1)) / 2;
        const float Lambda2 = (-1 + std::sqrt(1 + a2)) / 2;
        const float chi1 = (dot(v1, m) / v1.z) > 0 ? 1.0f : 0.0f;
        const float chi2 = (dot(v2, m) / v2.z) > 0 ? 1.0f : 0.0f;
        return chi1 * chi2 / (1 + Lambda1 + Lambda2);
    }
    static float ggxEvaluatePDF(float alpha_g, V3 v, V3 m) { // :505-507
        return ggxSmithG1(alpha_g, v, m) * std::fabs(dot(v, m)) * ggxD(alpha_g, m) / std::fabs(v.z);
    }
    static float ggxSample(float alpha_g, V3 v, float u0, float u1, V3* m, float* mPDensity) { // :470-504
        const V3 sv = normalize(V3(alpha_g * v.x, alpha_g * v.y, v.z));
        const float distIn2D = std::sqrt(sv.x * sv.x + sv.y * sv.y);
        const float recDistIn2D = 1.0f / distIn2D;
        const V3 T1 = (sv.z < 0.9999f) ? V3(sv.y * recDistIn2D, -sv.x * recDistIn2D, 0) : V3(1, 0, 0);
        const V3 T2(T1.y * sv.z, -T1.x * sv.z, distIn2D);
        const float a = 1.0f / (1.0f + sv.z);
        const float r = std::sqrt(u0);
        const float phi = kPi * ((u1 < a) ? u1 / a : 1 + (u1 - a) / (1.0f - a));
        float sinPhi, cosPhi;
        gm_sincos(phi, &sinPhi, &cosPhi);
        const float P1 = r * cosPhi;
        const float P2 = r * sinPhi * ((u1 < a) ? 1.0f : sv.z);
        *m = P1 * T1 + P2 * T2 + std::sqrt(1.0f - P1 * P1 - P2 * P2) * sv;
        *m = normalize(V3(alpha_g * m->x, alpha_g * m->y, m->z));
        const float D = ggxD(alpha_g, *m);
        *mPDensity = ggxSmithG1(alpha_g, v, *m) * std::fabs(dot(v, *m)) * D / std::fabs(v.z);
        return D;
    }

    RGB evaluate(V3 vGiven, V3 vSampled) const {
        if (type == 0) { // LambertBRDF::evaluate :358-363
            if (vGiven.z * vSampled.z > 0) return diffuseColor / kPi;
            return RGB(0.0f);
        }
        // DiffuseAndSpecularBRDF::evaluate :648-690
        const float alpha_g = roughness * roughness;
        if (vSampled.z * vGiven.z <= 0) return RGB(0.0f);
        const bool entering = vGiven.z >= 0.0f;
        const V3 dirV = entering ? vGiven : -vGiven;
        const V3 dirL = entering ? vSampled : -vSampled;
        const V3 m = halfVector(dirL, dirV);
        const float dotLH = dot(dirL, m);
        const float oneMinusDotLH5 = pow5(1 - dotLH);
        const float D = ggxD(alpha_g, m);
        const float G = ggxHeightCorrelatedSmithG(alpha_g, dirL, dirV, m);
        constexpr float F90 = 1.0f;
        const RGB F = lerp3(specularF0Color, RGB(F90), oneMinusDotLH5);
        const float microfacetDenom = 4 * dirL.z * dirV.z;
        RGB specularValue = F * ((D * G) / microfacetDenom);
        if (G == 0) specularValue = RGB(0.0f);
        const float F_D90 = 0.5f * roughness + 2 * roughness * dotLH * dotLH;
        const float oneMinusDotVN5 = pow5(1 - dirV.z);
        const float oneMinusDotLN5 = pow5(1 - dirL.z);
        const float diffuseFresnelOut = lerpf(1.0f, F_D90, oneMinusDotVN5);
        const float diffuseFresnelIn = lerpf(1.0f, F_D90, oneMinusDotLN5);
        const RGB diffuseValue = diffuseColor *
            (diffuseFresnelOut * diffuseFresnelIn * lerpf(1.0f, 1.0f / 1.51f, roughness) / kPi);
        return diffuseValue + specularValue;
    }

    RGB sampleThroughput(V3 vGiven, float uDir0, float uDir1, V3* vSampled, float* dirPDensity) const {
        if (type == 0) { // LambertBRDF::sampleThroughput :348-357
            *vSampled = cosineSampleHemisphere(uDir0, uDir1);
            *dirPDensity = vSampled->z / kPi;
            if (vGiven.z <= 0.0f) vSampled->z *= -1;
            return diffuseColor;
        }
        // DiffuseAndSpecularBRDF::sampleThroughput :532-647
        const float alpha_g = roughness * roughness;
        const bool entering = vGiven.z >= 0.0f;
        V3 dirL;
        const V3 dirV = entering ? vGiven : -vGiven;
        const float oneMinusDotVN5 = pow5(1 - dirV.z);
        const float expectedF_D90 = 0.5f * roughness + 2 * roughness * vGiven.z * vGiven.z;
        const float expectedDiffuseFresnel = lerpf(1.0f, expectedF_D90, oneMinusDotVN5);
        const float iBaseColor = sRGB_calcLuminance(diffuseColor) * pow2(expectedDiffuseFresnel) *
            lerpf(1.0f, 1.0f / 1.51f, roughness);
        const float expectedOneMinusDotVH5 = pow5(1 - dirV.z);
        const float iSpecularF0 = sRGB_calcLuminance(specularF0Color);
        const float diffuseWeight = iBaseColor;
        const float specularWeight = lerpf(iSpecularF0, 1.0f, expectedOneMinusDotVH5);
        const float sumWeights = diffuseWeight + specularWeight;
        if (sumWeights == 0.0f) { *dirPDensity = 0.0f; return RGB(0.0f); }
        const float uComponent = uDir1;
        float diffuseDirPDF, specularDirPDF;
        V3 m;
        float dotLH;
        float D;
        if (sumWeights * uComponent < diffuseWeight) {
            uDir1 = (sumWeights * uComponent - 0) / diffuseWeight;
            dirL = cosineSampleHemisphere(uDir0, uDir1);
            diffuseDirPDF = dirL.z / kPi;
            m = halfVector(dirL, dirV);
            dotLH = fmin2(dot(dirL, m), 1.0f);
            const float commonPDFTerm = 1.0f / (4 * dotLH);
            specularDirPDF = commonPDFTerm * ggxEvaluatePDF(alpha_g, dirV, m);
            D = ggxD(alpha_g, m);
        }
        else {
            uDir1 = (sumWeights * uComponent - diffuseWeight) / specularWeight;
            float mPDF;
            D = ggxSample(alpha_g, dirV, uDir0, uDir1, &m, &mPDF);
            const float dotVH = fmin2(dot(dirV, m), 1.0f);
            dotLH = dotVH;
            dirL = 2 * dotVH * m - dirV;
            if (dirL.z * dirV.z <= 0) { *dirPDensity = 0.0f; return RGB(0.0f); }
            const float commonPDFTerm = 1.0f / (4 * dotLH);
            specularDirPDF = commonPDFTerm * mPDF;
            diffuseDirPDF = dirL.z / kPi;
        }
        const float oneMinusDotLH5 = pow5(1 - dotLH);
        const float G = ggxHeightCorrelatedSmithG(alpha_g, dirL, dirV, m);
        constexpr float F90 = 1.0f;
        const RGB F = lerp3(specularF0Color, RGB(F90), oneMinusDotLH5);
        const float microfacetDenom = 4 * dirL.z * dirV.z;
        RGB specularValue = F * ((D * G) / microfacetDenom);
        if (G == 0) specularValue = RGB(0.0f);
        const float F_D90 = 0.5f * roughness + 2 * roughness * dotLH * dotLH;
        const float oneMinusDotLN5 = pow5(1 - dirL.z);
        const float diffuseFresnelOut = lerpf(1.0f, F_D90, oneMinusDotVN5);
        const float diffuseFresnelIn = lerpf(1.0f, F_D90, oneMinusDotLN5);
        const RGB diffuseValue = diffuseColor *
            (diffuseFresnelOut * diffuseFresnelIn * lerpf(1.0f, 1.0f / 1.51f, roughness) / kPi);
        RGB ret = diffuseValue + specularValue;
        *vSampled = entering ? dirL : -dirL;
        *dirPDensity = (diffuseDirPDF * diffuseWeight + specularDirPDF * specularWeight) / sumWeights;
        ret *= dirL.z / *dirPDensity;
        return ret;
    }

    float evaluatePDF(V3 vGiven, V3 vSampled) const {
        if (type == 0) { // :364-369
            if (vGiven.z * vSampled.z > 0) return std::fabs(vSampled.z) / kPi;
            return 0.0f;
        }
        // :691-734
        const float alpha_g = roughness * roughness;
        const bool entering = vGiven.z >= 0.0f;
        const V3 dirV = entering ? vGiven : -vGiven;
        const V3 dirL = entering ? vSampled : -vSampled;
        const V3 m = halfVector(dirL, dirV);
        const float dotLH = dot(dirL, m);
        const float commonPDFTerm = 1.0f / (4 * dotLH);
        const float expectedF_D90 = 0.5f * roughness + 2 * roughness * vGiven.z * vGiven.z;
        const float oneMinusDotVN5 = pow5(1 - dirV.z);
        const float expectedDiffuseFresnel = lerpf(1.0f, expectedF_D90, oneMinusDotVN5);
        const float iBaseColor = sRGB_calcLuminance(diffuseColor) * pow2(expectedDiffuseFresnel) *
            lerpf(1.0f, 1.0f / 1.51f, roughness);
        const float expectedOneMinusDotVH5 = pow5(1 - dirV.z);
        const float iSpecularF0 = sRGB_calcLuminance(specularF0Color);
        const float diffuseWeight = iBaseColor;
        const float specularWeight = lerpf(iSpecularF0, 1.0f, expectedOneMinusDotVH5);
        const float sumWeights = diffuseWeight + specularWeight;
        if (sumWeights == 0.0f) return 0.0f;
        const float diffuseDirPDF = dirL.z / kPi;
        const float specularDirPDF = commonPDFTerm * ggxEvaluatePDF(alpha_g, dirV, m);
        return (diffuseDirPDF * diffuseWeight + specularDirPDF * specularWeight) / sumWeights;
    }

    RGB evaluateDHReflectanceEstimate(V3 vGiven) const {
        if (type == 0) return diffuseColor; // :371-373
        // :736-764
        const bool entering = vGiven.z >= 0.0f;
        const V3 dirV = entering ? vGiven : -vGiven;
        const float expectedCosTheta_d = dirV.z;
        const float expectedF_D90 = 0.5f * roughness + 2 * roughness * pow2(expectedCosTheta_d);
        const float oneMinusDotVN5 = pow5(1 - dirV.z);
        const float expectedDiffFGiven = lerpf(1.0f, expectedF_D90, oneMinusDotVN5);
        const float expectedDiffFSampled = 1.0f;
        const RGB diffuseDHR = diffuseColor *
            expectedDiffFGiven * expectedDiffFSampled * lerpf(1.0f, 1.0f / 1.51f, roughness);
        const float expectedOneMinusDotVH5 = pow5(1 - dirV.z) * (1 - roughness);
        const RGB specularDHR = lerp3(specularF0Color, RGB(1.0f), expectedOneMinusDotVH5);
        return vmin(diffuseDHR + specularDHR, RGB(1.0f));
    }

    void getSurfaceParameters(RGB* diffuseReflectance, RGB* specularReflectance, float* rough) const {
        if (type == 0) { *diffuseReflectance = diffuseColor; *specularReflectance = RGB(0.0f); *rough = 1.0f; }
        else { *diffuseReflectance = diffuseColor; *specularReflectance = specularF0Color; *rough = roughness; }
    }
};

} // namespace orc
