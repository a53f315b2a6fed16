// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// orc_scene.h: flat scene description + the light-sampling / direct-lighting functions of
// restir_di/restir_di_shared.h:320-582 and the emitter-importance computation of
// common/gpu_kernels/compute_light_probs.cu.
#pragma once
#include <functional>
#include <vector>
#include "orc_shared.h"

namespace orc {

struct Vertex { // common/common_shared.h:1109-1114
    V3 position;
    V3 normal;
    V3 texCoord0Dir;
    V2 texCoord;
};
static_assert(sizeof(Vertex) == 44, "Vertex must be 44 bytes");
struct Triangle { uint32_t index0, index1, index2; }; // :1121-1123

struct GeometryInstanceData { // common/common_shared.h:1179-1193
    std::vector<Vertex> vertexBuffer;
    std::vector<Triangle> triangleBuffer;
    std::vector<float> emitterPrimWeights, emitterPrimCDF;
    DiscreteDistribution1D emitterPrimDist;
    uint32_t materialSlot = 0;
    uint32_t geomInstSlot = 0;
};

struct InstanceData { // common/common_shared.h:1243-1251
    M34 transform;
    M34 curToPrevTransform;
    M3 normalMatrix;
    float uniformScale = 1;
    std::vector<uint32_t> geomInstSlots;
    std::vector<float> lightGeomInstWeights, lightGeomInstCDF;
    DiscreteDistribution1D lightGeomInstDist;
};

struct PerspectiveCamera { // restir_di/restir_di_shared.h:45-60
    float aspect;
    float fovY;
    V3 position;
    M3 orientation;
    V2 calcScreenPosition(V3 posInWorld) const {
        const M3 invOri = invert(orientation);
        const V3 posInView = mul(invOri, posInWorld - position);
        const V2 posAtZ1{ posInView.x / posInView.z, posInView.y / posInView.z };
        const float h = 2 * gm_tan(fovY / 2);
        const float w = aspect * h;
        return V2{ 1 - (posAtZ1.x + 0.5f * w) / w, 1 - (posAtZ1.y + 0.5f * h) / h };
    }
};

struct EnvLight {
    const float* texels = nullptr; // float4 lat-long
    uint32_t w = 0, h = 0;
    RegularConstantContinuousDistribution2D importanceMap;
    bool present() const { return texels != nullptr; }
    // Nearest-texel point fetch: the build's definition of tex2DLod on the env map (the hardware
    // bilinear filter of the reference is not reproducible; SURVEY.md section 7 "Texture sampling").
    RGB fetch(float u, float v) const {
        uint32_t x = f2u(u * w); if (x > w - 1) x = w - 1;
        uint32_t y = f2u(v * h); if (y > h - 1) y = h - 1;
        const float* t = texels + 4 * (static_cast<size_t>(y) * w + x);
        return RGB(t[0], t[1], t[2]);
    }
};

struct Scene {
    std::vector<MaterialData> materials;
    TextureTable textures;   // indexed by slot, slot 0 unused
    std::vector<GeometryInstanceData> geomInsts;
    std::vector<InstanceData> insts;
    std::vector<float> lightInstWeights, lightInstCDF;
    DiscreteDistribution1D lightInstDist;
    EnvLight env;

    // exclusive scan + finalize (cubd::DeviceScan::ExclusiveSum + finalizeDiscreteDistribution1D,
    // common/common_host.h:1159-1163; compute_light_probs.cu:206-212).  Serial left-to-right sum.
    static void buildCDF(const std::vector<float>& w, std::vector<float>& cdf, DiscreteDistribution1D* d) {
        cdf.resize(w.size());
        float acc = 0.0f;
        for (size_t i = 0; i < w.size(); ++i) { cdf[i] = acc; acc += w[i]; }
        d->weights = w.data();
        d->CDF = cdf.data();
        d->numValues = static_cast<uint32_t>(w.size());
        d->integralValue = w.empty() ? 0.0f : cdf[w.size() - 1] + w[w.size() - 1];
    }

    // compute_light_probs.cu:22-46 computeTriangleImportance (constant emittance texture)
    float computeTriangleImportance(const GeometryInstanceData& g, uint32_t triIndex) const {
        const MaterialData& mat = materials[g.materialSlot];
        const Triangle& tri = g.triangleBuffer[triIndex];
        const Vertex& v0 = g.vertexBuffer[tri.index0];
        const Vertex& v1 = g.vertexBuffer[tri.index1];
        const Vertex& v2 = g.vertexBuffer[tri.index2];
        const V3 normal = cross(v1.position - v0.position, v2.position - v0.position);
        const float area = 0.5f * length(normal);
        RGB emittanceEstimate(0.0f, 0.0f, 0.0f);
        auto fetch = [&](const Vertex& v) {   // tex2DLod<float4>(mat.emittance, v.texCoord, 0)
            if (mat.texEmittance) { const Texel4 t = textures[mat.texEmittance].sample(v.texCoord.x, v.texCoord.y); return RGB(t.x, t.y, t.z); }
            return RGB(mat.emittance[0], mat.emittance[1], mat.emittance[2]);
        };
        emittanceEstimate += fetch(v0);
        emittanceEstimate += fetch(v1);
        emittanceEstimate += fetch(v2);
        emittanceEstimate /= 3;
        return sRGB_calcLuminance(emittanceEstimate) * area;
    }

    // common/common_host.h:1102-1266 setupLightGeomDistributions
    void setupLightGeomDistributions() {
        for (auto& g : geomInsts) {
            const MaterialData& mat = materials[g.materialSlot];
            g.emitterPrimWeights.clear();
            g.emitterPrimCDF.clear();
            g.emitterPrimDist = DiscreteDistribution1D();
            if (!mat.hasEmittance) continue; // only emitter geomInsts own a distribution (common_host.cpp:1877-1885)
            g.emitterPrimWeights.resize(g.triangleBuffer.size());
            for (uint32_t t = 0; t < g.triangleBuffer.size(); ++t)
                g.emitterPrimWeights[t] = computeTriangleImportance(g, t);
            buildCDF(g.emitterPrimWeights, g.emitterPrimCDF, &g.emitterPrimDist);
        }
        for (auto& inst : insts) {
            inst.lightGeomInstWeights.clear();
            inst.lightGeomInstCDF.clear();
            inst.lightGeomInstDist = DiscreteDistribution1D();
            bool hasEmitter = false;
            for (uint32_t slot : inst.geomInstSlots)
                if (materials[geomInsts[slot].materialSlot].hasEmittance) hasEmitter = true;
            if (!hasEmitter) continue; // common_host.cpp:2616-2625
            inst.lightGeomInstWeights.resize(inst.geomInstSlots.size());
            for (size_t i = 0; i < inst.geomInstSlots.size(); ++i)  // compute_light_probs.cu:86-93
                inst.lightGeomInstWeights[i] = geomInsts[inst.geomInstSlots[i]].emitterPrimDist.integral();
            buildCDF(inst.lightGeomInstWeights, inst.lightGeomInstCDF, &inst.lightGeomInstDist);
        }
    }
    // common/common_host.h:1268-1359 setupLightInstDistribution; compute_light_probs.cu:134-142
    void setupLightInstDistribution() {
        lightInstWeights.resize(insts.size());
        for (size_t i = 0; i < insts.size(); ++i) {
            const InstanceData& inst = insts[i];
            // Matrix4x4::decompose scale.x = length(column 0)   basic_types.h:4643-4646
            const float uniformScale = length(V3(inst.transform.m[0], inst.transform.m[4], inst.transform.m[8]));
            lightInstWeights[i] = pow2(uniformScale) * inst.lightGeomInstDist.integral();
        }
        buildCDF(lightInstWeights, lightInstCDF, &lightInstDist);
    }
};

// ---------------------------------------------------------------- sampleLight<false>
// restir_di/restir_di_shared.h:320-516 (useSolidAngleSampling = false)
// Spherical-triangle quantities of (pA, pB, pC) seen from a point: restir_di_shared.h:430-445 and
// path_tracing_shared.h:551-561 (acos / sin / cos through the math contract).
struct SphericalTriangle {
    V3 A, B, C;
    float cos_c, cosAlpha, alpha, sinAlpha, sphArea;
};
static inline SphericalTriangle sphericalTriangle(V3 pA, V3 pB, V3 pC, V3 refPoint) {
    SphericalTriangle t;
    t.A = normalize(pA - refPoint);
    t.B = normalize(pB - refPoint);
    t.C = normalize(pC - refPoint);
    const V3 cAB = normalize(cross(t.A, t.B));
    const V3 cBC = normalize(cross(t.B, t.C));
    const V3 cCA = normalize(cross(t.C, t.A));
    t.cos_c = dot(t.A, t.B);
    t.cosAlpha = -dot(cAB, cCA);
    const float cosBeta = -dot(cBC, cAB);
    const float cosGamma = -dot(cCA, cBC);
    t.alpha = gm_acos(t.cosAlpha);
    t.sinAlpha = std::sqrt(1 - pow2(t.cosAlpha));
    t.sphArea = t.alpha + gm_acos(cosBeta) + gm_acos(cosGamma) - kPi;
    return t;
}

static inline void sampleLight(
    const Scene& scene, float envLightRotation, float envLightPowerCoeff,
    V3 shadingPoint, float ul, bool sampleEnvLight, float u0, float u1,
    LightSample* lightSample, float* areaPDensity, bool useSolidAngleSampling = false)
{
    bool hasTexEmittance = false;
    RGB texValue(0.0f);
    RGB emittance(0.0f, 0.0f, 0.0f);
    if (sampleEnvLight) {
        float u, v, uvPDF;
        scene.env.importanceMap.sample(u0, u1, &u, &v, &uvPDF);
        const float phi = 2 * kPi * u;
        const float theta = kPi * v;
        float posPhi = phi - envLightRotation;
        posPhi = posPhi - std::floor(posPhi / (2 * kPi)) * 2 * kPi;
        const V3 direction = fromPolarYUp(posPhi, theta);
        const V3 position(direction.x, direction.y, direction.z);
        lightSample->position = position;
        lightSample->atInfinity = true;
        lightSample->normal = -position;
        const float sinTheta = gm_sin(theta);
        if (sinTheta == 0.0f) { *areaPDensity = 0.0f; return; }
        *areaPDensity = uvPDF / (2 * kPi * kPi * sinTheta);
        hasTexEmittance = true;
        texValue = scene.env.fetch(u, v);
        emittance = RGB(kPi * envLightPowerCoeff);
    }
    else {
        float lightProb = 1.0f;
        float instProb, uGeomInst;
        const uint32_t instSlot = scene.lightInstDist.sample(ul, &instProb, &uGeomInst);
        lightProb *= instProb;
        const InstanceData& inst = scene.insts[instSlot];
        if (instProb == 0.0f) { *areaPDensity = 0.0f; return; }

        float geomInstProb, uPrim;
        const uint32_t geomInstIndexInInst = inst.lightGeomInstDist.sample(uGeomInst, &geomInstProb, &uPrim);
        const uint32_t geomInstSlot = inst.geomInstSlots[geomInstIndexInInst];
        lightProb *= geomInstProb;
        const GeometryInstanceData& geomInst = scene.geomInsts[geomInstSlot];
        if (geomInstProb == 0.0f) { *areaPDensity = 0.0f; return; }

        float primProb;
        const uint32_t primIndex = geomInst.emitterPrimDist.sample(uPrim, &primProb);
        lightProb *= primProb;

        const MaterialData& mat = scene.materials[geomInst.materialSlot];
        const Triangle& tri = geomInst.triangleBuffer[primIndex];
        const Vertex& vA = geomInst.vertexBuffer[tri.index0];
        const Vertex& vB = geomInst.vertexBuffer[tri.index1];
        const Vertex& vC = geomInst.vertexBuffer[tri.index2];
        const V3 pA = xfmPoint(inst.transform, vA.position);
        const V3 pB = xfmPoint(inst.transform, vB.position);
        const V3 pC = xfmPoint(inst.transform, vC.position);
        const V3 geomNormal = cross(pB - pA, pC - pA);

        float bcA, bcB, bcC;
        if (useSolidAngleSampling) {   // :419-483: uniform in the solid angle the triangle subtends from the shading point
            const SphericalTriangle st = sphericalTriangle(pA, pB, pC, shadingPoint);
            const auto project = [](V3 a, V3 b) { return normalize(a - dot(a, b) * b); };
            const float sphAreaHat = st.sphArea * u0;
            const float s = gm_sin(sphAreaHat - st.alpha);
            const float t = gm_cos(sphAreaHat - st.alpha);
            const float uu = t - st.cosAlpha;
            const float vv = s + st.sinAlpha * st.cos_c;
            const float q = ((vv * t - uu * s) * st.cosAlpha - vv) / ((vv * s + uu * t) * st.sinAlpha);
            const V3 cHat = q * st.A + std::sqrt(1 - pow2(q)) * project(st.C, st.A);
            const float z = 1 - u1 * (1 - dot(cHat, st.B));
            const V3 P = z * st.B + std::sqrt(1 - pow2(z)) * project(cHat, st.B);
            const V3 dir = P;
            float dist;
            {   // restoreBarycentrics
                const V3 eAB = pB - pA;
                const V3 eAC = pC - pA;
                const V3 pVec = cross(dir, eAC);
                const float recDet = 1.0f / dot(eAB, pVec);
                const V3 tVec = shadingPoint - pA;
                bcB = dot(tVec, pVec) * recDet;
                const V3 qVec = cross(tVec, eAB);
                bcC = dot(dir, qVec) * recDet;
                dist = dot(eAC, qVec) * recDet;
            }
            bcA = 1 - (bcB + bcC);
            const float dirPDF = 1 / st.sphArea;
            const V3 gn = normalize(geomNormal);
            const float lpCos = -dot(dir, gn);
            if (lpCos > 0 && finitef(dirPDF)) *areaPDensity = lightProb * (dirPDF * lpCos / pow2(dist));
            else *areaPDensity = 0.0f;
        }
        else {
            // A Low-Distortion Map Between Triangle and Square (:485-498)
            bcA = 0.5f * u0;
            bcB = 0.5f * u1;
            const float offset = bcB - bcA;
            if (offset > 0) bcB += offset;
            else bcA -= offset;
            bcC = 1 - (bcA + bcB);
            const float recArea = 2.0f / length(geomNormal);
            *areaPDensity = lightProb * recArea;
        }

        lightSample->position = bcA * pA + bcB * pB + bcC * pC;
        lightSample->atInfinity = false;
        lightSample->normal = bcA * vA.normal + bcB * vB.normal + bcC * vC.normal;
        lightSample->normal = normalize(mul(inst.normalMatrix, lightSample->normal));
        if (mat.hasEmittance) {   // :504-508
            hasTexEmittance = true;
            emittance = RGB(1.0f, 1.0f, 1.0f);
            const V2 texCoord{ bcA * vA.texCoord.x + bcB * vB.texCoord.x + bcC * vC.texCoord.x,
                               bcA * vA.texCoord.y + bcB * vB.texCoord.y + bcC * vC.texCoord.y };
            if (mat.texEmittance) { const Texel4 t = scene.textures[mat.texEmittance].sample(texCoord.x, texCoord.y); texValue = RGB(t.x, t.y, t.z); }
            else texValue = RGB(mat.emittance[0], mat.emittance[1], mat.emittance[2]);
        }
    }
    if (hasTexEmittance)
        emittance *= texValue;
    lightSample->emittance = emittance;
}

// A visibility query: returns true iff the open segment (tmin, tmax) along (org, dir) is unoccluded.
using VisibilityFn = std::function<bool(V3 org, V3 dir, float tmin, float tmax)>;

// restir_di/restir_di_shared.h:518-557
static inline RGB performDirectLighting(
    bool withVisibility, const VisibilityFn& visFn,
    V3 shadingPoint, V3 vOutLocal, const ReferenceFrame& shadingFrame,
    const BSDF& bsdf, const LightSample& lightSample)
{
    V3 shadowRayDir = lightSample.atInfinity ? lightSample.position : (lightSample.position - shadingPoint);
    const float dist2 = sqLength(shadowRayDir);
    float dist = std::sqrt(dist2);
    shadowRayDir /= dist;
    const V3 shadowRayDirLocal = shadingFrame.toLocal(shadowRayDir);
    const float lpCos = dot(-shadowRayDir, lightSample.normal);
    const float spCos = shadowRayDirLocal.z;
    float visibility = 1.0f;
    if (withVisibility) {
        if (lightSample.atInfinity) dist = 1e+10f;
        visibility = visFn(shadingPoint, shadowRayDir, 0.0f, dist * 0.9999f) ? 1.0f : 0.0f;
    }
    if (visibility > 0 && lpCos > 0) {
        const RGB Le = lightSample.emittance / kPi;
        const RGB fsValue = bsdf.evaluate(vOutLocal, shadowRayDirLocal);
        const float G = lpCos * std::fabs(spCos) / dist2;
        return fsValue * Le * G;
    }
    return RGB(0.0f, 0.0f, 0.0f);
}

// restir_di/restir_di_shared.h:559-582
static inline bool evaluateVisibility(const VisibilityFn& visFn, V3 shadingPoint, const LightSample& lightSample) {
    V3 shadowRayDir = lightSample.atInfinity ? lightSample.position : (lightSample.position - shadingPoint);
    const float dist2 = sqLength(shadowRayDir);
    float dist = std::sqrt(dist2);
    shadowRayDir /= dist;
    if (lightSample.atInfinity) dist = 1e+10f;
    return visFn(shadingPoint, shadowRayDir, 0.0f, dist * 0.9999f);
}

// restir_di/restir_di_main.cpp:1487-1542: Halton(2,3) -> concentric disk, 1024 entries.
// (the host code uses cos/sin of <cmath>; the contract substitutes gm_sincos.)
static inline std::vector<V2> makeSpatialNeighborDeltas() {
    auto halton = [](uint32_t base, uint32_t idx) {
        const float recBase = 1.0f / base;
        float ret = 0.0f, scale = 1.0f;
        while (idx) { scale *= recBase; ret += (idx % base) * scale; idx /= base; }
        return ret;
    };
    std::vector<V2> t(1024);
    for (uint32_t i = 0; i < 1024; ++i)
        concentricSampleDisk(halton(2, i), halton(3, i), &t[i].x, &t[i].y);
    return t;
}

} // namespace orc
