// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// orc_pathtrace.h: CPU restatement of the baseline unidirectional path tracer (NEE + BSDF sampling
// with power-heuristic MIS, Russian roulette, path length limit):
//   performNextEventEstimation   path_tracing/gpu_kernels/optix_pathtracing_kernels.cu:18-72
//   pathTrace_rayGen_generic     :74-208
//   pathTrace_closestHit_generic :210-296
//   miss program                 :306-341
//   computeSurfacePoint (x2)     path_tracing/path_tracing_shared.h (same text as
//                                restir_di/restir_di_shared.h:584-721)
#pragma once
#include "orc_restir.h"
#include "orc_regir.h"

namespace orc {

// restir_di_shared.h:682-721 (the ray-generation variant: object-space interpolation)
static inline void computeSurfacePointRG(const InstanceData& inst, const GeometryInstanceData& geomInst,
                                         uint32_t primIndex, float bcB, float bcC,
                                         V3* positionInWorld, V3* shadingNormalInWorld, V3* texCoord0DirInWorld,
                                         V3* geometricNormalInWorld, V2* texCoord) {
    const Triangle& tri = geomInst.triangleBuffer[primIndex];
    const Vertex& vA = geomInst.vertexBuffer[tri.index0];
    const Vertex& vB = geomInst.vertexBuffer[tri.index1];
    const Vertex& vC = geomInst.vertexBuffer[tri.index2];
    const float bcA = 1 - (bcB + bcC);
    const V3 positionInObj = bcA * vA.position + bcB * vB.position + bcC * vC.position;
    *positionInWorld = xfmPoint(inst.transform, positionInObj);
    *geometricNormalInWorld = normalize(mul(inst.normalMatrix, cross(vB.position - vA.position, vC.position - vA.position)));
    const V3 shadingNormalInObj = bcA * vA.normal + bcB * vB.normal + bcC * vC.normal;
    const V3 texCoord0DirInObj = bcA * vA.texCoord0Dir + bcB * vB.texCoord0Dir + bcC * vC.texCoord0Dir;
    *texCoord = V2{ bcA * vA.texCoord.x + bcB * vB.texCoord.x + bcC * vC.texCoord.x,
                    bcA * vA.texCoord.y + bcB * vB.texCoord.y + bcC * vC.texCoord.y };
    *shadingNormalInWorld = normalize(mul(inst.normalMatrix, shadingNormalInObj));
    *texCoord0DirInWorld = xfmVector(inst.transform, texCoord0DirInObj);
    *texCoord0DirInWorld = normalize(*texCoord0DirInWorld - dot(*shadingNormalInWorld, *texCoord0DirInWorld) * *shadingNormalInWorld);
    if (!allFinite(*shadingNormalInWorld)) {
        *geometricNormalInWorld = V3(0, 0, 1);
        *shadingNormalInWorld = V3(0, 0, 1);
        *texCoord0DirInWorld = V3(1, 0, 0);
    }
    if (!allFinite(*texCoord0DirInWorld)) {
        V3 bitangent;
        makeCoordinateSystem(*shadingNormalInWorld, texCoord0DirInWorld, &bitangent);
    }
}

// restir_di_shared.h:584-680 with computeHypotheticalAreaPDensity = true, useSolidAngleSampling = false
// (the closest-hit variant: world-space interpolation + hypothetical light pdf)
static inline void computeSurfacePointCH(const Scene& scene, bool envEnabled, uint32_t instSlot,
                                         const InstanceData& inst, const GeometryInstanceData& geomInst,
                                         uint32_t primIndex, float bcB, float bcC,
                                         V3* positionInWorld, V3* shadingNormalInWorld, V3* texCoord0DirInWorld,
                                         V3* geometricNormalInWorld, V2* texCoord, float* hypAreaPDensity,
                                         bool useSolidAngleSampling = false, V3 referencePoint = V3(0.0f)) {
    (void)instSlot;
    const Triangle& tri = geomInst.triangleBuffer[primIndex];
    const Vertex& vA = geomInst.vertexBuffer[tri.index0];
    const Vertex& vB = geomInst.vertexBuffer[tri.index1];
    const Vertex& vC = geomInst.vertexBuffer[tri.index2];
    const V3 pA = xfmPoint(inst.transform, vA.position);
    const V3 pB = xfmPoint(inst.transform, vB.position);
    const V3 pC = xfmPoint(inst.transform, vC.position);
    const float bcA = 1 - (bcB + bcC);
    *positionInWorld = bcA * pA + bcB * pB + bcC * pC;
    const V3 shadingNormalInObj = bcA * vA.normal + bcB * vB.normal + bcC * vC.normal;
    const V3 texCoord0DirInObj = bcA * vA.texCoord0Dir + bcB * vB.texCoord0Dir + bcC * vC.texCoord0Dir;
    *texCoord = V2{ bcA * vA.texCoord.x + bcB * vB.texCoord.x + bcC * vC.texCoord.x,
                    bcA * vA.texCoord.y + bcB * vB.texCoord.y + bcC * vC.texCoord.y };
    *geometricNormalInWorld = cross(pB - pA, pC - pA);
    const float area = 0.5f * length(*geometricNormalInWorld);
    *geometricNormalInWorld = *geometricNormalInWorld / (2 * area);
    *shadingNormalInWorld = normalize(mul(inst.normalMatrix, shadingNormalInObj));
    *texCoord0DirInWorld = normalize(xfmVector(inst.transform, texCoord0DirInObj));
    if (!allFinite(*shadingNormalInWorld)) {
        *shadingNormalInWorld = V3(0, 0, 1);
        *texCoord0DirInWorld = V3(1, 0, 0);
    }
    if (!allFinite(*texCoord0DirInWorld)) {
        V3 bitangent;
        makeCoordinateSystem(*shadingNormalInWorld, texCoord0DirInWorld, &bitangent);
    }
    float lightProb = 1.0f;
    if (envEnabled) lightProb *= (1 - 0.25f);
    const float instImportance = inst.lightGeomInstDist.integral();
    lightProb *= (pow2(inst.uniformScale) * instImportance) / scene.lightInstDist.integral();
    lightProb *= geomInst.emitterPrimDist.integral() / instImportance;
    if (!finitef(lightProb)) { *hypAreaPDensity = 0.0f; return; }
    lightProb *= geomInst.emitterPrimDist.evaluatePMF(primIndex);
    if (useSolidAngleSampling) {   // path_tracing_shared.h:550-568
        const SphericalTriangle st = sphericalTriangle(pA, pB, pC, referencePoint);
        const float dirPDF = 1.0f / st.sphArea;
        V3 refDir = referencePoint - *positionInWorld;
        const float dist2ToRefPoint = sqLength(refDir);
        refDir /= std::sqrt(dist2ToRefPoint);
        const float lpCos = dot(refDir, *geometricNormalInWorld);
        if (lpCos > 0 && finitef(dirPDF)) *hypAreaPDensity = lightProb * (dirPDF * lpCos / dist2ToRefPoint);
        else *hypAreaPDensity = 0.0f;
    }
    else *hypAreaPDensity = lightProb / area;
}

struct PathTraceParams {
    const Scene* scene; const WorldAccel* accel;
    const gfx_restir_static_params* s; const gfx_restir_frame_params* f;
    PerspectiveCamera camera;
    uint32_t maxPathLength;
    const RegirState* regir = nullptr;   // non-null: pathTraceReGIR (regir/gpu_kernels/optix_pathtracing_kernels.cu:425-433)
    const Params* restir = nullptr;      // non-null (NRC tracer only): the first vertex's NEE is the pixel's ReSTIR DI reservoir (orc_nrc.h)
    bool envEnabled() const { return s->envLightTexture != nullptr && f->enableEnvLight; }
};

static inline float envEvaluatePDF(const EnvLight& env, float d0, float d1) { // common_shared.h:380-383
    const uint32_t idx1D = mapPrimarySampleToDiscrete(d1, env.importanceMap.top.numValues);
    return env.importanceMap.top.evaluatePDF(d1) * env.importanceMap.row(idx1D).evaluatePDF(d0);
}

// optix_pathtracing_kernels.cu:18-72
static inline RGB performNextEventEstimation(const PathTraceParams& p, const VisibilityFn& visFn, V3 shadingPoint, V3 vOutLocal,
                                             const ReferenceFrame& shadingFrame, const BSDF& bsdf, PCG32RNG& rng) {
    const Scene& scene = *p.scene;
    RGB ret(0.0f);
    if (p.regir) { // regir/gpu_kernels/optix_pathtracing_kernels.cu:90-102
        Params rp; rp.scene = p.scene; rp.accel = p.accel; rp.s = p.s; rp.f = p.f;
        LightSample lightSample;
        float recProbDensityEstimate;
        const RGB unshadowedContribution = sampleFromCell(rp, *p.regir, shadingPoint, vOutLocal, shadingFrame, bsdf, rng,
                                                          &lightSample, &recProbDensityEstimate);
        if (recProbDensityEstimate > 0.0f) {
            const float visibility = evaluateVisibility(visFn, shadingPoint, lightSample) ? 1.0f : 0.0f;
            ret = unshadowedContribution * (visibility * recProbDensityEstimate);
        }
        return ret;
    }
    float uLight = rng.getFloat0cTo1o();
    bool selectEnvLight = false;
    float probToSampleCurLightType = 1.0f;
    constexpr float probToSampleEnvLight = 0.25f;
    if (p.envEnabled()) {
        if (scene.lightInstDist.integral() > 0.0f) {
            if (uLight < probToSampleEnvLight) { probToSampleCurLightType = probToSampleEnvLight; uLight /= probToSampleCurLightType; selectEnvLight = true; }
            else { probToSampleCurLightType = 1.0f - probToSampleEnvLight; uLight = (uLight - probToSampleEnvLight) / probToSampleCurLightType; }
        }
        else selectEnvLight = true;
    }
    LightSample lightSample;
    float areaPDensity;
    const float u0 = rng.getFloat0cTo1o();
    const float u1 = rng.getFloat0cTo1o();
    sampleLight(scene, p.f->envLightRotation, p.f->envLightPowerCoeff, shadingPoint, uLight, selectEnvLight, u0, u1, &lightSample, &areaPDensity,
                p.f->useSolidAngleSampling != 0);
    areaPDensity *= probToSampleCurLightType;
    float misWeight = 1.0f;
    {
        V3 shadowRay = lightSample.atInfinity ? lightSample.position : (lightSample.position - shadingPoint);
        const float dist2 = sqLength(shadowRay);
        shadowRay /= std::sqrt(dist2);
        const V3 vInLocal = shadingFrame.toLocal(shadowRay);
        const float lpCos = std::fabs(dot(shadowRay, lightSample.normal));
        float bsdfPDensity = bsdf.evaluatePDF(vOutLocal, vInLocal) * lpCos / dist2;
        if (!finitef(bsdfPDensity)) bsdfPDensity = 0.0f;
        const float lightPDensity = areaPDensity;
        misWeight = pow2(lightPDensity) / (pow2(bsdfPDensity) + pow2(lightPDensity));
    }
    if (areaPDensity > 0.0f)
        ret = performDirectLighting(true, visFn, shadingPoint, vOutLocal, shadingFrame, bsdf, lightSample) * (misWeight / areaPDensity);
    return ret;
}

static inline void pathTracePixel(const PathTraceParams& p, int x, int y) {
    const Scene& scene = *p.scene;
    const uint32_t bufIdx = p.f->bufferIndex;
    const size_t i = static_cast<size_t>(y) * p.s->imageSizeX + x;
    const gfx_gbuffer0& gb0 = static_cast<const gfx_gbuffer0*>(p.s->gbuffer0[bufIdx])[i];
    const uint32_t instSlot = gb0.instSlot;
    const float bcB = decodeBarycentric(gb0.qbcB);
    const float bcC = decodeBarycentric(gb0.qbcC);
    const VisibilityFn visFn = [&p](V3 o, V3 d, float t0, float t1) { return !occluded(*p.accel, o, d, t0, t1); };
    const bool useEnvLight = p.envEnabled();
    RGB contribution(0.001f, 0.001f, 0.001f);
    if (instSlot != 0xFFFFFFFFu) {
        const InstanceData& inst = scene.insts[instSlot];
        const GeometryInstanceData& geomInst = scene.geomInsts[gb0.geomInstSlot];
        V3 positionInWorld, geometricNormalInWorld, shadingNormalInWorld, texCoord0DirInWorld;
        V2 texCoord;
        computeSurfacePointRG(inst, geomInst, gb0.primIndex, bcB, bcC, &positionInWorld, &shadingNormalInWorld,
                              &texCoord0DirInWorld, &geometricNormalInWorld, &texCoord);
        RGB alpha(1.0f);
        const float initImportance = sRGB_calcLuminance(alpha);
        uint64_t* rngBuf = static_cast<uint64_t*>(p.s->rngBuffer);
        PCG32RNG rng; rng.setState(rngBuf[i]);
        V3 vIn;
        float dirPDensity;
        {
            const MaterialData& mat = scene.materials[geomInst.materialSlot];
            const V3 vOut = normalize(p.camera.position - positionInWorld);
            const float frontHit = dot(vOut, geometricNormalInWorld) >= 0.0f ? 1.0f : -1.0f;
            positionInWorld = offsetRayOrigin(positionInWorld, frontHit * geometricNormalInWorld);
            ReferenceFrame shadingFrame(shadingNormalInWorld, texCoord0DirInWorld);
            if (p.f->enableBumpMapping) {   // optix_pathtracing_kernels.cu:117-121
                const V3 modLocalNormal = readModifiedNormal(scene.textures, mat, texCoord);
                applyBumpMapping(modLocalNormal, &shadingFrame);
            }
            const V3 vOutLocal = shadingFrame.toLocal(vOut);
            contribution = RGB(0.0f);
            if (vOutLocal.z > 0 && mat.hasEmittance) {
                const RGB emittance = materialEmittance(scene.textures, mat, texCoord);
                contribution += alpha * emittance / kPi;
            }
            BSDF bsdf; bsdf.setup(scene.textures, mat, texCoord);
            contribution += alpha * performNextEventEstimation(p, visFn, positionInWorld, vOutLocal, shadingFrame, bsdf, rng);
            V3 vInLocal;
            const float u0 = rng.getFloat0cTo1o();
            const float u1 = rng.getFloat0cTo1o();
            alpha *= bsdf.sampleThroughput(vOutLocal, u0, u1, &vInLocal, &dirPDensity);
            vIn = shadingFrame.fromLocal(vInLocal);
        }
        float prevDirPDensity = dirPDensity;
        uint32_t pathLength = 1;
        V3 rayOrg = positionInWorld, rayDir = vIn;
        while (true) {
            const bool isValidSampling = prevDirPDensity > 0.0f && finitef(prevDirPDensity);
            if (!isValidSampling) break;
            ++pathLength;
            const bool maxLengthTerminate = pathLength >= p.maxPathLength;
            if (p.regir) { // useReGIR: regir/gpu_kernels/optix_pathtracing_kernels.cu:247-256
                if (maxLengthTerminate) break;
                const float continueProb = std::fmin(sRGB_calcLuminance(alpha) / initImportance, 1.0f);
                if (rng.getFloat0cTo1o() >= continueProb) break;
                alpha /= continueProb;
            }
            const bvh::HitObject h = closestHitCanonical(*p.accel, rayOrg, rayDir, 0.0f, 3.402823466e+38f);
            if (!h.isHit()) { // miss program (the ReGIR ray type has an empty miss program, regir_main.cpp:250)
                if (useEnvLight && !p.regir) {
                    const V3 rd = normalize(rayDir);
                    float posPhi, theta;
                    toPolarYUp(rd, &posPhi, &theta);
                    float phi = posPhi + p.f->envLightRotation;
                    phi = phi - std::floor(phi / (2 * kPi)) * 2 * kPi;
                    const V2 tc{ phi / (2 * kPi), theta / kPi };
                    const RGB luminance = p.f->envLightPowerCoeff * scene.env.fetch(tc.x, tc.y);
                    const float uvPDF = envEvaluatePDF(scene.env, tc.x, tc.y);
                    const float hypAreaPDensity = uvPDF / (2 * kPi * kPi * gm_sin(theta));
                    const float lightPDensity = (scene.lightInstDist.integral() > 0.0f ? 0.25f : 1.0f) * hypAreaPDensity;
                    const float bsdfPDensity = prevDirPDensity;
                    const float misWeight = pow2(bsdfPDensity) / (pow2(bsdfPDensity) + pow2(lightPDensity));
                    contribution += alpha * luminance * misWeight;
                }
                break;
            }
            // closest-hit program
            const uint32_t hInst = p.accel->geomToInst[h.geomIndex], hGeom = p.accel->geomToGeomInst[h.geomIndex];
            const InstanceData& hi = scene.insts[hInst];
            const GeometryInstanceData& hg = scene.geomInsts[hGeom];
            V3 pos, sn, tc0, gn; V2 tc; float hypAreaPDensity;
            computeSurfacePointCH(scene, useEnvLight, hInst, hi, hg, h.primIndex, h.bcB, h.bcC, &pos, &sn, &tc0, &gn, &tc, &hypAreaPDensity,
                                  p.f->useSolidAngleSampling != 0, rayOrg);
            // pathTraceReGIR instantiates computeSurfacePoint<false, ..> and then READS hypAreaPDensity
            // uninitialised (regir/.../optix_pathtracing_kernels.cu:323-330, :356-361): undefined in the
            // reference; this build defines it as 0 (MIS weight 1 for implicit light hits).
            if (p.regir) hypAreaPDensity = 0.0f;
            const MaterialData& mat = scene.materials[hg.materialSlot];
            const V3 vOut = normalize(-rayDir);
            const float frontHit = dot(vOut, gn) >= 0.0f ? 1.0f : -1.0f;
            ReferenceFrame shadingFrame(sn, tc0);
            if (p.f->enableBumpMapping) {   // :251-255
                const V3 modLocalNormal = readModifiedNormal(scene.textures, mat, tc);
                applyBumpMapping(modLocalNormal, &shadingFrame);
            }
            const V3 posOff = offsetRayOrigin(pos, frontHit * gn);
            const V3 vOutLocal = shadingFrame.toLocal(vOut);
            if (vOutLocal.z > 0 && mat.hasEmittance) {
                const RGB emittance = materialEmittance(scene.textures, mat, tc);
                const V3 dd = rayOrg - posOff;   // sqDistance(rayOrigin, positionInWorld) AFTER the offset (:256-274)
                const float dist2 = sqLength(dd);
                const float lightPDensity = hypAreaPDensity * dist2 / vOutLocal.z;
                const float bsdfPDensity = prevDirPDensity;
                const float misWeight = pow2(bsdfPDensity) / (pow2(bsdfPDensity) + pow2(lightPDensity));
                contribution += alpha * emittance * (misWeight / kPi);
            }
            const float continueProb = std::fmin(sRGB_calcLuminance(alpha) / initImportance, 1.0f);
            if (rng.getFloat0cTo1o() >= continueProb || maxLengthTerminate) break;
            alpha /= continueProb;
            BSDF bsdf; bsdf.setup(scene.textures, mat, tc);
            contribution += alpha * performNextEventEstimation(p, visFn, posOff, vOutLocal, shadingFrame, bsdf, rng);
            V3 vInLocal;
            float dpd;
            const float u0 = rng.getFloat0cTo1o();
            const float u1 = rng.getFloat0cTo1o();
            alpha *= bsdf.sampleThroughput(vOutLocal, u0, u1, &vInLocal, &dpd);
            rayOrg = posOff;
            rayDir = shadingFrame.fromLocal(vInLocal);
            prevDirPDensity = dpd;
        }
        rngBuf[i] = rng.state;
    }
    else if (useEnvLight) {
        const RGB texValue = scene.env.fetch(bcB, bcC);
        contribution = p.f->envLightPowerCoeff * texValue;
    }
    float* beauty = static_cast<float*>(p.s->beautyAccumBuffer) + 4 * i;
    RGB prev(0.0f, 0.0f, 0.0f);
    if (p.f->numAccumFrames > 0) prev = RGB(beauty[0], beauty[1], beauty[2]);
    const float curWeight = 1.0f / (1 + p.f->numAccumFrames);
    const RGB colorResult = (1 - curWeight) * prev + curWeight * contribution;
    beauty[0] = colorResult.x; beauty[1] = colorResult.y; beauty[2] = colorResult.z; beauty[3] = 1.0f;
}

} // namespace orc
