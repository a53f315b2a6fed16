// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// orc_nrc.h: CPU restatement of the neural-radiance-cache plumbing around the network
//   createRadianceQuery / convertToPolar   neural_radiance_caching/gpu_kernels/optix_pathtracing_kernels.cu:12-32
//   performNextEventEstimation             :36-90
//   pathTrace_raygen_generic<true>         :92-378
//   pathTrace_closestHit_generic<true>     :380-632
//   pathTrace_miss_generic<true>           :634-677
//   visualizePrediction                    :703-778
//   preprocessNRC / accumulateInferredRadianceValues / propagateRadianceValues / shuffleTrainingData
//                                          neural_radiance_caching/gpu_kernels/nrc_setup_kernels.cu:6-216
// Buffer layouts are those of include/gfxexp.h (gfx_nrc_params); bit fields the reference leaves
// uninitialised (struct padding) are zero here.
#pragma once
#include "orc_pathtrace.h"

namespace orc {

constexpr float kPathTerminationFactor = 0.01f;              // neural_radiance_caching_shared.h:7
constexpr uint32_t kNumTrainingDataPerFrame = 1u << 16;      // :8
constexpr uint32_t kTrainBufferSize = 2 * kNumTrainingDataPerFrame; // :9
constexpr uint32_t kInvalidVertexDataIndex = 0x007FFFFFu;    // :146

struct RadianceQuery { // neural_radiance_caching_shared.h:118-137 (14 floats)
    float position[3];
    float normal_phi, normal_theta, vOut_phi, vOut_theta, roughness;
    float diffuseReflectance[3], specularReflectance[3];
    bool isValid() const {
        const float* f = &position[0];
        for (int i = 0; i < 14; ++i) if (!finitef(f[i])) return false;
        return true;
    }
};
static_assert(sizeof(RadianceQuery) == 56, "RadianceQuery must be 14 floats");

// TerminalInfo (:139-146): alpha RGB + {hasQuery:1, pathLength:8, isTrainingPixel:1, isUnbiasedTile:1}
static inline uint32_t packTerminalBits(bool hasQuery, uint32_t pathLength, bool isTrainingPixel, bool isUnbiasedTile) {
    return (hasQuery ? 1u : 0u) | ((pathLength & 0xFFu) << 1) | ((isTrainingPixel ? 1u : 0u) << 9) | ((isUnbiasedTile ? 1u : 0u) << 10);
}
// TrainingVertexInfo (:150-155): localThroughput RGB + {prevVertexDataIndex:23, pathLength:8}
static inline uint32_t packVertexBits(uint32_t prevVertexDataIndex, uint32_t pathLength) {
    return (prevVertexDataIndex & 0x7FFFFFu) | ((pathLength & 0xFFu) << 23);
}
// TrainingSuffixTerminalInfo (:157-162): {prevVertexDataIndex:23, hasQuery:1, pathLength:8}
static inline uint32_t packSuffixBits(uint32_t prevVertexDataIndex, bool hasQuery, uint32_t pathLength) {
    return (prevVertexDataIndex & 0x7FFFFFu) | ((hasQuery ? 1u : 0u) << 23) | ((pathLength & 0xFFu) << 24);
}
static inline int32_t floatToOrderedInt(float f) { const int32_t i = static_cast<int32_t>(f2bits(f)); return i >= 0 ? i : i ^ 0x7FFFFFFF; } // basic_types.h:411-418

struct NrcState {
    const gfx_nrc_params* n;
    uint32_t* numTrainingData(uint32_t b) const { return static_cast<uint32_t*>(n->numTrainingData[b]); }
    uint32_t* tileSize(uint32_t b) const { return static_cast<uint32_t*>(n->tileSize[b]); }
    RadianceQuery* inferenceQueries() const { return static_cast<RadianceQuery*>(n->inferenceRadianceQueryBuffer); }
    float* terminalInfos() const { return static_cast<float*>(n->inferenceTerminalInfoBuffer); }     // 4 words per pixel
    float* inferred() const { return static_cast<float*>(n->inferredRadianceBuffer); }               // 3 per entry
    float* perFrameContribution() const { return static_cast<float*>(n->perFrameContributionBuffer); }
    RadianceQuery* trainQueries(int i) const { return static_cast<RadianceQuery*>(n->trainRadianceQueryBuffer[i]); }
    float* trainTargets(int i) const { return static_cast<float*>(n->trainTargetBuffer[i]); }
    float* trainVertexInfos() const { return static_cast<float*>(n->trainVertexInfoBuffer); }        // 4 words
    uint32_t* suffixTerminals() const { return static_cast<uint32_t*>(n->trainSuffixTerminalInfoBuffer); }
    V3 aabbMin() const { return V3(n->sceneAabbMin[0], n->sceneAabbMin[1], n->sceneAabbMin[2]); }
    V3 aabbMax() const { return V3(n->sceneAabbMax[0], n->sceneAabbMax[1], n->sceneAabbMax[2]); }
};

static inline void convertToPolar(V3 dir, float* phi, float* theta) { // :12-16
    const float z = std::fmin(std::fmax(dir.z, -1.0f), 1.0f);
    *theta = gm_acos(z);
    *phi = gm_atan2(dir.y, dir.x);
}
static inline RadianceQuery createRadianceQuery(const NrcState& ns, V3 positionInWorld, V3 normalInWorld, V3 scatteredDirInWorld,
                                                float roughness, RGB diffuseReflectance, RGB specularReflectance) { // :18-32
    RadianceQuery q;
    const V3 num = positionInWorld - ns.aabbMin(), den = ns.aabbMax() - ns.aabbMin();   // AABB::normalize = safeDivide
    q.position[0] = den.x != 0 ? num.x / den.x : 0.0f;
    q.position[1] = den.y != 0 ? num.y / den.y : 0.0f;
    q.position[2] = den.z != 0 ? num.z / den.z : 0.0f;
    convertToPolar(normalInWorld, &q.normal_phi, &q.normal_theta);
    convertToPolar(scatteredDirInWorld, &q.vOut_phi, &q.vOut_theta);
    q.roughness = 1 - gm_exp(-roughness);
    q.diffuseReflectance[0] = diffuseReflectance.x; q.diffuseReflectance[1] = diffuseReflectance.y; q.diffuseReflectance[2] = diffuseReflectance.z;
    q.specularReflectance[0] = specularReflectance.x; q.specularReflectance[1] = specularReflectance.y; q.specularReflectance[2] = specularReflectance.z;
    return q;
}

// nrc_setup_kernels.cu:6-49
static inline void preprocessNRC(const NrcState& ns, const gfx_restir_frame_params& f) {
    const gfx_nrc_params& n = *ns.n;
    const uint32_t bufIdx = f.bufferIndex, prevBufIdx = (f.bufferIndex + 1) % 2;
    uint32_t nx, ny;
    if (n.isNewSequence) { nx = 8; ny = 8; }
    else {
        const uint32_t prevNumTrainingData = *ns.numTrainingData(prevBufIdx);
        const float r = std::sqrt(static_cast<float>(prevNumTrainingData) / kNumTrainingDataPerFrame);
        const uint32_t* cur = ns.tileSize(prevBufIdx);
        nx = f2u(cur[0] * r); ny = f2u(cur[1] * r);
        nx = nx < 4u ? 4u : (nx > 128u ? 128u : nx);
        ny = ny < 4u ? 4u : (ny > 128u ? 128u : ny);
    }
    ns.tileSize(bufIdx)[0] = nx; ns.tileSize(bufIdx)[1] = ny;
    *ns.numTrainingData(bufIdx) = 0;
    *static_cast<uint32_t*>(n.offsetToSelectUnbiasedTile) = n.preprocessOffsetToSelectUnbiasedTile;
    *static_cast<uint32_t*>(n.offsetToSelectTrainingPath) = n.preprocessOffsetToSelectTrainingPath;
    int32_t* mm = static_cast<int32_t*>(n.targetMinMax[bufIdx]);
    const int32_t pinf = floatToOrderedInt(INFINITY), ninf = floatToOrderedInt(-INFINITY);
    mm[0] = mm[1] = mm[2] = pinf; mm[3] = mm[4] = mm[5] = ninf;
    float* avg = static_cast<float*>(n.targetAvg[bufIdx]);
    avg[0] = avg[1] = avg[2] = 0.0f;
    for (uint32_t i = 0; i < n.maxNumTrainingSuffixes; ++i) ns.suffixTerminals()[i] = packSuffixBits(kInvalidVertexDataIndex, false, 0);
}

// performNextEventEstimation :36-90 (identical to the baseline path tracer's)
static inline RGB nrcNextEventEstimation(const PathTraceParams& p, const VisibilityFn& visFn, V3 shadingPoint, V3 vOutLocal,
                                         const ReferenceFrame& shadingFrame, const BSDF& bsdf, PCG32RNG& rng) {
    return performNextEventEstimation(p, visFn, shadingPoint, vOutLocal, shadingFrame, bsdf, rng);
}

struct NrcPayload { // PathTraceReadWritePayload<true>, neural_radiance_caching_shared.h:207-227
    PCG32RNG rng;
    float initImportance;
    RGB alpha, contribution;
    float prevDirPDensity;
    uint32_t linearTileIndex;
    float primaryPathSpread, curSqrtPathSpread;
    RGB prevLocalThroughput;
    uint32_t prevTrainDataIndex;
    bool renderingPathEndsWithCache, isTrainingPath, isUnbiasedTrainingTile, trainingSuffixEndsWithCache;
    bool maxLengthTerminate, terminate;
    uint32_t pathLength;
    V3 nextOrigin, nextDirection;
};

static inline void nrcWriteTrainVertex(const NrcState& ns, uint32_t idx, const RadianceQuery& q, RGB localThroughput,
                                       uint32_t prevIdx, uint32_t pathLength, RGB target) {
    ns.trainQueries(0)[idx] = q;
    float* vi = ns.trainVertexInfos() + 4 * static_cast<size_t>(idx);
    vi[0] = localThroughput.x; vi[1] = localThroughput.y; vi[2] = localThroughput.z; vi[3] = bits2f(packVertexBits(prevIdx, pathLength));
    float* t = ns.trainTargets(0) + 3 * static_cast<size_t>(idx);
    t[0] = target.x; t[1] = target.y; t[2] = target.z;
}
static inline void nrcWriteTerminal(const NrcState& ns, size_t linearIndex, RGB alpha, bool hasQuery, uint32_t pathLength,
                                    bool isTrainingPixel, bool isUnbiasedTile) {
    float* t = ns.terminalInfos() + 4 * linearIndex;
    t[0] = alpha.x; t[1] = alpha.y; t[2] = alpha.z; t[3] = bits2f(packTerminalBits(hasQuery, pathLength, isTrainingPixel, isUnbiasedTile));
}

// pathTrace_closestHit_generic<true> :380-632 and pathTrace_miss_generic<true> :634-677 for one traced ray
static inline void nrcTraceVertex(const PathTraceParams& p, const NrcState& ns, const VisibilityFn& visFn, uint32_t* trainCounter,
                                  int x, int y, V3 rayOrg, V3 rayDir, NrcPayload& pl) {
    const Scene& scene = *p.scene;
    const bool useEnvLight = p.envEnabled();
    const size_t numPixels = static_cast<size_t>(p.s->imageSizeX) * p.s->imageSizeY;
    const bvh::HitObject h = closestHitCanonical(*p.accel, rayOrg, rayDir, 0.0f, 3.402823466e+38f);
    // p.regir (GFX_PT_PATH_TRACE_NRC_REGIR, include/gfxexp.h): next-event estimation samples the ReGIR grid cell
    // (performNextEventEstimation's regir branch); that estimate has no evaluable density, so there is no MIS and emitters
    // found by BSDF sampling -- the environment in the miss program, emissive surfaces in the closest-hit program --
    // contribute nothing at path length >= 2.  An extension of this build: the reference lists the combination as open
    // (README.md:80-81).
    // p.restir (GFX_PT_PATH_TRACE_NRC_RESTIR): the first vertex's direct lighting is the pixel's ReSTIR DI reservoir, an estimate of ALL
    // the direct light at that vertex with no density to weight a BSDF-sampled emitter against -- so what the first extension ray (path
    // length 2) finds emitting, surface or environment, contributes nothing; deeper vertices keep the tracer's own NEE + MIS.
    const bool firstVertexIsRestir = p.restir != nullptr && pl.pathLength == 2;
    if (!h.isHit()) { // miss
        if (!useEnvLight || p.regir || firstVertexIsRestir) return;
        const V3 rd = normalize(rayDir);
        float posPhi, theta;
        toPolarYUp(rd, &posPhi, &theta);
        float phi = posPhi + p.f->envLightRotation;
        phi = phi - std::floor(phi / (2 * kPi)) * 2 * kPi;
        const V2 tc{ phi / (2 * kPi), theta / kPi };
        const RGB luminance = p.f->envLightPowerCoeff * scene.env.fetch(tc.x, tc.y);
        const float uvPDF = envEvaluatePDF(scene.env, tc.x, tc.y);
        const float hypAreaPDensity = uvPDF / (2 * kPi * kPi * gm_sin(theta));
        const float lightPDensity = 0.25f * hypAreaPDensity;
        const float bsdfPDensity = pl.prevDirPDensity;
        const float misWeight = pow2(bsdfPDensity) / (pow2(bsdfPDensity) + pow2(lightPDensity));
        const RGB directContImplicit = misWeight * luminance;
        pl.contribution += pl.alpha * directContImplicit;
        if (pl.isTrainingPath && pl.prevTrainDataIndex != kInvalidVertexDataIndex) {
            float* t = ns.trainTargets(0) + 3 * static_cast<size_t>(pl.prevTrainDataIndex);
            const RGB add = pl.prevLocalThroughput * directContImplicit;
            t[0] += add.x; t[1] += add.y; t[2] += add.z;
        }
        return;
    }
    const uint32_t hInst = p.accel->geomToInst[h.geomIndex], hGeom = p.accel->geomToGeomInst[h.geomIndex];
    const InstanceData& hi = scene.insts[hInst];
    const GeometryInstanceData& hg = scene.geomInsts[hGeom];
    V3 positionInWorld, shadingNormalInWorld, texCoord0DirInWorld, geometricNormalInWorld; V2 texCoord; float hypAreaPDensity;
    computeSurfacePointCH(scene, useEnvLight, hInst, hi, hg, h.primIndex, h.bcB, h.bcC, &positionInWorld, &shadingNormalInWorld,
                          &texCoord0DirInWorld, &geometricNormalInWorld, &texCoord, &hypAreaPDensity,
                          p.f->useSolidAngleSampling != 0, rayOrg);
    const MaterialData& mat = scene.materials[hg.materialSlot];
    const V3 vOut = normalize(-rayDir);
    const float frontHit = dot(vOut, geometricNormalInWorld) >= 0.0f ? 1.0f : -1.0f;
    ReferenceFrame shadingFrame(shadingNormalInWorld, texCoord0DirInWorld);
    if (p.f->enableBumpMapping) {
        const V3 modLocalNormal = readModifiedNormal(scene.textures, mat, texCoord);
        applyBumpMapping(modLocalNormal, &shadingFrame);
    }
    positionInWorld = offsetRayOrigin(positionInWorld, frontHit * geometricNormalInWorld);
    const V3 vOutLocal = shadingFrame.toLocal(vOut);
    const float dist2 = sqLength(rayOrg - positionInWorld);
    pl.curSqrtPathSpread += std::sqrt(dist2 / (pl.prevDirPDensity * std::fabs(vOutLocal.z)));

    if (!p.regir && !firstVertexIsRestir && vOutLocal.z > 0 && mat.hasEmittance) {
        const RGB emittance = materialEmittance(scene.textures, mat, texCoord);
        const float lightPDensity = hypAreaPDensity * dist2 / vOutLocal.z;
        const float bsdfPDensity = pl.prevDirPDensity;
        const float misWeight = pow2(bsdfPDensity) / (pow2(bsdfPDensity) + pow2(lightPDensity));
        const RGB directContImplicit = emittance * (misWeight / kPi);
        pl.contribution += pl.alpha * directContImplicit;
        if (pl.isTrainingPath && pl.prevTrainDataIndex != kInvalidVertexDataIndex) {
            float* t = ns.trainTargets(0) + 3 * static_cast<size_t>(pl.prevTrainDataIndex);
            const RGB add = pl.prevLocalThroughput * directContImplicit;
            t[0] += add.x; t[1] += add.y; t[2] += add.z;
        }
    }

    bool performRR = true;
    bool terminatedByRR = false;
    float recContinueProb = 1.0f;
    if (pl.isTrainingPath) performRR = pl.pathLength > 2;
    if (performRR) {
        const float continueProb = std::fmin(sRGB_calcLuminance(pl.alpha) / pl.initImportance, 1.0f);
        if (pl.rng.getFloat0cTo1o() >= continueProb || pl.maxLengthTerminate) {
            if (pl.renderingPathEndsWithCache && pl.isTrainingPath && pl.isUnbiasedTrainingTile) return;
            terminatedByRR = true;
        }
        recContinueProb = 1.0f / continueProb;
    }
    BSDF bsdf; bsdf.setup(scene.textures, mat, texCoord);
    {
        bool endsWithCache = false;
        const bool pathIsSpreadEnough = pow2(pl.curSqrtPathSpread) > kPathTerminationFactor * pl.primaryPathSpread;
        endsWithCache |= pathIsSpreadEnough;
        if (pl.renderingPathEndsWithCache && pl.isTrainingPath && pl.isUnbiasedTrainingTile) endsWithCache = false;
        if (endsWithCache) {
            const size_t linearIndex = static_cast<size_t>(y) * p.s->imageSizeX + x;
            float roughness; RGB diffuseReflectance, specularReflectance;
            bsdf.getSurfaceParameters(&diffuseReflectance, &specularReflectance, &roughness);
            const RadianceQuery radQuery = createRadianceQuery(ns, positionInWorld, shadingFrame.normal, vOut, roughness, diffuseReflectance, specularReflectance);
            if (!pl.renderingPathEndsWithCache) {
                ns.inferenceQueries()[linearIndex] = radQuery;
                nrcWriteTerminal(ns, linearIndex, pl.alpha, true, pl.pathLength, pl.isTrainingPath, pl.isUnbiasedTrainingTile);
                pl.renderingPathEndsWithCache = true;
                if (pl.isTrainingPath) pl.curSqrtPathSpread = 0;
                else return;
            }
            else {
                if (!pl.trainingSuffixEndsWithCache) {
                    ns.inferenceQueries()[numPixels + pl.linearTileIndex] = radQuery;
                    ns.suffixTerminals()[pl.linearTileIndex] = packSuffixBits(pl.prevTrainDataIndex, true, pl.pathLength);
                    pl.trainingSuffixEndsWithCache = true;
                }
                return;
            }
        }
    }
    if (terminatedByRR) return;
    pl.alpha *= recContinueProb;
    if (pl.isTrainingPath && pl.prevTrainDataIndex != kInvalidVertexDataIndex) {
        float* vi = ns.trainVertexInfos() + 4 * static_cast<size_t>(pl.prevTrainDataIndex);
        vi[0] *= recContinueProb; vi[1] *= recContinueProb; vi[2] *= recContinueProb;
    }
    const RGB directContNEE = nrcNextEventEstimation(p, visFn, positionInWorld, vOutLocal, shadingFrame, bsdf, pl.rng);
    pl.contribution += pl.alpha * directContNEE;
    V3 vInLocal; float dirPDensity;
    const float u0 = pl.rng.getFloat0cTo1o();
    const float u1 = pl.rng.getFloat0cTo1o();
    const RGB localThroughput = bsdf.sampleThroughput(vOutLocal, u0, u1, &vInLocal, &dirPDensity);
    pl.alpha *= localThroughput;
    pl.nextOrigin = positionInWorld;
    pl.nextDirection = shadingFrame.fromLocal(vInLocal);
    pl.prevDirPDensity = dirPDensity;
    pl.prevLocalThroughput = localThroughput;
    pl.terminate = false;
    if (pl.isTrainingPath && !pl.trainingSuffixEndsWithCache) {
        const uint32_t trainDataIndex = (*trainCounter)++;
        float roughness; RGB diffuseReflectance, specularReflectance;
        bsdf.getSurfaceParameters(&diffuseReflectance, &specularReflectance, &roughness);
        const RadianceQuery radQuery = createRadianceQuery(ns, positionInWorld, shadingFrame.normal, vOut, roughness, diffuseReflectance, specularReflectance);
        if (trainDataIndex < kTrainBufferSize) {
            nrcWriteTrainVertex(ns, trainDataIndex, radQuery, localThroughput, pl.prevTrainDataIndex, pl.pathLength, directContNEE);
            pl.prevTrainDataIndex = trainDataIndex;
        }
        else {
            ns.inferenceQueries()[numPixels + pl.linearTileIndex] = radQuery;
            ns.suffixTerminals()[pl.linearTileIndex] = packSuffixBits(pl.prevTrainDataIndex, true, pl.pathLength);
            pl.trainingSuffixEndsWithCache = true;
        }
    }
}

// pathTrace_raygen_generic<true> :92-378.  Pixels are visited row-major by ONE thread so the
// training-data indices (atomicAdd order in the reference, unspecified there) are deterministic here.
static inline void nrcPathTracePixel(const PathTraceParams& p, const NrcState& ns, uint32_t* trainCounter, int x, int y) {
    const Scene& scene = *p.scene;
    const uint32_t bufIdx = p.f->bufferIndex;
    const size_t i = static_cast<size_t>(y) * p.s->imageSizeX + x;
    const gfx_gbuffer0& gb0 = static_cast<const gfx_gbuffer0*>(p.s->gbuffer0[bufIdx])[i];
    const float bcB = decodeBarycentric(gb0.qbcB), bcC = decodeBarycentric(gb0.qbcC);
    const VisibilityFn visFn = [&p](V3 o, V3 d, float t0, float t1) { return !occluded(*p.accel, o, d, t0, t1); };
    const uint32_t tsx = ns.tileSize(bufIdx)[0], tsy = ns.tileSize(bufIdx)[1];
    const uint32_t numPixelsInTile = tsx * tsy;
    const uint32_t lx = static_cast<uint32_t>(x) % tsx, ly = static_cast<uint32_t>(y) % tsy;
    const uint32_t localLinearIndex = ly * tsx + lx;
    const bool isTrainingPath = (localLinearIndex + *static_cast<const uint32_t*>(ns.n->offsetToSelectTrainingPath)) % numPixelsInTile == 0;
    const uint32_t numTilesX = (static_cast<uint32_t>(p.s->imageSizeX) + tsx - 1) / tsx;
    const uint32_t tileX = static_cast<uint32_t>(x) / tsx, tileY = static_cast<uint32_t>(y) / tsy;
    const uint32_t linearTileIndex = tileY * numTilesX + tileX;
    const uint32_t localLinearTileIndex = (tileY % 4) * 4 + (tileX % 4);
    const bool isUnbiasedTrainingTile = (localLinearTileIndex + *static_cast<const uint32_t*>(ns.n->offsetToSelectUnbiasedTile)) % 16 == 0;

    const bool useEnvLight = p.envEnabled();
    RGB contribution(0.001f, 0.001f, 0.001f);
    bool renderingPathEndsWithCache = false;
    uint32_t pathLength = 1;
    if (gb0.instSlot != 0xFFFFFFFFu) {
        const InstanceData& inst = scene.insts[gb0.instSlot];
        const GeometryInstanceData& geomInst = scene.geomInsts[gb0.geomInstSlot];
        V3 positionInWorld, geometricNormalInWorld, shadingNormalInWorld, texCoord0DirInWorld; V2 texCoord;
        computeSurfacePointRG(inst, geomInst, gb0.primIndex, bcB, bcC, &positionInWorld, &shadingNormalInWorld,
                              &texCoord0DirInWorld, &geometricNormalInWorld, &texCoord);
        RGB alpha(1.0f);
        const float initImportance = sRGB_calcLuminance(alpha);
        uint64_t* rngBuf = static_cast<uint64_t*>(p.s->rngBuffer);
        PCG32RNG rng; rng.setState(rngBuf[i]);
        V3 vIn; float dirPDensity, primaryPathSpread; RGB localThroughput;
        uint32_t trainDataIndex = 0;
        {
            const MaterialData& mat = scene.materials[geomInst.materialSlot];
            V3 vOut = p.camera.position - positionInWorld;
            const float primaryDist2 = sqLength(vOut);
            vOut /= std::sqrt(primaryDist2);
            const float primaryDotVN = dot(vOut, geometricNormalInWorld);
            const float frontHit = primaryDotVN >= 0.0f ? 1.0f : -1.0f;
            positionInWorld = offsetRayOrigin(positionInWorld, frontHit * geometricNormalInWorld);
            primaryPathSpread = primaryDist2 / (4 * kPi * std::fabs(primaryDotVN));
            ReferenceFrame shadingFrame(shadingNormalInWorld, texCoord0DirInWorld);
            if (p.f->enableBumpMapping) {
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
            // first-vertex next-event estimation: the tracer's own light sample, or (p.restir) what the ReSTIR DI passes of this frame
            // left in the pixel's reservoir -- no random numbers are drawn here then
            const RGB directContNEE = p.restir ? restirDirectEstimate(*p.restir, x, y)
                                               : nrcNextEventEstimation(p, visFn, positionInWorld, vOutLocal, shadingFrame, bsdf, rng);
            contribution += alpha * directContNEE;
            V3 vInLocal;
            const float u0 = rng.getFloat0cTo1o();
            const float u1 = rng.getFloat0cTo1o();
            localThroughput = bsdf.sampleThroughput(vOutLocal, u0, u1, &vInLocal, &dirPDensity);
            alpha *= localThroughput;
            vIn = shadingFrame.fromLocal(vInLocal);
            if (isTrainingPath) {
                trainDataIndex = (*trainCounter)++;
                if (trainDataIndex < kTrainBufferSize) {
                    float roughness; RGB diffuseReflectance, specularReflectance;
                    bsdf.getSurfaceParameters(&diffuseReflectance, &specularReflectance, &roughness);
                    const RadianceQuery radQuery = createRadianceQuery(ns, positionInWorld, shadingFrame.normal, vOut, roughness, diffuseReflectance, specularReflectance);
                    nrcWriteTrainVertex(ns, trainDataIndex, radQuery, localThroughput, kInvalidVertexDataIndex, pathLength, directContNEE);
                }
                else trainDataIndex = kInvalidVertexDataIndex;
            }
        }
        NrcPayload pl;
        pl.rng = rng; pl.initImportance = initImportance; pl.alpha = alpha; pl.contribution = contribution;
        pl.prevDirPDensity = dirPDensity;
        pl.linearTileIndex = linearTileIndex; pl.primaryPathSpread = primaryPathSpread; pl.curSqrtPathSpread = 0.0f;
        pl.prevLocalThroughput = localThroughput; pl.prevTrainDataIndex = trainDataIndex;   // (indeterminate in the reference for non-training paths; never read there)
        pl.renderingPathEndsWithCache = false; pl.isTrainingPath = isTrainingPath; pl.isUnbiasedTrainingTile = isUnbiasedTrainingTile;
        pl.trainingSuffixEndsWithCache = false; pl.maxLengthTerminate = false; pl.terminate = false;
        pl.pathLength = pathLength;
        V3 rayOrg = positionInWorld, rayDir = vIn;
        while (true) {
            const bool isValidSampling = pl.prevDirPDensity > 0.0f && finitef(pl.prevDirPDensity);
            if (!isValidSampling) break;
            pl.pathLength = (pl.pathLength + 1) & 63u;               // 6-bit field
            if (pl.pathLength >= p.maxPathLength && p.maxPathLength > 0) pl.maxLengthTerminate = true;
            pl.terminate = true;
            nrcTraceVertex(p, ns, visFn, trainCounter, x, y, rayOrg, rayDir, pl);
            if (pl.terminate) break;
            rayOrg = pl.nextOrigin; rayDir = pl.nextDirection;
        }
        contribution = pl.contribution;
        rngBuf[i] = pl.rng.state;
        renderingPathEndsWithCache = pl.renderingPathEndsWithCache;
        pathLength = pl.pathLength;
        if (pl.isTrainingPath && !pl.trainingSuffixEndsWithCache)
            ns.suffixTerminals()[pl.linearTileIndex] = packSuffixBits(pl.prevTrainDataIndex, false, pl.pathLength);
    }
    else if (useEnvLight) {
        contribution = p.f->envLightPowerCoeff * scene.env.fetch(bcB, bcC);
    }
    if (!renderingPathEndsWithCache) nrcWriteTerminal(ns, i, RGB(0.0f, 0.0f, 0.0f), false, pathLength, isTrainingPath, isUnbiasedTrainingTile);
    float* c = ns.perFrameContribution() + 3 * i;
    c[0] = contribution.x; c[1] = contribution.y; c[2] = contribution.z;
}

static inline RGB nrcScaledPrediction(const NrcState& ns, size_t entry) {
    const float* r = ns.inferred() + 3 * entry;
    RGB radiance(fmax2(r[0], 0.0f), fmax2(r[1], 0.0f), fmax2(r[2], 0.0f));
    if (ns.n->radianceScale > 0) radiance /= ns.n->radianceScale;
    const RadianceQuery& q = ns.inferenceQueries()[entry];   // useReflectanceFactorization
    radiance *= RGB(q.diffuseReflectance[0] + q.specularReflectance[0], q.diffuseReflectance[1] + q.specularReflectance[1],
                    q.diffuseReflectance[2] + q.specularReflectance[2]);
    return radiance;
}

// nrc_setup_kernels.cu:51-93
static inline void accumulateInferredRadiancePixel(const NrcState& ns, const gfx_restir_static_params& s, const gfx_restir_frame_params& f, size_t i) {
    const float* t = ns.terminalInfos() + 4 * i;
    const RGB alpha(t[0], t[1], t[2]);
    const bool hasQuery = (f2bits(t[3]) & 1u) != 0;
    const float* d = ns.perFrameContribution() + 3 * i;
    const RGB directCont(d[0], d[1], d[2]);
    RGB radiance(0.0f, 0.0f, 0.0f);
    if (hasQuery) radiance = nrcScaledPrediction(ns, i);
    const RGB indirectCont = alpha * radiance;
    const RGB contribution = directCont + indirectCont;
    float* beauty = static_cast<float*>(s.beautyAccumBuffer) + 4 * i;
    RGB prev(0.0f, 0.0f, 0.0f);
    if (f.numAccumFrames > 0) prev = RGB(beauty[0], beauty[1], beauty[2]);
    const float curWeight = 1.0f / (1 + f.numAccumFrames);
    const RGB colorResult = (1 - curWeight) * prev + curWeight * contribution;
    beauty[0] = colorResult.x; beauty[1] = colorResult.y; beauty[2] = colorResult.z; beauty[3] = 1.0f;
}

// nrc_setup_kernels.cu:95-137
static inline void propagateRadianceSuffix(const NrcState& ns, const gfx_restir_static_params& s, uint32_t linearIndex) {
    const uint32_t bits = ns.suffixTerminals()[linearIndex];
    uint32_t last = bits & 0x7FFFFFu;
    if (last == kInvalidVertexDataIndex) return;
    RGB contribution(0.0f, 0.0f, 0.0f);
    if ((bits >> 23) & 1u) {
        const size_t offset = static_cast<size_t>(s.imageSizeX) * s.imageSizeY;
        contribution = nrcScaledPrediction(ns, offset + linearIndex);
    }
    while (last != kInvalidVertexDataIndex) {
        const float* vi = ns.trainVertexInfos() + 4 * static_cast<size_t>(last);
        float* t = ns.trainTargets(0) + 3 * static_cast<size_t>(last);
        const RGB indirectCont = RGB(vi[0], vi[1], vi[2]) * contribution;
        contribution = RGB(t[0], t[1], t[2]) + indirectCont;
        const RadianceQuery& q = ns.trainQueries(0)[last];
        const RGB refFactor(q.diffuseReflectance[0] + q.specularReflectance[0], q.diffuseReflectance[1] + q.specularReflectance[1],
                            q.diffuseReflectance[2] + q.specularReflectance[2]);
        t[0] = refFactor.x != 0 ? contribution.x / refFactor.x : 0.0f;
        t[1] = refFactor.y != 0 ? contribution.y / refFactor.y : 0.0f;
        t[2] = refFactor.z != 0 ? contribution.z / refFactor.z : 0.0f;
        last = f2bits(vi[3]) & 0x7FFFFFu;
    }
}

// nrc_setup_kernels.cu:139-216 (statistics: min / max exact, average in thread order)
static inline void shuffleTrainingData(const NrcState& ns, const gfx_restir_frame_params& f) {
    const gfx_nrc_params& n = *ns.n;
    const uint32_t bufIdx = f.bufferIndex;
    const uint32_t numTrainingData = *ns.numTrainingData(bufIdx);
    uint32_t* shufflers = static_cast<uint32_t*>(n.dataShufflerBuffer);
    int32_t* mm = static_cast<int32_t*>(n.targetMinMax[bufIdx]);
    float* avg = static_cast<float*>(n.targetAvg[bufIdx]);
    for (uint32_t linearIndex = 0; linearIndex < kNumTrainingDataPerFrame; ++linearIndex) {
        if (numTrainingData > 0) {
            shufflers[linearIndex] = (shufflers[linearIndex] * 1103515245u + 12345u) % (1u << 31);   // LCG :164-181
            const uint32_t dstIdx = shufflers[linearIndex] % kNumTrainingDataPerFrame;
            const uint32_t srcIdx = linearIndex % numTrainingData;
            RadianceQuery query = ns.trainQueries(0)[srcIdx];
            const float* tv = ns.trainTargets(0) + 3 * static_cast<size_t>(srcIdx);
            RGB targetValue(tv[0], tv[1], tv[2]);
            if (!query.isValid()) std::memset(&query, 0, sizeof(query));
            if (!allFinite(targetValue)) targetValue = RGB(0.0f);
            const float c[3] = { targetValue.x, targetValue.y, targetValue.z };
            for (int k = 0; k < 3; ++k) {
                const int32_t o = floatToOrderedInt(c[k]);
                if (o < mm[k]) mm[k] = o;
                if (o > mm[3 + k]) mm[3 + k] = o;
                avg[k] += c[k] / kNumTrainingDataPerFrame;
            }
            if (n.radianceScale > 0) targetValue *= n.radianceScale;
            targetValue = RGB(fmin2(targetValue.x, 1e+6f), fmin2(targetValue.y, 1e+6f), fmin2(targetValue.z, 1e+6f));
            ns.trainQueries(1)[dstIdx] = query;
            float* dst = ns.trainTargets(1) + 3 * static_cast<size_t>(dstIdx);
            dst[0] = targetValue.x; dst[1] = targetValue.y; dst[2] = targetValue.z;
        }
        else {
            std::memset(&ns.trainQueries(1)[linearIndex], 0, sizeof(RadianceQuery));
            float* dst = ns.trainTargets(1) + 3 * static_cast<size_t>(linearIndex);
            dst[0] = dst[1] = dst[2] = 0.0f;
        }
    }
}

// visualizePrediction :703-778
static inline void visualizePredictionPixel(const PathTraceParams& p, const NrcState& ns, int x, int y) {
    const Scene& scene = *p.scene;
    const uint32_t bufIdx = p.f->bufferIndex;
    const size_t i = static_cast<size_t>(y) * p.s->imageSizeX + x;
    const gfx_gbuffer0& gb0 = static_cast<const gfx_gbuffer0*>(p.s->gbuffer0[bufIdx])[i];
    if (gb0.instSlot != 0xFFFFFFFFu) {
        const float bcB = decodeBarycentric(gb0.qbcB), bcC = decodeBarycentric(gb0.qbcC);
        const InstanceData& inst = scene.insts[gb0.instSlot];
        const GeometryInstanceData& geomInst = scene.geomInsts[gb0.geomInstSlot];
        V3 positionInWorld, geometricNormalInWorld, shadingNormalInWorld, texCoord0DirInWorld; V2 texCoord;
        computeSurfacePointRG(inst, geomInst, gb0.primIndex, bcB, bcC, &positionInWorld, &shadingNormalInWorld,
                              &texCoord0DirInWorld, &geometricNormalInWorld, &texCoord);
        const MaterialData& mat = scene.materials[geomInst.materialSlot];
        V3 vOut = p.camera.position - positionInWorld;
        const float primaryDist2 = sqLength(vOut);
        vOut /= std::sqrt(primaryDist2);
        const float frontHit = dot(vOut, geometricNormalInWorld) >= 0.0f ? 1.0f : -1.0f;
        positionInWorld = offsetRayOrigin(positionInWorld, frontHit * geometricNormalInWorld);
        ReferenceFrame shadingFrame(shadingNormalInWorld, texCoord0DirInWorld);
        if (p.f->enableBumpMapping) {
            const V3 modLocalNormal = readModifiedNormal(scene.textures, mat, texCoord);
            applyBumpMapping(modLocalNormal, &shadingFrame);
        }
        BSDF bsdf; bsdf.setup(scene.textures, mat, texCoord);
        float roughness; RGB diffuseReflectance, specularReflectance;
        bsdf.getSurfaceParameters(&diffuseReflectance, &specularReflectance, &roughness);
        ns.inferenceQueries()[i] = createRadianceQuery(ns, positionInWorld, shadingFrame.normal, vOut, roughness, diffuseReflectance, specularReflectance);
    }
    nrcWriteTerminal(ns, i, RGB(1.0f), gb0.instSlot != 0xFFFFFFFFu, 1, false, false);
}

} // namespace orc
