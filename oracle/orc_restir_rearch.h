// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// orc_restir_rearch.h: CPU restatement of the rearchitected ReSTIR passes
//   performLightPreSampling / performPerPixelRIS   restir_di/gpu_kernels/per_pixel_ris.cu:6-128
//   traceShadowRays<T,S,U>                         restir_di/gpu_kernels/optix_restir_di_rearch_kernels.cu:14-225
//   computeMISWeight<type,T,S>                     :263-400
//   shadeAndResample<T,S>                          :402-664
// PreSampledLight (restir_di_shared.h:98-101, 44 B) is stored as three 16-byte words:
//   [emittance.rgb, position.x] [position.yz, normal.xy] [normal.z, atInfinity, areaPDensity, 0].
#pragma once
#include "orc_restir.h"

namespace orc {

constexpr uint32_t kNumLightSubsets = 128;   // restir_di_shared.h:8
constexpr uint32_t kLightSubsetSize = 1024;  // :9
constexpr int kTileSizeX = 8, kTileSizeY = 8; // :10-11

// SampleVisibility bit positions (restir_di_shared.h:146-164)
enum : uint32_t {
    SV_newSample = 1u << 0, SV_newSampleOnTemporal = 1u << 1, SV_newSampleOnSpatiotemporal = 1u << 2,
    SV_temporalPassedHeuristic = 1u << 3, SV_temporalSample = 1u << 4, SV_temporalSampleOnCurrent = 1u << 5,
    SV_temporalSampleOnSpatiotemporal = 1u << 6, SV_spatiotemporalPassedHeuristic = 1u << 7,
    SV_spatiotemporalSample = 1u << 8, SV_spatiotemporalSampleOnCurrent = 1u << 9,
    SV_spatiotemporalSampleOnTemporal = 1u << 10, SV_selectedSample = 1u << 11
};
static inline void svSet(uint32_t& sv, uint32_t bit, bool v) { sv = v ? (sv | bit) : (sv & ~bit); }

struct PreSampledLight { LightSample sample; float areaPDensity; };
static inline PreSampledLight readPreSampled(const void* buf, size_t i) {
    const float* q = static_cast<const float*>(buf) + 12 * i;
    PreSampledLight l;
    l.sample.emittance = RGB(q[0], q[1], q[2]);
    l.sample.position = V3(q[3], q[4], q[5]);
    l.sample.normal = V3(q[6], q[7], q[8]);
    l.sample.atInfinity = f2bits(q[9]) & 1u;
    l.areaPDensity = q[10];
    return l;
}
static inline void writePreSampled(void* buf, size_t i, const PreSampledLight& l) {
    float* q = static_cast<float*>(buf) + 12 * i;
    q[0] = l.sample.emittance.x; q[1] = l.sample.emittance.y; q[2] = l.sample.emittance.z;
    q[3] = l.sample.position.x; q[4] = l.sample.position.y; q[5] = l.sample.position.z;
    q[6] = l.sample.normal.x; q[7] = l.sample.normal.y; q[8] = l.sample.normal.z;
    q[9] = bits2f(l.sample.atInfinity & 1u); q[10] = l.areaPDensity; q[11] = 0.0f;
}

// per_pixel_ris.cu:6-40
static inline void lightPreSamplingThread(const Params& p, uint32_t linearThreadIndex) {
    const uint32_t indexInSubset = linearThreadIndex % kLightSubsetSize;
    uint64_t* rngs = static_cast<uint64_t*>(p.s->lightPreSamplingRngs);
    PCG32RNG rng; rng.setState(rngs[linearThreadIndex]);
    float probToSampleCurLightType = 1.0f;
    bool sampleEnvLight = false;
    if (p.envEnabled()) {
        if (p.scene->lightInstDist.integral() > 0.0f) {
            sampleEnvLight = indexInSubset < 0.25f * kLightSubsetSize;
            probToSampleCurLightType = sampleEnvLight ? 0.25f : (1 - 0.25f);
        }
        else sampleEnvLight = true;
    }
    PreSampledLight l;
    const float ul = rng.getFloat0cTo1o();
    const float u0 = rng.getFloat0cTo1o();
    const float u1 = rng.getFloat0cTo1o();
    sampleLight(*p.scene, p.f->envLightRotation, p.f->envLightPowerCoeff, V3(0.0f), ul, sampleEnvLight, u0, u1, &l.sample, &l.areaPDensity);
    l.areaPDensity *= probToSampleCurLightType;
    rngs[linearThreadIndex] = rng.state;
    writePreSampled(p.s->preSampledLights, linearThreadIndex, l);
}

// Shading point of the rearchitected passes: vOut = normalize(cam - p), frontHit from the unit vector
static inline void rearchShadingPoint(const Params& p, uint32_t bufIdx, size_t i, V3 camPos, ShadingPoint* sp) {
    const gfx_gbuffer2& gb2 = static_cast<const gfx_gbuffer2*>(p.s->gbuffer2[bufIdx])[i];
    const gfx_gbuffer3& gb3 = static_cast<const gfx_gbuffer3*>(p.s->gbuffer3[bufIdx])[i];
    V3 positionInWorld(gb2.positionInWorld[0], gb2.positionInWorld[1], gb2.positionInWorld[2]);
    const V3 geometricNormalInWorld = decodeNormal(gb2.qGeometricNormal);
    sp->vOut = normalize(camPos - positionInWorld);
    const float frontHit = dot(sp->vOut, geometricNormalInWorld) >= 0.0f ? 1.0f : -1.0f;
    sp->positionInWorld = offsetRayOrigin(positionInWorld, frontHit * geometricNormalInWorld);
    sp->shadingFrame = ReferenceFrame(decodeNormal(gb3.qShadingNormal), decodeVector(gb3.qShadingTangent));
    sp->vOutLocal = sp->shadingFrame.toLocal(sp->vOut);
    sp->bsdf.setup(p.scene->textures, p.scene->materials[gb3.matSlot], decodeTexCoords(gb3.qTexCoord));
    sp->dist = 0;
}

// per_pixel_ris.cu:44-128.  One call per 8x8 tile: the tile's (0,0) thread draws the subset index.
static inline void perPixelRISTile(const Params& p, int tileX, int tileY) {
    const uint32_t curBufIdx = p.f->bufferIndex;
    uint64_t* rngBuf = static_cast<uint64_t*>(p.s->rngBuffer);
    uint32_t perTileLightSubsetIndex = 0;
    {
        PCG32RNG rng0; rng0.setState(rngBuf[pix(p, tileX * kTileSizeX, tileY * kTileSizeY)]);
        perTileLightSubsetIndex = mapPrimarySampleToDiscrete(rng0.getFloat0cTo1o(), kNumLightSubsets);
    }
    const size_t subsetBase = static_cast<size_t>(perTileLightSubsetIndex) * kLightSubsetSize;
    for (int ty = 0; ty < kTileSizeY; ++ty)
        for (int tx = 0; tx < kTileSizeX; ++tx) {
            const int x = tileX * kTileSizeX + tx, y = tileY * kTileSizeY + ty;
            if (x >= p.s->imageSizeX || y >= p.s->imageSizeY) continue;
            const size_t i = pix(p, x, y);
            PCG32RNG rng; rng.setState(rngBuf[i]);
            if (tx == 0 && ty == 0) (void)rng.getFloat0cTo1o();
            const gfx_gbuffer0& gb0 = static_cast<const gfx_gbuffer0*>(p.s->gbuffer0[curBufIdx])[i];
            if (gb0.instSlot == 0xFFFFFFFFu) continue;
            ShadingPoint sp;
            rearchShadingPoint(p, curBufIdx, i, p.camera.position, &sp);
            Reservoir reservoir;
            reservoir.initialize(LightSample());
            float selectedTargetDensity = 0.0f;
            const uint32_t numCandidates = 1u << p.f->log2NumCandidateSamples;
            for (uint32_t c = 0; c < numCandidates; ++c) {
                const uint32_t lightIndex = mapPrimarySampleToDiscrete(rng.getFloat0cTo1o(), kLightSubsetSize);
                const PreSampledLight l = readPreSampled(p.s->preSampledLights, subsetBase + lightIndex);
                const RGB cont = performDirectLighting(false, VisibilityFn(), sp.positionInWorld, sp.vOutLocal, sp.shadingFrame, sp.bsdf, l.sample);
                const float targetDensity = convertToWeight(cont);
                const float weight = targetDensity / l.areaPDensity;
                if (reservoir.update(l.sample, weight, rng.getFloat0cTo1o())) selectedTargetDensity = targetDensity;
            }
            float recPDFEstimate = reservoir.sumWeights / (selectedTargetDensity * reservoir.streamLength);
            if (!finitef(recPDFEstimate)) { recPDFEstimate = 0.0f; selectedTargetDensity = 0.0f; }
            rngBuf[i] = rng.state;
            writeReservoir(p, p.currentReservoirIndex, i, reservoir);
            float* info = static_cast<float*>(p.s->reservoirInfoBuffer[p.currentReservoirIndex]) + 2 * i;
            info[0] = recPDFEstimate; info[1] = selectedTargetDensity;
        }
}

// The neighbour coordinates shared by traceShadowRays and shadeAndResample.
static inline void rearchTemporalCoord(const Params& p, uint32_t curBufIdx, size_t i, int x, int y, int* nx, int* ny) {
    const gfx_gbuffer1& gb1 = static_cast<const gfx_gbuffer1*>(p.s->gbuffer1[curBufIdx])[i];
    *nx = f2i(x + 0.5f - gb1.motionVector[0]);
    *ny = f2i(y + 0.5f - gb1.motionVector[1]);
}
static inline void rearchSpatialDelta(const Params& p, PCG32RNG& rng, int x, int y, float* deltaX, float* deltaY) {
    float radius = p.f->spatialNeighborRadius;
    if (p.f->useLowDiscrepancyNeighbors) {
        const uint32_t deltaIndex = p.spatialNeighborBaseIndex + 5u * static_cast<uint32_t>(x) + 7u * static_cast<uint32_t>(y);
        const float* d = static_cast<const float*>(p.s->spatialNeighborDeltas) + 2 * (deltaIndex % 1024);
        *deltaX = radius * d[0];
        *deltaY = radius * d[1];
    }
    else {
        radius *= std::sqrt(rng.getFloat0cTo1o());
        const float angle = 2 * kPi * rng.getFloat0cTo1o();
        float s, c; gm_sincos(angle, &s, &c);
        *deltaX = radius * c;
        *deltaY = radius * s;
    }
}

// optix_restir_di_rearch_kernels.cu:14-225
static inline void traceShadowRaysPixel(const Params& p, bool withTemporalRIS, bool withSpatialRIS, bool useUnbiasedEstimator, int x, int y) {
    const uint32_t curBufIdx = p.f->bufferIndex;
    const uint32_t prevBufIdx = (curBufIdx + 1) % 2;
    const uint32_t curRes = p.currentReservoirIndex, prevRes = (p.currentReservoirIndex + 1) % 2;
    const size_t i = pix(p, x, y);
    const gfx_gbuffer0& gb0 = static_cast<const gfx_gbuffer0*>(p.s->gbuffer0[curBufIdx])[i];
    if (gb0.instSlot == 0xFFFFFFFFu) return;
    const gfx_gbuffer2& gb2 = static_cast<const gfx_gbuffer2*>(p.s->gbuffer2[curBufIdx])[i];
    const gfx_gbuffer3& gb3 = static_cast<const gfx_gbuffer3*>(p.s->gbuffer3[curBufIdx])[i];
    V3 positionInWorld(gb2.positionInWorld[0], gb2.positionInWorld[1], gb2.positionInWorld[2]);
    const V3 geometricNormalInWorld = decodeNormal(gb2.qGeometricNormal);
    const V3 shadingNormalInWorld = decodeNormal(gb3.qShadingNormal);
    const V3 vOut = p.camera.position - positionInWorld;
    const float frontHit = dot(vOut, geometricNormalInWorld) >= 0.0f ? 1.0f : -1.0f;
    positionInWorld = offsetRayOrigin(positionInWorld, frontHit * geometricNormalInWorld);
    const float dist = length(vOut);
    const VisibilityFn visFn = [&p](V3 o, V3 d, float t0, float t1) { return !occluded(*p.accel, o, d, t0, t1); };
    const uint32_t* prevVisBuf = static_cast<const uint32_t*>(p.s->sampleVisibilityBuffer[prevBufIdx]);
    auto nbOrigin = [&](int nx, int ny) {
        const gfx_gbuffer2& n2 = static_cast<const gfx_gbuffer2*>(p.s->gbuffer2[prevBufIdx])[pix(p, nx, ny)];
        const V3 nbPos(n2.positionInWorld[0], n2.positionInWorld[1], n2.positionInWorld[2]);
        const V3 nbNg = decodeNormal(n2.qGeometricNormal);
        const V3 nbVOut = p.prevCamera.position - nbPos;
        const float nbFrontHit = dot(nbVOut, nbNg) >= 0.0f ? 1.0f : -1.0f;
        return offsetRayOrigin(nbPos, nbFrontHit * nbNg);
    };

    uint32_t sv = 0;
    LightSample newSample;
    bool newSampleIsValid;
    {
        const Reservoir reservoir = readReservoir(p, curRes, i);
        newSample = reservoir.sample;
        newSampleIsValid = reservoir.sumWeights > 0.0f;
        if (newSampleIsValid) svSet(sv, SV_newSample, evaluateVisibility(visFn, positionInWorld, newSample));
    }

    int tnx = 0, tny = 0;
    V3 tNbPositionInWorld(0.0f);
    bool temporalSampleIsValid = false;
    if (withTemporalRIS) {
        rearchTemporalCoord(p, curBufIdx, i, x, y, &tnx, &tny);
        svSet(sv, SV_temporalPassedHeuristic, testNeighbor(p, true, prevBufIdx, tnx, tny, dist, shadingNormalInWorld));
        if (sv & SV_temporalPassedHeuristic) {
            LightSample temporalSample;
            if (p.f->reuseVisibilityForTemporal && !useUnbiasedEstimator) {
                const uint32_t prevSv = prevVisBuf[pix(p, tnx, tny)];
                svSet(sv, SV_temporalSample, (prevSv & SV_selectedSample) != 0);
            }
            else {
                const Reservoir neighbor = readReservoir(p, prevRes, pix(p, tnx, tny));
                temporalSample = neighbor.sample;
                temporalSampleIsValid = neighbor.sumWeights > 0.0f;
                if (temporalSampleIsValid) svSet(sv, SV_temporalSample, evaluateVisibility(visFn, positionInWorld, temporalSample));
            }
            if (useUnbiasedEstimator) {
                tNbPositionInWorld = nbOrigin(tnx, tny);
                if (newSampleIsValid) svSet(sv, SV_newSampleOnTemporal, evaluateVisibility(visFn, tNbPositionInWorld, newSample));
                if (temporalSampleIsValid) svSet(sv, SV_temporalSampleOnCurrent, evaluateVisibility(visFn, positionInWorld, temporalSample));
            }
        }
    }

    int snx = 0, sny = 0;
    V3 stNbPositionInWorld(0.0f);
    bool spatiotemporalSampleIsValid = false;
    if (withSpatialRIS) {
        float deltaX, deltaY;
        PCG32RNG rng; rng.setState(static_cast<const uint64_t*>(p.s->rngBuffer)[i]);   // state change not stored (:150-152)
        rearchSpatialDelta(p, rng, x, y, &deltaX, &deltaY);
        snx = f2i(x + 0.5f + deltaX);
        sny = f2i(y + 0.5f + deltaY);
        bool passed = testNeighbor(p, true, prevBufIdx, snx, sny, dist, shadingNormalInWorld);
        passed &= snx != x || sny != y;
        svSet(sv, SV_spatiotemporalPassedHeuristic, passed);
        if (passed) {
            bool reused = false;
            if (p.f->reuseVisibilityForSpatiotemporal && !useUnbiasedEstimator) {
                const float threshold2 = pow2(p.f->radiusThresholdForSpatialVisReuse);
                const float dist2 = pow2(deltaX) + pow2(deltaY);
                reused = dist2 < threshold2;
            }
            LightSample spatiotemporalSample;
            if (reused) {
                const uint32_t prevSv = prevVisBuf[pix(p, snx, sny)];
                svSet(sv, SV_spatiotemporalSample, (prevSv & SV_selectedSample) != 0);
            }
            else {
                const Reservoir neighbor = readReservoir(p, prevRes, pix(p, snx, sny));
                spatiotemporalSample = neighbor.sample;
                spatiotemporalSampleIsValid = neighbor.sumWeights > 0.0f;
                if (spatiotemporalSampleIsValid)
                    svSet(sv, SV_spatiotemporalSample, evaluateVisibility(visFn, positionInWorld, spatiotemporalSample));
            }
            if (useUnbiasedEstimator) {
                stNbPositionInWorld = nbOrigin(snx, sny);
                if (newSampleIsValid) svSet(sv, SV_newSampleOnSpatiotemporal, evaluateVisibility(visFn, stNbPositionInWorld, newSample));
                if (spatiotemporalSampleIsValid)
                    svSet(sv, SV_spatiotemporalSampleOnCurrent, evaluateVisibility(visFn, positionInWorld, spatiotemporalSample));
            }
        }
    }

    if (useUnbiasedEstimator && withTemporalRIS && withSpatialRIS) {
        if ((sv & SV_temporalPassedHeuristic) && (sv & SV_spatiotemporalPassedHeuristic)) {
            if (temporalSampleIsValid) {
                const Reservoir tNeighbor = readReservoir(p, prevRes, pix(p, tnx, tny));
                svSet(sv, SV_temporalSampleOnSpatiotemporal, evaluateVisibility(visFn, stNbPositionInWorld, tNeighbor.sample));
            }
            if (spatiotemporalSampleIsValid) {
                const Reservoir stNeighbor = readReservoir(p, prevRes, pix(p, snx, sny));
                svSet(sv, SV_spatiotemporalSampleOnTemporal, evaluateVisibility(visFn, tNbPositionInWorld, stNeighbor.sample));
            }
        }
    }
    static_cast<uint32_t*>(p.s->sampleVisibilityBuffer[curBufIdx])[i] = sv;
}

enum class RearchSampleType { New = 0, Temporal, Spatiotemporal };

// optix_restir_di_rearch_kernels.cu:263-400 (useMIS_RIS = true)
static inline float computeMISWeight(const Params& p, RearchSampleType sampleType, bool withTemporalRIS, bool withSpatialRIS,
                                     uint32_t prevBufIdx, uint32_t prevRes, uint32_t maxPrevStreamLength, uint32_t sv,
                                     uint32_t selfStreamLength, const ShadingPoint& sp,
                                     int tnx, int tny, int snx, int sny,
                                     uint32_t streamLength, const LightSample& lightSample, float sampleTargetDensity) {
    const float numMisWeight = sampleTargetDensity;
    float denomMisWeight = numMisWeight * streamLength;
    if (sampleType != RearchSampleType::New) {
        const RGB cont = performDirectLighting(false, VisibilityFn(), sp.positionInWorld, sp.vOutLocal, sp.shadingFrame, sp.bsdf, lightSample);
        float targetDensity = convertToWeight(cont);
        if (p.f->useUnbiasedEstimator) {
            const uint32_t bit = sampleType == RearchSampleType::Temporal ? SV_temporalSampleOnCurrent : SV_spatiotemporalSampleOnCurrent;
            targetDensity *= (sv & bit) ? 1u : 0u;
        }
        denomMisWeight += targetDensity * selfStreamLength;
    }
    auto neighborTerm = [&](int nx, int ny, uint32_t visBit) {
        const size_t ni = pix(p, nx, ny);
        ShadingPoint nb;
        rearchShadingPoint(p, prevBufIdx, ni, p.prevCamera.position, &nb);
        const RGB cont = performDirectLighting(false, VisibilityFn(), nb.positionInWorld, nb.vOutLocal, nb.shadingFrame, nb.bsdf, lightSample);
        float nbTargetDensity = convertToWeight(cont);
        if (p.f->useUnbiasedEstimator) nbTargetDensity *= (sv & visBit) ? 1u : 0u;
        const Reservoir neighbor = readReservoir(p, prevRes, ni);
        const uint32_t nbStreamLength = neighbor.streamLength < maxPrevStreamLength ? neighbor.streamLength : maxPrevStreamLength;
        denomMisWeight += nbTargetDensity * nbStreamLength;
    };
    if (sampleType != RearchSampleType::Temporal && withTemporalRIS) {
        if (sv & SV_temporalPassedHeuristic)
            neighborTerm(tnx, tny, sampleType == RearchSampleType::New ? SV_newSampleOnTemporal : SV_spatiotemporalSampleOnTemporal);
    }
    if (sampleType != RearchSampleType::Spatiotemporal && withSpatialRIS) {
        if (sv & SV_spatiotemporalPassedHeuristic)
            neighborTerm(snx, sny, sampleType == RearchSampleType::New ? SV_newSampleOnSpatiotemporal : SV_temporalSampleOnSpatiotemporal);
    }
    return numMisWeight / denomMisWeight;
}

// optix_restir_di_rearch_kernels.cu:402-664
static inline void shadeAndResamplePixel(const Params& p, bool withTemporalRIS, bool withSpatialRIS, int x, int y) {
    const Scene& scene = *p.scene;
    const uint32_t curBufIdx = p.f->bufferIndex;
    const uint32_t prevBufIdx = (curBufIdx + 1) % 2;
    const uint32_t curRes = p.currentReservoirIndex, prevRes = (p.currentReservoirIndex + 1) % 2;
    const size_t i = pix(p, x, y);
    const gfx_gbuffer0& gb0 = static_cast<const gfx_gbuffer0*>(p.s->gbuffer0[curBufIdx])[i];
    const gfx_gbuffer3& gb3 = static_cast<const gfx_gbuffer3*>(p.s->gbuffer3[curBufIdx])[i];
    const V2 texCoord = decodeTexCoords(gb3.qTexCoord);
    RGB contribution(0.01f, 0.01f, 0.01f);
    if (gb0.instSlot != 0xFFFFFFFFu) {
        uint64_t* rngBuf = static_cast<uint64_t*>(p.s->rngBuffer);
        PCG32RNG rng; rng.setState(rngBuf[i]);
        int tnx = 0, tny = 0;
        if (withTemporalRIS) rearchTemporalCoord(p, curBufIdx, i, x, y, &tnx, &tny);
        int snx = 0, sny = 0;
        if (withSpatialRIS) {
            float deltaX, deltaY;
            rearchSpatialDelta(p, rng, x, y, &deltaX, &deltaY);
            snx = f2i(x + 0.5f + deltaX);
            sny = f2i(y + 0.5f + deltaY);
        }
        ShadingPoint sp;
        rearchShadingPoint(p, curBufIdx, i, p.camera.position, &sp);
        const MaterialData& mat = scene.materials[gb3.matSlot];
        contribution = RGB(0.0f);
        if (sp.vOutLocal.z > 0) {
            const RGB emittance = materialEmittance(scene.textures, mat, decodeTexCoords(gb3.qTexCoord));
            contribution += emittance / kPi;
        }
        uint32_t* visBuf = static_cast<uint32_t*>(p.s->sampleVisibilityBuffer[curBufIdx]);
        uint32_t sv = visBuf[i];
        float selectedTargetDensity = 0.0f;
        Reservoir combinedReservoir;
        uint32_t combinedStreamLength = 0;
        combinedReservoir.initialize(LightSample());
        RGB directCont(0.0f, 0.0f, 0.0f);
        float selectedMisWeight = 0.0f;
        const Reservoir selfRes = readReservoir(p, curRes, i);
        float* curInfo = static_cast<float*>(p.s->reservoirInfoBuffer[curRes]);
        const float* prevInfo = static_cast<const float*>(p.s->reservoirInfoBuffer[prevRes]);
        const ReservoirInfo selfResInfo{ curInfo[2 * i], curInfo[2 * i + 1] };
        const uint32_t selfStreamLength = selfRes.streamLength;
        const uint32_t maxPrevStreamLength = 20 * selfStreamLength;

        {
            if (selfResInfo.recPDFEstimate > 0.0f && (sv & SV_newSample)) {
                const LightSample lightSample = selfRes.sample;
                const RGB cont = performDirectLighting(false, VisibilityFn(), sp.positionInWorld, sp.vOutLocal, sp.shadingFrame, sp.bsdf, lightSample);
                const float targetDensity = convertToWeight(cont);
                float misWeight;
                if (withTemporalRIS || withSpatialRIS)
                    misWeight = computeMISWeight(p, RearchSampleType::New, withTemporalRIS, withSpatialRIS, prevBufIdx, prevRes,
                                                 maxPrevStreamLength, sv, selfStreamLength, sp, tnx, tny, snx, sny,
                                                 selfStreamLength, lightSample, selfResInfo.targetDensity);
                else
                    misWeight = 1.0f / selfStreamLength;
                directCont += (misWeight * selfResInfo.recPDFEstimate * selfStreamLength) * cont;
                combinedReservoir = selfRes;
                selectedTargetDensity = targetDensity;
                selectedMisWeight = misWeight;
                svSet(sv, SV_selectedSample, (sv & SV_newSample) != 0);
            }
            combinedStreamLength = selfStreamLength;
        }

        auto reuse = [&](RearchSampleType type, int nx, int ny, uint32_t sampleBit) {
            const size_t ni = pix(p, nx, ny);
            const Reservoir neighbor = readReservoir(p, prevRes, ni);
            const ReservoirInfo neighborInfo{ prevInfo[2 * ni], prevInfo[2 * ni + 1] };
            const uint32_t nbStreamLength = neighbor.streamLength < maxPrevStreamLength ? neighbor.streamLength : maxPrevStreamLength;
            if (neighborInfo.recPDFEstimate > 0.0f) {
                const LightSample nbLightSample = neighbor.sample;
                const RGB cont = performDirectLighting(false, VisibilityFn(), sp.positionInWorld, sp.vOutLocal, sp.shadingFrame, sp.bsdf, nbLightSample);
                const float targetDensity = convertToWeight(cont);
                const float misWeight = computeMISWeight(p, type, withTemporalRIS, withSpatialRIS, prevBufIdx, prevRes,
                                                         maxPrevStreamLength, sv, selfStreamLength, sp, tnx, tny, snx, sny,
                                                         nbStreamLength, nbLightSample, neighborInfo.targetDensity);
                const float weight = targetDensity * neighborInfo.recPDFEstimate * nbStreamLength;
                const uint32_t vis = (sv & sampleBit) ? 1u : 0u;
                directCont += (vis * misWeight * neighborInfo.recPDFEstimate * nbStreamLength) * cont;
                if (combinedReservoir.update(nbLightSample, weight, rng.getFloat0cTo1o())) {
                    selectedTargetDensity = targetDensity;
                    selectedMisWeight = misWeight;
                    svSet(sv, SV_selectedSample, vis != 0);
                }
            }
            combinedStreamLength += nbStreamLength;
        };
        if (withTemporalRIS && (sv & SV_temporalPassedHeuristic)) reuse(RearchSampleType::Temporal, tnx, tny, SV_temporalSample);
        if (withSpatialRIS && (sv & SV_spatiotemporalPassedHeuristic)) reuse(RearchSampleType::Spatiotemporal, snx, sny, SV_spatiotemporalSample);

        combinedReservoir.streamLength = combinedStreamLength;
        contribution += directCont;
        float recPDFEstimate = selectedMisWeight * combinedReservoir.sumWeights / selectedTargetDensity;
        if (!finitef(recPDFEstimate) || (p.f->reuseVisibility && !(sv & SV_selectedSample))) {
            recPDFEstimate = 0.0f;
            selectedTargetDensity = 0.0f;
        }
        visBuf[i] = sv;
        writeReservoir(p, curRes, i, combinedReservoir);
        curInfo[2 * i] = recPDFEstimate; curInfo[2 * i + 1] = selectedTargetDensity;
        rngBuf[i] = rng.state;
    }
    else if (p.envEnabled()) {
        contribution = p.f->envLightPowerCoeff * scene.env.fetch(texCoord.x, texCoord.y);
    }
    float* beauty = static_cast<float*>(p.s->beautyAccumBuffer) + 4 * i;
    RGB prev(0.0f, 0.0f, 0.0f);
    if (p.f->numAccumFrames > 0) prev = RGB(beauty[0], beauty[1], beauty[2]);
    const float curWeight = 1.0f / (1 + p.f->numAccumFrames);
    const RGB colorResult = (1 - curWeight) * prev + curWeight * contribution;
    beauty[0] = colorResult.x; beauty[1] = colorResult.y; beauty[2] = colorResult.z; beauty[3] = 1.0f;
}

} // namespace orc
