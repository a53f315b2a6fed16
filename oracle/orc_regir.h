// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// orc_regir.h: CPU restatement of the ReGIR grid-cell streaming-RIS kernels
//   sampleIntensity                          regir/gpu_kernels/build_cell_reservoirs.cu:6-68
//   buildCellReservoirsAndTemporalReuse<T>   :70-219
//   updateLastAccessFrameIndices             :229-243
//   calcCellLinearIndex                      regir/regir_shared.h:731-744
//   sampleFromCell                           regir/gpu_kernels/optix_pathtracing_kernels.cu:18-82
// Light-slot reservoirs use the same three-plane layout as the per-pixel reservoirs
// (include/gfxexp.h): plane k of buffer b at reservoirs[b] + 16 * (k * numLightSlots + slot).
#pragma once
#include "orc_restir.h"

namespace orc {

constexpr uint32_t kNumLightSlotsPerCell = 512; // regir_shared.h:7

struct RegirState {
    const gfx_regir_params* g;
    uint32_t numCells() const { return g->gridDimension[0] * g->gridDimension[1] * g->gridDimension[2]; }
    size_t numLightSlots() const { return static_cast<size_t>(numCells()) * kNumLightSlotsPerCell; }
    V3 gridOrigin() const { return V3(g->gridOrigin[0], g->gridOrigin[1], g->gridOrigin[2]); }
    V3 gridCellSize() const { return V3(g->gridCellSize[0], g->gridCellSize[1], g->gridCellSize[2]); }
    Reservoir readReservoir(uint32_t buf, size_t i) const {
        const size_t n = numLightSlots();
        const float* base = static_cast<const float*>(g->reservoirs[buf]);
        const float* p0 = base + 4 * i; const float* p1 = base + 4 * (n + i); const float* p2 = base + 4 * (2 * n + i);
        Reservoir r;
        r.sample.emittance = RGB(p0[0], p0[1], p0[2]);
        r.sample.position = V3(p0[3], p1[0], p1[1]);
        r.sample.normal = V3(p1[2], p1[3], p2[0]);
        r.sample.atInfinity = f2bits(p2[1]) & 1u;
        r.sumWeights = p2[2];
        r.streamLength = f2bits(p2[3]);
        return r;
    }
    void writeReservoir(uint32_t buf, size_t i, const Reservoir& r) const {
        const size_t n = numLightSlots();
        float* base = static_cast<float*>(g->reservoirs[buf]);
        float* p0 = base + 4 * i; float* p1 = base + 4 * (n + i); float* p2 = base + 4 * (2 * n + i);
        p0[0] = r.sample.emittance.x; p0[1] = r.sample.emittance.y; p0[2] = r.sample.emittance.z;
        p0[3] = r.sample.position.x; p1[0] = r.sample.position.y; p1[1] = r.sample.position.z;
        p1[2] = r.sample.normal.x; p1[3] = r.sample.normal.y; p2[0] = r.sample.normal.z;
        p2[1] = bits2f(r.sample.atInfinity & 1u); p2[2] = r.sumWeights; p2[3] = bits2f(r.streamLength);
    }
};

// build_cell_reservoirs.cu:6-68 (the half-space tests compare lpCos, which is still 1 there, with
// minSquaredDistance -- restated literally)
static inline RGB sampleIntensity(const Scene& scene, float envRotation, float envPowerCoeff,
                                  V3 cellCenter, V3 halfCellSize, float minSquaredDistance,
                                  float uLight, bool sampleEnvLight, float uPos0, float uPos1,
                                  LightSample* lightSample, float* probDensity) {
    sampleLight(scene, envRotation, envPowerCoeff, cellCenter, uLight, sampleEnvLight, uPos0, uPos1, lightSample, probDensity);
    float dist2 = minSquaredDistance;
    float lpCos = 1;
    const bool isOutsideCell =
        lightSample->atInfinity ||
        lightSample->position.x < cellCenter.x - halfCellSize.x || lightSample->position.x > cellCenter.x + halfCellSize.x ||
        lightSample->position.y < cellCenter.y - halfCellSize.y || lightSample->position.y > cellCenter.y + halfCellSize.y ||
        lightSample->position.z < cellCenter.z - halfCellSize.z || lightSample->position.z > cellCenter.z + halfCellSize.z;
    if (isOutsideCell) {
        const V3 shadowRayDir = lightSample->atInfinity ? lightSample->position : (lightSample->position - cellCenter);
        const float perpDistance = dot(-shadowRayDir, lightSample->normal);
        dist2 = sqLength(shadowRayDir);
        const float dist = std::sqrt(dist2);
        const bool cellIsInValidHalfSpace = lpCos > minSquaredDistance || lightSample->atInfinity;
        const bool cellIsInInvalidHalfSpace = lpCos < -minSquaredDistance;
        if (cellIsInValidHalfSpace) lpCos = perpDistance / dist;
        else if (cellIsInInvalidHalfSpace) lpCos = 0.0f;
    }
    if (lpCos > 0.0f) {
        const RGB Le = lightSample->emittance / kPi;
        return Le * (lpCos / dist2);
    }
    return RGB(0.0f, 0.0f, 0.0f);
}

// build_cell_reservoirs.cu:70-219, one light slot
static inline void buildCellReservoirThread(const Params& p, const RegirState& rs, bool useTemporalReuse, uint32_t linearThreadIndex) {
    const gfx_regir_params& g = *rs.g;
    const uint32_t frameIndex = p.f->frameIndex;
    const uint32_t bufferIndex = p.f->bufferIndex;
    const uint32_t cellLinearIndex = linearThreadIndex / kNumLightSlotsPerCell;
    const uint32_t lastAccessFrameIndex = static_cast<const uint32_t*>(g.lastAccessFrameIndices)[cellLinearIndex];
    if (frameIndex - lastAccessFrameIndex > 8) return;
    const uint32_t gx = g.gridDimension[0], gy = g.gridDimension[1];
    const uint32_t iz = cellLinearIndex / (gx * gy);
    const uint32_t iy = (cellLinearIndex % (gx * gy)) / gx;
    const uint32_t ix = cellLinearIndex % gx;
    const V3 cs = rs.gridCellSize();
    const V3 cellCenter = rs.gridOrigin() + V3((ix + 0.5f) * cs.x, (iy + 0.5f) * cs.y, (iz + 0.5f) * cs.z);
    const V3 halfCellSize = 0.5f * cs;
    const float minSquaredDistance = sqLength(0.5f * cs);
    uint64_t* rngs = static_cast<uint64_t*>(g.lightSlotRngs);
    PCG32RNG rng; rng.setState(rngs[linearThreadIndex]);
    float selectedTargetPDensity = 0.0f;
    Reservoir reservoir;
    reservoir.initialize(LightSample());
    const uint32_t numCandidates = 1u << g.log2NumCandidatesPerLightSlot;
    for (uint32_t candIdx = 0; candIdx < numCandidates; ++candIdx) {
        float uLight = rng.getFloat0cTo1o();
        bool sampleEnvLight = false;
        float probToSampleCurLightType = 1.0f;
        if (p.envEnabled()) {
            if (p.scene->lightInstDist.integral() > 0.0f) {
                const float prob = fmin2(fmax2(0.25f * numCandidates - candIdx, 0.0f), 1.0f);
                if (uLight < prob) { probToSampleCurLightType = 0.25f; uLight = uLight / prob; sampleEnvLight = true; }
                else { probToSampleCurLightType = 1.0f - 0.25f; uLight = (uLight - prob) / (1 - prob); }
            }
            else sampleEnvLight = true;
        }
        LightSample lightSample;
        float areaPDensity;
        const float u0 = rng.getFloat0cTo1o();
        const float u1 = rng.getFloat0cTo1o();
        const RGB cont = sampleIntensity(*p.scene, p.f->envLightRotation, p.f->envLightPowerCoeff, cellCenter, halfCellSize,
                                         minSquaredDistance, uLight, sampleEnvLight, u0, u1, &lightSample, &areaPDensity);
        areaPDensity *= probToSampleCurLightType;
        const float targetPDensity = convertToWeight(cont);
        const float weight = targetPDensity / areaPDensity;
        if (reservoir.update(lightSample, weight, rng.getFloat0cTo1o())) selectedTargetPDensity = targetPDensity;
    }
    float recPDFEstimate = reservoir.sumWeights / (selectedTargetPDensity * reservoir.streamLength);
    if (!finitef(recPDFEstimate)) { recPDFEstimate = 0.0f; selectedTargetPDensity = 0.0f; }
    if (useTemporalReuse) {
        const uint32_t prevBufferIndex = (bufferIndex + 1) % 2;
        const uint32_t selfStreamLength = reservoir.streamLength;
        if (recPDFEstimate == 0.0f) reservoir.initialize(LightSample());
        uint32_t combinedStreamLength = selfStreamLength;
        const uint32_t maxNumPrevSamples = 20 * selfStreamLength;
        const Reservoir prevReservoir = rs.readReservoir(prevBufferIndex, linearThreadIndex);
        const float prevTargetDensity = static_cast<const float*>(g.reservoirInfos[prevBufferIndex])[2 * linearThreadIndex + 1];
        const uint32_t prevStreamLength = prevReservoir.streamLength < maxNumPrevSamples ? prevReservoir.streamLength : maxNumPrevSamples;
        const float lengthCorrection = static_cast<float>(prevStreamLength) / prevReservoir.streamLength;
        const float weight = lengthCorrection * prevReservoir.sumWeights;
        if (reservoir.update(prevReservoir.sample, weight, rng.getFloat0cTo1o())) selectedTargetPDensity = prevTargetDensity;
        combinedStreamLength += prevStreamLength;
        reservoir.streamLength = combinedStreamLength;
        const float weightForEstimate = 1.0f / reservoir.streamLength;
        recPDFEstimate = weightForEstimate * reservoir.sumWeights / selectedTargetPDensity;
        if (!finitef(recPDFEstimate)) { recPDFEstimate = 0.0f; selectedTargetPDensity = 0.0f; }
    }
    rngs[linearThreadIndex] = rng.state;
    rs.writeReservoir(bufferIndex, linearThreadIndex, reservoir);
    float* info = static_cast<float*>(g.reservoirInfos[bufferIndex]) + 2 * linearThreadIndex;
    info[0] = recPDFEstimate; info[1] = selectedTargetPDensity;
}

// regir_shared.h:731-744
static inline uint32_t calcCellLinearIndex(const RegirState& rs, V3 positionInWorld) {
    const gfx_regir_params& g = *rs.g;
    const V3 relPos = positionInWorld - rs.gridOrigin();
    const V3 cs = rs.gridCellSize();
    auto clampIdx = [](float v, uint32_t dim) { const uint32_t i = f2u(v); return i < dim - 1 ? i : dim - 1; };
    const uint32_t ix = clampIdx(relPos.x / cs.x, g.gridDimension[0]);
    const uint32_t iy = clampIdx(relPos.y / cs.y, g.gridDimension[1]);
    const uint32_t iz = clampIdx(relPos.z / cs.z, g.gridDimension[2]);
    return iz * g.gridDimension[0] * g.gridDimension[1] + iy * g.gridDimension[0] + ix;
}

// regir/gpu_kernels/optix_pathtracing_kernels.cu:18-82
static inline RGB sampleFromCell(const Params& p, const RegirState& rs, V3 shadingPoint, V3 vOutLocal, const ReferenceFrame& shadingFrame,
                                 const BSDF& bsdf, PCG32RNG& rng, LightSample* lightSample, float* recProbDensityEstimate) {
    const gfx_regir_params& g = *rs.g;
    V3 randomOffset(0.0f);
    if (g.enableCellRandomization) {
        const float r0 = rng.getFloat0cTo1o();
        const float r1 = rng.getFloat0cTo1o();
        const float r2 = rng.getFloat0cTo1o();
        randomOffset = rs.gridCellSize() * V3(-0.5f + r0, -0.5f + r1, -0.5f + r2);
    }
    const uint32_t cellLinearIndex = calcCellLinearIndex(rs, shadingPoint + randomOffset);
    const size_t resStartIndex = static_cast<size_t>(kNumLightSlotsPerCell) * cellLinearIndex;
    uint32_t* accesses = static_cast<uint32_t*>(g.perCellNumAccesses);
#pragma omp atomic
    accesses[cellLinearIndex] += 1u;
    const uint32_t numResampling = 1u << g.log2NumCandidatesPerCell;
    Reservoir combinedReservoir;
    combinedReservoir.initialize(LightSample());
    uint32_t combinedStreamLength = 0;
    RGB selectedContribution(0.0f);
    float selectedTargetPDensity = 0.0f;
    const uint32_t bufferIndex = p.f->bufferIndex;
    for (uint32_t i = 0; i < numResampling; ++i) {
        const size_t lightSlotIdx = resStartIndex + mapPrimarySampleToDiscrete(rng.getFloat0cTo1o(), kNumLightSlotsPerCell);
        const Reservoir r = rs.readReservoir(bufferIndex, lightSlotIdx);
        const float recPDF = static_cast<const float*>(g.reservoirInfos[bufferIndex])[2 * lightSlotIdx];
        const uint32_t streamLength = r.streamLength;
        combinedStreamLength += streamLength;
        if (recPDF == 0.0f) continue;
        const RGB cont = performDirectLighting(false, VisibilityFn(), shadingPoint, vOutLocal, shadingFrame, bsdf, r.sample);
        const float targetPDensity = convertToWeight(cont);
        const float weight = targetPDensity * recPDF * streamLength;
        if (combinedReservoir.update(r.sample, weight, rng.getFloat0cTo1o())) {
            selectedContribution = cont;
            selectedTargetPDensity = targetPDensity;
        }
    }
    combinedReservoir.streamLength = combinedStreamLength;
    *lightSample = combinedReservoir.sample;
    const float weightForEstimate = 1.0f / combinedReservoir.streamLength;
    *recProbDensityEstimate = weightForEstimate * combinedReservoir.sumWeights / selectedTargetPDensity;
    if (!finitef(*recProbDensityEstimate)) *recProbDensityEstimate = 0.0f;
    return selectedContribution;
}

} // namespace orc
