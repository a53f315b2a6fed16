// restir_rearch.hip.h -- the rearchitected ReSTIR DI passes (included by restir.hip).
//
//   pass (gfx_restir_pass)                  kernels
//   LIGHT_PRESAMPLING                       k_light_presample            per_pixel_ris.cu:6-40
//   PER_PIXEL_RIS                           k_per_pixel_ris              per_pixel_ris.cu:44-128
//   TRACE_SHADOW_RAYS[_T|_S|_ST][_UNBIASED] k_rearch_emit<T,S,U> -> trace any -> k_rearch_vis_finish
//                                                                        optix_restir_di_rearch_kernels.cu:14-225
//   SHADE_AND_RESAMPLE[_T|_S|_ST]           k_rearch_shade<T,S>          optix_restir_di_rearch_kernels.cu:263-664
//
// The 8x8 pixel tile of the reference (restir_di_shared.h:10-11) is exactly one 64-lane wavefront:
// lane = ty * 8 + tx, lane 0 draws the tile's light-subset index and broadcasts it with a shuffle
// (the reference uses a __shared__ word + __syncthreads).
// traceShadowRays' up to nine visibility queries per pixel are seven distinct rays (temporalSample /
// temporalSampleOnCurrent and spatiotemporalSample / spatiotemporalSampleOnCurrent are the same ray);
// each ray kind is compacted into the any-hit queue with one ballot per wave, so consecutive queue
// entries are the same kind of ray from neighbouring pixels.
#pragma once
#include "restir_common.hip.h"

namespace gfx {

constexpr uint32_t kNumLightSubsets = 128;    // restir_di_shared.h:8
constexpr uint32_t kLightSubsetSize = 1024;   // :9
constexpr uint32_t kNumPreSampledLights = kNumLightSubsets * kLightSubsetSize;
constexpr int kRearchRayKinds = 7;

// SampleVisibility bits (restir_di_shared.h:146-164)
enum : uint32_t {
    SV_newSample = 1u << 0, SV_newSampleOnTemporal = 1u << 1, SV_newSampleOnSpatiotemporal = 1u << 2,
    SV_temporalPassedHeuristic = 1u << 3, SV_temporalSample = 1u << 4, SV_temporalSampleOnCurrent = 1u << 5,
    SV_temporalSampleOnSpatiotemporal = 1u << 6, SV_spatiotemporalPassedHeuristic = 1u << 7,
    SV_spatiotemporalSample = 1u << 8, SV_spatiotemporalSampleOnCurrent = 1u << 9,
    SV_spatiotemporalSampleOnTemporal = 1u << 10, SV_selectedSample = 1u << 11
};

// PreSampledLight (restir_di_shared.h:98-101) as three 16-byte words
struct PreSampled { LightSample sample; float areaPDensity; };
GFX_DEV PreSampled load_presampled(const void* buf, size_t i) {
    const float4* q = static_cast<const float4*>(buf) + 3 * i;
    const float4 a = q[0], b = q[1], c = q[2];
    PreSampled l;
    l.sample.emittance = f3(a.x, a.y, a.z);
    l.sample.position = f3(a.w, b.x, b.y);
    l.sample.normal = f3(b.z, b.w, c.x);
    l.sample.atInfinity = f2bits(c.y) & 1u;
    l.areaPDensity = c.z;
    return l;
}

__global__ __launch_bounds__(kBlock) void k_light_presample(RestirArgs a) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= kNumPreSampledLights) return;
    const uint32_t indexInSubset = i % kLightSubsetSize;
    uint64_t* rngs = static_cast<uint64_t*>(a.s.lightPreSamplingRngs);
    Pcg32 rng; rng.state = rngs[i];
    const EnvMap env = load_env(a.s);
    float probCurType = 1.0f;
    bool sampleEnv = false;
    if (env.present() && a.f.enableEnvLight) {
        if (*a.scene.lightInstIntegral > 0.0f) {
            sampleEnv = indexInSubset < 0.25f * kLightSubsetSize;
            probCurType = sampleEnv ? 0.25f : (1 - 0.25f);
        }
        else sampleEnv = true;
    }
    LightSample ls;
    ls.emittance = f3(0.0f); ls.position = f3(0.0f); ls.normal = f3(0.0f); ls.atInfinity = 0;
    float pd;
    const float ul = rng.uniform();
    const float u0 = rng.uniform();
    const float u1 = rng.uniform();
    sample_light(a.scene, env, a.f.envLightRotation, a.f.envLightPowerCoeff, ul, sampleEnv, u0, u1, ls, pd);
    pd *= probCurType;
    rngs[i] = rng.state;
    float4* q = static_cast<float4*>(a.s.preSampledLights) + 3ull * i;
    q[0] = make_float4(ls.emittance.x, ls.emittance.y, ls.emittance.z, ls.position.x);
    q[1] = make_float4(ls.position.y, ls.position.z, ls.normal.x, ls.normal.y);
    q[2] = make_float4(ls.normal.z, bits2f(ls.atInfinity & 1u), pd, 0.0f);
}

// One wave per 8x8 tile; tiles enumerated row-major over the band of rows [rowBegin, rowEnd).
__global__ __launch_bounds__(kBlock) void k_per_pixel_ris(RestirArgs a) {
    const size_t numPixels = static_cast<size_t>(a.s.imageSizeX) * a.s.imageSizeY;
    const int W = a.s.imageSizeX, H = a.s.imageSizeY;
    const int tilesX = (W + 7) / 8;
    const int rowBegin = static_cast<int>(a.px.rowBegin), rowEnd = static_cast<int>(a.px.rowEnd);
    const int tileRowBegin = rowBegin / 8, tileRowEnd = (rowEnd + 7) / 8;
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (tile >= tilesX * (tileRowEnd - tileRowBegin)) return;      // whole wave
    const int tileX = tile % tilesX, tileY = tileRowBegin + tile / tilesX;
    const int x = tileX * 8 + (lane & 7), y = tileY * 8 + (lane >> 3);
    const bool inImage = x < W && y < H;
    const size_t p = static_cast<size_t>(y) * W + x;
    const uint32_t bufIdx = a.f.bufferIndex;
    uint64_t* rngBuf = static_cast<uint64_t*>(a.s.rngBuffer);
    Pcg32 rng; rng.state = inImage ? rngBuf[p] : 0ull;
    uint32_t subset = 0;
    if (lane == 0) {
        subset = f2u_sat(rng.uniform() * kNumLightSubsets);
        if (subset > kNumLightSubsets - 1) subset = kNumLightSubsets - 1;
    }
    subset = __shfl(subset, 0);
    if (!inImage) return;
    if (static_cast<const uint4*>(a.s.gbuffer0[bufIdx])[p].x == 0xFFFFFFFFu) return;
    const size_t subsetBase = static_cast<size_t>(subset) * kLightSubsetSize;
    const Camera cam = load_camera(a.f.camera);
    ShadingPoint sp;
    make_shading_point(a, bufIdx, p, cam.pos, true, sp);
    Reservoir reservoir;
    reservoir.reset();
    float selectedTarget = 0.0f;
    const uint32_t numCandidates = 1u << a.f.log2NumCandidateSamples;
    for (uint32_t c = 0; c < numCandidates; ++c) {
        uint32_t lightIndex = f2u_sat(rng.uniform() * kLightSubsetSize);
        if (lightIndex > kLightSubsetSize - 1) lightIndex = kLightSubsetSize - 1;
        const PreSampled l = load_presampled(a.s.preSampledLights, subsetBase + lightIndex);
        const f3 cont = direct_lighting(sp.pos, sp.vOutLocal, sp.frame, sp.bsdf, l.sample);
        const float target = target_weight(cont);
        const float weight = target / l.areaPDensity;
        if (reservoir.update(l.sample, weight, rng.uniform())) selectedTarget = target;
    }
    float recPDF = reservoir.sumWeights / (selectedTarget * reservoir.streamLength);
    if (!is_finite(recPDF)) { recPDF = 0.0f; selectedTarget = 0.0f; }
    rngBuf[p] = rng.state;
    store_reservoir(a.s.reservoirBuffer[a.curRes], numPixels, p, reservoir);
    static_cast<float2*>(a.s.reservoirInfoBuffer[a.curRes])[p] = make_float2(recPDF, selectedTarget);
}

GFX_DEV void rearch_temporal_coord(const RestirArgs& a, uint32_t bufIdx, size_t p, int x, int y, int& nx, int& ny) {
    const float2 mv = static_cast<const float2*>(a.s.gbuffer1[bufIdx])[p];
    nx = f2i_sat(x + 0.5f - mv.x);
    ny = f2i_sat(y + 0.5f - mv.y);
}
GFX_DEV void rearch_spatial_delta(const RestirArgs& a, Pcg32& rng, int x, int y, float& dx, float& dy) {
    float radius = a.f.spatialNeighborRadius;
    if (a.f.useLowDiscrepancyNeighbors) {
        const uint32_t deltaIndex = a.baseIdx + 5u * static_cast<uint32_t>(x) + 7u * static_cast<uint32_t>(y);
        const float2 d = static_cast<const float2*>(a.s.spatialNeighborDeltas)[deltaIndex % 1024];
        dx = radius * d.x;
        dy = radius * d.y;
    }
    else {
        radius *= sqrtf(rng.uniform());
        const float angle = 2 * kPi * rng.uniform();
        float s, c; gm_sincos(angle, s, c);
        dx = radius * c;
        dy = radius * s;
    }
}
// offset ray origin of a previous-frame neighbour as traceShadowRays computes it (:96-102, :189-195)
GFX_DEV f3 rearch_neighbor_origin(const RestirArgs& a, uint32_t prevBuf, size_t np, f3 prevCamPos) {
    const float4 g2 = static_cast<const float4*>(a.s.gbuffer2[prevBuf])[np];
    const f3 pos(g2.x, g2.y, g2.z);
    const f3 ng = decode_dir(f2bits(g2.w));
    const f3 vOut = prevCamPos - pos;
    const float frontHit = dot(vOut, ng) >= 0.0f ? 1.0f : -1.0f;
    return offset_ray_origin(pos, frontHit * ng);
}

// Ray kinds: 0 new@current 1 temporal@current 2 new@temporal 3 spatiotemporal@current
//            4 new@spatiotemporal 5 temporal@spatiotemporal 6 spatiotemporal@temporal
template <bool TEMPORAL, bool SPATIAL, bool UNBIASED>
__global__ __launch_bounds__(kBlock) void k_rearch_emit(RestirArgs a) {
    const size_t numPixels = static_cast<size_t>(a.s.imageSizeX) * a.s.imageSizeY;
    const PixelId px = pixel_of_thread(a.px);
    const size_t p = px.p;
    const uint32_t bufIdx = a.f.bufferIndex;
    const uint32_t prevBuf = (bufIdx + 1) % 2, prevRes = (a.curRes + 1) % 2;
    bool surface = false;
    if (px.valid) surface = static_cast<const uint4*>(a.s.gbuffer0[bufIdx])[p].x != 0xFFFFFFFFu;
    uint32_t sv = 0;
    bool want[kRearchRayKinds] = { false, false, false, false, false, false, false };
    f3 pos(0.0f), tPos(0.0f), stPos(0.0f);
    LightSample newS, tS, stS;
    newS.emittance = newS.position = newS.normal = f3(0.0f); newS.atInfinity = 0;
    tS = newS; stS = newS;
    if (surface) {
        const int x = px.x, y = px.y;
        const Camera cam = load_camera(a.f.camera);
        const f3 prevCamPos(a.f.prevCamera.position[0], a.f.prevCamera.position[1], a.f.prevCamera.position[2]);
        const float4 g2 = static_cast<const float4*>(a.s.gbuffer2[bufIdx])[p];
        const uint32_t qns = static_cast<const uint4*>(a.s.gbuffer3[bufIdx])[p].x;
        pos = f3(g2.x, g2.y, g2.z);
        const f3 ng = decode_dir(f2bits(g2.w));
        const f3 ns = decode_dir(qns);
        const f3 vOut = cam.pos - pos;
        const float frontHit = dot(vOut, ng) >= 0.0f ? 1.0f : -1.0f;
        pos = offset_ray_origin(pos, frontHit * ng);
        const float dist = len(vOut);
        const uint32_t* prevVis = static_cast<const uint32_t*>(a.s.sampleVisibilityBuffer[prevBuf]);

        bool newValid;
        {
            const Reservoir r = load_reservoir(a.s.reservoirBuffer[a.curRes], numPixels, p);
            newS = r.sample;
            newValid = r.sumWeights > 0.0f;
            want[0] = newValid;
        }
        int tnx = 0, tny = 0;
        bool tValid = false, tPassed = false;
        if (TEMPORAL) {
            rearch_temporal_coord(a, bufIdx, p, x, y, tnx, tny);
            tPassed = test_neighbor(a, true, prevBuf, tnx, tny, dist, ns, cam.pos);
            if (tPassed) {
                sv |= SV_temporalPassedHeuristic;
                const size_t np = static_cast<size_t>(tny) * a.s.imageSizeX + tnx;
                if (a.f.reuseVisibilityForTemporal && !UNBIASED) {
                    if (prevVis[np] & SV_selectedSample) sv |= SV_temporalSample;
                }
                else {
                    const Reservoir nb = load_reservoir(a.s.reservoirBuffer[prevRes], numPixels, np);
                    tS = nb.sample;
                    tValid = nb.sumWeights > 0.0f;
                    want[1] = tValid;
                }
                if (UNBIASED) {
                    tPos = rearch_neighbor_origin(a, prevBuf, np, prevCamPos);
                    want[2] = newValid;
                }
            }
        }
        int snx = 0, sny = 0;
        bool stValid = false, stPassed = false;
        if (SPATIAL) {
            float dx, dy;
            Pcg32 rng; rng.state = static_cast<const uint64_t*>(a.s.rngBuffer)[p];   // state change not stored
            rearch_spatial_delta(a, rng, x, y, dx, dy);
            snx = f2i_sat(x + 0.5f + dx);
            sny = f2i_sat(y + 0.5f + dy);
            stPassed = test_neighbor(a, true, prevBuf, snx, sny, dist, ns, cam.pos);
            stPassed = stPassed && (snx != x || sny != y);
            if (stPassed) {
                sv |= SV_spatiotemporalPassedHeuristic;
                const size_t np = static_cast<size_t>(sny) * a.s.imageSizeX + snx;
                bool reused = false;
                if (a.f.reuseVisibilityForSpatiotemporal && !UNBIASED) {
                    const float threshold2 = a.f.radiusThresholdForSpatialVisReuse * a.f.radiusThresholdForSpatialVisReuse;
                    const float dist2 = dx * dx + dy * dy;
                    reused = dist2 < threshold2;
                }
                if (reused) {
                    if (prevVis[np] & SV_selectedSample) sv |= SV_spatiotemporalSample;
                }
                else {
                    const Reservoir nb = load_reservoir(a.s.reservoirBuffer[prevRes], numPixels, np);
                    stS = nb.sample;
                    stValid = nb.sumWeights > 0.0f;
                    want[3] = stValid;
                }
                if (UNBIASED) {
                    stPos = rearch_neighbor_origin(a, prevBuf, np, prevCamPos);
                    want[4] = newValid;
                }
            }
        }
        if (UNBIASED && TEMPORAL && SPATIAL) {
            if (tPassed && stPassed) {
                want[5] = tValid;
                want[6] = stValid;
            }
        }
    }
    // kind -> (origin, sample); every lane of the wave takes part in every append
    const f3* orgs[kRearchRayKinds] = { &pos, &pos, &tPos, &pos, &stPos, &stPos, &tPos };
    const LightSample* smps[kRearchRayKinds] = { &newS, &tS, &newS, &stS, &newS, &tS, &stS };
    // kinds this instantiation never emits stay want = false; one reservation for all seven kinds
    uint32_t slots[kRearchRayKinds];
    queue_reserve<kRearchRayKinds>(want, a.rayCount, slots);
#pragma unroll
    for (int k = 0; k < kRearchRayKinds; ++k) {
        if (k == 1 && !TEMPORAL) continue;
        if ((k == 2 || k == 5 || k == 6) && !(TEMPORAL && UNBIASED)) continue;
        if (k == 3 && !SPATIAL) continue;
        if ((k == 4 || k == 5 || k == 6) && !(SPATIAL && UNBIASED)) continue;
        if (slots[k] != GFX_INVALID_SLOT) {
            const ShadowRay sr = shadow_ray(*orgs[k], *smps[k]);
            queue_write(slots[k], *orgs[k], sr.dir, 0.0f, sr.tmax, a.rayOrg, a.rayDir);
        }
        if (px.valid) a.rearchSlots[static_cast<size_t>(k) * numPixels + p] = slots[k];
    }
    if (px.valid) a.pixelRaySlot[p] = sv;
}

template <bool TEMPORAL, bool SPATIAL, bool UNBIASED>
__global__ __launch_bounds__(kBlock) void k_rearch_vis_finish(RestirArgs a) {
    const size_t numPixels = static_cast<size_t>(a.s.imageSizeX) * a.s.imageSizeY;
    const PixelId px = pixel_of_thread(a.px);
    if (!px.valid) return;
    const size_t p = px.p;
    const uint32_t bufIdx = a.f.bufferIndex;
    if (static_cast<const uint4*>(a.s.gbuffer0[bufIdx])[p].x == 0xFFFFFFFFu) return;   // the reference returns before the write
    uint32_t sv = a.pixelRaySlot[p];
    auto visible = [&](int k) {
        const uint32_t slot = a.rearchSlots[static_cast<size_t>(k) * numPixels + p];
        return slot != GFX_INVALID_SLOT && !a.occluded[slot];
    };
    if (visible(0)) sv |= SV_newSample;
    if (TEMPORAL) {
        if (visible(1)) sv |= SV_temporalSample | (UNBIASED ? SV_temporalSampleOnCurrent : 0u);
        if (UNBIASED && visible(2)) sv |= SV_newSampleOnTemporal;
    }
    if (SPATIAL) {
        if (visible(3)) sv |= SV_spatiotemporalSample | (UNBIASED ? SV_spatiotemporalSampleOnCurrent : 0u);
        if (UNBIASED && visible(4)) sv |= SV_newSampleOnSpatiotemporal;
    }
    if (UNBIASED && TEMPORAL && SPATIAL) {
        if (visible(5)) sv |= SV_temporalSampleOnSpatiotemporal;
        if (visible(6)) sv |= SV_spatiotemporalSampleOnTemporal;
    }
    static_cast<uint32_t*>(a.s.sampleVisibilityBuffer[bufIdx])[p] = sv;
}

enum { kSampleNew = 0, kSampleTemporal = 1, kSampleSpatiotemporal = 2 };

// computeMISWeight<sampleType, T, S>, optix_restir_di_rearch_kernels.cu:263-400 (useMIS_RIS = true)
template <int TYPE, bool TEMPORAL, bool SPATIAL>
GFX_DEV float rearch_mis_weight(const RestirArgs& a, size_t numPixels, uint32_t prevBuf, uint32_t prevRes, uint32_t maxPrevStreamLength,
                                uint32_t sv, uint32_t selfStreamLength, const ShadingPoint& sp, size_t tnp, size_t snp, f3 prevCamPos,
                                uint32_t streamLength, const LightSample& ls, float sampleTarget) {
    const float num = sampleTarget;
    float denom = num * streamLength;
    if (TYPE != kSampleNew) {
        float target = target_weight(direct_lighting(sp.pos, sp.vOutLocal, sp.frame, sp.bsdf, ls));
        if (a.f.useUnbiasedEstimator) {
            const uint32_t bit = TYPE == kSampleTemporal ? SV_temporalSampleOnCurrent : SV_spatiotemporalSampleOnCurrent;
            target *= (sv & bit) ? 1u : 0u;
        }
        denom += target * selfStreamLength;
    }
    if (TYPE != kSampleTemporal && TEMPORAL) {
        if (sv & SV_temporalPassedHeuristic) {
            ShadingPoint nb;
            make_shading_point(a, prevBuf, tnp, prevCamPos, true, nb);
            float nbTarget = target_weight(direct_lighting(nb.pos, nb.vOutLocal, nb.frame, nb.bsdf, ls));
            if (a.f.useUnbiasedEstimator) {
                const uint32_t bit = TYPE == kSampleNew ? SV_newSampleOnTemporal : SV_spatiotemporalSampleOnTemporal;
                nbTarget *= (sv & bit) ? 1u : 0u;
            }
            const uint32_t nbLen = f2bits(static_cast<const float4*>(a.s.reservoirBuffer[prevRes])[2 * numPixels + tnp].w);
            denom += nbTarget * (nbLen < maxPrevStreamLength ? nbLen : maxPrevStreamLength);
        }
    }
    if (TYPE != kSampleSpatiotemporal && SPATIAL) {
        if (sv & SV_spatiotemporalPassedHeuristic) {
            ShadingPoint nb;
            make_shading_point(a, prevBuf, snp, prevCamPos, true, nb);
            float nbTarget = target_weight(direct_lighting(nb.pos, nb.vOutLocal, nb.frame, nb.bsdf, ls));
            if (a.f.useUnbiasedEstimator) {
                const uint32_t bit = TYPE == kSampleNew ? SV_newSampleOnSpatiotemporal : SV_temporalSampleOnSpatiotemporal;
                nbTarget *= (sv & bit) ? 1u : 0u;
            }
            const uint32_t nbLen = f2bits(static_cast<const float4*>(a.s.reservoirBuffer[prevRes])[2 * numPixels + snp].w);
            denom += nbTarget * (nbLen < maxPrevStreamLength ? nbLen : maxPrevStreamLength);
        }
    }
    return num / denom;
}

template <bool TEMPORAL, bool SPATIAL>
__global__ __launch_bounds__(kBlock) void k_rearch_shade(RestirArgs a) {
    const size_t numPixels = static_cast<size_t>(a.s.imageSizeX) * a.s.imageSizeY;
    const PixelId px = pixel_of_thread(a.px);
    if (!px.valid) return;
    const size_t p = px.p;
    const int x = px.x, y = px.y;
    const uint32_t bufIdx = a.f.bufferIndex;
    const uint32_t prevBuf = (bufIdx + 1) % 2, prevRes = (a.curRes + 1) % 2;
    const uint32_t instSlot = static_cast<const uint4*>(a.s.gbuffer0[bufIdx])[p].x;
    const uint4 g3 = static_cast<const uint4*>(a.s.gbuffer3[bufIdx])[p];
    f3 contribution(0.01f, 0.01f, 0.01f);
    if (instSlot != 0xFFFFFFFFu) {
        uint64_t* rngBuf = static_cast<uint64_t*>(a.s.rngBuffer);
        Pcg32 rng; rng.state = rngBuf[p];
        int tnx = 0, tny = 0, snx = 0, sny = 0;
        if (TEMPORAL) rearch_temporal_coord(a, bufIdx, p, x, y, tnx, tny);
        if (SPATIAL) {
            float dx, dy;
            rearch_spatial_delta(a, rng, x, y, dx, dy);
            snx = f2i_sat(x + 0.5f + dx);
            sny = f2i_sat(y + 0.5f + dy);
        }
        const size_t tnp = static_cast<size_t>(tny) * a.s.imageSizeX + tnx;
        const size_t snp = static_cast<size_t>(sny) * a.s.imageSizeX + snx;
        const Camera cam = load_camera(a.f.camera);
        const f3 prevCamPos(a.f.prevCamera.position[0], a.f.prevCamera.position[1], a.f.prevCamera.position[2]);
        ShadingPoint sp;
        make_shading_point(a, bufIdx, p, cam.pos, true, sp);
        const gfx_material& mat = a.scene.materials[g3.w];
        contribution = f3(0.0f);
        if (sp.vOutLocal.z > 0) {
            float tu, tv;
            decode_uv(g3.z, tu, tv);
            const f3 e = material_emittance(a.scene, mat, tu, tv);
            contribution = contribution + e / kPi;
        }
        uint32_t* visBuf = static_cast<uint32_t*>(a.s.sampleVisibilityBuffer[bufIdx]);
        uint32_t sv = visBuf[p];
        float selectedTarget = 0.0f;
        Reservoir combined;
        combined.reset();
        uint32_t combinedStreamLength = 0;
        f3 directCont(0.0f);
        float selectedMisWeight = 0.0f;
        const Reservoir selfRes = load_reservoir(a.s.reservoirBuffer[a.curRes], numPixels, p);
        float2* curInfo = static_cast<float2*>(a.s.reservoirInfoBuffer[a.curRes]);
        const float2* prevInfo = static_cast<const float2*>(a.s.reservoirInfoBuffer[prevRes]);
        const float2 selfInfo = curInfo[p];
        const uint32_t selfStreamLength = selfRes.streamLength;
        const uint32_t maxPrevStreamLength = 20 * selfStreamLength;

        if (selfInfo.x > 0.0f && (sv & SV_newSample)) {
            const f3 cont = direct_lighting(sp.pos, sp.vOutLocal, sp.frame, sp.bsdf, selfRes.sample);
            const float target = target_weight(cont);
            float misWeight;
            if (TEMPORAL || SPATIAL)
                misWeight = rearch_mis_weight<kSampleNew, TEMPORAL, SPATIAL>(a, numPixels, prevBuf, prevRes, maxPrevStreamLength, sv,
                                                                             selfStreamLength, sp, tnp, snp, prevCamPos,
                                                                             selfStreamLength, selfRes.sample, selfInfo.y);
            else
                misWeight = 1.0f / selfStreamLength;
            directCont = directCont + (misWeight * selfInfo.x * selfStreamLength) * cont;
            combined = selfRes;
            selectedTarget = target;
            selectedMisWeight = misWeight;
            sv |= SV_selectedSample;
        }
        combinedStreamLength = selfStreamLength;

        if (TEMPORAL && (sv & SV_temporalPassedHeuristic)) {
            const Reservoir nb = load_reservoir(a.s.reservoirBuffer[prevRes], numPixels, tnp);
            const float2 nbInfo = prevInfo[tnp];
            const uint32_t nbLen = nb.streamLength < maxPrevStreamLength ? nb.streamLength : maxPrevStreamLength;
            if (nbInfo.x > 0.0f) {
                const f3 cont = direct_lighting(sp.pos, sp.vOutLocal, sp.frame, sp.bsdf, nb.sample);
                const float target = target_weight(cont);
                const float misWeight = rearch_mis_weight<kSampleTemporal, TEMPORAL, SPATIAL>(a, numPixels, prevBuf, prevRes, maxPrevStreamLength, sv,
                                                                                             selfStreamLength, sp, tnp, snp, prevCamPos,
                                                                                             nbLen, nb.sample, nbInfo.y);
                const float weight = target * nbInfo.x * nbLen;
                const uint32_t vis = (sv & SV_temporalSample) ? 1u : 0u;
                directCont = directCont + (vis * misWeight * nbInfo.x * nbLen) * cont;
                if (combined.update(nb.sample, weight, rng.uniform())) {
                    selectedTarget = target;
                    selectedMisWeight = misWeight;
                    sv = vis ? (sv | SV_selectedSample) : (sv & ~SV_selectedSample);
                }
            }
            combinedStreamLength += nbLen;
        }
        if (SPATIAL && (sv & SV_spatiotemporalPassedHeuristic)) {
            const Reservoir nb = load_reservoir(a.s.reservoirBuffer[prevRes], numPixels, snp);
            const float2 nbInfo = prevInfo[snp];
            const uint32_t nbLen = nb.streamLength < maxPrevStreamLength ? nb.streamLength : maxPrevStreamLength;
            if (nbInfo.x > 0.0f) {
                const f3 cont = direct_lighting(sp.pos, sp.vOutLocal, sp.frame, sp.bsdf, nb.sample);
                const float target = target_weight(cont);
                const float misWeight = rearch_mis_weight<kSampleSpatiotemporal, TEMPORAL, SPATIAL>(a, numPixels, prevBuf, prevRes, maxPrevStreamLength, sv,
                                                                                                   selfStreamLength, sp, tnp, snp, prevCamPos,
                                                                                                   nbLen, nb.sample, nbInfo.y);
                const float weight = target * nbInfo.x * nbLen;
                const uint32_t vis = (sv & SV_spatiotemporalSample) ? 1u : 0u;
                directCont = directCont + (vis * misWeight * nbInfo.x * nbLen) * cont;
                if (combined.update(nb.sample, weight, rng.uniform())) {
                    selectedTarget = target;
                    selectedMisWeight = misWeight;
                    sv = vis ? (sv | SV_selectedSample) : (sv & ~SV_selectedSample);
                }
            }
            combinedStreamLength += nbLen;
        }

        combined.streamLength = combinedStreamLength;
        contribution = contribution + directCont;
        float recPDF = selectedMisWeight * combined.sumWeights / selectedTarget;
        if (!is_finite(recPDF) || (a.f.reuseVisibility && !(sv & SV_selectedSample))) { recPDF = 0.0f; selectedTarget = 0.0f; }
        visBuf[p] = sv;
        store_reservoir(a.s.reservoirBuffer[a.curRes], numPixels, p, combined);
        curInfo[p] = make_float2(recPDF, selectedTarget);
        rngBuf[p] = rng.state;
    }
    else {
        const EnvMap env = load_env(a.s);
        if (env.present() && a.f.enableEnvLight) {
            const float u = (g3.z & 0xFFFF) / 65535.0f, v = (g3.z >> 16) / 65535.0f;
            contribution = a.f.envLightPowerCoeff * env.fetch(u, v);
        }
    }
    float4* beauty = static_cast<float4*>(a.s.beautyAccumBuffer) + p;
    f3 prev(0.0f);
    if (a.f.numAccumFrames > 0) { const float4 b = *beauty; prev = f3(b.x, b.y, b.z); }
    const float curWeight = 1.0f / (1 + a.f.numAccumFrames);
    const f3 result = (1 - curWeight) * prev + curWeight * contribution;
    *beauty = make_float4(result.x, result.y, result.z, 1.0f);
}

} // namespace gfx
