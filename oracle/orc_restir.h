// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// orc_restir.h: CPU restatement of the original-ReSTIR passes, one pixel at a time, over the
// same HBM layouts the product uses (include/gfxexp.h) so buffers compare byte for byte.
//   setupGBuffers RG/CH/MS        restir_di/gpu_kernels/optix_gbuffer_kernels.cu:5-243
//   performInitialAndTemporalRIS  restir_di/gpu_kernels/optix_restir_di_kernels.cu:14-287
//   performSpatialRIS             restir_di/gpu_kernels/optix_restir_di_kernels.cu:303-547
//   shading                       restir_di/gpu_kernels/optix_restir_di_kernels.cu:559-637
//   testNeighbor                  restir_di/restir_di_shared.h:747-771
#pragma once
#include "../include/gfxexp.h"
#include "orc_bvh.h"
#include "orc_scene.h"

namespace orc {

struct WorldAccel {
    // one bvh::Geometry per (instance, geomInst) in (instSlot asc, list order) enumeration
    std::vector<bvh::Geometry> geoms;
    std::vector<uint32_t> geomToInst, geomToGeomInst, primOffsets;
    bvh::GeometryBVH bvh;
    bool useBruteForce = false;
};

struct PixelHit {
    bool hit = false;
    uint32_t instSlot = 0xFFFFFFFFu, geomInstSlot = 0xFFFFFFFFu, primIndex = 0xFFFFFFFFu;
    float dist = 0, bcB = 0, bcC = 0;
};

// Canonical closest hit: minimum dist, ties broken by the lowest flattened triangle index
// (the product's documented rule; the reference's own result on exact ties is traversal-order
// dependent, common/bvh_builder.cpp:1486-1497).
static inline bvh::HitObject closestHitCanonical(const WorldAccel& a, V3 org, V3 dir, float tmin, float tmax) {
    if (a.useBruteForce) return bvh::bruteForce(a.bvh.triStorages, org, dir, tmin, tmax, false);
    bvh::HitObject h = bvh::traverse(a.bvh, org, dir, tmin, tmax, nullptr, false, true);
    if (!h.isHit()) return h;
    // re-run over the closed interval end to collect exact ties: any triangle with the same dist
    // and a lower flattened index replaces the winner.
    const float upper = std::nextafter(h.dist, INFINITY);
    bvh::HitObject best = h;
    uint32_t bestFlat = a.primOffsets[h.geomIndex] + h.primIndex;
    // narrow traversal restricted to (prev(dist), next(dist)): only tied triangles can be accepted.
    const float lower = std::nextafter(h.dist, -INFINITY);
    struct Rec { static void run(const WorldAccel& a, V3 org, V3 dir, float lo, float hi, float d,
                                 bvh::HitObject* best, uint32_t* bestFlat) {
        // explicit stack walk testing every triangle whose box overlaps [lo, hi]
        std::vector<uint32_t> st; st.push_back(0);
        while (!st.empty()) {
            const bvh::InternalNode& n = a.bvh.intNodes[st.back()]; st.pop_back();
            for (uint32_t slot = 0; slot < bvh::arity; ++slot) {
                if (!n.getChildIsValid(slot)) break;
                float t0, t1;
                if (!n.getChildAabb(slot).intersect(org, dir, lo, hi, &t0, &t1, true)) continue;
                if (!n.getChildIsLeaf(slot)) { st.push_back(n.intNodeChildBaseIndex + n.getInternalChildNumber(slot)); continue; }
                uint32_t idx = n.leafBaseIndex + n.childMetas[slot];
                while (true) {
                    const bvh::PrimitiveReference pr = a.bvh.primRefs[idx];
                    const bvh::TriangleStorage& ts = a.bvh.triStorages[pr.storageIndex];
                    float t, b, c;
                    if (bvh::testRayVsTriangle(org, dir, lo, hi, ts.pA, ts.pB, ts.pC, &t, &b, &c) && t == d) {
                        const uint32_t flat = a.primOffsets[ts.geomIndex] + ts.primIndex;
                        if (flat < *bestFlat) {
                            *bestFlat = flat;
                            best->geomIndex = ts.geomIndex; best->primIndex = ts.primIndex;
                            best->bcA = 1.0f - (b + c); best->bcB = b; best->bcC = c;
                        }
                    }
                    if (pr.isLeafEnd) break;
                    ++idx;
                }
            }
        }
    } };
    Rec::run(a, org, dir, lower, upper, h.dist, &best, &bestFlat);
    return best;
}

static inline bool occluded(const WorldAccel& a, V3 org, V3 dir, float tmin, float tmax) {
    if (a.useBruteForce) return bvh::bruteForce(a.bvh.triStorages, org, dir, tmin, tmax, true).isHit();
    return bvh::traverse(a.bvh, org, dir, tmin, tmax, nullptr, true, true).isHit();
}

// ---------------------------------------------------------------- pixel-buffer accessors
struct Params {
    const Scene* scene;
    const WorldAccel* accel;
    const gfx_restir_static_params* s;
    const gfx_restir_frame_params* f;
    uint32_t currentReservoirIndex;
    uint32_t spatialNeighborBaseIndex;
    PerspectiveCamera camera, prevCamera;
    bool envEnabled() const { return s->envLightTexture != nullptr && f->enableEnvLight; }
};

static inline PerspectiveCamera toCamera(const gfx_camera& c) {
    PerspectiveCamera p;
    p.aspect = c.aspect; p.fovY = c.fovY;
    p.position = V3(c.position[0], c.position[1], c.position[2]);
    p.orientation.r0 = V3(c.orientation[0], c.orientation[1], c.orientation[2]);
    p.orientation.r1 = V3(c.orientation[3], c.orientation[4], c.orientation[5]);
    p.orientation.r2 = V3(c.orientation[6], c.orientation[7], c.orientation[8]);
    return p;
}

static inline size_t pix(const Params& p, int x, int y) { return static_cast<size_t>(y) * p.s->imageSizeX + x; }

static inline Reservoir readReservoir(const Params& p, uint32_t bufIdx, size_t i) {
    const size_t n = static_cast<size_t>(p.s->imageSizeX) * p.s->imageSizeY;
    const float* base = static_cast<const float*>(p.s->reservoirBuffer[bufIdx]);
    const float* p0 = base + 4 * i;
    const float* p1 = base + 4 * (n + i);
    const float* p2 = base + 4 * (2 * n + i);
    Reservoir r;
    r.sample.emittance = RGB(p0[0], p0[1], p0[2]);
    r.sample.position = V3(p0[3], p1[0], p1[1]);
    r.sample.normal = V3(p1[2], p1[3], p2[0]);
    r.sample.atInfinity = f2bits(p2[1]) & 1u;
    r.sumWeights = p2[2];
    r.streamLength = f2bits(p2[3]);
    return r;
}
static inline void writeReservoir(const Params& p, uint32_t bufIdx, size_t i, const Reservoir& r) {
    const size_t n = static_cast<size_t>(p.s->imageSizeX) * p.s->imageSizeY;
    float* base = static_cast<float*>(p.s->reservoirBuffer[bufIdx]);
    float* p0 = base + 4 * i;
    float* p1 = base + 4 * (n + i);
    float* p2 = base + 4 * (2 * n + i);
    p0[0] = r.sample.emittance.x; p0[1] = r.sample.emittance.y; p0[2] = r.sample.emittance.z;
    p0[3] = r.sample.position.x; p1[0] = r.sample.position.y; p1[1] = r.sample.position.z;
    p1[2] = r.sample.normal.x; p1[3] = r.sample.normal.y; p2[0] = r.sample.normal.z;
    p2[1] = bits2f(r.sample.atInfinity & 1u);
    p2[2] = r.sumWeights;
    p2[3] = bits2f(r.streamLength);
}

// restir_di/restir_di_shared.h:747-771
static inline bool testNeighbor(const Params& p, bool testGeometry, uint32_t nbBufIdx, int nx, int ny,
                                float dist, V3 normalInWorld) {
    if (nx < 0 || nx >= p.s->imageSizeX || ny < 0 || ny >= p.s->imageSizeY) return false;
    const size_t ni = pix(p, nx, ny);
    const gfx_gbuffer0& g0 = static_cast<const gfx_gbuffer0*>(p.s->gbuffer0[nbBufIdx])[ni];
    if (g0.instSlot == 0xFFFFFFFFu) return false;
    if (testGeometry) {
        const gfx_gbuffer2& g2 = static_cast<const gfx_gbuffer2*>(p.s->gbuffer2[nbBufIdx])[ni];
        const gfx_gbuffer3& g3 = static_cast<const gfx_gbuffer3*>(p.s->gbuffer3[nbBufIdx])[ni];
        const V3 nbPos(g2.positionInWorld[0], g2.positionInWorld[1], g2.positionInWorld[2]);
        const V3 nbNormal = decodeNormal(g3.qShadingNormal);
        const float nbDist = length(p.camera.position - nbPos);
        if (std::fabs(nbDist - dist) / dist > 0.1f || dot(normalInWorld, nbNormal) < 0.9f) return false;
    }
    return true;
}

// ---------------------------------------------------------------- setupGBuffers
static inline void setupGBuffersPixel(const Params& p, int x, int y) {
    const Scene& scene = *p.scene;
    const uint32_t bufIdx = p.f->bufferIndex;
    const size_t i = pix(p, x, y);
    const PerspectiveCamera& camera = p.camera;
    float jx = 0.5f, jy = 0.5f;
    uint64_t* rngBuf = static_cast<uint64_t*>(p.s->rngBuffer);
    if (p.f->enableJittering) {
        PCG32RNG rng; rng.setState(rngBuf[i]);
        jx = rng.getFloat0cTo1o();
        jy = rng.getFloat0cTo1o();
        rngBuf[i] = rng.state;
    }
    const float fx = (x + jx) / p.s->imageSizeX;
    const float fy = (y + jy) / p.s->imageSizeY;
    const float vh = 2 * gm_tan(camera.fovY * 0.5f);
    const float vw = camera.aspect * vh;
    const V3 origin = camera.position;
    const V3 direction = normalize(mul(camera.orientation, V3(vw * (0.5f - fx), vh * (0.5f - fy), 1)));

    RGB albedo(0.0f);
    V3 positionInWorld(NAN), prevPositionInWorld(NAN), shadingNormalInWorld(NAN);
    uint32_t qGeometricNormalInWorld = 0, qTexCoord0DirInWorld = 0, qTexCoord = 0;
    uint32_t matSlot = 0xFFFFFFFFu, instSlot = 0xFFFFFFFFu, geomInstSlot = 0xFFFFFFFFu, primIndex = 0xFFFFFFFFu;
    uint16_t qbcB = 0, qbcC = 0;

    const bvh::HitObject h = closestHitCanonical(*p.accel, origin, direction, 0.0f, 3.402823466e+38f);
    if (h.isHit()) { // RT_CH_NAME(setupGBuffers) :112-199
        instSlot = p.accel->geomToInst[h.geomIndex];
        geomInstSlot = p.accel->geomToGeomInst[h.geomIndex];
        primIndex = h.primIndex;
        const InstanceData& inst = scene.insts[instSlot];
        const GeometryInstanceData& geomInst = scene.geomInsts[geomInstSlot];
        const MaterialData& mat = scene.materials[geomInst.materialSlot];
        matSlot = geomInst.materialSlot;
        const Triangle& tri = geomInst.triangleBuffer[primIndex];
        const Vertex& vA = geomInst.vertexBuffer[tri.index0];
        const Vertex& vB = geomInst.vertexBuffer[tri.index1];
        const Vertex& vC = geomInst.vertexBuffer[tri.index2];
        const float bcB = h.bcB, bcC = h.bcC;
        const float bcA = 1 - (bcB + bcC);
        qbcB = encodeBarycentric(bcB);
        qbcC = encodeBarycentric(bcC);
        const V3 positionInObj = bcA * vA.position + bcB * vB.position + bcC * vC.position;
        const V3 shadingNormalInObj = bcA * vA.normal + bcB * vB.normal + bcC * vC.normal;
        const V3 texCoord0DirInObj = bcA * vA.texCoord0Dir + bcB * vB.texCoord0Dir + bcC * vC.texCoord0Dir;
        const V2 texCoord{ bcA * vA.texCoord.x + bcB * vB.texCoord.x + bcC * vC.texCoord.x,
                           bcA * vA.texCoord.y + bcB * vB.texCoord.y + bcC * vC.texCoord.y };
        const V3 geometricNormalInObj = cross(vB.position - vA.position, vC.position - vA.position);
        positionInWorld = xfmPoint(inst.transform, positionInObj);
        prevPositionInWorld = xfmPoint(inst.curToPrevTransform, positionInWorld);
        V3 geometricNormalInWorld = normalize(mul(inst.normalMatrix, geometricNormalInObj));
        shadingNormalInWorld = normalize(mul(inst.normalMatrix, shadingNormalInObj));
        V3 texCoord0DirInWorld = xfmVector(inst.transform, texCoord0DirInObj);
        texCoord0DirInWorld = normalize(
            texCoord0DirInWorld - dot(shadingNormalInWorld, texCoord0DirInWorld) * shadingNormalInWorld);
        if (!allFinite(shadingNormalInWorld)) {
            geometricNormalInWorld = V3(0, 0, 1);
            shadingNormalInWorld = V3(0, 0, 1);
            texCoord0DirInWorld = V3(1, 0, 0);
        }
        qGeometricNormalInWorld = encodeNormal(geometricNormalInWorld);
        qTexCoord = encodeTexCoords(texCoord);
        BSDF bsdf; bsdf.setup(scene.textures, mat, texCoord);
        ReferenceFrame shadingFrame(shadingNormalInWorld, texCoord0DirInWorld);
        if (p.f->enableBumpMapping) {   // :170-173
            const V3 modLocalNormal = readModifiedNormal(scene.textures, mat, texCoord);
            applyBumpMapping(modLocalNormal, &shadingFrame);
        }
        const V3 vOut = -direction;
        const V3 vOutLocal = shadingFrame.toLocal(normalize(vOut));
        shadingNormalInWorld = shadingFrame.normal;
        qTexCoord0DirInWorld = encodeVector(shadingFrame.tangent);
        albedo = bsdf.evaluateDHReflectanceEstimate(vOutLocal);
    }
    else { // RT_MS_NAME(setupGBuffers) :201-243
        const V3 vOut = -direction;
        const V3 pp = -vOut;
        float posPhi, posTheta;
        toPolarYUp(pp, &posPhi, &posTheta);
        const float phi = posPhi + p.f->envLightRotation;
        float u = phi / (2 * kPi);
        u -= std::floor(u);
        const float v = posTheta / kPi;
        positionInWorld = pp;
        prevPositionInWorld = pp;
        qGeometricNormalInWorld = encodeNormal(vOut);
        shadingNormalInWorld = vOut;
        qTexCoord0DirInWorld = encodeVector(V3(-gm_cos(posPhi), 0, -gm_sin(posPhi)));
        qTexCoord = encodeTexCoords(V2{ u, v });
        qbcB = encodeBarycentric(u);
        qbcC = encodeBarycentric(v);
    }

    const V2 curRasterPos{ x + 0.5f, y + 0.5f };
    const V2 sp = p.prevCamera.calcScreenPosition(prevPositionInWorld);
    const V2 prevRasterPos{ sp.x * p.s->imageSizeX, sp.y * p.s->imageSizeY };
    V2 motionVector{ curRasterPos.x - prevRasterPos.x, curRasterPos.y - prevRasterPos.y };
    if (p.f->resetFlowBuffer || std::isnan(prevPositionInWorld.x)) motionVector = V2{ 0.0f, 0.0f };

    gfx_gbuffer0 g0; g0.instSlot = instSlot; g0.geomInstSlot = geomInstSlot; g0.primIndex = primIndex; g0.qbcB = qbcB; g0.qbcC = qbcC;
    gfx_gbuffer1 g1; g1.motionVector[0] = motionVector.x; g1.motionVector[1] = motionVector.y;
    gfx_gbuffer2 g2; g2.positionInWorld[0] = positionInWorld.x; g2.positionInWorld[1] = positionInWorld.y; g2.positionInWorld[2] = positionInWorld.z;
    g2.qGeometricNormal = qGeometricNormalInWorld;
    gfx_gbuffer3 g3; g3.qShadingNormal = encodeNormal(shadingNormalInWorld); g3.qShadingTangent = qTexCoord0DirInWorld;
    g3.qTexCoord = qTexCoord; g3.matSlot = matSlot;
    static_cast<gfx_gbuffer0*>(p.s->gbuffer0[bufIdx])[i] = g0;
    static_cast<gfx_gbuffer1*>(p.s->gbuffer1[bufIdx])[i] = g1;
    static_cast<gfx_gbuffer2*>(p.s->gbuffer2[bufIdx])[i] = g2;
    static_cast<gfx_gbuffer3*>(p.s->gbuffer3[bufIdx])[i] = g3;

    float* albedoAcc = static_cast<float*>(p.s->albedoAccumBuffer) + 4 * i;
    float* normalAcc = static_cast<float*>(p.s->normalAccumBuffer) + 4 * i;
    RGB prevAlbedo(0.0f); V3 prevNormal(0.0f);
    if (p.f->numAccumFrames > 0) {
        prevAlbedo = RGB(albedoAcc[0], albedoAcc[1], albedoAcc[2]);
        prevNormal = V3(normalAcc[0], normalAcc[1], normalAcc[2]);
    }
    const float curWeight = 1.0f / (1 + p.f->numAccumFrames);
    const RGB albedoResult = (1 - curWeight) * prevAlbedo + curWeight * albedo;
    const V3 normalResult = (1 - curWeight) * prevNormal + curWeight * shadingNormalInWorld;
    albedoAcc[0] = albedoResult.x; albedoAcc[1] = albedoResult.y; albedoAcc[2] = albedoResult.z; albedoAcc[3] = 1.0f;
    normalAcc[0] = normalResult.x; normalAcc[1] = normalResult.y; normalAcc[2] = normalResult.z; normalAcc[3] = 1.0f;
}

// Shared prologue of the per-pixel passes: re-derive the shading point from the quantised G-buffer.
struct ShadingPoint {
    V3 positionInWorld;       // offset ray origin
    V3 vOut; float dist;
    ReferenceFrame shadingFrame;
    V3 vOutLocal;
    BSDF bsdf;
};

// ---------------------------------------------------------------- performInitialAndTemporalRIS
static inline void initialAndTemporalRISPixel(const Params& p, bool withTemporalRIS, bool useUnbiasedEstimator, int x, int y) {
    constexpr bool useMIS_RIS = true; // optix_restir_di_kernels.cu:10
    const Scene& scene = *p.scene;
    const uint32_t curBufIdx = p.f->bufferIndex;
    const size_t i = pix(p, x, y);
    const gfx_gbuffer0& gb0 = static_cast<const gfx_gbuffer0*>(p.s->gbuffer0[curBufIdx])[i];
    if (gb0.instSlot == 0xFFFFFFFFu) return;
    const PerspectiveCamera& camera = p.camera;
    const gfx_gbuffer2& gb2 = static_cast<const gfx_gbuffer2*>(p.s->gbuffer2[curBufIdx])[i];
    const gfx_gbuffer3& gb3 = static_cast<const gfx_gbuffer3*>(p.s->gbuffer3[curBufIdx])[i];
    V3 positionInWorld(gb2.positionInWorld[0], gb2.positionInWorld[1], gb2.positionInWorld[2]);
    const V3 geometricNormalInWorld = decodeNormal(gb2.qGeometricNormal);
    const MaterialData& mat = scene.materials[gb3.matSlot];
    uint64_t* rngBuf = static_cast<uint64_t*>(p.s->rngBuffer);
    PCG32RNG rng; rng.setState(rngBuf[i]);

    V3 vOut = camera.position - positionInWorld;
    const float frontHit = dot(vOut, geometricNormalInWorld) >= 0.0f ? 1.0f : -1.0f;
    positionInWorld = offsetRayOrigin(positionInWorld, frontHit * geometricNormalInWorld);
    const float dist = length(vOut);
    vOut /= dist;
    const V3 shadingNormalInWorld = decodeNormal(gb3.qShadingNormal);
    const V3 shadingTangentInWorld = decodeVector(gb3.qShadingTangent);
    const ReferenceFrame shadingFrame(shadingNormalInWorld, shadingTangentInWorld);
    const V3 vOutLocal = shadingFrame.toLocal(vOut);
    BSDF bsdf; bsdf.setup(scene.textures, mat, decodeTexCoords(gb3.qTexCoord));

    const VisibilityFn visFn = [&p](V3 o, V3 d, float t0, float t1) { return !occluded(*p.accel, o, d, t0, t1); };
    const uint32_t curResIndex = p.currentReservoirIndex;
    Reservoir reservoir;
    reservoir.initialize(LightSample());

    float selectedTargetDensity = 0.0f;
    const uint32_t numCandidates = 1u << p.f->log2NumCandidateSamples;
    for (uint32_t ci = 0; ci < numCandidates; ++ci) {
        float ul = rng.getFloat0cTo1o();
        float probToSampleCurLightType = 1.0f;
        bool sampleEnvLight = false;
        if (p.envEnabled()) {
            if (scene.lightInstDist.integral() > 0.0f) {
                constexpr float probToSampleEnvLight = 0.25f; // restir_di_shared.h:6
                const float prob = fmin2(fmax2(probToSampleEnvLight * numCandidates - ci, 0.0f), 1.0f);
                if (ul < prob) { probToSampleCurLightType = probToSampleEnvLight; ul = ul / prob; sampleEnvLight = true; }
                else { probToSampleCurLightType = 1.0f - probToSampleEnvLight; ul = (ul - prob) / (1 - prob); }
            }
            else sampleEnvLight = true;
        }
        LightSample lightSample;
        float probDensity;
        const float u0 = rng.getFloat0cTo1o(); // left-to-right argument evaluation (contract)
        const float u1 = rng.getFloat0cTo1o();
        sampleLight(scene, p.f->envLightRotation, p.f->envLightPowerCoeff, positionInWorld,
                    ul, sampleEnvLight, u0, u1, &lightSample, &probDensity);
        const RGB cont = performDirectLighting(false, visFn, positionInWorld, vOutLocal, shadingFrame, bsdf, lightSample);
        probDensity *= probToSampleCurLightType;
        const float targetDensity = convertToWeight(cont);
        const float weight = targetDensity / probDensity;
        if (reservoir.update(lightSample, weight, rng.getFloat0cTo1o()))
            selectedTargetDensity = targetDensity;
    }

    float recPDFEstimate = reservoir.sumWeights / (selectedTargetDensity * reservoir.streamLength);
    if (!finitef(recPDFEstimate)) { recPDFEstimate = 0.0f; selectedTargetDensity = 0.0f; }

    if (p.f->reuseVisibility && selectedTargetDensity > 0.0f) {
        if (!evaluateVisibility(visFn, positionInWorld, reservoir.sample)) {
            recPDFEstimate = 0.0f; selectedTargetDensity = 0.0f;
        }
    }

    if (withTemporalRIS) {
        const uint32_t prevBufIdx = (curBufIdx + 1) % 2;
        const uint32_t prevResIndex = (curResIndex + 1) % 2;
        bool neighborIsSelected = false;
        const uint32_t selfStreamLength = reservoir.streamLength;
        if (recPDFEstimate == 0.0f) reservoir.initialize(LightSample());
        uint32_t combinedStreamLength = selfStreamLength;
        const uint32_t maxPrevStreamLength = 20 * selfStreamLength;
        const gfx_gbuffer1& gb1 = static_cast<const gfx_gbuffer1*>(p.s->gbuffer1[curBufIdx])[i];
        const int nbx = f2i(x + 0.5f - gb1.motionVector[0]);
        const int nby = f2i(y + 0.5f - gb1.motionVector[1]);
        const bool acceptedNeighbor = testNeighbor(p, !useUnbiasedEstimator, prevBufIdx, nbx, nby, dist, shadingNormalInWorld);
        size_t ni = 0;
        if (acceptedNeighbor) {
            ni = pix(p, nbx, nby);
            const Reservoir neighbor = readReservoir(p, prevResIndex, ni);
            const gfx_reservoir_info neighborInfo = static_cast<const gfx_reservoir_info*>(p.s->reservoirInfoBuffer[prevResIndex])[ni];
            const LightSample nbLightSample = neighbor.sample;
            const RGB cont = performDirectLighting(false, visFn, positionInWorld, vOutLocal, shadingFrame, bsdf, nbLightSample);
            const float targetDensity = convertToWeight(cont);
            const uint32_t nbStreamLength = std::min(neighbor.streamLength, maxPrevStreamLength);
            const float weight = targetDensity * neighborInfo.recPDFEstimate * nbStreamLength;
            if (reservoir.update(nbLightSample, weight, rng.getFloat0cTo1o())) {
                selectedTargetDensity = targetDensity;
                if (useUnbiasedEstimator) neighborIsSelected = true;
            }
            combinedStreamLength += nbStreamLength;
        }
        reservoir.streamLength = combinedStreamLength;

        float weightForEstimate;
        if (useUnbiasedEstimator) {
            const LightSample selectedLightSample = reservoir.sample;
            float numWeight, denomWeight;
            {
                const RGB cont = performDirectLighting(false, visFn, positionInWorld, vOutLocal, shadingFrame, bsdf, selectedLightSample);
                const float targetDensityForSelf = convertToWeight(cont);
                if (useMIS_RIS) { numWeight = targetDensityForSelf; denomWeight = targetDensityForSelf * selfStreamLength; }
                else { numWeight = 1.0f; denomWeight = 0.0f; if (targetDensityForSelf > 0.0f) denomWeight = selfStreamLength; }
            }
            if (acceptedNeighbor) {
                const gfx_gbuffer2& nb2 = static_cast<const gfx_gbuffer2*>(p.s->gbuffer2[prevBufIdx])[ni];
                const gfx_gbuffer3& nb3 = static_cast<const gfx_gbuffer3*>(p.s->gbuffer3[prevBufIdx])[ni];
                V3 nbPositionInWorld(nb2.positionInWorld[0], nb2.positionInWorld[1], nb2.positionInWorld[2]);
                const V3 nbGeometricNormalInWorld = decodeNormal(nb2.qGeometricNormal);
                const V3 nbVOut = normalize(p.prevCamera.position - nbPositionInWorld);
                const float nbFrontHit = dot(nbVOut, nbGeometricNormalInWorld) >= 0.0f ? 1.0f : -1.0f;
                nbPositionInWorld = offsetRayOrigin(nbPositionInWorld, nbFrontHit * nbGeometricNormalInWorld);
                const MaterialData& nbMat = scene.materials[nb3.matSlot];
                BSDF nbBsdf; nbBsdf.setup(scene.textures, nbMat, decodeTexCoords(nb3.qTexCoord));
                const V3 nbShadingNormalInWorld = decodeNormal(nb3.qShadingNormal);
                const V3 nbShadingTangentInWorld = decodeVector(nb3.qShadingTangent);
                const ReferenceFrame nbShadingFrame(nbShadingNormalInWorld, nbShadingTangentInWorld);
                const V3 nbVOutLocal = nbShadingFrame.toLocal(nbVOut);
                const Reservoir neighbor = readReservoir(p, prevResIndex, ni);
                const RGB cont = performDirectLighting(false, visFn, nbPositionInWorld, nbVOutLocal, nbShadingFrame, nbBsdf, selectedLightSample);
                const float nbTargetDensity = convertToWeight(cont);
                const uint32_t nbStreamLength = std::min(neighbor.streamLength, maxPrevStreamLength);
                if (useMIS_RIS) { denomWeight += nbTargetDensity * nbStreamLength; if (neighborIsSelected) numWeight = nbTargetDensity; }
                else { if (nbTargetDensity > 0.0f) denomWeight += nbStreamLength; }
            }
            weightForEstimate = numWeight / denomWeight;
        }
        else weightForEstimate = 1.0f / reservoir.streamLength;

        recPDFEstimate = weightForEstimate * reservoir.sumWeights / selectedTargetDensity;
        if (!finitef(recPDFEstimate)) { recPDFEstimate = 0.0f; selectedTargetDensity = 0.0f; }
    }

    rngBuf[i] = rng.state;
    writeReservoir(p, curResIndex, i, reservoir);
    gfx_reservoir_info info; info.recPDFEstimate = recPDFEstimate; info.targetDensity = selectedTargetDensity;
    static_cast<gfx_reservoir_info*>(p.s->reservoirInfoBuffer[curResIndex])[i] = info;
}

// ---------------------------------------------------------------- performSpatialRIS
static inline void spatialNeighborCoord(const Params& p, PCG32RNG& rng, uint32_t nIdx, int x, int y, int* nbx, int* nby) {
    float radius = p.f->spatialNeighborRadius;
    float deltaX, deltaY;
    if (p.f->useLowDiscrepancyNeighbors) {
        const float* d = static_cast<const float*>(p.s->spatialNeighborDeltas) + 2 * ((p.spatialNeighborBaseIndex + nIdx) % 1024);
        deltaX = radius * d[0];
        deltaY = radius * d[1];
    }
    else {
        radius *= std::sqrt(rng.getFloat0cTo1o());
        const float angle = 2 * kPi * rng.getFloat0cTo1o();
        float s, c; gm_sincos(angle, &s, &c);
        deltaX = radius * c;
        deltaY = radius * s;
    }
    *nbx = f2i(x + 0.5f + deltaX);
    *nby = f2i(y + 0.5f + deltaY);
}

static inline void spatialRISPixel(const Params& p, bool useUnbiasedEstimator, int x, int y) {
    constexpr bool useMIS_RIS = true;
    const Scene& scene = *p.scene;
    const uint32_t bufIdx = p.f->bufferIndex;
    const size_t i = pix(p, x, y);
    const gfx_gbuffer0& gb0 = static_cast<const gfx_gbuffer0*>(p.s->gbuffer0[bufIdx])[i];
    if (gb0.instSlot == 0xFFFFFFFFu) return;
    const gfx_gbuffer2& gb2 = static_cast<const gfx_gbuffer2*>(p.s->gbuffer2[bufIdx])[i];
    const gfx_gbuffer3& gb3 = static_cast<const gfx_gbuffer3*>(p.s->gbuffer3[bufIdx])[i];
    V3 positionInWorld(gb2.positionInWorld[0], gb2.positionInWorld[1], gb2.positionInWorld[2]);
    const V3 geometricNormalInWorld = decodeNormal(gb2.qGeometricNormal);
    uint64_t* rngBuf = static_cast<uint64_t*>(p.s->rngBuffer);
    PCG32RNG rng; rng.setState(rngBuf[i]);
    V3 vOut = p.camera.position - positionInWorld;
    const float frontHit = dot(vOut, geometricNormalInWorld) >= 0.0f ? 1.0f : -1.0f;
    positionInWorld = offsetRayOrigin(positionInWorld, frontHit * geometricNormalInWorld);
    const float dist = length(vOut);
    vOut /= dist;
    const ReferenceFrame shadingFrame(decodeNormal(gb3.qShadingNormal), decodeVector(gb3.qShadingTangent));
    const V3 vOutLocal = shadingFrame.toLocal(vOut);
    const MaterialData& mat = scene.materials[gb3.matSlot];
    BSDF bsdf; bsdf.setup(scene.textures, mat, decodeTexCoords(gb3.qTexCoord));
    const VisibilityFn visFn = [&p](V3 o, V3 d, float t0, float t1) { return !occluded(*p.accel, o, d, t0, t1); };

    const uint32_t srcResIndex = p.currentReservoirIndex;
    const uint32_t dstResIndex = (srcResIndex + 1) % 2;
    Reservoir combinedReservoir;
    combinedReservoir.initialize(LightSample());
    float selectedTargetDensity = 0.0f;
    int32_t selectedNeighborIndex = -1;

    const Reservoir self = readReservoir(p, srcResIndex, i);
    const gfx_reservoir_info selfResInfo = static_cast<const gfx_reservoir_info*>(p.s->reservoirInfoBuffer[srcResIndex])[i];
    if (selfResInfo.recPDFEstimate > 0.0f) { combinedReservoir = self; selectedTargetDensity = selfResInfo.targetDensity; }
    uint32_t combinedStreamLength = self.streamLength;

    for (uint32_t nIdx = 0; nIdx < p.f->numSpatialNeighbors; ++nIdx) {
        int nbx, nby;
        spatialNeighborCoord(p, rng, nIdx, x, y, &nbx, &nby);
        const bool acceptedNeighbor = testNeighbor(p, !useUnbiasedEstimator, bufIdx, nbx, nby, dist, shadingFrame.normal)
            && (nbx != x || nby != y);
        if (acceptedNeighbor) {
            const size_t ni = pix(p, nbx, nby);
            const Reservoir neighbor = readReservoir(p, srcResIndex, ni);
            const gfx_reservoir_info neighborInfo = static_cast<const gfx_reservoir_info*>(p.s->reservoirInfoBuffer[srcResIndex])[ni];
            const LightSample nbLightSample = neighbor.sample;
            const RGB cont = performDirectLighting(false, visFn, positionInWorld, vOutLocal, shadingFrame, bsdf, nbLightSample);
            const float targetDensity = convertToWeight(cont);
            const uint32_t nbStreamLength = neighbor.streamLength;
            const float weight = targetDensity * neighborInfo.recPDFEstimate * nbStreamLength;
            if (combinedReservoir.update(nbLightSample, weight, rng.getFloat0cTo1o())) {
                selectedTargetDensity = targetDensity;
                if (useUnbiasedEstimator) selectedNeighborIndex = static_cast<int32_t>(nIdx);
            }
            combinedStreamLength += nbStreamLength;
        }
    }
    combinedReservoir.streamLength = combinedStreamLength;

    float weightForEstimate = 0.0f;
    if (useUnbiasedEstimator) {
        if (selectedTargetDensity > 0.0f) {
            const LightSample selectedLightSample = combinedReservoir.sample;
            float numWeight, denomWeight;
            bool visibility = true;
            {
                const RGB cont = performDirectLighting(p.f->reuseVisibility != 0, visFn, positionInWorld, vOutLocal, shadingFrame, bsdf, selectedLightSample);
                const float targetDensityForSelf = convertToWeight(cont);
                if (p.f->reuseVisibility) visibility = targetDensityForSelf > 0.0f;
                if (useMIS_RIS) { numWeight = targetDensityForSelf; denomWeight = targetDensityForSelf * self.streamLength; }
                else { numWeight = 1.0f; denomWeight = 0.0f; if (targetDensityForSelf > 0.0f) denomWeight = self.streamLength; }
            }
            for (uint32_t nIdx = 0; nIdx < p.f->numSpatialNeighbors; ++nIdx) {
                int nbx, nby;
                spatialNeighborCoord(p, rng, nIdx, x, y, &nbx, &nby);
                const bool acceptedNeighbor = (nbx >= 0 && nbx < p.s->imageSizeX && nby >= 0 && nby < p.s->imageSizeY)
                    && (nbx != x || nby != y);
                if (acceptedNeighbor) {
                    const size_t ni = pix(p, nbx, nby);
                    const gfx_gbuffer0& nb0 = static_cast<const gfx_gbuffer0*>(p.s->gbuffer0[bufIdx])[ni];
                    if (nb0.instSlot == 0xFFFFFFFFu) continue;
                    const gfx_gbuffer2& nb2 = static_cast<const gfx_gbuffer2*>(p.s->gbuffer2[bufIdx])[ni];
                    const gfx_gbuffer3& nb3 = static_cast<const gfx_gbuffer3*>(p.s->gbuffer3[bufIdx])[ni];
                    V3 nbPositionInWorld(nb2.positionInWorld[0], nb2.positionInWorld[1], nb2.positionInWorld[2]);
                    const V3 nbGeometricNormalInWorld = decodeNormal(nb2.qGeometricNormal);
                    const V3 nbVOut = normalize(p.prevCamera.position - nbPositionInWorld); // prevCamera: as in the reference (:487)
                    const float nbFrontHit = dot(nbVOut, nbGeometricNormalInWorld) >= 0.0f ? 1.0f : -1.0f;
                    nbPositionInWorld = offsetRayOrigin(nbPositionInWorld, nbFrontHit * nbGeometricNormalInWorld);
                    const MaterialData& nbMat = scene.materials[nb3.matSlot];
                    BSDF nbBsdf; nbBsdf.setup(scene.textures, nbMat, decodeTexCoords(nb3.qTexCoord));
                    const ReferenceFrame nbShadingFrame(decodeNormal(nb3.qShadingNormal), decodeVector(nb3.qShadingTangent));
                    const V3 nbVOutLocal = nbShadingFrame.toLocal(nbVOut);
                    const Reservoir neighbor = readReservoir(p, srcResIndex, ni);
                    const RGB cont = performDirectLighting(p.f->reuseVisibility != 0, visFn, nbPositionInWorld, nbVOutLocal, nbShadingFrame, nbBsdf, selectedLightSample);
                    const float nbTargetDensity = convertToWeight(cont);
                    const uint32_t nbStreamLength = neighbor.streamLength;
                    if (useMIS_RIS) {
                        denomWeight += nbTargetDensity * nbStreamLength;
                        if (static_cast<int32_t>(nIdx) == selectedNeighborIndex) numWeight = nbTargetDensity;
                    }
                    else { if (nbTargetDensity > 0.0f) denomWeight += nbStreamLength; }
                }
            }
            weightForEstimate = numWeight / denomWeight;
            if (p.f->reuseVisibility && !visibility) weightForEstimate = 0.0f;
        }
    }
    else weightForEstimate = 1.0f / combinedReservoir.streamLength;

    gfx_reservoir_info info;
    info.recPDFEstimate = weightForEstimate * combinedReservoir.sumWeights / selectedTargetDensity;
    info.targetDensity = selectedTargetDensity;
    if (!finitef(info.recPDFEstimate)) { info.recPDFEstimate = 0.0f; info.targetDensity = 0.0f; }

    rngBuf[i] = rng.state;
    writeReservoir(p, dstResIndex, i, combinedReservoir);
    static_cast<gfx_reservoir_info*>(p.s->reservoirInfoBuffer[dstResIndex])[i] = info;
}

// ---------------------------------------------------------------- shading
// The direct-lighting term of the shading pass for pixel (x, y): recPDFEstimate x performDirectLighting of the pixel's final reservoir
// sample at the shading point the G-buffer holds (the statements of shadingPixel, :574-606, with the visibility always tested).  Used by
// the NRC path tracer whose first-vertex next-event estimation is the ReSTIR DI reservoir (orc_nrc.h, GFX_PT_PATH_TRACE_NRC_RESTIR):
// a composition of this build -- the reference names it as open work (README.md:80-81) and has no code for it.
// `shadowRayOrigin` / `sample`: what the caller needs to restate the visibility test itself.
static inline RGB restirDirectEstimate(const Params& p, int x, int y) {
    const Scene& scene = *p.scene;
    const uint32_t bufIdx = p.f->bufferIndex;
    const size_t i = pix(p, x, y);
    const gfx_gbuffer0& gb0 = static_cast<const gfx_gbuffer0*>(p.s->gbuffer0[bufIdx])[i];
    if (gb0.instSlot == 0xFFFFFFFFu) return RGB(0.0f);
    const gfx_gbuffer3& gb3 = static_cast<const gfx_gbuffer3*>(p.s->gbuffer3[bufIdx])[i];
    const gfx_gbuffer2& gb2 = static_cast<const gfx_gbuffer2*>(p.s->gbuffer2[bufIdx])[i];
    V3 positionInWorld(gb2.positionInWorld[0], gb2.positionInWorld[1], gb2.positionInWorld[2]);
    const V3 geometricNormalInWorld = decodeNormal(gb2.qGeometricNormal);
    const V3 vOut = normalize(p.camera.position - positionInWorld);
    const float frontHit = dot(vOut, geometricNormalInWorld) >= 0.0f ? 1.0f : -1.0f;
    positionInWorld = offsetRayOrigin(positionInWorld, frontHit * geometricNormalInWorld);
    const ReferenceFrame shadingFrame(decodeNormal(gb3.qShadingNormal), decodeVector(gb3.qShadingTangent));
    const V3 vOutLocal = shadingFrame.toLocal(vOut);
    const MaterialData& mat = scene.materials[gb3.matSlot];
    BSDF bsdf; bsdf.setup(scene.textures, mat, decodeTexCoords(gb3.qTexCoord));
    const VisibilityFn visFn = [&p](V3 o, V3 d, float t0, float t1) { return !occluded(*p.accel, o, d, t0, t1); };
    const Reservoir reservoir = readReservoir(p, p.currentReservoirIndex, i);
    const float recPDFEstimate = static_cast<const gfx_reservoir_info*>(p.s->reservoirInfoBuffer[p.currentReservoirIndex])[i].recPDFEstimate;
    RGB directCont(0.0f);
    if (recPDFEstimate > 0 && finitef(recPDFEstimate))
        directCont = performDirectLighting(true, visFn, positionInWorld, vOutLocal, shadingFrame, bsdf, reservoir.sample);
    return recPDFEstimate * directCont;
}

static inline void shadingPixel(const Params& p, int x, int y) {
    const Scene& scene = *p.scene;
    const uint32_t bufIdx = p.f->bufferIndex;
    const size_t i = pix(p, x, y);
    const gfx_gbuffer0& gb0 = static_cast<const gfx_gbuffer0*>(p.s->gbuffer0[bufIdx])[i];
    const gfx_gbuffer3& gb3 = static_cast<const gfx_gbuffer3*>(p.s->gbuffer3[bufIdx])[i];
    const V2 texCoord = decodeTexCoords(gb3.qTexCoord);
    RGB contribution(0.01f, 0.01f, 0.01f);
    if (gb0.instSlot != 0xFFFFFFFFu) {
        const gfx_gbuffer2& gb2 = static_cast<const gfx_gbuffer2*>(p.s->gbuffer2[bufIdx])[i];
        V3 positionInWorld(gb2.positionInWorld[0], gb2.positionInWorld[1], gb2.positionInWorld[2]);
        const V3 geometricNormalInWorld = decodeNormal(gb2.qGeometricNormal);
        const V3 vOut = normalize(p.camera.position - positionInWorld);
        const float frontHit = dot(vOut, geometricNormalInWorld) >= 0.0f ? 1.0f : -1.0f;
        positionInWorld = offsetRayOrigin(positionInWorld, frontHit * geometricNormalInWorld);
        const ReferenceFrame shadingFrame(decodeNormal(gb3.qShadingNormal), decodeVector(gb3.qShadingTangent));
        const V3 vOutLocal = shadingFrame.toLocal(vOut);
        const MaterialData& mat = scene.materials[gb3.matSlot];
        BSDF bsdf; bsdf.setup(scene.textures, mat, decodeTexCoords(gb3.qTexCoord));
        const VisibilityFn visFn = [&p](V3 o, V3 d, float t0, float t1) { return !occluded(*p.accel, o, d, t0, t1); };
        const uint32_t curResIndex = p.currentReservoirIndex;
        const Reservoir reservoir = readReservoir(p, curResIndex, i);
        const gfx_reservoir_info reservoirInfo = static_cast<const gfx_reservoir_info*>(p.s->reservoirInfoBuffer[curResIndex])[i];
        contribution = RGB(0.0f);
        if (vOutLocal.z > 0) {
            const RGB emittance = materialEmittance(scene.textures, mat, decodeTexCoords(gb3.qTexCoord));   // :594-599
            contribution += emittance / kPi;
        }
        const LightSample lightSample = reservoir.sample;
        RGB directCont(0.0f);
        const float recPDFEstimate = reservoirInfo.recPDFEstimate;
        if (recPDFEstimate > 0 && finitef(recPDFEstimate)) {
            const bool visDone = p.f->reuseVisibility &&
                (!p.f->enableTemporalReuse || (p.f->enableSpatialReuse && p.f->useUnbiasedEstimator));
            directCont = performDirectLighting(!visDone, visFn, positionInWorld, vOutLocal, shadingFrame, bsdf, lightSample);
        }
        contribution += recPDFEstimate * directCont;
    }
    else {
        if (p.envEnabled()) {
            const RGB texValue = scene.env.fetch(texCoord.x, texCoord.y);
            contribution = p.f->envLightPowerCoeff * texValue;
        }
    }
    float* beauty = static_cast<float*>(p.s->beautyAccumBuffer) + 4 * i;
    RGB prevColorResult(0.0f, 0.0f, 0.0f);
    if (p.f->numAccumFrames > 0) prevColorResult = RGB(beauty[0], beauty[1], beauty[2]);
    const float curWeight = 1.0f / (1 + p.f->numAccumFrames);
    const RGB colorResult = (1 - curWeight) * prevColorResult + curWeight * contribution;
    beauty[0] = colorResult.x; beauty[1] = colorResult.y; beauty[2] = colorResult.z; beauty[3] = 1.0f;
}

} // namespace orc
