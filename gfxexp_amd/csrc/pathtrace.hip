// pathtrace.hip -- the baseline unidirectional path tracer as a wavefront pipeline.
//
// The reference runs one OptiX ray-generation megakernel per pixel that loops over path vertices,
// tracing a shadow ray (NEE) and an extension ray per vertex inline
// (path_tracing/gpu_kernels/optix_pathtracing_kernels.cu:18-341).  Here a frame is
//
//   k_pt_first                     first-hit shading from the G-buffer: NEE ray + extension ray
//   repeat for pathLength = 2..max(2, maxPathLength):
//     trace any   (NEE queue)      -> k_pt_apply_nee   contribution += alpha * NEE  (if unoccluded)
//     trace closest (ext queue)    -> k_pt_bounce      closest-hit / miss program of that vertex
//   k_pt_finish                    running mean into the beauty buffer
//
// Both ray queues are dense (ballot/popcount compaction, one atomic per wave) and shrink every
// bounce; every launch covers the queue capacity and reads the live count from the device, so the
// whole frame is enqueued without a host sync.  Per-path arithmetic, RNG draw order and the order
// of the floating-point adds into `contribution` are those of the reference, so the beauty buffer
// and the RNG states match the CPU restatement (oracle/orc_pathtrace.h) bit for bit.
//
// ReGIR (regir/gpu_kernels/build_cell_reservoirs.cu, regir/gpu_kernels/optix_pathtracing_kernels.cu)
// reuses the same pipeline with REGIR = true: NEE resamples the light-slot reservoirs of the grid cell
// under the shading point, Russian roulette moves to the loop head, and three small kernels maintain
// the grid: k_regir_build<temporal>, k_regir_update_last_access.
#include <cstring>
#include "internal.h"
#include "shading.hip.h"
#include "pass_common.hip.h"
#include "restir_common.hip.h"

namespace gfx {

constexpr int kPtBlock = 256;
constexpr uint32_t kNumLightSlotsPerCell = 512;   // regir_shared.h:7

struct PtArgs {
    DevScene scene;
    gfx_restir_static_params s;
    gfx_restir_frame_params f;
    size_t pixelBegin, pixelEnd;
    uint32_t pathLength;          // pathLength of the vertices k_pt_bounce processes
    uint32_t maxLengthTerminate;  // pathLength >= maxPathLength
    // NEE (any-hit) queue, rebuilt every bounce
    float4* neeOrg; float4* neeDir; float4* neePending;   // pending = alpha * NEE (rgb), owner pixel (w)
    uint32_t* neeCount;
    const uint32_t* occluded;
    // extension (closest-hit) queues: [in] produced by the previous vertex, [out] by this one
    const float4* extOrgIn; const float4* extDirIn; const uint32_t* extOwnerIn; const uint32_t* extCountIn;
    float4* extOrgOut; float4* extDirOut; uint32_t* extOwnerOut; uint32_t* extCountOut;
    const gfx_hit* hits;
    const Bvh8Tri* tris;
    float4* state;                // per pixel: [2p] = alpha.rgb, prevDirPDensity; [2p+1] = contribution.rgb
    gfx_regir_params g;           // ReGIR grid (REGIR kernels only)
    uint32_t nextMaxLengthTerminate;   // pathLength + 1 >= maxPathLength (ReGIR loop head)
};

// ---------------------------------------------------------------- ReGIR
GFX_DEV uint32_t regir_cell_index(const gfx_regir_params& g, f3 pw) {   // regir_shared.h:731-744
    const float rx = (pw.x - g.gridOrigin[0]) / g.gridCellSize[0];
    const float ry = (pw.y - g.gridOrigin[1]) / g.gridCellSize[1];
    const float rz = (pw.z - g.gridOrigin[2]) / g.gridCellSize[2];
    uint32_t ix = f2u_sat(rx); if (ix > g.gridDimension[0] - 1) ix = g.gridDimension[0] - 1;
    uint32_t iy = f2u_sat(ry); if (iy > g.gridDimension[1] - 1) iy = g.gridDimension[1] - 1;
    uint32_t iz = f2u_sat(rz); if (iz > g.gridDimension[2] - 1) iz = g.gridDimension[2] - 1;
    return iz * g.gridDimension[0] * g.gridDimension[1] + iy * g.gridDimension[0] + ix;
}

// atomicAdd(&perCellNumAccesses[cell], 1) with the lanes of a wave that touch the same cell merged
// into one atomic (same totals; neighbouring pixels mostly share a cell).
GFX_DEV void regir_count_access(uint32_t* perCellNumAccesses, bool active, uint32_t cell) {
    unsigned long long todo = __ballot(active);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const uint32_t leaderCell = __shfl(cell, leader);
        const unsigned long long same = __ballot(active && cell == leaderCell) & todo;
        if (lane == leader) atomicAdd(perCellNumAccesses + leaderCell, static_cast<uint32_t>(__popcll(same)));
        todo &= ~same;
    }
}

// sampleFromCell, regir/gpu_kernels/optix_pathtracing_kernels.cu:18-82.  Every lane of the wave must
// call it (`active` = lanes that shade a vertex).
GFX_DEV f3 regir_sample_from_cell(const PtArgs& a, bool active, f3 pos, f3 vOutLocal, const Frame& frame, const Bsdf& bsdf, Pcg32& rng,
                                  LightSample& ls, float& recPDF) {
    const gfx_regir_params& g = a.g;
    uint32_t cell = 0;
    if (active) {
        f3 off(0.0f);
        if (g.enableCellRandomization) {
            const float r0 = rng.uniform();
            const float r1 = rng.uniform();
            const float r2 = rng.uniform();
            off = f3(g.gridCellSize[0], g.gridCellSize[1], g.gridCellSize[2]) * f3(-0.5f + r0, -0.5f + r1, -0.5f + r2);
        }
        cell = regir_cell_index(g, pos + off);
    }
    regir_count_access(static_cast<uint32_t*>(g.perCellNumAccesses), active, cell);
    f3 selectedContribution(0.0f);
    ls.emittance = f3(0.0f); ls.position = f3(0.0f); ls.normal = f3(0.0f); ls.atInfinity = 0;
    recPDF = 0.0f;
    if (!active) return selectedContribution;
    const size_t numLightSlots = static_cast<size_t>(g.gridDimension[0]) * g.gridDimension[1] * g.gridDimension[2] * kNumLightSlotsPerCell;
    const size_t resStart = static_cast<size_t>(kNumLightSlotsPerCell) * cell;
    const uint32_t bufferIndex = a.f.bufferIndex;
    const uint32_t numResampling = 1u << g.log2NumCandidatesPerCell;
    Reservoir combined;
    combined.reset();
    uint32_t combinedStreamLength = 0;
    float selectedTarget = 0.0f;
    for (uint32_t i = 0; i < numResampling; ++i) {
        uint32_t k = f2u_sat(rng.uniform() * kNumLightSlotsPerCell);
        if (k > kNumLightSlotsPerCell - 1) k = kNumLightSlotsPerCell - 1;
        const size_t slot = resStart + k;
        const Reservoir r = load_reservoir(g.reservoirs[bufferIndex], numLightSlots, slot);
        const float slotRecPDF = static_cast<const float2*>(g.reservoirInfos[bufferIndex])[slot].x;
        combinedStreamLength += r.streamLength;
        if (slotRecPDF == 0.0f) continue;
        const f3 cont = direct_lighting(pos, vOutLocal, frame, bsdf, r.sample);
        const float target = target_weight(cont);
        const float weight = target * slotRecPDF * r.streamLength;
        if (combined.update(r.sample, weight, rng.uniform())) { selectedContribution = cont; selectedTarget = target; }
    }
    combined.streamLength = combinedStreamLength;
    ls = combined.sample;
    const float weightForEstimate = 1.0f / combined.streamLength;
    recPDF = weightForEstimate * combined.sumWeights / selectedTarget;
    if (!is_finite(recPDF)) recPDF = 0.0f;
    return selectedContribution;
}

struct PtVertexOut {              // what one shaded vertex hands to the queues
    bool wantNee; f3 neeDir; float neeTmax; f3 pending;
    bool wantExt; f3 extDir;
};

// performNextEventEstimation (optix_pathtracing_kernels.cu:18-72) without the trace: returns the
// unshadowed estimate and the shadow ray; + BSDF sampling of the next direction (:140-147, :283-295)
// and the head of the path extension loop.  Every lane of the wave must call it (`active` = lanes
// that shade a vertex) because the ReGIR variant merges its cell-access atomics across the wave.
template <bool REGIR>
GFX_DEV void shade_vertex(const PtArgs& a, bool active, const EnvMap& env, bool envEnabled, f3 pos, f3 vOutLocal, const Frame& frame,
                          const Bsdf& bsdf, Pcg32& rng, f3& alpha, f3& contribution, float& dirPDensity, PtVertexOut& o) {
    f3 ret(0.0f);
    LightSample ls;
    ls.emittance = f3(0.0f); ls.position = f3(0.0f); ls.normal = f3(0.0f); ls.atInfinity = 0;
    if (REGIR) {
        float recPDF;
        const f3 unshadowed = regir_sample_from_cell(a, active, pos, vOutLocal, frame, bsdf, rng, ls, recPDF);
        if (!active) return;
        if (recPDF > 0.0f) ret = unshadowed * (1.0f * recPDF);     // visibility * recProbDensityEstimate (:100)
    }
    else {
        if (!active) return;
        const float* instWeights = a.scene.lightWeights + a.scene.lightInstDistOffset;
        const float* instCDF = a.scene.lightCDF + a.scene.lightInstDistOffset;
        float ul = rng.uniform();
        bool selectEnv = false;
        float probCurType = 1.0f;
        if (envEnabled) {
            if (*a.scene.lightInstIntegral > 0.0f) {
                if (ul < 0.25f) { probCurType = 0.25f; ul = ul / probCurType; selectEnv = true; }
                else { probCurType = 1.0f - 0.25f; ul = (ul - 0.25f) / probCurType; }
            }
            else selectEnv = true;
        }
        float areaPDensity;
        const float u0 = rng.uniform();
        const float u1 = rng.uniform();
        sample_light(a.scene, instWeights, instCDF, env, a.f.envLightRotation, a.f.envLightPowerCoeff, ul, selectEnv, u0, u1, ls, areaPDensity);
        areaPDensity *= probCurType;
        const ShadowRay sr = shadow_ray(pos, ls);
        float misWeight;
        {
            const f3 vInLocal = frame.to_local(sr.dir);
            const float lpCos = fabsf(dot(sr.dir, ls.normal));
            float bsdfPDensity = bsdf.evaluate_pdf(vOutLocal, vInLocal) * lpCos / sr.dist2;
            if (!is_finite(bsdfPDensity)) bsdfPDensity = 0.0f;
            misWeight = (areaPDensity * areaPDensity) / (bsdfPDensity * bsdfPDensity + areaPDensity * areaPDensity);
        }
        if (areaPDensity > 0.0f) ret = direct_lighting(pos, vOutLocal, frame, bsdf, ls) * (misWeight / areaPDensity);
    }
    const ShadowRay sr = shadow_ray(pos, ls);
    o.wantNee = ret.x != 0.0f || ret.y != 0.0f || ret.z != 0.0f;
    o.neeDir = sr.dir; o.neeTmax = sr.tmax;
    o.pending = alpha * ret;
    if (!o.wantNee) contribution = contribution + o.pending;   // nothing to trace: the add happens here

    f3 vInLocal;
    const float b0 = rng.uniform();
    const float b1 = rng.uniform();
    alpha = alpha * bsdf.sample_throughput(vOutLocal, b0, b1, vInLocal, dirPDensity);
    o.extDir = frame.from_local(vInLocal);
    // loop head of the ray-generation program (:163-166): only valid samples are extended
    o.wantExt = dirPDensity > 0.0f && is_finite(dirPDensity);
    if (REGIR && o.wantExt) {   // regir/gpu_kernels/optix_pathtracing_kernels.cu:247-256
        if (a.nextMaxLengthTerminate) o.wantExt = false;
        else {
            const float continueProb = fminf(luminance_srgb(alpha) / luminance_srgb(f3(1.0f)), 1.0f);
            if (rng.uniform() >= continueProb) o.wantExt = false;
            else alpha = alpha / continueProb;
        }
    }
}

GFX_DEV void push_vertex(const PtArgs& a, uint32_t pixel, f3 pos, const PtVertexOut& o) {
    const uint32_t ns = queue_append(o.wantNee, pos, o.neeDir, 0.0f, o.neeTmax, a.neeOrg, a.neeDir, a.neeCount);
    if (o.wantNee) a.neePending[ns] = make_float4(o.pending.x, o.pending.y, o.pending.z, bits2f(pixel));
    const uint32_t es = queue_append(o.wantExt, pos, o.extDir, 0.0f, 3.402823466e+38f, a.extOrgOut, a.extDirOut, a.extCountOut);
    if (o.wantExt) a.extOwnerOut[es] = pixel;
}

// pathTrace_rayGen_generic up to the path extension loop (optix_pathtracing_kernels.cu:74-160)
template <bool REGIR>
__global__ __launch_bounds__(kPtBlock) void k_pt_first(PtArgs a) {
    const size_t p = a.pixelBegin + static_cast<size_t>(blockIdx.x) * kPtBlock + threadIdx.x;
    const uint32_t bufIdx = a.f.bufferIndex;
    PtVertexOut o;
    o.wantNee = false; o.wantExt = false; o.neeDir = f3(0.0f); o.extDir = f3(0.0f); o.neeTmax = 0; o.pending = f3(0.0f);
    f3 pos(0.0f), vOutLocal(0.0f);
    Frame frame(f3(0, 0, 1), f3(1, 0, 0));
    Bsdf bsdf;
    Pcg32 rng; rng.state = 0;
    const EnvMap env = load_env(a.s);
    const bool envEnabled = env.present() && a.f.enableEnvLight;
    f3 contribution(0.001f, 0.001f, 0.001f);
    f3 alpha(1.0f);
    float dirPDensity = 0.0f;
    bool surface = false;
    uint4 g0 = make_uint4(0xFFFFFFFFu, 0, 0, 0);
    if (p < a.pixelEnd) g0 = static_cast<const uint4*>(a.s.gbuffer0[bufIdx])[p];
    surface = g0.x != 0xFFFFFFFFu;
    if (surface) {
        const float bcB = decode_bc(g0.w & 0xFFFF), bcC = decode_bc(g0.w >> 16);
        const DevInstance* inst = a.scene.insts + g0.x;
        const DevGeomInst g = a.scene.geomInsts[g0.y];
        // computeSurfacePoint, path_tracing_shared.h:582-621
        const uint32_t* tri = a.scene.triangles + 3ull * (g.triangleOffset + g0.z);
        const DevVertex vA = load_vertex(a.scene.vertices + g.vertexOffset + tri[0]);
        const DevVertex vB = load_vertex(a.scene.vertices + g.vertexOffset + tri[1]);
        const DevVertex vC = load_vertex(a.scene.vertices + g.vertexOffset + tri[2]);
        const float bcA = 1 - (bcB + bcC);
        const f3 pAo(vA.px, vA.py, vA.pz), pBo(vB.px, vB.py, vB.pz), pCo(vC.px, vC.py, vC.pz);
        const m34 xfm = load_m34(inst->transform);
        const m33 nrm = load_m33_rows(inst->normalMatrix);
        pos = xfm_point(xfm, bcA * pAo + bcB * pBo + bcC * pCo);
        f3 ng = unit(mul(nrm, cross(pBo - pAo, pCo - pAo)));
        const f3 nsObj = bcA * f3(vA.nx, vA.ny, vA.nz) + bcB * f3(vB.nx, vB.ny, vB.nz) + bcC * f3(vC.nx, vC.ny, vC.nz);
        const f3 tcObj = bcA * f3(vA.tx, vA.ty, vA.tz) + bcB * f3(vB.tx, vB.ty, vB.tz) + bcC * f3(vC.tx, vC.ty, vC.tz);
        f3 ns = unit(mul(nrm, nsObj));
        f3 tc0 = xfm_vector(xfm, tcObj);
        tc0 = unit(tc0 - dot(ns, tc0) * ns);
        if (!all_finite(ns)) { ng = f3(0, 0, 1); ns = f3(0, 0, 1); tc0 = f3(1, 0, 0); }
        if (!all_finite(tc0)) { f3 bt; make_coordinate_system(ns, tc0, bt); }

        const Camera cam = load_camera(a.f.camera);
        rng.state = static_cast<const uint64_t*>(a.s.rngBuffer)[p];
        const gfx_material& mat = a.scene.materials[g.materialSlot];
        const f3 vOut = unit(cam.pos - pos);
        const float frontHit = dot(vOut, ng) >= 0.0f ? 1.0f : -1.0f;
        pos = offset_ray_origin(pos, frontHit * ng);
        frame = Frame(ns, tc0);
        vOutLocal = frame.to_local(vOut);
        contribution = f3(0.0f);
        if (vOutLocal.z > 0 && mat.hasEmittance)
            contribution = contribution + alpha * f3(mat.emittance[0], mat.emittance[1], mat.emittance[2]) / kPi;
        bsdf.setup(mat);
    }
    else if (p < a.pixelEnd && envEnabled) {
        contribution = a.f.envLightPowerCoeff * env.fetch(decode_bc(g0.w & 0xFFFF), decode_bc(g0.w >> 16));
    }
    shade_vertex<REGIR>(a, surface, env, envEnabled, pos, vOutLocal, frame, bsdf, rng, alpha, contribution, dirPDensity, o);
    if (surface) static_cast<uint64_t*>(a.s.rngBuffer)[p] = rng.state;
    if (p < a.pixelEnd) {
        a.state[2 * p] = make_float4(alpha.x, alpha.y, alpha.z, dirPDensity);
        a.state[2 * p + 1] = make_float4(contribution.x, contribution.y, contribution.z, 0.0f);
    }
    push_vertex(a, static_cast<uint32_t>(p), pos, o);
}

// The visibility factor of performDirectLighting<.., true> applied after the any-hit trace:
// contribution += alpha * NEE  (optix_pathtracing_kernels.cu:136-137, :279-280)
__global__ __launch_bounds__(kPtBlock) void k_pt_apply_nee(PtArgs a) {
    const uint32_t i = blockIdx.x * kPtBlock + threadIdx.x;
    if (i >= *a.neeCount) return;
    const float4 pend = a.neePending[i];
    const uint32_t pixel = f2bits(pend.w);
    f3 add(pend.x, pend.y, pend.z);
    if (a.occluded[i]) add = add * 0.0f;      // alpha * RGB(0): zero unless the throughput is not finite
    float4* c = a.state + 2ull * pixel + 1;
    const float4 cur = *c;
    *c = make_float4(cur.x + add.x, cur.y + add.y, cur.z + add.z, 0.0f);
}

// closest-hit + miss programs of one path vertex, then the loop head of the ray-generation program
// (optix_pathtracing_kernels.cu:210-296, :306-341, :161-201; ReGIR: regir/.../optix_pathtracing_kernels.cu:302-392)
template <bool REGIR>
__global__ __launch_bounds__(kPtBlock) void k_pt_bounce(PtArgs a) {
    const uint32_t i = blockIdx.x * kPtBlock + threadIdx.x;
    const uint32_t count = *a.extCountIn;
    PtVertexOut o;
    o.wantNee = false; o.wantExt = false; o.neeDir = f3(0.0f); o.extDir = f3(0.0f); o.neeTmax = 0; o.pending = f3(0.0f);
    f3 pos(0.0f), vOutLocal(0.0f);
    Frame frame(f3(0, 0, 1), f3(1, 0, 0));
    Bsdf bsdf;
    Pcg32 rng; rng.state = 0;
    const EnvMap env = load_env(a.s);
    const bool envEnabled = env.present() && a.f.enableEnvLight;
    f3 alpha(0.0f), contribution(0.0f);
    float dirPDensity = 0.0f;
    uint32_t pixel = 0;
    bool hitSurface = false, shade = false;
    if (i < count) {
        pixel = a.extOwnerIn[i];
        const gfx_hit h = a.hits[i];
        const float4 ro4 = a.extOrgIn[i], rd4 = a.extDirIn[i];
        const f3 rayOrg(ro4.x, ro4.y, ro4.z), rayDir(rd4.x, rd4.y, rd4.z);
        const float4 s0 = a.state[2ull * pixel], s1 = a.state[2ull * pixel + 1];
        alpha = f3(s0.x, s0.y, s0.z);
        const float prevDirPDensity = s0.w;
        dirPDensity = prevDirPDensity;
        contribution = f3(s1.x, s1.y, s1.z);
        if (h.triIndex == GFX_INVALID_SLOT) {
            if (envEnabled && !REGIR) {   // the ReGIR ray type has an empty miss program (regir_main.cpp:250)
                const f3 rd = unit(rayDir);
                float posPhi, theta;
                to_polar_yup(rd, posPhi, theta);
                float phi = posPhi + a.f.envLightRotation;
                phi = phi - floorf(phi / (2 * kPi)) * 2 * kPi;
                const float tu = phi / (2 * kPi), tv = theta / kPi;
                const f3 luminance = a.f.envLightPowerCoeff * env.fetch(tu, tv);
                const float uvPDF = env.evaluate_pdf(tu, tv);
                const float hypAreaPDensity = uvPDF / (2 * kPi * kPi * gm_sin(theta));
                const float lightPDensity = (*a.scene.lightInstIntegral > 0.0f ? 0.25f : 1.0f) * hypAreaPDensity;
                const float misWeight = (prevDirPDensity * prevDirPDensity) / (prevDirPDensity * prevDirPDensity + lightPDensity * lightPDensity);
                contribution = contribution + alpha * luminance * misWeight;
                a.state[2ull * pixel + 1] = make_float4(contribution.x, contribution.y, contribution.z, 0.0f);
            }
        }
        else {
            hitSurface = true;
            const Bvh8Tri* tr = a.tris + h.triIndex;
            const uint32_t instSlot = tr->instSlot, geomInstSlot = tr->geomInstSlot, primIndex = tr->primIndex;
            const DevInstance* inst = a.scene.insts + instSlot;
            const DevGeomInst g = a.scene.geomInsts[geomInstSlot];
            // computeSurfacePoint<true, false>, path_tracing_shared.h:485-580
            const uint32_t* tri = a.scene.triangles + 3ull * (g.triangleOffset + primIndex);
            const DevVertex vA = load_vertex(a.scene.vertices + g.vertexOffset + tri[0]);
            const DevVertex vB = load_vertex(a.scene.vertices + g.vertexOffset + tri[1]);
            const DevVertex vC = load_vertex(a.scene.vertices + g.vertexOffset + tri[2]);
            const m34 xfm = load_m34(inst->transform);
            const m33 nrm = load_m33_rows(inst->normalMatrix);
            const f3 pA = xfm_point(xfm, f3(vA.px, vA.py, vA.pz));
            const f3 pB = xfm_point(xfm, f3(vB.px, vB.py, vB.pz));
            const f3 pC = xfm_point(xfm, f3(vC.px, vC.py, vC.pz));
            const float bcB = h.bcB, bcC = h.bcC;
            const float bcA = 1 - (bcB + bcC);
            pos = bcA * pA + bcB * pB + bcC * pC;
            const f3 nsObj = bcA * f3(vA.nx, vA.ny, vA.nz) + bcB * f3(vB.nx, vB.ny, vB.nz) + bcC * f3(vC.nx, vC.ny, vC.nz);
            const f3 tcObj = bcA * f3(vA.tx, vA.ty, vA.tz) + bcB * f3(vB.tx, vB.ty, vB.tz) + bcC * f3(vC.tx, vC.ty, vC.tz);
            f3 ng = cross(pB - pA, pC - pA);
            const float area = 0.5f * len(ng);
            ng = ng / (2 * area);
            f3 ns = unit(mul(nrm, nsObj));
            f3 tc0 = unit(xfm_vector(xfm, tcObj));
            if (!all_finite(ns)) { ns = f3(0, 0, 1); tc0 = f3(1, 0, 0); }
            if (!all_finite(tc0)) { f3 bt; make_coordinate_system(ns, tc0, bt); }
            // hypothetical light density of the hit point; pathTraceReGIR reads it uninitialised in the
            // reference (undefined) -- this build defines it as 0 there
            float hypAreaPDensity = 0.0f;
            if (!REGIR) {
                float lightProb = 1.0f;
                if (envEnabled) lightProb *= (1 - 0.25f);
                const float instImportance = inst->distIntegral;
                lightProb *= (inst->uniformScale * inst->uniformScale * instImportance) / *a.scene.lightInstIntegral;
                lightProb *= g.distIntegral / instImportance;
                if (is_finite(lightProb)) {
                    float pmf = 0.0f;
                    if (g.distOffset != 0xFFFFFFFFu && g.distIntegral != 0.0f) pmf = a.scene.lightWeights[g.distOffset + primIndex] / g.distIntegral;
                    lightProb *= pmf;
                    hypAreaPDensity = lightProb / area;
                }
            }
            const gfx_material& mat = a.scene.materials[g.materialSlot];
            const f3 vOut = unit(-rayDir);
            const float frontHit = dot(vOut, ng) >= 0.0f ? 1.0f : -1.0f;
            frame = Frame(ns, tc0);
            pos = offset_ray_origin(pos, frontHit * ng);
            vOutLocal = frame.to_local(vOut);
            if (vOutLocal.z > 0 && mat.hasEmittance) {
                const f3 emittance(mat.emittance[0], mat.emittance[1], mat.emittance[2]);
                const float dist2 = len2(rayOrg - pos);
                const float lightPDensity = hypAreaPDensity * dist2 / vOutLocal.z;
                const float misWeight = (prevDirPDensity * prevDirPDensity) / (prevDirPDensity * prevDirPDensity + lightPDensity * lightPDensity);
                contribution = contribution + alpha * emittance * (misWeight / kPi);
            }
            rng.state = static_cast<const uint64_t*>(a.s.rngBuffer)[pixel];
            // Russian roulette; initImportance = sRGB_calcLuminance(RGB(1))
            const float continueProb = fminf(luminance_srgb(alpha) / luminance_srgb(f3(1.0f)), 1.0f);
            if (!(rng.uniform() >= continueProb || a.maxLengthTerminate)) {
                alpha = alpha / continueProb;
                bsdf.setup(mat);
                shade = true;
            }
        }
    }
    shade_vertex<REGIR>(a, shade, env, envEnabled, pos, vOutLocal, frame, bsdf, rng, alpha, contribution, dirPDensity, o);
    if (hitSurface) {
        static_cast<uint64_t*>(a.s.rngBuffer)[pixel] = rng.state;
        a.state[2ull * pixel] = make_float4(alpha.x, alpha.y, alpha.z, dirPDensity);
        a.state[2ull * pixel + 1] = make_float4(contribution.x, contribution.y, contribution.z, 0.0f);
    }
    push_vertex(a, pixel, pos, o);
}

// ---------------------------------------------------------------- ReGIR grid maintenance
// sampleIntensity, regir/gpu_kernels/build_cell_reservoirs.cu:6-68 (the half-space tests compare lpCos,
// still 1 at that point, with minSquaredDistance -- restated literally)
GFX_DEV f3 regir_sample_intensity(const LightSample& ls, f3 cellCenter, f3 halfCellSize, float minSquaredDistance) {
    float dist2 = minSquaredDistance;
    float lpCos = 1;
    const bool outside = ls.atInfinity ||
        ls.position.x < cellCenter.x - halfCellSize.x || ls.position.x > cellCenter.x + halfCellSize.x ||
        ls.position.y < cellCenter.y - halfCellSize.y || ls.position.y > cellCenter.y + halfCellSize.y ||
        ls.position.z < cellCenter.z - halfCellSize.z || ls.position.z > cellCenter.z + halfCellSize.z;
    if (outside) {
        const f3 d = ls.atInfinity ? ls.position : (ls.position - cellCenter);
        const float perpDistance = dot(-d, ls.normal);
        dist2 = len2(d);
        const float dist = sqrtf(dist2);
        const bool valid = lpCos > minSquaredDistance || ls.atInfinity;
        const bool invalid = lpCos < -minSquaredDistance;
        if (valid) lpCos = perpDistance / dist;
        else if (invalid) lpCos = 0.0f;
    }
    if (lpCos > 0.0f) {
        const f3 Le = ls.emittance / kPi;
        return Le * (lpCos / dist2);
    }
    return f3(0.0f);
}

// buildCellReservoirsAndTemporalReuse<TEMPORAL>, build_cell_reservoirs.cu:70-219: one thread per light slot
template <bool TEMPORAL>
__global__ __launch_bounds__(kPtBlock) void k_regir_build(PtArgs a) {
    const gfx_regir_params& g = a.g;
    const uint32_t numCells = g.gridDimension[0] * g.gridDimension[1] * g.gridDimension[2];
    const size_t numLightSlots = static_cast<size_t>(numCells) * kNumLightSlotsPerCell;
    const size_t i = static_cast<size_t>(blockIdx.x) * kPtBlock + threadIdx.x;
    if (i >= numLightSlots) return;
    const uint32_t bufferIndex = a.f.bufferIndex;
    const uint32_t cell = static_cast<uint32_t>(i / kNumLightSlotsPerCell);
    const uint32_t lastAccess = static_cast<const uint32_t*>(g.lastAccessFrameIndices)[cell];
    if (i == 0) *static_cast<uint32_t*>(g.numActiveCells[bufferIndex]) = 0;
    if (i % kNumLightSlotsPerCell == 0) static_cast<uint32_t*>(g.perCellNumAccesses)[cell] = 0;
    if (a.f.frameIndex - lastAccess > 8) return;
    const uint32_t gx = g.gridDimension[0], gy = g.gridDimension[1];
    const uint32_t iz = cell / (gx * gy), iy = (cell % (gx * gy)) / gx, ix = cell % gx;
    const f3 cs(g.gridCellSize[0], g.gridCellSize[1], g.gridCellSize[2]);
    const f3 cellCenter = f3(g.gridOrigin[0], g.gridOrigin[1], g.gridOrigin[2]) + f3((ix + 0.5f) * cs.x, (iy + 0.5f) * cs.y, (iz + 0.5f) * cs.z);
    const f3 halfCellSize = 0.5f * cs;
    const float minSquaredDistance = len2(0.5f * cs);
    uint64_t* rngs = static_cast<uint64_t*>(g.lightSlotRngs);
    Pcg32 rng; rng.state = rngs[i];
    const EnvMap env = load_env(a.s);
    const bool envEnabled = env.present() && a.f.enableEnvLight;
    const float* instWeights = a.scene.lightWeights + a.scene.lightInstDistOffset;
    const float* instCDF = a.scene.lightCDF + a.scene.lightInstDistOffset;
    float selectedTarget = 0.0f;
    Reservoir reservoir;
    reservoir.reset();
    const uint32_t numCandidates = 1u << g.log2NumCandidatesPerLightSlot;
    for (uint32_t c = 0; c < numCandidates; ++c) {
        float ul = rng.uniform();
        bool sampleEnv = false;
        float probCurType = 1.0f;
        if (envEnabled) {
            if (*a.scene.lightInstIntegral > 0.0f) {
                const float prob = fmin2(fmax2(0.25f * numCandidates - c, 0.0f), 1.0f);
                if (ul < prob) { probCurType = 0.25f; ul = ul / prob; sampleEnv = true; }
                else { probCurType = 1.0f - 0.25f; ul = (ul - prob) / (1 - prob); }
            }
            else sampleEnv = true;
        }
        LightSample ls;
        ls.emittance = f3(0.0f); ls.position = f3(0.0f); ls.normal = f3(0.0f); ls.atInfinity = 0;
        float pd;
        const float u0 = rng.uniform();
        const float u1 = rng.uniform();
        sample_light(a.scene, instWeights, instCDF, env, a.f.envLightRotation, a.f.envLightPowerCoeff, ul, sampleEnv, u0, u1, ls, pd);
        const f3 cont = regir_sample_intensity(ls, cellCenter, halfCellSize, minSquaredDistance);
        pd *= probCurType;
        const float target = target_weight(cont);
        const float weight = target / pd;
        if (reservoir.update(ls, weight, rng.uniform())) selectedTarget = target;
    }
    float recPDF = reservoir.sumWeights / (selectedTarget * reservoir.streamLength);
    if (!is_finite(recPDF)) { recPDF = 0.0f; selectedTarget = 0.0f; }
    if (TEMPORAL) {
        const uint32_t prevBuffer = (bufferIndex + 1) % 2;
        const uint32_t selfStreamLength = reservoir.streamLength;
        if (recPDF == 0.0f) reservoir.reset();
        uint32_t combinedStreamLength = selfStreamLength;
        const uint32_t maxNumPrevSamples = 20 * selfStreamLength;
        const Reservoir prev = load_reservoir(g.reservoirs[prevBuffer], numLightSlots, i);
        const float prevTarget = static_cast<const float2*>(g.reservoirInfos[prevBuffer])[i].y;
        const uint32_t prevLen = prev.streamLength < maxNumPrevSamples ? prev.streamLength : maxNumPrevSamples;
        const float lengthCorrection = static_cast<float>(prevLen) / prev.streamLength;
        const float weight = lengthCorrection * prev.sumWeights;
        if (reservoir.update(prev.sample, weight, rng.uniform())) selectedTarget = prevTarget;
        combinedStreamLength += prevLen;
        reservoir.streamLength = combinedStreamLength;
        const float weightForEstimate = 1.0f / reservoir.streamLength;
        recPDF = weightForEstimate * reservoir.sumWeights / selectedTarget;
        if (!is_finite(recPDF)) { recPDF = 0.0f; selectedTarget = 0.0f; }
    }
    rngs[i] = rng.state;
    store_reservoir(g.reservoirs[bufferIndex], numLightSlots, i, reservoir);
    static_cast<float2*>(g.reservoirInfos[bufferIndex])[i] = make_float2(recPDF, selectedTarget);
}

// updateLastAccessFrameIndices, build_cell_reservoirs.cu:229-243
__global__ __launch_bounds__(kPtBlock) void k_regir_update_last_access(PtArgs a) {
    const gfx_regir_params& g = a.g;
    const uint32_t numCells = g.gridDimension[0] * g.gridDimension[1] * g.gridDimension[2];
    const uint32_t cell = blockIdx.x * kPtBlock + threadIdx.x;
    bool accessed = false;
    if (cell < numCells) {
        accessed = static_cast<const uint32_t*>(g.perCellNumAccesses)[cell] > 0;
        if (accessed) static_cast<uint32_t*>(g.lastAccessFrameIndices)[cell] = a.f.frameIndex;
    }
    const unsigned long long mask = __ballot(accessed);
    if ((threadIdx.x & 63) == 0 && mask) atomicAdd(static_cast<uint32_t*>(g.numActiveCells[a.f.bufferIndex]), static_cast<uint32_t>(__popcll(mask)));
}

// running mean (optix_pathtracing_kernels.cu:203-208)
__global__ __launch_bounds__(kPtBlock) void k_pt_finish(PtArgs a) {
    const size_t p = a.pixelBegin + static_cast<size_t>(blockIdx.x) * kPtBlock + threadIdx.x;
    if (p >= a.pixelEnd) return;
    const float4 c = a.state[2 * p + 1];
    const f3 contribution(c.x, c.y, c.z);
    float4* beauty = static_cast<float4*>(a.s.beautyAccumBuffer) + p;
    f3 prev(0.0f);
    if (a.f.numAccumFrames > 0) { const float4 b = *beauty; prev = f3(b.x, b.y, b.z); }
    const float curWeight = 1.0f / (1 + a.f.numAccumFrames);
    const f3 result = (1 - curWeight) * prev + curWeight * contribution;
    *beauty = make_float4(result.x, result.y, result.z, 1.0f);
}

// ---------------------------------------------------------------- host sequencing
void pathtrace_launch(Context& ctx, hipStream_t stream, int pass, uint32_t width, uint32_t height,
                      uint32_t maxPathLength, uint32_t rowBegin, uint32_t rowEnd) {
    if (pass == GFX_PT_SETUP_GBUFFERS) {
        // path_tracing/gpu_kernels/optix_gbuffer_kernels.cu: same program text as ReSTIR's G-buffer
        // pass (GBuffer0/1 + albedo/normal accumulation); GBuffer2/3 are simply not read afterwards.
        restir_launch(ctx, stream, GFX_RESTIR_SETUP_GBUFFERS, width, height, rowBegin, rowEnd);
        return;
    }
    const RestirParams& rp = ctx.restir;
    if (!rp.valid) throw HipError("gfx_pt_launch: gfx_restir_set_params has not been called");
    const bool regirPass = pass >= GFX_PT_REGIR_BUILD_CELL_RESERVOIRS && pass <= GFX_PT_REGIR_UPDATE_LAST_ACCESS;
    if (!regirPass && pass != GFX_PT_PATH_TRACE_BASELINE) throw HipError("gfx_pt_launch: unknown pass");
    if (regirPass && !ctx.regirValid) throw HipError("gfx_pt_launch: gfx_regir_set_params has not been called");
    PtArgs a;
    std::memset(&a, 0, sizeof(a));
    a.scene = ctx.devScene();
    a.s = rp.s; a.f = rp.f;
    if (regirPass) a.g = ctx.regir;
    if (pass == GFX_PT_REGIR_BUILD_CELL_RESERVOIRS || pass == GFX_PT_REGIR_BUILD_CELL_RESERVOIRS_TEMPORAL) {
        const size_t numLightSlots = static_cast<size_t>(a.g.gridDimension[0]) * a.g.gridDimension[1] * a.g.gridDimension[2] * kNumLightSlotsPerCell;
        ScopedKernelTimer timer(ctx, stream, "regir_build_cells");
        const dim3 grid(static_cast<uint32_t>((numLightSlots + kPtBlock - 1) / kPtBlock));
        if (pass == GFX_PT_REGIR_BUILD_CELL_RESERVOIRS) hipLaunchKernelGGL(k_regir_build<false>, grid, dim3(kPtBlock), 0, stream, a);
        else hipLaunchKernelGGL(k_regir_build<true>, grid, dim3(kPtBlock), 0, stream, a);
        GFX_HIP(hipGetLastError());
        return;
    }
    if (pass == GFX_PT_REGIR_UPDATE_LAST_ACCESS) {
        const uint32_t numCells = a.g.gridDimension[0] * a.g.gridDimension[1] * a.g.gridDimension[2];
        ScopedKernelTimer timer(ctx, stream, "regir_update_last_access");
        hipLaunchKernelGGL(k_regir_update_last_access, dim3((numCells + kPtBlock - 1) / kPtBlock), dim3(kPtBlock), 0, stream, a);
        GFX_HIP(hipGetLastError());
        return;
    }
    const bool regir = pass == GFX_PT_PATH_TRACE_REGIR;
    if (static_cast<uint32_t>(rp.s.imageSizeX) != width || static_cast<uint32_t>(rp.s.imageSizeY) != height)
        throw HipError("gfx_pt_launch: launch size differs from imageSize in the static parameters");
    const uint64_t h = rp.f.travHandle;
    if (h == 0 || h > ctx.accels.size() || !ctx.accels[h - 1]) throw HipError("gfx_pt_launch: invalid travHandle");
    if (rowEnd > height || rowBegin > rowEnd) throw HipError("gfx_pt_launch: row range outside the image");
    maxPathLength &= 15u;   // 4-bit bitfield, path_tracing_shared.h:165
    const size_t numPixels = static_cast<size_t>(width) * height;
    const size_t bandPixels = static_cast<size_t>(rowEnd - rowBegin) * width;
    if (bandPixels == 0) return;
    ctx.rayOrg.reserve(16 * numPixels); ctx.rayDir.reserve(16 * numPixels);   // NEE queue
    ctx.rayOut.reserve(4 * numPixels);
    ctx.rayHits.reserve(sizeof(gfx_hit) * numPixels);
    ctx.ptPending.reserve(16 * bandPixels);
    ctx.ptExtOrg.reserve(2 * 16 * bandPixels); ctx.ptExtDir.reserve(2 * 16 * bandPixels); ctx.ptExtOwner.reserve(2 * 4 * bandPixels);
    ctx.ptState.reserve(32 * numPixels);
    ctx.smallCounters.reserve(256);
    uint32_t* counters = ctx.smallCounters.as<uint32_t>() + 8;   // [0] nee, [1] ext ping, [2] ext pong
    GFX_HIP(hipMemsetAsync(counters, 0, 3 * sizeof(uint32_t), stream));

    a.pixelBegin = static_cast<size_t>(rowBegin) * width;
    a.pixelEnd = static_cast<size_t>(rowEnd) * width;
    a.neeOrg = ctx.rayOrg.as<float4>(); a.neeDir = ctx.rayDir.as<float4>(); a.neePending = ctx.ptPending.as<float4>();
    a.neeCount = counters;
    a.occluded = ctx.rayOut.as<uint32_t>();
    a.hits = ctx.rayHits.as<gfx_hit>();
    a.tris = ctx.accels[h - 1]->tris.as<Bvh8Tri>();
    a.state = ctx.ptState.as<float4>();
    float4* extOrg[2] = { ctx.ptExtOrg.as<float4>(), ctx.ptExtOrg.as<float4>() + bandPixels };
    float4* extDir[2] = { ctx.ptExtDir.as<float4>(), ctx.ptExtDir.as<float4>() + bandPixels };
    uint32_t* extOwner[2] = { ctx.ptExtOwner.as<uint32_t>(), ctx.ptExtOwner.as<uint32_t>() + bandPixels };
    const uint32_t grid = static_cast<uint32_t>((bandPixels + kPtBlock - 1) / kPtBlock);
    const DevAccel accel = ctx.accels[h - 1]->dev();
    auto set_queues = [&](int in, int out) {
        a.extOrgIn = extOrg[in]; a.extDirIn = extDir[in]; a.extOwnerIn = extOwner[in]; a.extCountIn = counters + 1 + in;
        a.extOrgOut = extOrg[out]; a.extDirOut = extDir[out]; a.extOwnerOut = extOwner[out]; a.extCountOut = counters + 1 + out;
    };
    auto launch = [&](const char* name, void (*kernel)(PtArgs)) {
        ScopedKernelTimer timer(ctx, stream, name);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kPtBlock), 0, stream, a);
        GFX_HIP(hipGetLastError());
    };
    auto trace = [&](int mode, const float4* org, const float4* dir, const uint32_t* count, void* out) {
        TraceLaunch t;
        t.accel = accel; t.rayOrgTmin = org; t.rayDirTmax = dir; t.numRays = 0; t.numRaysPtr = count; t.out = out; t.mode = mode;
        trace_launch(ctx, stream, t);
    };

    int cur = 0;                 // the queue k_pt_first fills
    set_queues(1, cur);
    a.pathLength = 1; a.maxLengthTerminate = 0;
    a.nextMaxLengthTerminate = 2 >= maxPathLength ? 1u : 0u;
    launch("pt_first", regir ? k_pt_first<true> : k_pt_first<false>);
    // while (true) { ++pathLength; trace; }.  Baseline: at least one extension even when maxPathLength < 2,
    // the terminal vertex (implicit light only) emits no NEE ray.  ReGIR: the loop head breaks before the
    // trace at the length limit, so the last vertex's NEE ray is resolved after the loop.
    for (uint32_t pathLength = 2;; ++pathLength) {
        trace(GFX_TRACE_ANY, a.neeOrg, a.neeDir, a.neeCount, ctx.rayOut.p);
        launch("pt_apply_nee", k_pt_apply_nee);
        if (regir && pathLength >= maxPathLength) break;
        trace(GFX_TRACE_CLOSEST, extOrg[cur], extDir[cur], counters + 1 + cur, ctx.rayHits.p);
        GFX_HIP(hipMemsetAsync(counters, 0, sizeof(uint32_t), stream));
        GFX_HIP(hipMemsetAsync(counters + 1 + (cur ^ 1), 0, sizeof(uint32_t), stream));
        set_queues(cur, cur ^ 1);
        a.pathLength = pathLength;
        a.maxLengthTerminate = pathLength >= maxPathLength ? 1u : 0u;
        a.nextMaxLengthTerminate = pathLength + 1 >= maxPathLength ? 1u : 0u;
        launch("pt_bounce", regir ? k_pt_bounce<true> : k_pt_bounce<false>);
        cur ^= 1;
        if (!regir && a.maxLengthTerminate) break;
    }
    launch("pt_finish", k_pt_finish);
}

} // namespace gfx
