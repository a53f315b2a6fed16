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
#include "trace_local.hip.h"

namespace gfx {

constexpr int kPtBlock = 256;
constexpr uint32_t kNumLightSlotsPerCell = 512;   // regir_shared.h:7

struct PtArgs {
    DevScene scene;
    gfx_restir_static_params s;
    gfx_restir_frame_params f;
    PixelGrid px;                 // per-pixel launches: rows [rowBegin, rowEnd) (restir_common.hip.h)
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
    gfx_nrc_params nrc;           // NRC render state (NRC kernels only)
    float4* nrcState;             // per pixel: [2p] = prevLocalThroughput.rgb, primaryPathSpread;
                                  //            [2p+1] = curSqrtPathSpread, prevTrainDataIndex, flags, -
    uint32_t* neeTrainIdx;        // per NEE slot: training record initialised with that NEE estimate (or invalid)
    uint32_t curRes;              // GFX_PT_PATH_TRACE_NRC_RESTIR: the reservoir buffer the frame's ReSTIR passes finished in
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
    f3 neeRet;                    // unshadowed NEE estimate (NRC: initial training target)
    f3 localThroughput;           // BSDF sampling throughput of this vertex
    uint32_t trainIdx;            // NRC: training record tied to the NEE ray
    bool neeFromOrg; f3 neeOrg;   // the NEE ray starts at neeOrg instead of the vertex (ReSTIR-driven first vertex: the G-buffer's shading point)
};

// performNextEventEstimation (optix_pathtracing_kernels.cu:18-72) without the trace: returns the
// unshadowed estimate and the shadow ray; + BSDF sampling of the next direction (:140-147, :283-295)
// and the head of the path extension loop.  Every lane of the wave must call it (`active` = lanes
// that shade a vertex) because the ReGIR variant merges its cell-access atomics across the wave.
// REGIR: NEE from the grid cell (sampleFromCell); REGIR_LOOP: Russian roulette at the loop head as pathTraceReGIR does
// (the NRC tracer with ReGIR NEE keeps its own Russian roulette: REGIR = true, REGIR_LOOP = false).
// What the ReSTIR DI passes of the frame left for pixel p (GFX_PT_PATH_TRACE_NRC_RESTIR): the direct term of the shading pass,
// recPDFEstimate x unshadowed performDirectLighting of the final reservoir sample at the G-buffer's shading point (restir.hip
// k_shade_prepare; optix_restir_di_kernels.cu:574-606), and the shadow ray that decides it.
struct ReservoirNee { f3 ret; f3 org; LightSample ls; };
GFX_DEV ReservoirNee restir_reservoir_nee(const PtArgs& a, uint32_t bufIdx, size_t p) {
    ReservoirNee r;
    r.ret = f3(0.0f); r.org = f3(0.0f);
    r.ls.emittance = f3(0.0f); r.ls.position = f3(0.0f); r.ls.normal = f3(0.0f); r.ls.atInfinity = 0;
    const size_t numPixels = static_cast<size_t>(a.s.imageSizeX) * a.s.imageSizeY;
    const Camera cam = load_camera(a.f.camera);
    ShadingPoint sp;
    make_shading_point(a, bufIdx, p, cam.pos, true, sp);
    const Reservoir reservoir = load_reservoir(a.s.reservoirBuffer[a.curRes], numPixels, p);
    const float recPDF = static_cast<const float2*>(a.s.reservoirInfoBuffer[a.curRes])[p].x;
    r.org = sp.pos; r.ls = reservoir.sample;
    if (recPDF > 0 && is_finite(recPDF)) r.ret = recPDF * direct_lighting(sp.pos, sp.vOutLocal, sp.frame, sp.bsdf, reservoir.sample);
    return r;
}

// `given`: the vertex's next-event estimation is already made (the ReSTIR-driven first vertex of the NRC tracer): no light is
// sampled and no random number drawn for it here.
template <bool REGIR, bool REGIR_LOOP = REGIR>
GFX_DEV void shade_vertex(const PtArgs& a, bool active, const EnvMap& env, bool envEnabled, f3 pos, f3 vOutLocal, const Frame& frame,
                          const Bsdf& bsdf, Pcg32& rng, f3& alpha, f3& contribution, float& dirPDensity, PtVertexOut& o,
                          const ReservoirNee* given = nullptr, int nextMaxLengthTerminate = -1 /* -1: a.nextMaxLengthTerminate */) {
    f3 ret(0.0f);
    LightSample ls;
    ls.emittance = f3(0.0f); ls.position = f3(0.0f); ls.normal = f3(0.0f); ls.atInfinity = 0;
    if (given) {
        if (!active) return;
        ret = given->ret; ls = given->ls;
    }
    else if (REGIR) {
        float recPDF;
        const f3 unshadowed = regir_sample_from_cell(a, active, pos, vOutLocal, frame, bsdf, rng, ls, recPDF);
        if (!active) return;
        if (recPDF > 0.0f) ret = unshadowed * (1.0f * recPDF);     // visibility * recProbDensityEstimate (:100)
    }
    else {
        if (!active) return;
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
        if (a.f.useSolidAngleSampling) sample_light_solid_angle(a.scene, env, a.f.envLightRotation, a.f.envLightPowerCoeff, pos, ul, selectEnv, u0, u1, ls, areaPDensity);
        else sample_light(a.scene, env, a.f.envLightRotation, a.f.envLightPowerCoeff, ul, selectEnv, u0, u1, ls, areaPDensity);
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
    const ShadowRay sr = shadow_ray(given ? given->org : pos, ls);
    if (given) { o.neeFromOrg = true; o.neeOrg = given->org; }
    o.wantNee = ret.x != 0.0f || ret.y != 0.0f || ret.z != 0.0f;
    o.neeDir = sr.dir; o.neeTmax = sr.tmax;
    o.pending = alpha * ret;
    o.neeRet = ret;
    if (!o.wantNee) contribution = contribution + o.pending;   // nothing to trace: the add happens here

    f3 vInLocal;
    const float b0 = rng.uniform();
    const float b1 = rng.uniform();
    o.localThroughput = bsdf.sample_throughput(vOutLocal, b0, b1, vInLocal, dirPDensity);
    alpha = alpha * o.localThroughput;
    o.extDir = frame.from_local(vInLocal);
    // loop head of the ray-generation program (:163-166): only valid samples are extended
    o.wantExt = dirPDensity > 0.0f && is_finite(dirPDensity);
    if (REGIR_LOOP && o.wantExt) {   // regir/gpu_kernels/optix_pathtracing_kernels.cu:247-256
        if (nextMaxLengthTerminate < 0 ? a.nextMaxLengthTerminate != 0u : nextMaxLengthTerminate != 0) o.wantExt = false;
        else {
            const float continueProb = fminf(luminance_srgb(alpha) / luminance_srgb(f3(1.0f)), 1.0f);
            if (rng.uniform() >= continueProb) o.wantExt = false;
            else alpha = alpha / continueProb;
        }
    }
}

GFX_DEV void push_vertex(const PtArgs& a, uint32_t pixel, f3 pos, const PtVertexOut& o) {
    // the vertex's NEE (any-hit) ray and extension (closest-hit) ray: both queue heads in one block-level step
    const bool want[2] = { o.wantNee, o.wantExt };
    uint32_t* const counters[2] = { a.neeCount, a.extCountOut };
    uint32_t slots[2];
    queue_reserve_each<2>(want, counters, slots);
    const uint32_t ns = slots[0], es = slots[1];
    queue_write(ns, o.neeFromOrg ? o.neeOrg : pos, o.neeDir, 0.0f, o.neeTmax, a.neeOrg, a.neeDir);
    if (o.wantNee) {
        a.neePending[ns] = make_float4(o.pending.x, o.pending.y, o.pending.z, bits2f(pixel));
        if (a.neeTrainIdx) a.neeTrainIdx[ns] = o.trainIdx;
    }
    queue_write(es, pos, o.extDir, 0.0f, 3.402823466e+38f, a.extOrgOut, a.extDirOut);
    if (o.wantExt) a.extOwnerOut[es] = pixel;
}

// pathTrace_rayGen_generic up to the path extension loop (optix_pathtracing_kernels.cu:74-160)
// One path in registers between two vertices: throughput, radiance so far, the density its last direction was sampled with, the
// pixel's RNG, the vertex position and what the vertex asked for (its NEE ray and its extension ray).
struct PtPath {
    f3 alpha, contribution; float dirPDensity;
    Pcg32 rng;
    f3 pos;
    PtVertexOut o;
};
GFX_DEV void reset_vertex_out(PtVertexOut& o) {
    o.wantNee = false; o.wantExt = false; o.neeDir = f3(0.0f); o.extDir = f3(0.0f); o.neeTmax = 0; o.pending = f3(0.0f);
    o.neeRet = f3(0.0f); o.localThroughput = f3(0.0f); o.trainIdx = 0x007FFFFFu; o.neeFromOrg = false; o.neeOrg = f3(0.0f);
}
// The first vertex from the G-buffer.  Returns whether the pixel shows a surface (its RNG state moved).  EVERY lane must call
// (ReGIR merges its cell-access atomics across the wave).
// A vertex ready to be shaded: what pt_first_setup / pt_next_setup leave for shade_vertex (the path's own state lives in PtPath).
struct PtSetup {
    f3 vOutLocal; Frame frame; Bsdf bsdf; bool shade;
    GFX_DEV PtSetup() : vOutLocal(0.0f), frame(f3(0, 0, 1), f3(1, 0, 0)), shade(false) {}
};
// pt_first_vertex up to the shading of the vertex (shade_vertex): the surface point of the pixel from the G-buffer, emission, BSDF.
template <bool REGIR>
GFX_DEV bool pt_first_setup(const PtArgs& a, const PixelId& px, PtPath& path, PtSetup& su) {
    const size_t p = px.p;
    const uint32_t bufIdx = a.f.bufferIndex;
    PtVertexOut& o = path.o;
    reset_vertex_out(o);
    f3& pos = path.pos;
    pos = f3(0.0f);
    f3& vOutLocal = su.vOutLocal;
    vOutLocal = f3(0.0f);
    Frame& frame = su.frame;
    frame = Frame(f3(0, 0, 1), f3(1, 0, 0));
    Bsdf& bsdf = su.bsdf;
    Pcg32& rng = path.rng; rng.state = 0;
    const EnvMap env = load_env(a.s);
    const bool envEnabled = env.present() && a.f.enableEnvLight;
    f3& contribution = path.contribution;
    f3& alpha = path.alpha;
    float& dirPDensity = path.dirPDensity;
    contribution = f3(0.001f, 0.001f, 0.001f);
    alpha = f3(1.0f);
    dirPDensity = 0.0f;
    bool surface = false;
    uint4 g0 = make_uint4(0xFFFFFFFFu, 0, 0, 0);
    if (px.valid) g0 = static_cast<const uint4*>(a.s.gbuffer0[bufIdx])[p];
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
        const float tu = bcA * vA.u + bcB * vB.u + bcC * vC.u, tv = bcA * vA.v + bcB * vB.v + bcC * vC.v;
        frame = Frame(ns, tc0);
        if (a.f.enableBumpMapping) apply_bump_mapping(read_modified_normal(a.scene, mat, tu, tv), frame);
        vOutLocal = frame.to_local(vOut);
        contribution = f3(0.0f);
        if (vOutLocal.z > 0 && mat.hasEmittance)
            contribution = contribution + alpha * material_emittance(a.scene, mat, tu, tv) / kPi;
        bsdf.setup(a.scene, mat, tu, tv);
    }
    else if (px.valid && envEnabled) {
        contribution = a.f.envLightPowerCoeff * env.fetch(decode_bc(g0.w & 0xFFFF), decode_bc(g0.w >> 16));
    }
    su.shade = surface;
    return surface;
}
template <bool REGIR>
GFX_DEV bool pt_first_vertex(const PtArgs& a, const PixelId& px, PtPath& path) {
    PtSetup su;
    const bool surface = pt_first_setup<REGIR>(a, px, path, su);
    const EnvMap env = load_env(a.s);
    const bool envEnabled = env.present() && a.f.enableEnvLight;
    shade_vertex<REGIR>(a, su.shade, env, envEnabled, path.pos, su.vOutLocal, su.frame, su.bsdf, path.rng, path.alpha, path.contribution, path.dirPDensity, path.o);
    return surface;
}
template <bool REGIR>
__global__ __launch_bounds__(kPtBlock) void k_pt_first(PtArgs a) {
    const PixelId px = pixel_of_thread(a.px);
    const size_t p = px.p;
    PtPath path;
    const bool surface = pt_first_vertex<REGIR>(a, px, path);
    if (surface) static_cast<uint64_t*>(a.s.rngBuffer)[p] = path.rng.state;
    if (px.valid) {
        a.state[2 * p] = make_float4(path.alpha.x, path.alpha.y, path.alpha.z, path.dirPDensity);
        a.state[2 * p + 1] = make_float4(path.contribution.x, path.contribution.y, path.contribution.z, 0.0f);
    }
    push_vertex(a, static_cast<uint32_t>(p), path.pos, path.o);
}

// The visibility factor of performDirectLighting<.., true> applied after the any-hit trace:
// contribution += alpha * NEE  (optix_pathtracing_kernels.cu:136-137, :279-280)
__global__ __launch_bounds__(kPtBlock) void k_pt_apply_nee(PtArgs a) {
    const uint32_t i = blockIdx.x * kPtBlock + threadIdx.x;
    if (i >= *a.neeCount) return;
    const float4 pend = a.neePending[i];
    const uint32_t pixel = f2bits(pend.w);
    f3 add(pend.x, pend.y, pend.z);
    if (a.occluded[i]) {
        add = add * 0.0f;                     // alpha * RGB(0): zero unless the throughput is not finite
        if (a.neeTrainIdx) {                  // NRC: the record's initial target is the SHADOWED estimate
            const uint32_t t = a.neeTrainIdx[i];
            if (t != 0x007FFFFFu) {
                float* tgt = static_cast<float*>(a.nrc.trainTargetBuffer[0]) + 3ull * t;
                tgt[0] = 0.0f; tgt[1] = 0.0f; tgt[2] = 0.0f;
            }
        }
    }
    float4* c = a.state + 2ull * pixel + 1;
    const float4 cur = *c;
    *c = make_float4(cur.x + add.x, cur.y + add.y, cur.z + add.z, 0.0f);
}

// closest-hit + miss programs of one path vertex, then the loop head of the ray-generation program
// (optix_pathtracing_kernels.cu:210-296, :306-341, :161-201; ReGIR: regir/.../optix_pathtracing_kernels.cu:302-392)
// What the extension ray of the previous vertex found (h; `active`: the lane holds such a ray, with path.alpha / contribution /
// dirPDensity as that vertex left them): implicit light or environment with its MIS weight, Russian roulette, the next vertex.
// rngBuf: where the pixel's RNG state is read when the ray hit a surface (the wavefront kernels), or null: path.rng holds it.
// Returns kPtHitSurface (the RNG state moved and the whole state changed) | kPtMissAdded (only the contribution changed).
// EVERY lane must call (ReGIR merges its cell-access atomics across the wave).
constexpr uint32_t kPtHitSurface = 1u, kPtMissAdded = 2u;
// pt_next_vertex up to the shading of the vertex: what the extension ray found (implicit light / environment with MIS), Russian roulette, the
// surface point and its BSDF when the path goes on (su.shade).
template <bool REGIR>
GFX_DEV uint32_t pt_next_setup(const PtArgs& a, bool active, const gfx_hit& h, f3 rayOrg, f3 rayDir, const uint64_t* rngBuf, PtPath& path, PtSetup& su,
                               int maxLengthTerminate = -1) {
    PtVertexOut& o = path.o;
    reset_vertex_out(o);
    f3& pos = path.pos;
    pos = f3(0.0f);
    f3& vOutLocal = su.vOutLocal;
    vOutLocal = f3(0.0f);
    Frame& frame = su.frame;
    frame = Frame(f3(0, 0, 1), f3(1, 0, 0));
    Bsdf& bsdf = su.bsdf;
    Pcg32& rng = path.rng;
    const EnvMap env = load_env(a.s);
    const bool envEnabled = env.present() && a.f.enableEnvLight;
    f3& alpha = path.alpha;
    f3& contribution = path.contribution;
    float& dirPDensity = path.dirPDensity;
    bool hitSurface = false, shade = false, missAdded = false;
    if (active) {
        const float prevDirPDensity = dirPDensity;
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
                missAdded = true;
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
                    if (a.f.useSolidAngleSampling) {   // path_tracing_shared.h:550-568, reference point = the ray origin
                        const SphericalTriangle st = spherical_triangle(pA, pB, pC, rayOrg);
                        const float dirPDF = 1.0f / st.sphArea;
                        f3 refDir = rayOrg - pos;
                        const float dist2ToRef = len2(refDir);
                        refDir = refDir / sqrtf(dist2ToRef);
                        const float lpCosRef = dot(refDir, ng);
                        hypAreaPDensity = (lpCosRef > 0 && is_finite(dirPDF)) ? lightProb * (dirPDF * lpCosRef / dist2ToRef) : 0.0f;
                    }
                    else hypAreaPDensity = lightProb / area;
                }
            }
            const gfx_material& mat = a.scene.materials[g.materialSlot];
            const f3 vOut = unit(-rayDir);
            const float frontHit = dot(vOut, ng) >= 0.0f ? 1.0f : -1.0f;
            const float tu = bcA * vA.u + bcB * vB.u + bcC * vC.u, tv = bcA * vA.v + bcB * vB.v + bcC * vC.v;
            frame = Frame(ns, tc0);
            if (a.f.enableBumpMapping) apply_bump_mapping(read_modified_normal(a.scene, mat, tu, tv), frame);
            pos = offset_ray_origin(pos, frontHit * ng);
            vOutLocal = frame.to_local(vOut);
            if (vOutLocal.z > 0 && mat.hasEmittance) {
                const f3 emittance = material_emittance(a.scene, mat, tu, tv);
                const float dist2 = len2(rayOrg - pos);
                const float lightPDensity = hypAreaPDensity * dist2 / vOutLocal.z;
                const float misWeight = (prevDirPDensity * prevDirPDensity) / (prevDirPDensity * prevDirPDensity + lightPDensity * lightPDensity);
                contribution = contribution + alpha * emittance * (misWeight / kPi);
            }
            if (rngBuf) rng.state = *rngBuf;
            // Russian roulette; initImportance = sRGB_calcLuminance(RGB(1))
            const float continueProb = fminf(luminance_srgb(alpha) / luminance_srgb(f3(1.0f)), 1.0f);
            if (!(rng.uniform() >= continueProb || (maxLengthTerminate < 0 ? a.maxLengthTerminate != 0u : maxLengthTerminate != 0))) {
                alpha = alpha / continueProb;
                bsdf.setup(a.scene, mat, tu, tv);
                shade = true;
            }
        }
    }
    su.shade = shade;
    return (hitSurface ? kPtHitSurface : 0u) | (missAdded ? kPtMissAdded : 0u);
}
template <bool REGIR>
GFX_DEV uint32_t pt_next_vertex(const PtArgs& a, bool active, const gfx_hit& h, f3 rayOrg, f3 rayDir, const uint64_t* rngBuf, PtPath& path,
                                int maxLengthTerminate = -1, int nextMaxLengthTerminate = -1 /* -1: the launch's (a.*) */) {
    PtSetup su;
    const uint32_t what = pt_next_setup<REGIR>(a, active, h, rayOrg, rayDir, rngBuf, path, su, maxLengthTerminate);
    const EnvMap env = load_env(a.s);
    const bool envEnabled = env.present() && a.f.enableEnvLight;
    shade_vertex<REGIR>(a, su.shade, env, envEnabled, path.pos, su.vOutLocal, su.frame, su.bsdf, path.rng, path.alpha, path.contribution, path.dirPDensity, path.o, nullptr, nextMaxLengthTerminate);
    return what;
}
template <bool REGIR>
__global__ __launch_bounds__(kPtBlock) void k_pt_bounce(PtArgs a) {
    const uint32_t i = blockIdx.x * kPtBlock + threadIdx.x;
    const uint32_t count = *a.extCountIn;
    PtPath path;
    path.alpha = f3(0.0f); path.contribution = f3(0.0f); path.dirPDensity = 0.0f; path.rng.state = 0;
    uint32_t pixel = 0;
    gfx_hit h; h.dist = 0.0f; h.bcB = 0.0f; h.bcC = 0.0f; h.triIndex = GFX_INVALID_SLOT;
    f3 rayOrg(0.0f), rayDir(0.0f);
    const bool active = i < count;
    if (active) {
        pixel = a.extOwnerIn[i];
        h = a.hits[i];
        const float4 ro4 = a.extOrgIn[i], rd4 = a.extDirIn[i];
        rayOrg = f3(ro4.x, ro4.y, ro4.z); rayDir = f3(rd4.x, rd4.y, rd4.z);
        const float4 s0 = a.state[2ull * pixel], s1 = a.state[2ull * pixel + 1];
        path.alpha = f3(s0.x, s0.y, s0.z);
        path.dirPDensity = s0.w;
        path.contribution = f3(s1.x, s1.y, s1.z);
    }
    const uint32_t what = pt_next_vertex<REGIR>(a, active, h, rayOrg, rayDir, static_cast<const uint64_t*>(a.s.rngBuffer) + pixel, path);
    if (what & kPtHitSurface) {
        static_cast<uint64_t*>(a.s.rngBuffer)[pixel] = path.rng.state;
        a.state[2ull * pixel] = make_float4(path.alpha.x, path.alpha.y, path.alpha.z, path.dirPDensity);
    }
    if (what & (kPtHitSurface | kPtMissAdded)) a.state[2ull * pixel + 1] = make_float4(path.contribution.x, path.contribution.y, path.contribution.z, 0.0f);
    push_vertex(a, pixel, path.pos, path.o);
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
    if (a.f.frameIndex - lastAccess > 8) return;      // block-uniform: a cell owns 512 = 2 x kPtBlock consecutive slots
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
                // prob is 0 or 1 for every candidate count >= 4: x / 1 and (x - 0) / (1 - 0) are x, no division needed
                if (prob == 1.0f) { probCurType = 0.25f; sampleEnv = true; }
                else if (prob == 0.0f) probCurType = 1.0f - 0.25f;
                else if (ul < prob) { probCurType = 0.25f; ul = ul / prob; sampleEnv = true; }
                else { probCurType = 1.0f - 0.25f; ul = (ul - prob) / (1 - prob); }
            }
            else sampleEnv = true;
        }
        LightSample ls;
        ls.emittance = f3(0.0f); ls.position = f3(0.0f); ls.normal = f3(0.0f); ls.atInfinity = 0;
        float pd;
        const float u0 = rng.uniform();
        const float u1 = rng.uniform();
        sample_light(a.scene, env, a.f.envLightRotation, a.f.envLightPowerCoeff, ul, sampleEnv, u0, u1, ls, pd);
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
    const PixelId px = pixel_of_thread(a.px);
    if (!px.valid) return;
    const size_t p = px.p;
    const float4 c = a.state[2 * p + 1];
    const f3 contribution(c.x, c.y, c.z);
    float4* beauty = static_cast<float4*>(a.s.beautyAccumBuffer) + p;
    f3 prev(0.0f);
    if (a.f.numAccumFrames > 0) { const float4 b = *beauty; prev = f3(b.x, b.y, b.z); }
    const float curWeight = 1.0f / (1 + a.f.numAccumFrames);
    const f3 result = (1 - curWeight) * prev + curWeight * contribution;
    *beauty = make_float4(result.x, result.y, result.z, 1.0f);
}

// The whole path of a pixel in ONE kernel, for small launches (trace_local.hip.h): first vertex, then per further vertex the NEE ray
// (any hit), its visibility applied, the extension ray (closest hit), the next vertex -- what k_pt_first, the two k_trace launches,
// k_pt_apply_nee, k_pt_bounce and k_pt_finish do through queues, here in the registers of the pixel's lane, in the same order per pixel
// (so the same sums).  A 512 x 512 frame is one round of waves; its wavefront form is ~20 launches of which most last as long as their
// slowest wave.  A lane whose path has ended idles until its wave's longest path has (no compaction: that is the price, and why large
// launches keep the wavefront form).
template <bool REGIR>
__global__ __launch_bounds__(kPtBlock) void k_pt_fused(PtArgs a, DevAccel accel, uint2* spill, int spillCap, uint32_t maxPathLength, uint32_t* __restrict__ blockCost,
                                                       unsigned long long* __restrict__ diag) {
    __shared__ uint2 ldsStack[kLdsStackDepth * kPtBlock];
    __shared__ __attribute__((aligned(16))) uint4 fetchBuf[(kPtBlock / 64) * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    uint4* waveBuf = fetchBuf + 256 * __builtin_amdgcn_readfirstlane(tid >> 6);
    uint2* stackLds = ldsStack + tid;
    uint2* stackSpill = spill + (static_cast<size_t>(blockIdx.x) * kPtBlock + tid) * spillCap;
    const PixelId px = pixel_of_thread(a.px);                   // (a.px.order: the blocks whose paths took most steps one frame ago start first)
    uint32_t steps = 0, totalSteps = 0;
    uint32_t diagIterations = 0, diagLanes = 0;                 // "pt_diag": bounce iterations of the wave, lanes that held a ray in them
    PtPath path;
    bool rngMoved = pt_first_vertex<REGIR>(a, px, path);       // a.pathLength = 1 (set by the host)
    for (uint32_t pathLength = 2;; ++pathLength) {
        if (diag) { ++diagIterations; diagLanes += static_cast<uint32_t>(__popcll(__ballot(path.o.wantNee || path.o.wantExt))); }
        const RayHit shadow = trace_wave_local<true>(accel, path.o.wantNee, path.o.neeFromOrg ? path.o.neeOrg : path.pos, path.o.neeDir, 0.0f, path.o.neeTmax,
                                                     stackLds, kPtBlock, stackSpill, spillCap, waveBuf, lane, 0xFFFFFFFFu, &steps);
        totalSteps += steps;
        if (path.o.wantNee) {                                   // k_pt_apply_nee
            f3 add = path.o.pending;
            if (shadow.tri != GFX_INVALID_SLOT) add = add * 0.0f;
            path.contribution = path.contribution + add;
        }
        if (REGIR && pathLength >= maxPathLength) break;
        if (__ballot(path.o.wantExt) == 0ull) break;            // no path of the wave goes on: nothing further can be added
        const bool extend = path.o.wantExt;
        const f3 rayOrg = path.pos, rayDir = path.o.extDir;
        const RayHit hit = trace_wave_local<false>(accel, extend, rayOrg, rayDir, 0.0f, 3.402823466e+38f, stackLds, kPtBlock, stackSpill, spillCap, waveBuf, lane, 0xFFFFFFFFu, &steps);
        totalSteps += steps;
        gfx_hit h; h.dist = hit.t; h.bcB = hit.bcB; h.bcC = hit.bcC; h.triIndex = hit.tri;
        const bool lastVertex = pathLength >= maxPathLength;     // what the host sets per bounce launch in the wavefront form
        if (pt_next_vertex<REGIR>(a, extend, h, rayOrg, rayDir, nullptr, path, lastVertex ? 1 : 0, pathLength + 1 >= maxPathLength ? 1 : 0) & kPtHitSurface) rngMoved = true;
        if (!REGIR && lastVertex) break;
    }
    if (blockCost && lane == 0) atomicMax(blockCost + launch_block(a.px), totalSteps >> 1);    // (<= 255 after the sort's clamp: 510 steps)
    if (diag && lane == 0) { atomicAdd(diag, diagIterations); atomicAdd(diag + 1, diagLanes); atomicAdd(diag + 2, totalSteps); atomicAdd(diag + 3, 1ull); }
    if (!px.valid) return;
    const size_t p = px.p;
    if (rngMoved) static_cast<uint64_t*>(a.s.rngBuffer)[p] = path.rng.state;
    float4* beauty = static_cast<float4*>(a.s.beautyAccumBuffer) + p;          // k_pt_finish
    f3 prev(0.0f);
    if (a.f.numAccumFrames > 0) { const float4 bb = *beauty; prev = f3(bb.x, bb.y, bb.z); }
    const float curWeight = 1.0f / (1 + a.f.numAccumFrames);
    const f3 result = (1 - curWeight) * prev + curWeight * path.contribution;
    *beauty = make_float4(result.x, result.y, result.z, 1.0f);
}

// k_pt_fused with path regeneration.  In k_pt_fused a lane whose path has ended idles until the longest path of its wave has: 0.30
// lanes per instruction on the 512 x 512 bunny frame (profiles/r05_pmc_config1.json: most pixels miss or end after one or two vertices,
// a few run to the length limit).  Here the launch is the waves the GPU holds at once, and a lane whose path has ended writes its pixel
// and draws the next launch slot from a ticket counter (one wave-aggregated atomic per refill), takes that pixel's first vertex from
// the G-buffer and joins the wave's next trace: the wave keeps full lanes until the ticket runs out.  A pixel's path is the same
// sequence of operations on the same RNG stream whichever lane runs it and whenever: the frame is bit-identical to k_pt_fused's.
// Baseline path tracer only (the ReGIR tracer merges its cell-access atomics across lanes that are at the same vertex).
__global__ __launch_bounds__(kPtBlock) void k_pt_regen(PtArgs a, DevAccel accel, uint2* spill, int spillCap, uint32_t maxPathLength,
                                                       uint32_t* __restrict__ ticket, uint32_t numSlots, int minRefill, unsigned long long* __restrict__ diag) {
    __shared__ uint2 ldsStack[kLdsStackDepth * kPtBlock];
    __shared__ __attribute__((aligned(16))) uint4 fetchBuf[(kPtBlock / 64) * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    uint4* waveBuf = fetchBuf + 256 * __builtin_amdgcn_readfirstlane(tid >> 6);
    uint2* stackLds = ldsStack + tid;
    uint2* stackSpill = spill + (static_cast<size_t>(blockIdx.x) * kPtBlock + tid) * spillCap;
    PixelId px; px.p = 0; px.x = 0; px.y = 0; px.slot = 0; px.valid = false;
    PtPath path;
    reset_vertex_out(path.o);
    path.alpha = f3(0.0f); path.contribution = f3(0.0f); path.dirPDensity = 0.0f; path.rng.state = 0; path.pos = f3(0.0f);
    bool active = false, rngMoved = false, exhausted = false;
    uint32_t pathLength = 2;
    uint32_t diagIterations = 0, diagLanes = 0, diagSteps = 0, diagRefills = 0, steps = 0;
    auto finish = [&]() {                                       // the pixel's RNG state and its running mean (k_pt_finish)
        if (px.valid) {
            const size_t p = px.p;
            if (rngMoved) static_cast<uint64_t*>(a.s.rngBuffer)[p] = path.rng.state;
            float4* beauty = static_cast<float4*>(a.s.beautyAccumBuffer) + p;
            f3 prev(0.0f);
            if (a.f.numAccumFrames > 0) { const float4 bb = *beauty; prev = f3(bb.x, bb.y, bb.z); }
            const float curWeight = 1.0f / (1 + a.f.numAccumFrames);
            const f3 result = (1 - curWeight) * prev + curWeight * path.contribution;
            *beauty = make_float4(result.x, result.y, result.z, 1.0f);
        }
        active = false;
    };
    const EnvMap env = load_env(a.s);
    const bool envEnabled = env.present() && a.f.enableEnvLight;
    for (;;) {
        // ---- the rays of the vertices the lanes stand on: NEE (any hit), its visibility applied, then the extension (closest hit)
        PtSetup su;                                             // the vertex a lane shades at the end of this iteration
        if (__ballot(active) != 0ull) {
            if (diag) { ++diagIterations; diagLanes += static_cast<uint32_t>(__popcll(__ballot(active))); }
            const bool nee = active && path.o.wantNee;
            const RayHit shadow = trace_wave_local<true>(accel, nee, path.o.neeFromOrg ? path.o.neeOrg : path.pos, path.o.neeDir, 0.0f, path.o.neeTmax,
                                                         stackLds, kPtBlock, stackSpill, spillCap, waveBuf, lane, 0xFFFFFFFFu, &steps);
            diagSteps += steps;
            if (nee) {                                          // k_pt_apply_nee
                f3 add = path.o.pending;
                if (shadow.tri != GFX_INVALID_SLOT) add = add * 0.0f;
                path.contribution = path.contribution + add;
            }
            bool done = active && !path.o.wantExt;
            const bool extend = active && path.o.wantExt;
            if (__ballot(extend) != 0ull) {
                const f3 rayOrg = path.pos, rayDir = path.o.extDir;
                const RayHit hit = trace_wave_local<false>(accel, extend, rayOrg, rayDir, 0.0f, 3.402823466e+38f, stackLds, kPtBlock, stackSpill, spillCap, waveBuf, lane, 0xFFFFFFFFu, &steps);
                diagSteps += steps;
                if (extend) {                                   // what the ray found: implicit light with MIS, Russian roulette, the next surface point
                    gfx_hit h; h.dist = hit.t; h.bcB = hit.bcB; h.bcC = hit.bcC; h.triIndex = hit.tri;
                    const bool lastVertex = pathLength >= maxPathLength;
                    if (pt_next_setup<false>(a, true, h, rayOrg, rayDir, nullptr, path, su, lastVertex ? 1 : 0) & kPtHitSurface) rngMoved = true;
                    if (lastVertex || !su.shade) done = true;   // (a path that is not shaded asks for no ray: it has ended)
                    ++pathLength;
                }
            }
            if (done) finish();
        }
        // ---- refill: when at least `minRefill` lanes are idle (or all of them), the idle lanes draw launch slots -- one wave-aggregated
        // atomic -- and set their pixel's first vertex up from the G-buffer.  A pixel without a surface (background) is finished on the spot
        // and its lane draws again, up to four times per refill.
#pragma nounroll
        for (int round = 0; round < 4 && !exhausted; ++round) {
            const unsigned long long need = __ballot(!active);
            const int idle = __popcll(need);
            if (idle == 0 || (idle < minRefill && idle < 64 && round == 0)) break;
            const uint32_t count = static_cast<uint32_t>(idle);
            ++diagRefills;
            uint32_t base = 0;
            if (lane == __builtin_ctzll(need)) base = atomicAdd(ticket, count);
            base = __shfl(base, __builtin_ctzll(need));
            if (!active) {
                const uint32_t slot = base + static_cast<uint32_t>(__popcll(need & ((1ull << lane) - 1ull)));
                if (slot < numSlots) {
                    px = pixel_of_block_thread(a.px, slot >> 8, slot & 255u);      // (kPtBlock = 256 slots per launch block)
                    rngMoved = pt_first_setup<false>(a, px, path, su);
                    pathLength = 2;
                    active = true;
                    if (!su.shade) finish();                    // nothing to shade: no ray will be asked for
                }
            }
            exhausted = base + count >= numSlots;
        }
        if (__ballot(active) == 0ull) { if (exhausted) break; else continue; }
        // ---- ONE shading step for the wave: the vertices the extension rays reached and the first vertices of the pixels just drawn
        // (light sample + MIS + BSDF sample: the code a wave executes once per iteration whichever lanes take part)
        shade_vertex<false>(a, active && su.shade, env, envEnabled, path.pos, su.vOutLocal, su.frame, su.bsdf, path.rng, path.alpha, path.contribution, path.dirPDensity, path.o);
    }
    if (diag && lane == 0) { atomicAdd(diag, diagIterations); atomicAdd(diag + 1, diagLanes); atomicAdd(diag + 2, diagSteps); atomicAdd(diag + 3, 1ull); atomicAdd(diag + 4, diagRefills); }
}

// ---------------------------------------------------------------- neural radiance caching (render side)
// neural_radiance_caching/gpu_kernels/optix_pathtracing_kernels.cu + nrc_setup_kernels.cu on the
// same wavefront pipeline.  Training-record indices come from one wave-aggregated atomicAdd per
// kernel wave (the reference's per-thread atomicAdd order is unspecified as well); the initial
// target of a record is the NEE estimate of its vertex, so the apply kernel zeroes it when the
// shadow ray turns out occluded.
constexpr uint32_t kInvalidVertexDataIndex = 0x007FFFFFu;     // neural_radiance_caching_shared.h:146
constexpr uint32_t kNumTrainingDataPerFrame = 1u << 16, kTrainBufferSize = 2u << 16;   // :8-9
constexpr uint32_t kNrcFlagRenderEnds = 1u, kNrcFlagSuffixEnds = 2u;

GFX_DEV uint32_t nrc_terminal_bits(bool hasQuery, uint32_t pathLength, bool training, bool unbiased) {
    return (hasQuery ? 1u : 0u) | ((pathLength & 0xFFu) << 1) | ((training ? 1u : 0u) << 9) | ((unbiased ? 1u : 0u) << 10);
}
GFX_DEV uint32_t nrc_suffix_bits(uint32_t prevIdx, bool hasQuery, uint32_t pathLength) {
    return (prevIdx & 0x7FFFFFu) | ((hasQuery ? 1u : 0u) << 23) | ((pathLength & 0xFFu) << 24);
}
struct NrcTile { uint32_t linearTileIndex; bool training, unbiased; };
GFX_DEV NrcTile nrc_tile_of(const PtArgs& a, uint32_t pixel) {   // optix_pathtracing_kernels.cu:107-131
    const uint32_t W = static_cast<uint32_t>(a.s.imageSizeX);
    const uint32_t x = pixel % W, y = pixel / W;
    const uint2 ts = *static_cast<const uint2*>(a.nrc.tileSize[a.f.bufferIndex]);
    const uint32_t local = (y % ts.y) * ts.x + (x % ts.x);
    NrcTile t;
    t.training = (local + *static_cast<const uint32_t*>(a.nrc.offsetToSelectTrainingPath)) % (ts.x * ts.y) == 0;
    const uint32_t numTilesX = (W + ts.x - 1) / ts.x;
    const uint32_t tx = x / ts.x, ty = y / ts.y;
    t.linearTileIndex = ty * numTilesX + tx;
    t.unbiased = ((ty % 4) * 4 + (tx % 4) + *static_cast<const uint32_t*>(a.nrc.offsetToSelectUnbiasedTile)) % 16 == 0;
    return t;
}
struct NrcQuery { float v[14]; };
GFX_DEV NrcQuery nrc_make_query(const PtArgs& a, f3 pos, f3 normal, f3 vOut, const Bsdf& bsdf) {   // :12-32
    NrcQuery q;
    const float* lo = a.nrc.sceneAabbMin; const float* hi = a.nrc.sceneAabbMax;
    const float p[3] = { pos.x, pos.y, pos.z };
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float den = hi[k] - lo[k]; q.v[k] = den != 0 ? (p[k] - lo[k]) / den : 0.0f; }
    q.v[4] = gm_acos(fmin2(fmax2(normal.z, -1.0f), 1.0f)); q.v[3] = gm_atan2(normal.y, normal.x);
    q.v[6] = gm_acos(fmin2(fmax2(vOut.z, -1.0f), 1.0f)); q.v[5] = gm_atan2(vOut.y, vOut.x);
    q.v[7] = 1 - gm_exp(-bsdf.roughness);
    q.v[8] = bsdf.diffuse.x; q.v[9] = bsdf.diffuse.y; q.v[10] = bsdf.diffuse.z;
    q.v[11] = bsdf.specularF0.x; q.v[12] = bsdf.specularF0.y; q.v[13] = bsdf.specularF0.z;
    return q;
}
GFX_DEV void nrc_store_query(void* buf, size_t idx, const NrcQuery& q) {
    float2* dst = reinterpret_cast<float2*>(static_cast<float*>(buf) + 14 * idx);
#pragma unroll
    for (int k = 0; k < 7; ++k) dst[k] = make_float2(q.v[2 * k], q.v[2 * k + 1]);
}
// trainDataIndex = atomicAdd(numTrainingData, 1) for the lanes with want, one atomic per wave
GFX_DEV uint32_t nrc_alloc_train_index(uint32_t* counter, bool want) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return kInvalidVertexDataIndex;
    const int lane = threadIdx.x & 63;
    const int leader = __builtin_ctzll(mask);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, static_cast<uint32_t>(__popcll(mask)));
    base = __shfl(base, leader);
    return base + __popcll(mask & ((1ull << lane) - 1ull));
}
GFX_DEV void nrc_write_train_vertex(const PtArgs& a, uint32_t idx, const NrcQuery& q, f3 localThroughput, uint32_t prevIdx,
                                    uint32_t pathLength, f3 target) {
    nrc_store_query(a.nrc.trainRadianceQueryBuffer[0], idx, q);
    static_cast<float4*>(a.nrc.trainVertexInfoBuffer)[idx] =
        make_float4(localThroughput.x, localThroughput.y, localThroughput.z, bits2f((prevIdx & 0x7FFFFFu) | ((pathLength & 0xFFu) << 23)));
    float* t = static_cast<float*>(a.nrc.trainTargetBuffer[0]) + 3ull * idx;
    t[0] = target.x; t[1] = target.y; t[2] = target.z;
}
// tail of the ray-generation program for a path that stops here (:339-373)
GFX_DEV void nrc_end_path(const PtArgs& a, uint32_t pixel, const NrcTile& tile, uint32_t flags, uint32_t prevTrainIdx, uint32_t pathLength) {
    if (tile.training && !(flags & kNrcFlagSuffixEnds))
        static_cast<uint32_t*>(a.nrc.trainSuffixTerminalInfoBuffer)[tile.linearTileIndex] = nrc_suffix_bits(prevTrainIdx, false, pathLength);
    if (!(flags & kNrcFlagRenderEnds))
        static_cast<float4*>(a.nrc.inferenceTerminalInfoBuffer)[pixel] =
            make_float4(0.0f, 0.0f, 0.0f, bits2f(nrc_terminal_bits(false, pathLength, tile.training, tile.unbiased)));
}

// preprocessNRC, nrc_setup_kernels.cu:6-49
__global__ __launch_bounds__(kPtBlock) void k_nrc_preprocess(PtArgs a) {
    const uint32_t i = blockIdx.x * kPtBlock + threadIdx.x;
    if (i >= a.nrc.maxNumTrainingSuffixes) return;
    const uint32_t bufIdx = a.f.bufferIndex, prevBufIdx = (a.f.bufferIndex + 1) % 2;
    if (i == 0) {
        uint32_t nx = 8, ny = 8;
        if (!a.nrc.isNewSequence) {
            const uint32_t prevNum = *static_cast<const uint32_t*>(a.nrc.numTrainingData[prevBufIdx]);
            const float r = sqrtf(static_cast<float>(prevNum) / kNumTrainingDataPerFrame);
            const uint2 cur = *static_cast<const uint2*>(a.nrc.tileSize[prevBufIdx]);
            nx = f2u_sat(cur.x * r); ny = f2u_sat(cur.y * r);
            nx = nx < 4u ? 4u : (nx > 128u ? 128u : nx);
            ny = ny < 4u ? 4u : (ny > 128u ? 128u : ny);
        }
        *static_cast<uint2*>(a.nrc.tileSize[bufIdx]) = make_uint2(nx, ny);
        *static_cast<uint32_t*>(a.nrc.numTrainingData[bufIdx]) = 0;
        *static_cast<uint32_t*>(a.nrc.offsetToSelectUnbiasedTile) = a.nrc.preprocessOffsetToSelectUnbiasedTile;
        *static_cast<uint32_t*>(a.nrc.offsetToSelectTrainingPath) = a.nrc.preprocessOffsetToSelectTrainingPath;
        int32_t* mm = static_cast<int32_t*>(a.nrc.targetMinMax[bufIdx]);
        mm[0] = mm[1] = mm[2] = 0x7F800000;                         // floatToOrderedInt(+inf)
        mm[3] = mm[4] = mm[5] = static_cast<int32_t>(0xFF800000u) ^ 0x7FFFFFFF;   // floatToOrderedInt(-inf)
        float* avg = static_cast<float*>(a.nrc.targetAvg[bufIdx]);
        avg[0] = avg[1] = avg[2] = 0.0f;
    }
    static_cast<uint32_t*>(a.nrc.trainSuffixTerminalInfoBuffer)[i] = nrc_suffix_bits(kInvalidVertexDataIndex, false, 0);
}

// pathTrace_raygen_generic<true> up to the path extension loop (:133-318).  REGIR: next-event estimation from the ReGIR
// grid (GFX_PT_PATH_TRACE_NRC_REGIR, include/gfxexp.h).
// NEE: 0 the tracer's own light sample, 1 ReGIR cell sampling, 2 the pixel's ReSTIR DI reservoir (first vertex only).
template <int NEE>
__global__ __launch_bounds__(kPtBlock) void k_nrc_pt_first(PtArgs a) {
    constexpr bool REGIR = NEE == 1;
    const PixelId px = pixel_of_thread(a.px);
    const size_t p = px.p;
    const uint32_t bufIdx = a.f.bufferIndex;
    PtVertexOut o;
    o.wantNee = false; o.wantExt = false; o.neeDir = f3(0.0f); o.extDir = f3(0.0f); o.neeTmax = 0; o.pending = f3(0.0f);
    o.neeRet = f3(0.0f); o.localThroughput = f3(0.0f); o.trainIdx = kInvalidVertexDataIndex; o.neeFromOrg = false; o.neeOrg = f3(0.0f);
    f3 pos(0.0f), vOutLocal(0.0f), vOut(0.0f);
    Frame frame(f3(0, 0, 1), f3(1, 0, 0));
    Bsdf bsdf; bsdf.type = 0; bsdf.diffuse = f3(0.0f); bsdf.specularF0 = f3(0.0f); bsdf.roughness = 1.0f;
    Pcg32 rng; rng.state = 0;
    const EnvMap env = load_env(a.s);
    const bool envEnabled = env.present() && a.f.enableEnvLight;
    f3 contribution(0.001f, 0.001f, 0.001f);
    f3 alpha(1.0f);
    float dirPDensity = 0.0f, primaryPathSpread = 0.0f;
    const bool inImage = px.valid;
    uint4 g0 = make_uint4(0xFFFFFFFFu, 0, 0, 0);
    if (inImage) g0 = static_cast<const uint4*>(a.s.gbuffer0[bufIdx])[p];
    const bool surface = g0.x != 0xFFFFFFFFu;
    NrcTile tile; tile.linearTileIndex = 0; tile.training = false; tile.unbiased = false;
    if (inImage) tile = nrc_tile_of(a, static_cast<uint32_t>(p));
    if (surface) {
        const float bcB = decode_bc(g0.w & 0xFFFF), bcC = decode_bc(g0.w >> 16);
        const DevInstance* inst = a.scene.insts + g0.x;
        const DevGeomInst g = a.scene.geomInsts[g0.y];
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
        vOut = cam.pos - pos;
        const float primaryDist2 = len2(vOut);
        vOut = vOut / sqrtf(primaryDist2);
        const float primaryDotVN = dot(vOut, ng);
        const float frontHit = primaryDotVN >= 0.0f ? 1.0f : -1.0f;
        pos = offset_ray_origin(pos, frontHit * ng);
        primaryPathSpread = primaryDist2 / (4 * kPi * fabsf(primaryDotVN));
        const float tu = bcA * vA.u + bcB * vB.u + bcC * vC.u, tv = bcA * vA.v + bcB * vB.v + bcC * vC.v;
        frame = Frame(ns, tc0);
        if (a.f.enableBumpMapping) apply_bump_mapping(read_modified_normal(a.scene, mat, tu, tv), frame);
        vOutLocal = frame.to_local(vOut);
        contribution = f3(0.0f);
        if (vOutLocal.z > 0 && mat.hasEmittance)
            contribution = contribution + alpha * material_emittance(a.scene, mat, tu, tv) / kPi;
        bsdf.setup(a.scene, mat, tu, tv);
    }
    else if (inImage && envEnabled) {
        contribution = a.f.envLightPowerCoeff * env.fetch(decode_bc(g0.w & 0xFFFF), decode_bc(g0.w >> 16));
    }
    if (NEE == 2) {
        ReservoirNee given;
        given.ret = f3(0.0f); given.org = f3(0.0f);
        given.ls.emittance = f3(0.0f); given.ls.position = f3(0.0f); given.ls.normal = f3(0.0f); given.ls.atInfinity = 0;
        if (surface) given = restir_reservoir_nee(a, bufIdx, p);
        shade_vertex<false, false>(a, surface, env, envEnabled, pos, vOutLocal, frame, bsdf, rng, alpha, contribution, dirPDensity, o, &given);
    }
    else shade_vertex<REGIR, false>(a, surface, env, envEnabled, pos, vOutLocal, frame, bsdf, rng, alpha, contribution, dirPDensity, o);
    // training record of the first vertex (:238-277)
    uint32_t trainIdx = nrc_alloc_train_index(static_cast<uint32_t*>(a.nrc.numTrainingData[bufIdx]), surface && tile.training);
    if (surface && tile.training) {
        if (trainIdx < kTrainBufferSize)
            nrc_write_train_vertex(a, trainIdx, nrc_make_query(a, pos, frame.n, vOut, bsdf), o.localThroughput, kInvalidVertexDataIndex, 1, o.neeRet);
        else trainIdx = kInvalidVertexDataIndex;
    }
    else trainIdx = kInvalidVertexDataIndex;   // (never read for non-training paths)
    o.trainIdx = trainIdx;
    if (surface) static_cast<uint64_t*>(a.s.rngBuffer)[p] = rng.state;
    if (inImage) {
        a.state[2 * p] = make_float4(alpha.x, alpha.y, alpha.z, dirPDensity);
        a.state[2 * p + 1] = make_float4(contribution.x, contribution.y, contribution.z, 0.0f);
        a.nrcState[2 * p] = make_float4(o.localThroughput.x, o.localThroughput.y, o.localThroughput.z, primaryPathSpread);
        a.nrcState[2 * p + 1] = make_float4(0.0f, bits2f(trainIdx), bits2f(0u), 0.0f);
        if (!o.wantExt) {
            NrcTile t = tile;
            if (!surface) t.training = false;      // the suffix terminal is only written inside the surface branch (:344-350)
            nrc_end_path(a, static_cast<uint32_t>(p), t, 0u, trainIdx, 1);
            if (!surface)   // ... but the terminal info of a background pixel still records the tile flags (:365-372)
                static_cast<float4*>(a.nrc.inferenceTerminalInfoBuffer)[p] =
                    make_float4(0.0f, 0.0f, 0.0f, bits2f(nrc_terminal_bits(false, 1, tile.training, tile.unbiased)));
        }
    }
    push_vertex(a, static_cast<uint32_t>(p), pos, o);
}

// pathTrace_closestHit_generic<true> / pathTrace_miss_generic<true> for one extension ray, then the
// loop head and, when the path stops, the tail of the ray-generation program (:380-677, :320-373).  REGIR: emitters
// found by BSDF sampling contribute nothing (the ReGIR NEE has no density to weight them against).
template <int NEE>
__global__ __launch_bounds__(kPtBlock) void k_nrc_pt_bounce(PtArgs a) {
    constexpr bool REGIR = NEE == 1;
    // NEE == 2: the first vertex's direct light came from the ReSTIR reservoir, which has no density to weight a BSDF-sampled emitter
    // against: what the first extension ray finds emitting counts nothing (GFX_PT_PATH_TRACE_NRC_RESTIR, include/gfxexp.h)
    const bool noImplicit = REGIR || (NEE == 2 && a.pathLength == 2);
    const uint32_t i = blockIdx.x * kPtBlock + threadIdx.x;
    const uint32_t count = *a.extCountIn;
    const uint32_t bufIdx = a.f.bufferIndex;
    const size_t numPixels = static_cast<size_t>(a.s.imageSizeX) * a.s.imageSizeY;
    PtVertexOut o;
    o.wantNee = false; o.wantExt = false; o.neeDir = f3(0.0f); o.extDir = f3(0.0f); o.neeTmax = 0; o.pending = f3(0.0f);
    o.neeRet = f3(0.0f); o.localThroughput = f3(0.0f); o.trainIdx = kInvalidVertexDataIndex; o.neeFromOrg = false; o.neeOrg = f3(0.0f);
    f3 pos(0.0f), vOutLocal(0.0f), vOut(0.0f);
    Frame frame(f3(0, 0, 1), f3(1, 0, 0));
    Bsdf bsdf; bsdf.type = 0; bsdf.diffuse = f3(0.0f); bsdf.specularF0 = f3(0.0f); bsdf.roughness = 1.0f;
    Pcg32 rng; rng.state = 0;
    const EnvMap env = load_env(a.s);
    const bool envEnabled = env.present() && a.f.enableEnvLight;
    f3 alpha(0.0f), contribution(0.0f), prevLocalThroughput(0.0f);
    float dirPDensity = 0.0f, primaryPathSpread = 0.0f, curSqrtPathSpread = 0.0f;
    uint32_t pixel = 0, prevTrainIdx = kInvalidVertexDataIndex, flags = 0;
    NrcTile tile; tile.linearTileIndex = 0; tile.training = false; tile.unbiased = false;
    const uint32_t pathLength = a.pathLength;
    bool active = i < count, hitSurface = false, shade = false;
    if (active) {
        pixel = a.extOwnerIn[i];
        tile = nrc_tile_of(a, pixel);
        const gfx_hit h = a.hits[i];
        const float4 ro4 = a.extOrgIn[i], rd4 = a.extDirIn[i];
        const f3 rayOrg(ro4.x, ro4.y, ro4.z), rayDir(rd4.x, rd4.y, rd4.z);
        const float4 s0 = a.state[2ull * pixel], s1 = a.state[2ull * pixel + 1];
        const float4 n0 = a.nrcState[2ull * pixel], n1 = a.nrcState[2ull * pixel + 1];
        alpha = f3(s0.x, s0.y, s0.z);
        const float prevDirPDensity = s0.w;
        dirPDensity = prevDirPDensity;
        contribution = f3(s1.x, s1.y, s1.z);
        prevLocalThroughput = f3(n0.x, n0.y, n0.z); primaryPathSpread = n0.w;
        curSqrtPathSpread = n1.x; prevTrainIdx = f2bits(n1.y); flags = f2bits(n1.z);
        float* prevTarget = static_cast<float*>(a.nrc.trainTargetBuffer[0]) + 3ull * prevTrainIdx;
        const bool linkPrev = tile.training && prevTrainIdx != kInvalidVertexDataIndex;
        if (h.triIndex == GFX_INVALID_SLOT) {
            if (envEnabled && !noImplicit) {
                const f3 rd = unit(rayDir);
                float posPhi, theta;
                to_polar_yup(rd, posPhi, theta);
                float phi = posPhi + a.f.envLightRotation;
                phi = phi - floorf(phi / (2 * kPi)) * 2 * kPi;
                const float tu = phi / (2 * kPi), tv = theta / kPi;
                const f3 luminance = a.f.envLightPowerCoeff * env.fetch(tu, tv);
                const float uvPDF = env.evaluate_pdf(tu, tv);
                const float hypAreaPDensity = uvPDF / (2 * kPi * kPi * gm_sin(theta));
                const float lightPDensity = 0.25f * hypAreaPDensity;
                const float misWeight = (prevDirPDensity * prevDirPDensity) / (prevDirPDensity * prevDirPDensity + lightPDensity * lightPDensity);
                const f3 implicit = misWeight * luminance;
                contribution = contribution + alpha * implicit;
                if (linkPrev) {
                    const f3 add = prevLocalThroughput * implicit;
                    prevTarget[0] += add.x; prevTarget[1] += add.y; prevTarget[2] += add.z;
                }
            }
        }
        else {
            hitSurface = true;
            const Bvh8Tri* tr = a.tris + h.triIndex;
            const uint32_t instSlot = tr->instSlot, geomInstSlot = tr->geomInstSlot, primIndex = tr->primIndex;
            const DevInstance* inst = a.scene.insts + instSlot;
            const DevGeomInst g = a.scene.geomInsts[geomInstSlot];
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
                    if (a.f.useSolidAngleSampling) {   // path_tracing_shared.h:550-568, reference point = the ray origin
                        const SphericalTriangle st = spherical_triangle(pA, pB, pC, rayOrg);
                        const float dirPDF = 1.0f / st.sphArea;
                        f3 refDir = rayOrg - pos;
                        const float dist2ToRef = len2(refDir);
                        refDir = refDir / sqrtf(dist2ToRef);
                        const float lpCosRef = dot(refDir, ng);
                        hypAreaPDensity = (lpCosRef > 0 && is_finite(dirPDF)) ? lightProb * (dirPDF * lpCosRef / dist2ToRef) : 0.0f;
                    }
                    else hypAreaPDensity = lightProb / area;
                }
            }
            const gfx_material& mat = a.scene.materials[g.materialSlot];
            vOut = unit(-rayDir);
            const float frontHit = dot(vOut, ng) >= 0.0f ? 1.0f : -1.0f;
            const float tu = bcA * vA.u + bcB * vB.u + bcC * vC.u, tv = bcA * vA.v + bcB * vB.v + bcC * vC.v;
            frame = Frame(ns, tc0);
            if (a.f.enableBumpMapping) apply_bump_mapping(read_modified_normal(a.scene, mat, tu, tv), frame);
            pos = offset_ray_origin(pos, frontHit * ng);
            vOutLocal = frame.to_local(vOut);
            const float dist2 = len2(rayOrg - pos);
            curSqrtPathSpread += sqrtf(dist2 / (prevDirPDensity * fabsf(vOutLocal.z)));
            if (!noImplicit && vOutLocal.z > 0 && mat.hasEmittance) {
                const f3 emittance = material_emittance(a.scene, mat, tu, tv);
                const float lightPDensity = hypAreaPDensity * dist2 / vOutLocal.z;
                const float misWeight = (prevDirPDensity * prevDirPDensity) / (prevDirPDensity * prevDirPDensity + lightPDensity * lightPDensity);
                const f3 implicit = emittance * (misWeight / kPi);
                contribution = contribution + alpha * implicit;
                if (linkPrev) {
                    const f3 add = prevLocalThroughput * implicit;
                    prevTarget[0] += add.x; prevTarget[1] += add.y; prevTarget[2] += add.z;
                }
            }
            rng.state = static_cast<const uint64_t*>(a.s.rngBuffer)[pixel];
            // Russian roulette (:455-474)
            bool performRR = true, terminatedByRR = false, stop = false;
            float recContinueProb = 1.0f;
            if (tile.training) performRR = pathLength > 2;
            const bool unbiasedSuffix = (flags & kNrcFlagRenderEnds) && tile.training && tile.unbiased;
            if (performRR) {
                const float continueProb = fminf(luminance_srgb(alpha) / luminance_srgb(f3(1.0f)), 1.0f);
                if (rng.uniform() >= continueProb || a.maxLengthTerminate) {
                    if (unbiasedSuffix) stop = true;
                    terminatedByRR = true;
                }
                recContinueProb = 1.0f / continueProb;
            }
            bsdf.setup(a.scene, mat, tu, tv);
            if (!stop) {   // cache termination heuristic (:479-538)
                bool endsWithCache = curSqrtPathSpread * curSqrtPathSpread > 0.01f * primaryPathSpread;
                if (unbiasedSuffix) endsWithCache = false;
                if (endsWithCache) {
                    const NrcQuery q = nrc_make_query(a, pos, frame.n, vOut, bsdf);
                    if (!(flags & kNrcFlagRenderEnds)) {
                        nrc_store_query(a.nrc.inferenceRadianceQueryBuffer, pixel, q);
                        static_cast<float4*>(a.nrc.inferenceTerminalInfoBuffer)[pixel] =
                            make_float4(alpha.x, alpha.y, alpha.z, bits2f(nrc_terminal_bits(true, pathLength, tile.training, tile.unbiased)));
                        flags |= kNrcFlagRenderEnds;
                        if (tile.training) curSqrtPathSpread = 0;
                        else stop = true;
                    }
                    else {
                        if (!(flags & kNrcFlagSuffixEnds)) {
                            nrc_store_query(a.nrc.inferenceRadianceQueryBuffer, numPixels + tile.linearTileIndex, q);
                            static_cast<uint32_t*>(a.nrc.trainSuffixTerminalInfoBuffer)[tile.linearTileIndex] = nrc_suffix_bits(prevTrainIdx, true, pathLength);
                            flags |= kNrcFlagSuffixEnds;
                        }
                        stop = true;
                    }
                }
            }
            if (!stop && terminatedByRR) stop = true;
            if (!stop) {
                alpha = alpha * recContinueProb;
                if (linkPrev) {
                    float4* vi = static_cast<float4*>(a.nrc.trainVertexInfoBuffer) + prevTrainIdx;
                    float4 v = *vi;
                    v.x *= recContinueProb; v.y *= recContinueProb; v.z *= recContinueProb;
                    *vi = v;
                }
                shade = true;
            }
        }
    }
    shade_vertex<REGIR, false>(a, shade, env, envEnabled, pos, vOutLocal, frame, bsdf, rng, alpha, contribution, dirPDensity, o);
    // training record of this vertex (:574-631)
    const bool wantRecord = shade && tile.training && !(flags & kNrcFlagSuffixEnds);
    uint32_t trainIdx = nrc_alloc_train_index(static_cast<uint32_t*>(a.nrc.numTrainingData[bufIdx]), wantRecord);
    if (wantRecord) {
        const NrcQuery q = nrc_make_query(a, pos, frame.n, vOut, bsdf);
        if (trainIdx < kTrainBufferSize) {
            nrc_write_train_vertex(a, trainIdx, q, o.localThroughput, prevTrainIdx, pathLength, o.neeRet);
            prevTrainIdx = trainIdx;
            o.trainIdx = trainIdx;
        }
        else {
            nrc_store_query(a.nrc.inferenceRadianceQueryBuffer, numPixels + tile.linearTileIndex, q);
            static_cast<uint32_t*>(a.nrc.trainSuffixTerminalInfoBuffer)[tile.linearTileIndex] = nrc_suffix_bits(prevTrainIdx, true, pathLength);
            flags |= kNrcFlagSuffixEnds;
        }
    }
    if (shade) prevLocalThroughput = o.localThroughput;
    if (active) {
        if (hitSurface) {
            static_cast<uint64_t*>(a.s.rngBuffer)[pixel] = rng.state;
            a.state[2ull * pixel] = make_float4(alpha.x, alpha.y, alpha.z, dirPDensity);
        }
        a.state[2ull * pixel + 1] = make_float4(contribution.x, contribution.y, contribution.z, 0.0f);
        a.nrcState[2ull * pixel] = make_float4(prevLocalThroughput.x, prevLocalThroughput.y, prevLocalThroughput.z, primaryPathSpread);
        a.nrcState[2ull * pixel + 1] = make_float4(curSqrtPathSpread, bits2f(prevTrainIdx), bits2f(flags), 0.0f);
        if (!o.wantExt) nrc_end_path(a, pixel, tile, flags, prevTrainIdx, pathLength);
    }
    push_vertex(a, pixel, pos, o);
}

// numInferenceQueries of the frame on the device (neural_radiance_caching_main.cpp:2293-2303 does this on the host)
__global__ void k_nrc_count_queries(PtArgs a, uint32_t* out) {
    const uint2 ts = *static_cast<const uint2*>(a.nrc.tileSize[a.f.bufferIndex]);
    const uint32_t W = static_cast<uint32_t>(a.s.imageSizeX), H = static_cast<uint32_t>(a.s.imageSizeY);
    const uint32_t n = W * H + ((W + ts.x - 1) / ts.x) * ((H + ts.y - 1) / ts.y);
    *out = (n + 127u) / 128u * 128u;
}

// perFrameContributionBuffer = contribution (:375)
__global__ __launch_bounds__(kPtBlock) void k_nrc_pt_finish(PtArgs a) {
    const PixelId px = pixel_of_thread(a.px);
    if (!px.valid) return;
    const size_t p = px.p;
    const float4 c = a.state[2 * p + 1];
    float* dst = static_cast<float*>(a.nrc.perFrameContributionBuffer) + 3 * p;
    dst[0] = c.x; dst[1] = c.y; dst[2] = c.z;
}

GFX_DEV f3 nrc_scaled_prediction(const PtArgs& a, size_t entry) {
    const float* r = static_cast<const float*>(a.nrc.inferredRadianceBuffer) + 3 * entry;
    f3 radiance(fmax2(r[0], 0.0f), fmax2(r[1], 0.0f), fmax2(r[2], 0.0f));
    if (a.nrc.radianceScale > 0) radiance = radiance / a.nrc.radianceScale;
    const float* q = static_cast<const float*>(a.nrc.inferenceRadianceQueryBuffer) + 14 * entry;
    return radiance * f3(q[8] + q[11], q[9] + q[12], q[10] + q[13]);
}

// accumulateInferredRadianceValues, nrc_setup_kernels.cu:51-93
__global__ __launch_bounds__(kPtBlock) void k_nrc_accumulate(PtArgs a) {
    const PixelId px = pixel_of_thread(a.px);
    if (!px.valid) return;
    const size_t p = px.p;
    const float4 t = static_cast<const float4*>(a.nrc.inferenceTerminalInfoBuffer)[p];
    const float* d = static_cast<const float*>(a.nrc.perFrameContributionBuffer) + 3 * p;
    const f3 directCont(d[0], d[1], d[2]);
    f3 radiance(0.0f);
    if (f2bits(t.w) & 1u) radiance = nrc_scaled_prediction(a, p);
    const f3 contribution = directCont + f3(t.x, t.y, t.z) * radiance;
    float4* beauty = static_cast<float4*>(a.s.beautyAccumBuffer) + p;
    f3 prev(0.0f);
    if (a.f.numAccumFrames > 0) { const float4 b = *beauty; prev = f3(b.x, b.y, b.z); }
    const float curWeight = 1.0f / (1 + a.f.numAccumFrames);
    const f3 result = (1 - curWeight) * prev + curWeight * contribution;
    *beauty = make_float4(result.x, result.y, result.z, 1.0f);
}

// propagateRadianceValues, nrc_setup_kernels.cu:95-137
__global__ __launch_bounds__(kPtBlock) void k_nrc_propagate(PtArgs a) {
    const uint32_t i = blockIdx.x * kPtBlock + threadIdx.x;
    if (i >= a.nrc.maxNumTrainingSuffixes) return;
    const uint32_t bits = static_cast<const uint32_t*>(a.nrc.trainSuffixTerminalInfoBuffer)[i];
    uint32_t last = bits & 0x7FFFFFu;
    if (last == kInvalidVertexDataIndex) return;
    f3 contribution(0.0f);
    if ((bits >> 23) & 1u) contribution = nrc_scaled_prediction(a, static_cast<size_t>(a.s.imageSizeX) * a.s.imageSizeY + i);
    while (last != kInvalidVertexDataIndex) {
        const float4 vi = static_cast<const float4*>(a.nrc.trainVertexInfoBuffer)[last];
        float* t = static_cast<float*>(a.nrc.trainTargetBuffer[0]) + 3ull * last;
        contribution = f3(t[0], t[1], t[2]) + f3(vi.x, vi.y, vi.z) * contribution;
        const float* q = static_cast<const float*>(a.nrc.trainRadianceQueryBuffer[0]) + 14ull * last;
        const f3 ref(q[8] + q[11], q[9] + q[12], q[10] + q[13]);
        t[0] = ref.x != 0 ? contribution.x / ref.x : 0.0f;
        t[1] = ref.y != 0 ? contribution.y / ref.y : 0.0f;
        t[2] = ref.z != 0 ? contribution.z / ref.z : 0.0f;
        last = f2bits(vi.w) & 0x7FFFFFu;
    }
}

GFX_DEV int32_t float_to_ordered_int(float f) { const int32_t i = static_cast<int32_t>(f2bits(f)); return i >= 0 ? i : i ^ 0x7FFFFFFF; }

// shuffleTrainingData, nrc_setup_kernels.cu:139-216 (one thread per destination candidate, 65536 threads)
__global__ __launch_bounds__(kPtBlock) void k_nrc_shuffle(PtArgs a) {
    const uint32_t i = blockIdx.x * kPtBlock + threadIdx.x;
    const uint32_t bufIdx = a.f.bufferIndex;
    const uint32_t numTrainingData = *static_cast<const uint32_t*>(a.nrc.numTrainingData[bufIdx]);
    float* dstQ = static_cast<float*>(a.nrc.trainRadianceQueryBuffer[1]);
    float* dstT = static_cast<float*>(a.nrc.trainTargetBuffer[1]);
    if (numTrainingData == 0) {
        for (int k = 0; k < 14; ++k) dstQ[14ull * i + k] = 0.0f;
        dstT[3ull * i] = 0.0f; dstT[3ull * i + 1] = 0.0f; dstT[3ull * i + 2] = 0.0f;
        return;
    }
    uint32_t* shuffler = static_cast<uint32_t*>(a.nrc.dataShufflerBuffer) + i;
    const uint32_t state = (*shuffler * 1103515245u + 12345u) % (1u << 31);    // LinearCongruentialGenerator, :164-181
    *shuffler = state;
    const uint32_t dstIdx = state % kNumTrainingDataPerFrame;
    const uint32_t srcIdx = i % numTrainingData;
    const float* sq = static_cast<const float*>(a.nrc.trainRadianceQueryBuffer[0]) + 14ull * srcIdx;
    const float* st = static_cast<const float*>(a.nrc.trainTargetBuffer[0]) + 3ull * srcIdx;
    float q[14];
    bool valid = true;
#pragma unroll
    for (int k = 0; k < 14; ++k) { q[k] = sq[k]; valid = valid && is_finite(q[k]); }
    if (!valid) for (int k = 0; k < 14; ++k) q[k] = 0.0f;
    f3 target(st[0], st[1], st[2]);
    if (!all_finite(target)) target = f3(0.0f);
    // statistics (min / max exact; the average accumulates in arrival order)
    int32_t* mm = static_cast<int32_t*>(a.nrc.targetMinMax[bufIdx]);
    float* avg = static_cast<float*>(a.nrc.targetAvg[bufIdx]);
    const float c[3] = { target.x, target.y, target.z };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int32_t lo = float_to_ordered_int(c[k]), hi = lo;
        float sum = c[k] / kNumTrainingDataPerFrame;
        for (int off = 32; off >= 1; off >>= 1) {
            lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off)); sum += __shfl_xor(sum, off);
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(mm + k, lo); atomicMax(mm + 3 + k, hi); atomicAdd(avg + k, sum); }
    }
    if (a.nrc.radianceScale > 0) target = target * a.nrc.radianceScale;
    target = f3(fmin2(target.x, 1e+6f), fmin2(target.y, 1e+6f), fmin2(target.z, 1e+6f));
#pragma unroll
    for (int k = 0; k < 14; ++k) dstQ[14ull * dstIdx + k] = q[k];
    dstT[3ull * dstIdx] = target.x; dstT[3ull * dstIdx + 1] = target.y; dstT[3ull * dstIdx + 2] = target.z;
}

// visualizePrediction ray generation, optix_pathtracing_kernels.cu:705-778
__global__ __launch_bounds__(kPtBlock) void k_nrc_visualize(PtArgs a) {
    const size_t p = static_cast<size_t>(blockIdx.x) * kPtBlock + threadIdx.x;
    if (p >= static_cast<size_t>(a.s.imageSizeX) * a.s.imageSizeY) return;
    const uint4 g0 = static_cast<const uint4*>(a.s.gbuffer0[a.f.bufferIndex])[p];
    const bool surface = g0.x != 0xFFFFFFFFu;
    if (surface) {
        const float bcB = decode_bc(g0.w & 0xFFFF), bcC = decode_bc(g0.w >> 16);
        const DevInstance* inst = a.scene.insts + g0.x;
        const DevGeomInst g = a.scene.geomInsts[g0.y];
        const uint32_t* tri = a.scene.triangles + 3ull * (g.triangleOffset + g0.z);
        const DevVertex vA = load_vertex(a.scene.vertices + g.vertexOffset + tri[0]);
        const DevVertex vB = load_vertex(a.scene.vertices + g.vertexOffset + tri[1]);
        const DevVertex vC = load_vertex(a.scene.vertices + g.vertexOffset + tri[2]);
        const float bcA = 1 - (bcB + bcC);
        const f3 pAo(vA.px, vA.py, vA.pz), pBo(vB.px, vB.py, vB.pz), pCo(vC.px, vC.py, vC.pz);
        const m34 xfm = load_m34(inst->transform);
        const m33 nrm = load_m33_rows(inst->normalMatrix);
        f3 pos = xfm_point(xfm, bcA * pAo + bcB * pBo + bcC * pCo);
        f3 ng = unit(mul(nrm, cross(pBo - pAo, pCo - pAo)));
        const f3 nsObj = bcA * f3(vA.nx, vA.ny, vA.nz) + bcB * f3(vB.nx, vB.ny, vB.nz) + bcC * f3(vC.nx, vC.ny, vC.nz);
        const f3 tcObj = bcA * f3(vA.tx, vA.ty, vA.tz) + bcB * f3(vB.tx, vB.ty, vB.tz) + bcC * f3(vC.tx, vC.ty, vC.tz);
        f3 ns = unit(mul(nrm, nsObj));
        f3 tc0 = xfm_vector(xfm, tcObj);
        tc0 = unit(tc0 - dot(ns, tc0) * ns);
        if (!all_finite(ns)) { ng = f3(0, 0, 1); ns = f3(0, 0, 1); tc0 = f3(1, 0, 0); }
        if (!all_finite(tc0)) { f3 bt; make_coordinate_system(ns, tc0, bt); }
        const Camera cam = load_camera(a.f.camera);
        f3 vOut = cam.pos - pos;
        vOut = vOut / sqrtf(len2(vOut));
        const float frontHit = dot(vOut, ng) >= 0.0f ? 1.0f : -1.0f;
        pos = offset_ray_origin(pos, frontHit * ng);
        const gfx_material& mat = a.scene.materials[g.materialSlot];
        const float tu = bcA * vA.u + bcB * vB.u + bcC * vC.u, tv = bcA * vA.v + bcB * vB.v + bcC * vC.v;
        Frame frame(ns, tc0);
        if (a.f.enableBumpMapping) apply_bump_mapping(read_modified_normal(a.scene, mat, tu, tv), frame);
        Bsdf bsdf; bsdf.setup(a.scene, mat, tu, tv);
        nrc_store_query(a.nrc.inferenceRadianceQueryBuffer, p, nrc_make_query(a, pos, frame.n, vOut, bsdf));
    }
    static_cast<float4*>(a.nrc.inferenceTerminalInfoBuffer)[p] = make_float4(1.0f, 1.0f, 1.0f, bits2f(nrc_terminal_bits(surface, 1, false, false)));
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
    const bool nrcRegir = pass == GFX_PT_PATH_TRACE_NRC_REGIR, nrcRestir = pass == GFX_PT_PATH_TRACE_NRC_RESTIR;
    const bool regirPass = (pass >= GFX_PT_REGIR_BUILD_CELL_RESERVOIRS && pass <= GFX_PT_REGIR_UPDATE_LAST_ACCESS) || nrcRegir;
    const bool nrcPass = (pass >= GFX_PT_NRC_PREPROCESS && pass <= GFX_PT_NRC_COUNT_QUERIES) || nrcRegir || nrcRestir;
    if (!regirPass && !nrcPass && pass != GFX_PT_PATH_TRACE_BASELINE) throw HipError("gfx_pt_launch: unknown pass");
    if (regirPass && !ctx.regirValid) throw HipError("gfx_pt_launch: gfx_regir_set_params has not been called");
    if (nrcPass && !ctx.nrcRenderValid) throw HipError("gfx_pt_launch: gfx_nrc_set_render_params has not been called");
    PtArgs a;
    std::memset(&a, 0, sizeof(a));
    a.scene = ctx.devScene();
    a.s = rp.s; a.f = rp.f;
    if (nrcPass) {
        a.nrc = ctx.nrcRender;
        const size_t np = static_cast<size_t>(rp.s.imageSizeX) * rp.s.imageSizeY;
        auto simple = [&](const char* name, void (*kernel)(PtArgs), size_t threads) {
            ScopedKernelTimer timer(ctx, stream, name);
            hipLaunchKernelGGL(kernel, dim3(static_cast<uint32_t>((threads + kPtBlock - 1) / kPtBlock)), dim3(kPtBlock), 0, stream, a);
            GFX_HIP(hipGetLastError());
        };
        if (pass == GFX_PT_NRC_PREPROCESS) { simple("nrc_preprocess", k_nrc_preprocess, a.nrc.maxNumTrainingSuffixes); return; }
        if (pass == GFX_PT_NRC_ACCUMULATE) {   // the one per-pixel NRC pass besides the path tracing: honours the row band
            if (rowEnd > height || rowBegin > rowEnd) throw HipError("gfx_pt_launch: row range outside the image");
            a.px = make_pixel_grid(ctx, width, rowBegin, (rowBegin == 0 && rowEnd == 0) ? height : rowEnd);
            if (a.px.rowEnd == a.px.rowBegin) return;
            simple("nrc_accumulate", k_nrc_accumulate, static_cast<size_t>(a.px.launchBlocks) * kPtBlock);
            return;
        }
        if (pass == GFX_PT_NRC_PROPAGATE) { simple("nrc_propagate", k_nrc_propagate, a.nrc.maxNumTrainingSuffixes); return; }
        if (pass == GFX_PT_NRC_SHUFFLE) { simple("nrc_shuffle", k_nrc_shuffle, kNumTrainingDataPerFrame); return; }
        if (pass == GFX_PT_NRC_VISUALIZE_PREDICTION) { simple("nrc_visualize", k_nrc_visualize, np); return; }
        if (pass == GFX_PT_NRC_COUNT_QUERIES) {
            if (!ctx.nrcQueryCount.p) { ctx.nrcQueryCount.reserve(256); GFX_HIP(hipMemsetAsync(ctx.nrcQueryCount.p, 0, 256, stream)); }
            hipLaunchKernelGGL(k_nrc_count_queries, dim3(1), dim3(1), 0, stream, a, ctx.nrcQueryCount.as<uint32_t>());
            GFX_HIP(hipGetLastError());
            return;
        }
    }
    if (regirPass) a.g = ctx.regir;
    if (pass == GFX_PT_REGIR_BUILD_CELL_RESERVOIRS || pass == GFX_PT_REGIR_BUILD_CELL_RESERVOIRS_TEMPORAL) {
        const size_t numLightSlots = static_cast<size_t>(a.g.gridDimension[0]) * a.g.gridDimension[1] * a.g.gridDimension[2] * kNumLightSlotsPerCell;
        ScopedKernelTimer timer(ctx, stream, "regir_build_cells");
        const dim3 grid(static_cast<uint32_t>((numLightSlots + kPtBlock - 1) / kPtBlock));
        const bool temporal = pass != GFX_PT_REGIR_BUILD_CELL_RESERVOIRS;
        if (temporal) hipLaunchKernelGGL(k_regir_build<true>, grid, dim3(kPtBlock), 0, stream, a);
        else hipLaunchKernelGGL(k_regir_build<false>, grid, dim3(kPtBlock), 0, stream, a);
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
    const bool nrc = pass == GFX_PT_PATH_TRACE_NRC || nrcRegir || nrcRestir;
    if (nrcRestir) {
        if (!rp.s.reservoirBuffer[0] || !rp.s.reservoirBuffer[1] || !rp.s.reservoirInfoBuffer[0] || !rp.s.reservoirInfoBuffer[1] ||
            !rp.s.gbuffer2[0] || !rp.s.gbuffer3[0])
            throw HipError("gfx_pt_launch: GFX_PT_PATH_TRACE_NRC_RESTIR reads the reservoirs and G-buffers 2 / 3 of the ReSTIR passes (static parameters)");
        if (!(rowBegin == 0 && (rowEnd == 0 || rowEnd == height)))
            throw HipError("gfx_pt_launch: GFX_PT_PATH_TRACE_NRC_RESTIR is a whole-frame pass");
        a.curRes = rp.currentReservoirIndex & 1u;
    }
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
    if (nrc) { ctx.nrcState.reserve(32 * numPixels); ctx.neeTrainIdx.reserve(4 * numPixels); }
    ctx.smallCounters.reserve(kSmallCountersBytes);
    uint32_t* counters = ctx.smallCounters.as<uint32_t>() + 8;   // [0] nee (even bounces), [1] ext ping, [2] ext pong, [3] nee (odd bounces)
    GFX_HIP(hipMemsetAsync(counters, 0, 5 * sizeof(uint32_t), stream));   // [4]: the slot ticket of k_pt_regen
    uint32_t* const neeCounts[2] = { counters, counters + 3 };

    // The NEE trace of a bounce (any-hit) and the kernel that applies its result touch the NEE queue, the occlusion words and the
    // per-pixel contribution; the extension trace of the same bounce (closest-hit) touches the extension queue and the hit records:
    // disjoint, and both only needed by the bounce kernel.  So the first two run on a second stream (own stack-spill / ticket scratch)
    // underneath the third -- small launches are mostly ramp-up and the tail of their longest rays (profiles/r04_experiments.txt 8),
    // which now overlap.  The NEE queue head alternates between two words so that the extension trace can zero the one the bounce kernel
    // appends to while the NEE trace still reads the other.
    const bool overlap = ctx.tune.ptOverlap != 0;
    if (overlap && !ctx.auxStream) GFX_HIP(hipStreamCreateWithFlags(&ctx.auxStream, hipStreamNonBlocking));
    if (overlap && !ctx.auxFork) {
        GFX_HIP(hipEventCreateWithFlags(&ctx.auxFork, hipEventDisableTiming));
        GFX_HIP(hipEventCreateWithFlags(&ctx.auxJoin, hipEventDisableTiming));
    }
    hipStream_t neeStream = overlap ? ctx.auxStream : stream;

    a.px = make_pixel_grid(ctx, width, rowBegin, rowEnd);
    a.neeOrg = ctx.rayOrg.as<float4>(); a.neeDir = ctx.rayDir.as<float4>(); a.neePending = ctx.ptPending.as<float4>();
    a.neeCount = neeCounts[0];
    a.occluded = ctx.rayOut.as<uint32_t>();
    a.hits = ctx.rayHits.as<gfx_hit>();
    a.tris = ctx.accels[h - 1]->trisPtr();
    a.state = ctx.ptState.as<float4>();
    if (nrc) { a.nrcState = ctx.nrcState.as<float4>(); a.neeTrainIdx = ctx.neeTrainIdx.as<uint32_t>(); }
    float4* extOrg[2] = { ctx.ptExtOrg.as<float4>(), ctx.ptExtOrg.as<float4>() + bandPixels };
    float4* extDir[2] = { ctx.ptExtDir.as<float4>(), ctx.ptExtDir.as<float4>() + bandPixels };
    uint32_t* extOwner[2] = { ctx.ptExtOwner.as<uint32_t>(), ctx.ptExtOwner.as<uint32_t>() + bandPixels };
    const uint32_t grid = static_cast<uint32_t>((bandPixels + kPtBlock - 1) / kPtBlock);
    const DevAccel accel = ctx.accels[h - 1]->dev();
    auto set_queues = [&](int in, int out) {
        a.extOrgIn = extOrg[in]; a.extDirIn = extDir[in]; a.extOwnerIn = extOwner[in]; a.extCountIn = counters + 1 + in;
        a.extOrgOut = extOrg[out]; a.extDirOut = extDir[out]; a.extOwnerOut = extOwner[out]; a.extCountOut = counters + 1 + out;
    };
    auto launch = [&](hipStream_t s, const char* name, void (*kernel)(PtArgs)) {   // one thread per queue entry (capacity = the band's pixels)
        ScopedKernelTimer timer(ctx, s, name);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kPtBlock), 0, s, a);
        GFX_HIP(hipGetLastError());
    };
    auto launch_pixels = [&](const char* name, void (*kernel)(PtArgs)) {   // one thread per pixel of the band (a.px)
        ScopedKernelTimer timer(ctx, stream, name);
        hipLaunchKernelGGL(kernel, dim3(a.px.launchBlocks), dim3(kPtBlock), 0, stream, a);
        GFX_HIP(hipGetLastError());
    };
    auto trace = [&](hipStream_t s, bool auxScratch, int mode, const float4* org, const float4* dir, const uint32_t* count, void* out,
                     uint32_t* zero0 = nullptr, uint32_t* zero1 = nullptr) {
        TraceLaunch t;
        t.accel = accel; t.rayOrgTmin = org; t.rayDirTmax = dir; t.numRays = 0; t.numRaysPtr = count; t.out = out; t.mode = mode;
        t.zeroWords[0] = zero0; t.zeroWords[1] = zero1;
        if (auxScratch) { t.spill = &ctx.auxSpill; t.counters = &ctx.auxCounters; }
        trace_launch(ctx, s, t);
    };
    auto join = [&]() { if (overlap) GFX_HIP(hipStreamWaitEvent(stream, ctx.auxJoin, 0)); };

    int cur = 0;                 // the queue k_pt_first fills
    set_queues(1, cur);
    a.pathLength = 1; a.maxLengthTerminate = 0;
    a.nextMaxLengthTerminate = 2 >= maxPathLength ? 1u : 0u;
    // a launch of about one round of waves (512 x 512; a band of an 8-way split full-HD frame): the path of a pixel in one kernel.
    // "fuse_passes" 1 never, 2 always; counting launches keep the wavefront form (the counters live in k_trace).
    {
        const uint32_t launchWaves = a.px.launchBlocks * (kPtBlock / 64), waveSlots = static_cast<uint32_t>(ctx.numCUs) * 16u;
        const int spillCap = static_cast<int>(local_spill_depth(ctx.accels[h - 1]->maxDepth));
        const size_t spillBytes = sizeof(uint2) * static_cast<size_t>(a.px.launchBlocks) * kPtBlock * spillCap;
        const bool small = launchWaves <= waveSlots + waveSlots / 2;
        if (!nrc && !ctx.countersEnabled && spillBytes <= (size_t(1) << 30) && (ctx.tune.fusePasses == 2 || (ctx.tune.fusePasses == 0 && small))) {
            ctx.spill.reserve(spillBytes);
            unsigned long long* ptDiag = nullptr;             // "pt_diag": wave iterations / lanes with a ray / traversal steps / waves / refills (gfx_pt_diag_read)
            if (ctx.tune.ptDiag) {
                if (!ctx.ptDiag.p) { ctx.ptDiag.reserve(64); GFX_HIP(hipMemsetAsync(ctx.ptDiag.p, 0, 64, stream)); }
                ptDiag = ctx.ptDiag.as<unsigned long long>();
            }
            // Path regeneration (k_pt_regen): when the launch is more blocks than the GPU holds at once, launch what it holds and let the
            // lanes draw pixels until none are left ("pt_regen": resident blocks per CU, 0 = off)
            const uint32_t regenBlocks = static_cast<uint32_t>(ctx.tune.ptRegen) * static_cast<uint32_t>(ctx.numCUs);
            if (!regir && ctx.tune.ptRegen > 0 && a.px.launchBlocks > regenBlocks) {
                a.px.order = nullptr;
                ScopedKernelTimer timer(ctx, stream, "pt_regen");
                hipLaunchKernelGGL(k_pt_regen, dim3(regenBlocks), dim3(kPtBlock), 0, stream, a, accel, ctx.spill.as<uint2>(), spillCap, maxPathLength,
                                   counters + 4, a.px.launchBlocks * static_cast<uint32_t>(kPtBlock), ctx.tune.ptRegenMin, ptDiag);
                GFX_HIP(hipGetLastError());
                return;
            }
            // 159 VGPRs: three blocks per CU are resident, so a 512 x 512 frame (1 024 blocks) is already more than one round
            const uint32_t* order = nullptr;
            uint32_t* cost = nullptr;
            const uint64_t key = (static_cast<uint64_t>(rowBegin) << 44) ^ (static_cast<uint64_t>(rowEnd) << 24) ^ (static_cast<uint64_t>(width) << 4) ^ (regir ? 1u : 0u);
            block_order_begin(ctx, stream, 4, a.px.launchBlocks, key, 3u * static_cast<uint32_t>(ctx.numCUs), order, cost);
            a.px.order = order;
            {
                ScopedKernelTimer timer(ctx, stream, regir ? "pt_regir_fused" : "pt_fused");
                hipLaunchKernelGGL(regir ? k_pt_fused<true> : k_pt_fused<false>, dim3(a.px.launchBlocks), dim3(kPtBlock), 0, stream, a, accel, ctx.spill.as<uint2>(), spillCap, maxPathLength, cost, ptDiag);
                GFX_HIP(hipGetLastError());
            }
            block_order_end(ctx, stream, 4, a.px.launchBlocks, cost);
            return;
        }
    }
    if (nrc) launch_pixels("nrc_pt_first", nrcRegir ? k_nrc_pt_first<1> : nrcRestir ? k_nrc_pt_first<2> : k_nrc_pt_first<0>);
    else launch_pixels("pt_first", regir ? k_pt_first<true> : k_pt_first<false>);
    // while (true) { ++pathLength; trace; }.  Baseline: at least one extension even when maxPathLength < 2,
    // the terminal vertex (implicit light only) emits no NEE ray.  ReGIR: the loop head breaks before the
    // trace at the length limit, so the last vertex's NEE ray is resolved after the loop.  NRC:
    // maxPathLength == 0 means unlimited bounces (neural_radiance_caching_main.cpp:2246) -- the live
    // path count is read back every fourth bounce to stop (the 6-bit pathLength field caps it at 63).
    int nee = 0;                 // the NEE queue head the last vertex kernel appended to
    for (uint32_t pathLength = 2;; ++pathLength) {
        if (overlap) { GFX_HIP(hipEventRecord(ctx.auxFork, stream)); GFX_HIP(hipStreamWaitEvent(neeStream, ctx.auxFork, 0)); }
        a.neeCount = neeCounts[nee];
        trace(neeStream, overlap, GFX_TRACE_ANY, a.neeOrg, a.neeDir, a.neeCount, ctx.rayOut.p);
        launch(neeStream, "pt_apply_nee", k_pt_apply_nee);
        if (overlap) GFX_HIP(hipEventRecord(ctx.auxJoin, neeStream));
        if (regir && pathLength >= maxPathLength) { join(); break; }
        // NRC: training paths skip Russian roulette (and with it the length test) at pathLength 2 (:459-462),
        // so the last vertex that can exist is max(maxPathLength, 3)
        if (nrc && maxPathLength > 0 && pathLength > (maxPathLength > 3 ? maxPathLength : 3)) { join(); break; }
        if (nrc && pathLength >= 63) { join(); break; }
        if (nrc && maxPathLength == 0 && pathLength > 2 && (pathLength & 3u) == 2u) {
            uint32_t live = 0;
            GFX_HIP(hipMemcpyAsync(&live, counters + 1 + cur, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            GFX_HIP(hipStreamSynchronize(stream));
            if (live == 0) { join(); break; }
        }
        // the extension trace also resets the two queue heads the bounce kernel appends to (the NEE head of the NEXT bounce -- not the
        // one the NEE trace above is reading -- and the other extension queue): no memset between the kernels of a bounce
        trace(stream, false, GFX_TRACE_CLOSEST, extOrg[cur], extDir[cur], counters + 1 + cur, ctx.rayHits.p, neeCounts[nee ^ 1], counters + 1 + (cur ^ 1));
        join();                  // the bounce kernel adds to the contribution after k_pt_apply_nee has, and rewrites the NEE queue
        set_queues(cur, cur ^ 1);
        nee ^= 1;
        a.neeCount = neeCounts[nee];
        a.pathLength = pathLength;
        a.maxLengthTerminate = (pathLength >= maxPathLength && (!nrc || maxPathLength > 0)) ? 1u : 0u;
        a.nextMaxLengthTerminate = pathLength + 1 >= maxPathLength ? 1u : 0u;
        if (nrc) launch(stream, "nrc_pt_bounce", nrcRegir ? k_nrc_pt_bounce<1> : nrcRestir ? k_nrc_pt_bounce<2> : k_nrc_pt_bounce<0>);
        else launch(stream, "pt_bounce", regir ? k_pt_bounce<true> : k_pt_bounce<false>);
        cur ^= 1;
        if (!regir && !nrc && a.maxLengthTerminate) break;
    }
    if (nrc) { launch_pixels("nrc_pt_finish", k_nrc_pt_finish); return; }
    launch_pixels("pt_finish", k_pt_finish);
}

} // namespace gfx
