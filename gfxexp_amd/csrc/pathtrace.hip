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
#include "internal.h"
#include "shading.hip.h"
#include "pass_common.hip.h"

namespace gfx {

constexpr int kPtBlock = 256;

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
};

struct PtVertexOut {              // what one shaded vertex hands to the queues
    bool wantNee; f3 neeDir; float neeTmax; f3 pending;
    bool wantExt; f3 extDir;
};

// performNextEventEstimation (optix_pathtracing_kernels.cu:18-72) without the trace: returns the
// unshadowed estimate and the shadow ray; + BSDF sampling of the next direction (:140-147, :283-295).
GFX_DEV void shade_vertex(const PtArgs& a, const EnvMap& env, bool envEnabled, f3 pos, f3 vOutLocal, const Frame& frame,
                          const Bsdf& bsdf, Pcg32& rng, f3& alpha, f3& contribution, float& dirPDensity, PtVertexOut& o) {
    const float* instWeights = a.scene.lightWeights + a.scene.lightInstDistOffset;
    const float* instCDF = a.scene.lightCDF + a.scene.lightInstDistOffset;
    f3 ret(0.0f);
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
    LightSample ls;
    ls.emittance = f3(0.0f); ls.position = f3(0.0f); ls.normal = f3(0.0f); ls.atInfinity = 0;
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
}

GFX_DEV void push_vertex(const PtArgs& a, uint32_t pixel, f3 pos, const PtVertexOut& o) {
    const uint32_t ns = queue_append(o.wantNee, pos, o.neeDir, 0.0f, o.neeTmax, a.neeOrg, a.neeDir, a.neeCount);
    if (o.wantNee) a.neePending[ns] = make_float4(o.pending.x, o.pending.y, o.pending.z, bits2f(pixel));
    const uint32_t es = queue_append(o.wantExt, pos, o.extDir, 0.0f, 3.402823466e+38f, a.extOrgOut, a.extDirOut, a.extCountOut);
    if (o.wantExt) a.extOwnerOut[es] = pixel;
}

// pathTrace_rayGen_generic up to the path extension loop (optix_pathtracing_kernels.cu:74-160)
__global__ __launch_bounds__(kPtBlock) void k_pt_first(PtArgs a) {
    const size_t p = a.pixelBegin + static_cast<size_t>(blockIdx.x) * kPtBlock + threadIdx.x;
    const uint32_t bufIdx = a.f.bufferIndex;
    PtVertexOut o;
    o.wantNee = false; o.wantExt = false; o.neeDir = f3(0.0f); o.extDir = f3(0.0f); o.neeTmax = 0; o.pending = f3(0.0f);
    f3 pos(0.0f);
    if (p < a.pixelEnd) {
        const uint4 g0 = static_cast<const uint4*>(a.s.gbuffer0[bufIdx])[p];
        const EnvMap env = load_env(a.s);
        const bool envEnabled = env.present() && a.f.enableEnvLight;
        f3 contribution(0.001f, 0.001f, 0.001f);
        f3 alpha(1.0f);
        float dirPDensity = 0.0f;
        if (g0.x != 0xFFFFFFFFu) {
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
            uint64_t* rngBuf = static_cast<uint64_t*>(a.s.rngBuffer);
            Pcg32 rng; rng.state = rngBuf[p];
            const gfx_material& mat = a.scene.materials[g.materialSlot];
            const f3 vOut = unit(cam.pos - pos);
            const float frontHit = dot(vOut, ng) >= 0.0f ? 1.0f : -1.0f;
            pos = offset_ray_origin(pos, frontHit * ng);
            const Frame frame(ns, tc0);
            const f3 vOutLocal = frame.to_local(vOut);
            contribution = f3(0.0f);
            if (vOutLocal.z > 0 && mat.hasEmittance)
                contribution = contribution + alpha * f3(mat.emittance[0], mat.emittance[1], mat.emittance[2]) / kPi;
            Bsdf bsdf; bsdf.setup(mat);
            shade_vertex(a, env, envEnabled, pos, vOutLocal, frame, bsdf, rng, alpha, contribution, dirPDensity, o);
            rngBuf[p] = rng.state;
        }
        else if (envEnabled) {
            contribution = a.f.envLightPowerCoeff * env.fetch(decode_bc(g0.w & 0xFFFF), decode_bc(g0.w >> 16));
        }
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
// (optix_pathtracing_kernels.cu:210-296, :306-341, :161-201)
__global__ __launch_bounds__(kPtBlock) void k_pt_bounce(PtArgs a) {
    const uint32_t i = blockIdx.x * kPtBlock + threadIdx.x;
    const uint32_t count = *a.extCountIn;
    PtVertexOut o;
    o.wantNee = false; o.wantExt = false; o.neeDir = f3(0.0f); o.extDir = f3(0.0f); o.neeTmax = 0; o.pending = f3(0.0f);
    f3 pos(0.0f);
    uint32_t pixel = 0;
    if (i < count) {
        pixel = a.extOwnerIn[i];
        const gfx_hit h = a.hits[i];
        const float4 ro4 = a.extOrgIn[i], rd4 = a.extDirIn[i];
        const f3 rayOrg(ro4.x, ro4.y, ro4.z), rayDir(rd4.x, rd4.y, rd4.z);
        const float4 s0 = a.state[2ull * pixel], s1 = a.state[2ull * pixel + 1];
        f3 alpha(s0.x, s0.y, s0.z);
        const float prevDirPDensity = s0.w;
        f3 contribution(s1.x, s1.y, s1.z);
        const EnvMap env = load_env(a.s);
        const bool envEnabled = env.present() && a.f.enableEnvLight;
        if (h.triIndex == GFX_INVALID_SLOT) {
            if (envEnabled) {
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
            float hypAreaPDensity;
            {
                float lightProb = 1.0f;
                if (envEnabled) lightProb *= (1 - 0.25f);
                const float instImportance = inst->distIntegral;
                lightProb *= (inst->uniformScale * inst->uniformScale * instImportance) / *a.scene.lightInstIntegral;
                lightProb *= g.distIntegral / instImportance;
                if (!is_finite(lightProb)) hypAreaPDensity = 0.0f;
                else {
                    float pmf = 0.0f;
                    if (g.distOffset != 0xFFFFFFFFu && g.distIntegral != 0.0f) pmf = a.scene.lightWeights[g.distOffset + primIndex] / g.distIntegral;
                    lightProb *= pmf;
                    hypAreaPDensity = lightProb / area;
                }
            }
            const gfx_material& mat = a.scene.materials[g.materialSlot];
            const f3 vOut = unit(-rayDir);
            const float frontHit = dot(vOut, ng) >= 0.0f ? 1.0f : -1.0f;
            const Frame frame(ns, tc0);
            pos = offset_ray_origin(pos, frontHit * ng);
            const f3 vOutLocal = frame.to_local(vOut);
            if (vOutLocal.z > 0 && mat.hasEmittance) {
                const f3 emittance(mat.emittance[0], mat.emittance[1], mat.emittance[2]);
                const float dist2 = len2(rayOrg - pos);
                const float lightPDensity = hypAreaPDensity * dist2 / vOutLocal.z;
                const float misWeight = (prevDirPDensity * prevDirPDensity) / (prevDirPDensity * prevDirPDensity + lightPDensity * lightPDensity);
                contribution = contribution + alpha * emittance * (misWeight / kPi);
            }
            uint64_t* rngBuf = static_cast<uint64_t*>(a.s.rngBuffer);
            Pcg32 rng; rng.state = rngBuf[pixel];
            // Russian roulette; initImportance = sRGB_calcLuminance(RGB(1))
            const float initImportance = luminance_srgb(f3(1.0f));
            const float continueProb = fminf(luminance_srgb(alpha) / initImportance, 1.0f);
            float dirPDensity = prevDirPDensity;
            if (!(rng.uniform() >= continueProb || a.maxLengthTerminate)) {
                alpha = alpha / continueProb;
                Bsdf bsdf; bsdf.setup(mat);
                shade_vertex(a, env, envEnabled, pos, vOutLocal, frame, bsdf, rng, alpha, contribution, dirPDensity, o);
            }
            rngBuf[pixel] = rng.state;
            a.state[2ull * pixel] = make_float4(alpha.x, alpha.y, alpha.z, dirPDensity);
            a.state[2ull * pixel + 1] = make_float4(contribution.x, contribution.y, contribution.z, 0.0f);
        }
    }
    push_vertex(a, pixel, pos, o);
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
    if (pass != GFX_PT_PATH_TRACE_BASELINE) throw HipError("gfx_pt_launch: unknown pass");
    const RestirParams& rp = ctx.restir;
    if (!rp.valid) throw HipError("gfx_pt_launch: gfx_restir_set_params has not been called");
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

    PtArgs a;
    a.scene = ctx.devScene();
    a.s = rp.s; a.f = rp.f;
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
    launch("pt_first", k_pt_first);
    // while (true) { ++pathLength; trace; } -- at least one extension even when maxPathLength < 2
    for (uint32_t pathLength = 2;; ++pathLength) {
        trace(GFX_TRACE_ANY, a.neeOrg, a.neeDir, a.neeCount, ctx.rayOut.p);
        launch("pt_apply_nee", k_pt_apply_nee);
        trace(GFX_TRACE_CLOSEST, extOrg[cur], extDir[cur], counters + 1 + cur, ctx.rayHits.p);
        GFX_HIP(hipMemsetAsync(counters, 0, sizeof(uint32_t), stream));
        GFX_HIP(hipMemsetAsync(counters + 1 + (cur ^ 1), 0, sizeof(uint32_t), stream));
        set_queues(cur, cur ^ 1);
        a.pathLength = pathLength;
        a.maxLengthTerminate = pathLength >= maxPathLength ? 1u : 0u;
        launch("pt_bounce", k_pt_bounce);
        cur ^= 1;
        if (a.maxLengthTerminate) break;
    }
    launch("pt_finish", k_pt_finish);
}

} // namespace gfx
