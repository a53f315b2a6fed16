// restir.hip -- the original ReSTIR DI passes as a wavefront pipeline of HIP kernels.
//
// The reference runs each pass as one OptiX ray-generation megakernel that traces inline
// (restir_di/gpu_kernels/optix_gbuffer_kernels.cu, optix_restir_di_kernels.cu).  Here every
// optixTrace becomes: producer kernel -> dense ray queue (ballot/mbcnt compaction, one atomic per
// wave) -> trace.hip -> consumer kernel.  Per-pixel arithmetic and RNG draw order are those of the
// reference, so reservoirs match the CPU restatement bit for bit.
//
//   pass (gfx_restir_pass)          kernels
//   SETUP_GBUFFERS                  k_primary_rays -> trace closest -> k_gbuffer_resolve
//   INITIAL_RIS / +TEMPORAL_*       k_initial_candidates -> trace any -> k_temporal<mode>
//   SPATIAL_BIASED                  k_spatial<false>
//   SPATIAL_UNBIASED                k_spatial<true> (select + emit MIS rays) -> trace any -> k_spatial_mis_finish
//   SHADING                         k_shade_prepare -> trace any -> k_shade_finish
//   SPATIAL_BIASED_AND_SHADING      the two above, back to back
// (a small launch runs each of the ray passes as one kernel: k_gbuffer_fused / k_initial_fused / k_shading_fused below)
#include "internal.h"
#include "shading.hip.h"
#include "pass_common.hip.h"
#include "restir_common.hip.h"
#include "coop_fetch.hip.h"
#include "trace_local.hip.h"
#include "restir_rearch.hip.h"

namespace gfx {

// ---------------------------------------------------------------- SETUP_GBUFFERS
// ray generation of optix_gbuffer_kernels.cu:5-27
// (org.xyz | tmin, dir.xyz | tmax) of the pixel's primary ray; a launch slot without a pixel holds an empty-interval ray (an immediate miss)
struct RayPair { float4 org, dir; };
GFX_DEV RayPair primary_ray(const RestirArgs& a, const PixelId& px) {
    RayPair r;
    r.org = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    r.dir = make_float4(0.0f, 0.0f, 1.0f, -1.0f);
    if (!px.valid) return r;
    const size_t p = px.p;
    const int x = px.x, y = px.y;
    const Camera cam = load_camera(a.f.camera);
    float jx = 0.5f, jy = 0.5f;
    if (a.f.enableJittering) {
        uint64_t* rngBuf = static_cast<uint64_t*>(a.s.rngBuffer);
        Pcg32 rng; rng.state = rngBuf[p];
        jx = rng.uniform();
        jy = rng.uniform();
        rngBuf[p] = rng.state;
    }
    const float fx = (x + jx) / a.s.imageSizeX;
    const float fy = (y + jy) / a.s.imageSizeY;
    const float vh = 2 * gm_tan(cam.fovY * 0.5f);
    const float vw = cam.aspect * vh;
    const f3 dir = unit(mul(cam.ori, f3(vw * (0.5f - fx), vh * (0.5f - fy), 1)));
    r.org = make_float4(cam.pos.x, cam.pos.y, cam.pos.z, 0.0f);
    r.dir = make_float4(dir.x, dir.y, dir.z, 3.402823466e+38f);
    return r;
}
__global__ __launch_bounds__(kBlock) void k_primary_rays(RestirArgs a) {
    const PixelId px = pixel_of_thread(a.px);
    const RayPair r = primary_ray(a, px);          // the queue of this pass is indexed by launch slot
    a.rayOrg[px.slot] = r.org;
    a.rayDir[px.slot] = r.dir;
}

// PerspectiveCamera::calcScreenPosition, restir_di_shared.h:51-59
GFX_DEV void calc_screen_position(const Camera& cam, f3 pw, float& sx, float& sy) {
    const m33 invOri = inverse(cam.ori);
    const f3 pv = mul(invOri, pw - cam.pos);
    const float ax = pv.x / pv.z, ay = pv.y / pv.z;
    const float h = 2 * gm_tan(cam.fovY / 2);
    const float w = cam.aspect * h;
    sx = 1 - (ax + 0.5f * w) / w;
    sy = 1 - (ay + 0.5f * h) / h;
}

// closest-hit / miss programs + the tail of the ray-generation program (optix_gbuffer_kernels.cu:56-243)
GFX_DEV void gbuffer_resolve(const RestirArgs& a, const PixelId& px, const gfx_hit& h, f3 direction) {
    const size_t p = px.p;
    const int x = px.x, y = px.y;
    const uint32_t bufIdx = a.f.bufferIndex;

    f3 albedo(0.0f);
    const float qnan = bits2f(0x7FC00000u);
    f3 positionInWorld(qnan), prevPositionInWorld(qnan), shadingNormalInWorld(qnan);
    uint32_t qGeomNormal = 0, qTangent = 0, qTexCoord = 0;
    uint32_t matSlot = 0xFFFFFFFFu, instSlot = 0xFFFFFFFFu, geomInstSlot = 0xFFFFFFFFu, primIndex = 0xFFFFFFFFu;
    uint32_t qbcB = 0, qbcC = 0;

    if (h.triIndex != GFX_INVALID_SLOT) {
        const Bvh8Tri* tr = a.tris + h.triIndex;
        instSlot = tr->instSlot; geomInstSlot = tr->geomInstSlot; primIndex = tr->primIndex;
        const DevInstance* inst = a.scene.insts + instSlot;
        const DevGeomInst g = a.scene.geomInsts[geomInstSlot];
        matSlot = g.materialSlot;
        const uint32_t* tri = a.scene.triangles + 3ull * (g.triangleOffset + primIndex);
        const DevVertex vA = load_vertex(a.scene.vertices + g.vertexOffset + tri[0]);
        const DevVertex vB = load_vertex(a.scene.vertices + g.vertexOffset + tri[1]);
        const DevVertex vC = load_vertex(a.scene.vertices + g.vertexOffset + tri[2]);
        const float bcB = h.bcB, bcC = h.bcC;
        const float bcA = 1 - (bcB + bcC);
        qbcB = encode_bc(bcB);
        qbcC = encode_bc(bcC);
        const f3 pAo(vA.px, vA.py, vA.pz), pBo(vB.px, vB.py, vB.pz), pCo(vC.px, vC.py, vC.pz);
        const f3 positionInObj = bcA * pAo + bcB * pBo + bcC * pCo;
        const f3 shadingNormalInObj = bcA * f3(vA.nx, vA.ny, vA.nz) + bcB * f3(vB.nx, vB.ny, vB.nz) + bcC * f3(vC.nx, vC.ny, vC.nz);
        const f3 tc0DirInObj = bcA * f3(vA.tx, vA.ty, vA.tz) + bcB * f3(vB.tx, vB.ty, vB.tz) + bcC * f3(vC.tx, vC.ty, vC.tz);
        const float tu = bcA * vA.u + bcB * vB.u + bcC * vC.u;
        const float tv = bcA * vA.v + bcB * vB.v + bcC * vC.v;
        const f3 geomNormalInObj = cross(pBo - pAo, pCo - pAo);
        const m34 xfm = load_m34(inst->transform);
        const m33 nrm = load_m33_rows(inst->normalMatrix);
        positionInWorld = xfm_point(xfm, positionInObj);
        prevPositionInWorld = xfm_point(load_m34(inst->curToPrevTransform), positionInWorld);
        f3 geomNormalInWorld = unit(mul(nrm, geomNormalInObj));
        shadingNormalInWorld = unit(mul(nrm, shadingNormalInObj));
        f3 tc0DirInWorld = xfm_vector(xfm, tc0DirInObj);
        tc0DirInWorld = unit(tc0DirInWorld - dot(shadingNormalInWorld, tc0DirInWorld) * shadingNormalInWorld);
        if (!all_finite(shadingNormalInWorld)) {
            geomNormalInWorld = f3(0, 0, 1);
            shadingNormalInWorld = f3(0, 0, 1);
            tc0DirInWorld = f3(1, 0, 0);
        }
        qGeomNormal = encode_dir(geomNormalInWorld);
        qTexCoord = encode_uv(tu, tv);
        const gfx_material& mat = a.scene.materials[matSlot];
        Bsdf bsdf; bsdf.setup(a.scene, mat, tu, tv);
        Frame frame(shadingNormalInWorld, tc0DirInWorld);
        if (a.f.enableBumpMapping) apply_bump_mapping(read_modified_normal(a.scene, mat, tu, tv), frame);
        const f3 vOutLocal = frame.to_local(unit(-direction));
        shadingNormalInWorld = frame.n;
        qTangent = encode_dir(frame.t);
        albedo = bsdf.dh_reflectance_estimate(vOutLocal);
    }
    else {
        const f3 vOut = -direction;
        const f3 pp = -vOut;
        float posPhi, posTheta;
        to_polar_yup(pp, posPhi, posTheta);
        const float phi = posPhi + a.f.envLightRotation;
        float u = phi / (2 * kPi);
        u -= floorf(u);
        const float v = posTheta / kPi;
        positionInWorld = pp;
        prevPositionInWorld = pp;
        qGeomNormal = encode_dir(vOut);
        shadingNormalInWorld = vOut;
        qTangent = encode_dir(f3(-gm_cos(posPhi), 0, -gm_sin(posPhi)));
        qTexCoord = encode_uv(u, v);
        qbcB = encode_bc(u);
        qbcC = encode_bc(v);
    }

    const Camera prevCam = load_camera(a.f.prevCamera);
    float sx, sy;
    calc_screen_position(prevCam, prevPositionInWorld, sx, sy);
    float mvx = (x + 0.5f) - sx * a.s.imageSizeX;
    float mvy = (y + 0.5f) - sy * a.s.imageSizeY;
    if (a.f.resetFlowBuffer || prevPositionInWorld.x != prevPositionInWorld.x) { mvx = 0.0f; mvy = 0.0f; }

    static_cast<uint4*>(a.s.gbuffer0[bufIdx])[p] = make_uint4(instSlot, geomInstSlot, primIndex, qbcB | (qbcC << 16));
    static_cast<float2*>(a.s.gbuffer1[bufIdx])[p] = make_float2(mvx, mvy);
    static_cast<float4*>(a.s.gbuffer2[bufIdx])[p] = make_float4(positionInWorld.x, positionInWorld.y, positionInWorld.z, bits2f(qGeomNormal));
    static_cast<uint4*>(a.s.gbuffer3[bufIdx])[p] = make_uint4(encode_dir(shadingNormalInWorld), qTangent, qTexCoord, matSlot);

    float4* albedoAcc = static_cast<float4*>(a.s.albedoAccumBuffer) + p;
    float4* normalAcc = static_cast<float4*>(a.s.normalAccumBuffer) + p;
    f3 prevAlbedo(0.0f), prevNormal(0.0f);
    if (a.f.numAccumFrames > 0) {
        const float4 pa = *albedoAcc, pn = *normalAcc;
        prevAlbedo = f3(pa.x, pa.y, pa.z);
        prevNormal = f3(pn.x, pn.y, pn.z);
    }
    const float curWeight = 1.0f / (1 + a.f.numAccumFrames);
    const f3 albedoResult = (1 - curWeight) * prevAlbedo + curWeight * albedo;
    const f3 normalResult = (1 - curWeight) * prevNormal + curWeight * shadingNormalInWorld;
    *albedoAcc = make_float4(albedoResult.x, albedoResult.y, albedoResult.z, 1.0f);
    *normalAcc = make_float4(normalResult.x, normalResult.y, normalResult.z, 1.0f);
}
__global__ __launch_bounds__(kBlock) void k_gbuffer_resolve(RestirArgs a) {
    const PixelId px = pixel_of_thread(a.px);
    if (!px.valid) return;
    const float4 rd = a.rayDir[px.slot];
    gbuffer_resolve(a, px, a.hits[px.slot], f3(rd.x, rd.y, rd.z));
}

// ---------------------------------------------------------------- INITIAL (+ TEMPORAL)
// candidate loop + visibility-ray emission: optix_restir_di_kernels.cu:57-133
// The light of a candidate comes out of the emitter interval table (emitter_spans.h): one guided search
// instead of the reference's three nested ones, identical result.
#ifndef GFX_INIT_WAVES   // experiment switch (tools/sessions): waves per SIMD the register allocation of the kernel targets
#define GFX_INIT_WAVES 4
#endif
// The emitter record (64 B) of a candidate is gathered by the wave cooperatively (coop_fetch.hip.h): 16 line requests per 16 lanes
// instead of 4 per lane; the normal matrix comes out of the deduplicated table the record's flags index.  At 11 gathers per lane and candidate
// the kernel ran exactly at the CU's one-line-request-per-clock time (profiles/r03_experiments.txt).  The candidate loop is therefore
// wave-uniform: lanes without a surface take part in the fetch and in nothing else.
//
// SPLIT > 1 (small launches -- a row band of a multi-GPU frame is ONE round of waves, each a chain of 32 dependent candidate
// iterations): SPLIT neighbouring lanes share a pixel and take its candidates round robin (lane j: candidates j, j + SPLIT, ...), so the
// launch has SPLIT times the waves, each 1 / SPLIT as long.  Same result, bit for bit:
//   * every candidate draws exactly four numbers, so lane j starts 4 j draws into the pixel's PCG32 stream and skips 4 (SPLIT - 1)
//     draws after each candidate (an LCG jumps n steps with one multiply-add by constants);
//   * the weights of the SPLIT candidates of a step are added to the running sum left to right through the lanes (quad-permute DPP), so
//     every candidate sees the fp32 sum the sequential loop has at its turn (Reservoir::update, restir_di_shared.h:118-125);
//   * each lane keeps the last candidate IT accepted; the pixel's sample is the one with the highest candidate index, and the lane that
//     holds it writes the pixel's outputs.
struct LcgJump { uint64_t mul, add; };
constexpr LcgJump lcg_jump(uint32_t n) {      // n steps of state -> state * 6364136223846793005 + 1 (shading.hip.h Pcg32)
    LcgJump j = { 1ull, 0ull };
    for (uint32_t k = 0; k < n; ++k) { j.add = j.add * 6364136223846793005ULL + 1ull; j.mul = j.mul * 6364136223846793005ULL; }
    return j;
}
template <int CTRL> GFX_DEV float quad_perm(float v) {      // v of the lane quad_perm names, within every aligned group of four lanes
    return bits2f(static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(f2bits(v)), CTRL, 0xF, 0xF, false)));
}
template <int CTRL> GFX_DEV int quad_perm(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
// The pixel of launch thread t when SPLIT lanes share a pixel: thread t / SPLIT of the one-lane-per-pixel launch (same pixel, same ray slot).
template <int SPLIT>
GFX_DEV PixelId pixel_of_split_thread(const PixelGrid& g, uint32_t& sub, uint32_t block) {
    if (SPLIT == 1) { sub = 0u; return pixel_of_block_thread(g, block, threadIdx.x); }
    const uint32_t launchThread = block * kBlock + threadIdx.x;
    sub = launchThread & (SPLIT - 1);
    return pixel_of_block_thread(g, (launchThread / SPLIT) / kBlock, (launchThread / SPLIT) % kBlock);
}
// What the candidate loop leaves in registers: the visibility ray of the selected candidate (want: there is one), held by the lane
// that wrote the pixel's reservoir (writer; with one lane per pixel every lane is one).
struct CandidateRay { bool writer, want; f3 org, dir; float tmax; };
// waveBuf: 256 x 16 B of LDS private to the wave (64 emitter records).  EVERY lane of the wave must call.
template <bool EMITTER_TEX, int SPLIT>
GFX_DEV CandidateRay initial_candidates(const RestirArgs& a, uint4* waveBuf, int lane, const PixelId& px, uint32_t sub) {
    static_assert(SPLIT == 1 || SPLIT == 2 || SPLIT == 4, "a pixel's lanes are an aligned pair or quad");
    const size_t numPixels = static_cast<size_t>(a.s.imageSizeX) * a.s.imageSizeY;
    const size_t p = px.p;
    const uint32_t bufIdx = a.f.bufferIndex;
    bool surface = false;
    if (px.valid) surface = static_cast<const uint4*>(a.s.gbuffer0[bufIdx])[p].x != 0xFFFFFFFFu;

    bool wantRay = false;
    f3 rayO(0.0f), rayD(0.0f);
    float rayTmax = 0;
    const EnvMap env = load_env(a.s);
    const bool envEnabled = env.present() && a.f.enableEnvLight;
    ShadingPoint sp;
    uint64_t* rngBuf = static_cast<uint64_t*>(a.s.rngBuffer);
    Pcg32 rng; rng.state = 0;
    if (surface) {
        const Camera cam = load_camera(a.f.camera);
        make_shading_point(a, bufIdx, p, cam.pos, false, sp);
        rng.state = rngBuf[p];
        if (SPLIT > 1 && sub > 0) {                       // 4 draws per candidate: this lane's first candidate is `sub`
            constexpr LcgJump j1 = lcg_jump(4), j2 = lcg_jump(8), j3 = lcg_jump(12);
            rng.state = rng.state * (sub == 1 ? j1.mul : sub == 2 ? j2.mul : j3.mul) + (sub == 1 ? j1.add : sub == 2 ? j2.add : j3.add);
        }
    }
    Reservoir reservoir;
    reservoir.reset();
    float selectedTarget = 0.0f;
    float sumWeights = 0.0f;             // SPLIT > 1: the pixel's running sum (the same in its lanes between two steps)
    int lastAccepted = -1;               // SPLIT > 1: the last candidate this lane accepted
    const uint32_t numCandidates = 1u << a.f.log2NumCandidateSamples;
    GFX_CYC_BEGIN
    for (uint32_t i = sub; i < numCandidates; i += SPLIT) {
        GFX_PROF(0);
        GFX_CYC(0);   // random numbers, light type, table lookup
        // ---- what this lane's candidate needs from the tables
        float probCurType = 1.0f, u0 = 0.0f, u1 = 0.0f;
        bool sampleEnv = false;
        LightPick pk; pk.rec = 0; pk.instSlot = 0; pk.density = 0.0f; pk.partialProb = 0.0f; pk.ok = false; pk.table = true;
        if (surface) {
            if (SPLIT > 1 && i >= SPLIT) {                // past the draws of the other lanes' candidates since this lane's last one
                constexpr LcgJump skip = lcg_jump(4 * (SPLIT - 1));
                rng.state = rng.state * skip.mul + skip.add;
            }
            float ul = rng.uniform();
            if (envEnabled) {
                if (*a.scene.lightInstIntegral > 0.0f) {
                    const float prob = fmin2(fmax2(0.25f * numCandidates - i, 0.0f), 1.0f);
                    // prob is 0 or 1 for every candidate count >= 4: x / 1 and (x - 0) / (1 - 0) are x, no division needed
                    if (prob == 1.0f) { probCurType = 0.25f; sampleEnv = true; }
                    else if (prob == 0.0f) probCurType = 1.0f - 0.25f;
                    else if (ul < prob) { probCurType = 0.25f; ul = ul / prob; sampleEnv = true; }
                    else { probCurType = 1.0f - 0.25f; ul = (ul - prob) / (1 - prob); }
                }
                else sampleEnv = true;
            }
            u0 = rng.uniform();
            u1 = rng.uniform();
            if (!sampleEnv) pk = light_select(a.scene, ul);
        }
        // ---- the wave gathers the records, then -- their flags name them -- the normal matrices
        GFX_CYC(1);   // cooperative record fetch (issue, wait, read back), matrix loads issued
        const bool fetch = surface && !sampleEnv && pk.ok;
        uint4 q0 = make_uint4(0u, 0u, 0u, 0u), q1 = q0, q2 = q0, q3 = q0;
        m33 normalMatrix;
        normalMatrix.r0 = normalMatrix.r1 = normalMatrix.r2 = f3(0.0f);
        if (__ballot(fetch) != 0ull) {
            coop_fetch64_issue(fetch ? pk.rec : kCoopNone, reinterpret_cast<const char*>(a.scene.emitterRecs), waveBuf, lane);
            coop_fetch64_wait();
            if (fetch) coop_fetch64_read(waveBuf, lane, q0, q1, q2, q3);
            // the matrix: three 16-byte loads per lane out of the deduplicated table.  Measured alternatives, all slower: a second
            // cooperative round (+6 %: its wait cannot hide behind arithmetic the way these loads do); the table in LDS with one
            // 16-wave block per CU (+5 %, 2 100 matrices: bank conflicts of the scattered reads and the coarser block granularity
            // cost more than the saved L2 sector) -- profiles/r03_experiments.txt
            if (fetch) normalMatrix = load_m33_rows(a.scene.lightNormalMatrices + 16u * emitter_matrix_index(q3.w));
        }
        // ---- the candidate itself
        GFX_CYC(2);   // point on the emitter (waits for the matrix)
        if (surface) {
            LightSample ls;
            ls.emittance = f3(0.0f); ls.position = f3(0.0f); ls.normal = f3(0.0f); ls.atInfinity = 0;
            float pd = 0.0f;
            // the emittance texture of the candidate is read only when f * G is non-zero: a zero-weight candidate is
            // never accepted by the reservoir, so its emittance is never observed (finite texels: 0 * Le = 0)
            PendingEmittance pending; pending.tex = 0u; pending.rec = 0u; pending.bcA = pending.bcB = pending.bcC = 0.0f;
            if (sampleEnv) sample_env_light(env, a.f.envLightRotation, a.f.envLightPowerCoeff, u0, u1, ls, pd);
            else if (pk.ok) {
                GFX_PROF(1);
                light_from_record<EMITTER_TEX, false>(a.scene, pk, as_float4(q0), as_float4(q1), as_float4(q2), as_float4(q3), normalMatrix, u0, u1, ls, pd,
                                                      f3(0.0f), EMITTER_TEX ? &pending : nullptr);
            }
            GFX_CYC(3);   // shadow-ray geometry, BSDF evaluation, emittance texture
            const f3 cont = EMITTER_TEX ? direct_lighting_pending(a.scene, sp.pos, sp.vOutLocal, sp.frame, sp.bsdf, ls, pending)
                                        : direct_lighting(sp.pos, sp.vOutLocal, sp.frame, sp.bsdf, ls);
            GFX_CYC(4);   // reservoir update
            pd *= probCurType;
            const float target = target_weight(cont);
            const float weight = target / pd;
            if (SPLIT == 1) {
                if (reservoir.update(ls, weight, rng.uniform())) { GFX_PROF(5); selectedTarget = target; }
            }
            else {
                const float u = rng.uniform();
                // the lanes of a pixel share `surface`, so they are all here: candidate i - sub + j of lane j joins the sum after j - 1's
                float s = sumWeights + weight;
#pragma unroll
                for (int j = 1; j < SPLIT; ++j) {
                    const float before = SPLIT == 2 ? quad_perm<0xA0>(s) : quad_perm<0x90>(s);   // lane sub - 1 of the pair / quad
                    if (sub >= static_cast<uint32_t>(j)) s = before + weight;
                }
                if (u < weight / s) { GFX_PROF(5); reservoir.sample = ls; selectedTarget = target; lastAccepted = static_cast<int>(i); }
                sumWeights = SPLIT == 2 ? quad_perm<0xF5>(s) : quad_perm<0xFF>(s);                 // the last lane's sum: after all SPLIT candidates
            }
        }
    }
    bool writer = true;                  // SPLIT > 1: the lane that holds the pixel's sample (lane 0 when no candidate was accepted)
    if (SPLIT > 1) {
        int last = lastAccepted;
        last = max(last, quad_perm<0xB1>(last));                                                   // lane ^ 1
        if (SPLIT == 4) last = max(last, quad_perm<0x4E>(last));                                   // lane ^ 2
        writer = last < 0 ? sub == 0 : lastAccepted == last;
        reservoir.sumWeights = sumWeights;
        reservoir.streamLength = numCandidates;
        if (surface && sub == SPLIT - 1) rngBuf[p] = rng.state;                                    // the state after the last candidate's fourth draw
    }
    GFX_CYC(5);       // after the loop
    if (surface && writer) {
        GFX_PROF(8);
        float recPDF = reservoir.sumWeights / (selectedTarget * reservoir.streamLength);
        if (!is_finite(recPDF)) { recPDF = 0.0f; selectedTarget = 0.0f; }

        if (a.f.reuseVisibility && selectedTarget > 0.0f) {
            const ShadowRay sr = shadow_ray(sp.pos, reservoir.sample);
            wantRay = true; rayO = sp.pos; rayD = sr.dir; rayTmax = sr.tmax;
        }
        if (SPLIT == 1) rngBuf[p] = rng.state;
        store_reservoir(a.s.reservoirBuffer[a.curRes], numPixels, p, reservoir);
        static_cast<float2*>(a.s.reservoirInfoBuffer[a.curRes])[p] = make_float2(recPDF, selectedTarget);
    }
    GFX_CYC_END;
    CandidateRay r;
    r.writer = writer; r.want = wantRay; r.org = rayO; r.dir = rayD; r.tmax = rayTmax;
    return r;
}
template <bool EMITTER_TEX, int SPLIT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(GFX_INIT_WAVES, GFX_INIT_WAVES))) void k_initial_candidates(RestirArgs a, uint32_t* __restrict__ blockCost) {
    __shared__ __attribute__((aligned(16))) uint4 fetchBuf[(kBlock / 64) * 256];   // per wave: 256 x 16 B = 64 records
    const int lane = threadIdx.x & 63;
    uint4* waveBuf = fetchBuf + 256 * __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, kept scalar
    // costliest blocks of one frame ago first (a.px.order, k_order_blocks): sky tiles cost nothing, tiles in reach of many textured emitters
    // the most, and the launch is eight rounds of blocks; the cost of a block = the clock its slowest wave spent here, in units of 2048 cycles
    const uint32_t block = launch_block(a.px);
    const unsigned long long t0 = blockCost ? __builtin_amdgcn_s_memtime() : 0ull;
    uint32_t sub;
    const PixelId px = pixel_of_split_thread<SPLIT>(a.px, sub, block);
    const CandidateRay r = initial_candidates<EMITTER_TEX, SPLIT>(a, waveBuf, lane, px, sub);
    if (r.writer) {
        const uint32_t slot = emit_ray_at_slot(px, r.want, r.org, r.dir, 0.0f, r.tmax, a);
        if (px.valid) a.pixelRaySlot[px.p] = slot;
    }
    if (blockCost && lane == 0) atomicMax(blockCost + block, static_cast<uint32_t>((__builtin_amdgcn_s_memtime() - t0) >> 11));
}

// visibility application + temporal reuse: optix_restir_di_kernels.cu:128-286
// MODE 0: performInitialRIS, 1: ...TemporalRISBiased, 2: ...TemporalRISUnbiased
// `occluded`: the visibility ray of the pixel's selected candidate found an occluder (false when the pixel had no ray)
template <int MODE>
GFX_DEV void temporal_reuse(const RestirArgs& a, const PixelId& px, bool occluded) {
    const size_t numPixels = static_cast<size_t>(a.s.imageSizeX) * a.s.imageSizeY;
    if (!px.valid) return;
    const size_t p = px.p;
    const uint32_t bufIdx = a.f.bufferIndex;
    if (static_cast<const uint4*>(a.s.gbuffer0[bufIdx])[p].x == 0xFFFFFFFFu) return;
    const int x = px.x, y = px.y;

    float2* infoBuf = static_cast<float2*>(a.s.reservoirInfoBuffer[a.curRes]);
    float2 info = infoBuf[p];
    float recPDF = info.x, selectedTarget = info.y;
    if (occluded) { recPDF = 0.0f; selectedTarget = 0.0f; }
    if (MODE == 0) {
        infoBuf[p] = make_float2(recPDF, selectedTarget);
        return;
    }
    constexpr bool unbiased = MODE == 2;
    const Camera cam = load_camera(a.f.camera);
    ShadingPoint sp;
    make_shading_point(a, bufIdx, p, cam.pos, false, sp);
    uint64_t* rngBuf = static_cast<uint64_t*>(a.s.rngBuffer);
    Pcg32 rng; rng.state = rngBuf[p];
    Reservoir reservoir = load_reservoir(a.s.reservoirBuffer[a.curRes], numPixels, p);

    const uint32_t prevBuf = (bufIdx + 1) % 2;
    const uint32_t prevRes = (a.curRes + 1) % 2;
    bool neighborIsSelected = false;
    const uint32_t selfStreamLength = reservoir.streamLength;
    if (recPDF == 0.0f) reservoir.reset();
    uint32_t combinedStreamLength = selfStreamLength;
    const uint32_t maxPrevStreamLength = 20 * selfStreamLength;
    const float2 mv = static_cast<const float2*>(a.s.gbuffer1[bufIdx])[p];
    const int nbx = f2i_sat(x + 0.5f - mv.x);
    const int nby = f2i_sat(y + 0.5f - mv.y);
    const bool accepted = test_neighbor(a, !unbiased, prevBuf, nbx, nby, sp.dist, sp.frame.n, cam.pos);
    size_t np = 0;
    Reservoir neighbor;
    if (accepted) {
        np = static_cast<size_t>(nby) * a.s.imageSizeX + nbx;
        neighbor = load_reservoir(a.s.reservoirBuffer[prevRes], numPixels, np);
        const float2 nbInfo = static_cast<const float2*>(a.s.reservoirInfoBuffer[prevRes])[np];
        const f3 cont = direct_lighting(sp.pos, sp.vOutLocal, sp.frame, sp.bsdf, neighbor.sample);
        const float target = target_weight(cont);
        const uint32_t nbStreamLength = min(neighbor.streamLength, maxPrevStreamLength);
        const float weight = target * nbInfo.x * nbStreamLength;
        if (reservoir.update(neighbor.sample, weight, rng.uniform())) {
            selectedTarget = target;
            if (unbiased) neighborIsSelected = true;
        }
        combinedStreamLength += nbStreamLength;
    }
    reservoir.streamLength = combinedStreamLength;

    float weightForEstimate;
    if (unbiased) {
        const LightSample selected = reservoir.sample;
        float numWeight, denomWeight;
        {
            const f3 cont = direct_lighting(sp.pos, sp.vOutLocal, sp.frame, sp.bsdf, selected);
            const float targetSelf = target_weight(cont);
            numWeight = targetSelf;                          // useMIS_RIS = true (:10, 210-213)
            denomWeight = targetSelf * selfStreamLength;
        }
        if (accepted) {
            const Camera prevCam = load_camera(a.f.prevCamera);
            ShadingPoint nsp;
            make_shading_point(a, prevBuf, np, prevCam.pos, true, nsp);
            const f3 cont = direct_lighting(nsp.pos, nsp.vOutLocal, nsp.frame, nsp.bsdf, selected);
            const float nbTarget = target_weight(cont);
            const uint32_t nbStreamLength = min(neighbor.streamLength, maxPrevStreamLength);
            denomWeight += nbTarget * nbStreamLength;
            if (neighborIsSelected) numWeight = nbTarget;
        }
        weightForEstimate = numWeight / denomWeight;
    }
    else weightForEstimate = 1.0f / reservoir.streamLength;

    recPDF = weightForEstimate * reservoir.sumWeights / selectedTarget;
    if (!is_finite(recPDF)) { recPDF = 0.0f; selectedTarget = 0.0f; }

    rngBuf[p] = rng.state;
    store_reservoir(a.s.reservoirBuffer[a.curRes], numPixels, p, reservoir);
    infoBuf[p] = make_float2(recPDF, selectedTarget);
}
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_temporal(RestirArgs a) {
    const PixelId px = pixel_of_thread(a.px);
    if (!px.valid) return;
    const uint32_t slot = a.pixelRaySlot[px.p];
    temporal_reuse<MODE>(a, px, slot != GFX_INVALID_SLOT && a.occluded[slot] != 0u);
}

// ---------------------------------------------------------------- SPATIAL
GFX_DEV void spatial_neighbor(const RestirArgs& a, Pcg32& rng, uint32_t nIdx, int x, int y, int& nbx, int& nby) {
    float radius = a.f.spatialNeighborRadius;
    float dx, dy;
    if (a.f.useLowDiscrepancyNeighbors) {
        const float2 d = static_cast<const float2*>(a.s.spatialNeighborDeltas)[(a.baseIdx + nIdx) % 1024];
        dx = radius * d.x;
        dy = radius * d.y;
    }
    else {
        radius *= sqrtf(rng.uniform());
        const float angle = 2 * kPi * rng.uniform();
        float s, c;
        gm_sincos(angle, s, c);
        dx = radius * c;
        dy = radius * s;
    }
    nbx = f2i_sat(x + 0.5f + dx);
    nby = f2i_sat(y + 0.5f + dy);
}

// optix_restir_di_kernels.cu:303-547.  UNBIASED: combines, then emits the MIS-denominator rays.
// EVERY thread of the block must call the UNBIASED form (its ray queue reservation is a block-wide operation).
// What the combination of a pixel's own reservoir with its neighbours' leaves (the first half of the pass)
struct SpatialSelection {
    bool surface;
    ShadingPoint sp;
    Pcg32 rng;
    Reservoir combined;
    float selectedTarget;
    int32_t selectedNeighborIndex;      // UNBIASED: which neighbour's sample was selected last, -1 = the pixel's own
    uint32_t selfStreamLength;
};
template <bool UNBIASED>
GFX_DEV void spatial_select(const RestirArgs& a, const PixelId& px, const Camera& cam, SpatialSelection& s) {
    const size_t numPixels = static_cast<size_t>(a.s.imageSizeX) * a.s.imageSizeY;
    const size_t p = px.p;
    const uint32_t bufIdx = a.f.bufferIndex;
    const uint32_t numNb = a.f.numSpatialNeighbors;
    s.surface = false;
    if (px.valid) s.surface = static_cast<const uint4*>(a.s.gbuffer0[bufIdx])[p].x != 0xFFFFFFFFu;
    const int x = px.x, y = px.y;
    const uint32_t srcRes = a.curRes;
    s.rng.state = 0;
    s.combined.reset();
    s.selectedTarget = 0.0f;
    s.selectedNeighborIndex = -1;
    s.selfStreamLength = 0;
    const uint64_t* rngBuf = static_cast<const uint64_t*>(a.s.rngBuffer);
    if (s.surface) {
        make_shading_point(a, bufIdx, p, cam.pos, false, s.sp);
        s.rng.state = rngBuf[p];
        const Reservoir self = load_reservoir(a.s.reservoirBuffer[srcRes], numPixels, p);
        const float2 selfInfo = static_cast<const float2*>(a.s.reservoirInfoBuffer[srcRes])[p];
        if (selfInfo.x > 0.0f) { s.combined = self; s.selectedTarget = selfInfo.y; }
        s.selfStreamLength = self.streamLength;
        uint32_t combinedStreamLength = self.streamLength;
        for (uint32_t nIdx = 0; nIdx < numNb; ++nIdx) {
            int nbx, nby;
            spatial_neighbor(a, s.rng, nIdx, x, y, nbx, nby);
            const bool accepted = test_neighbor(a, !UNBIASED, bufIdx, nbx, nby, s.sp.dist, s.sp.frame.n, cam.pos) && (nbx != x || nby != y);
            if (accepted) {
                const size_t np = static_cast<size_t>(nby) * a.s.imageSizeX + nbx;
                const Reservoir neighbor = load_reservoir(a.s.reservoirBuffer[srcRes], numPixels, np);
                const float2 nbInfo = static_cast<const float2*>(a.s.reservoirInfoBuffer[srcRes])[np];
                const f3 cont = direct_lighting(s.sp.pos, s.sp.vOutLocal, s.sp.frame, s.sp.bsdf, neighbor.sample);
                const float target = target_weight(cont);
                const uint32_t nbStreamLength = neighbor.streamLength;
                const float weight = target * nbInfo.x * nbStreamLength;
                if (s.combined.update(neighbor.sample, weight, s.rng.uniform())) {
                    s.selectedTarget = target;
                    if (UNBIASED) s.selectedNeighborIndex = static_cast<int32_t>(nIdx);
                }
                combinedStreamLength += nbStreamLength;
            }
        }
        s.combined.streamLength = combinedStreamLength;
    }
}
// MIS term k of the unbiased pass (0 = the pixel itself, 1 + nIdx = neighbour nIdx): the target density of the selected sample there, that
// pixel's stream length, and the visibility ray the term needs.  Terms are formed in the order k = 0, 1, ...: a neighbour term draws its
// position from the pixel's stream (random-neighbour mode).
struct MisTerm { float target; uint32_t streamLength; bool want, evaluated; f3 ro, rd; float tmax; };
GFX_DEV MisTerm spatial_mis_term(const RestirArgs& a, const PixelId& px, SpatialSelection& s, const Camera& prevCam, uint32_t k) {
    const size_t numPixels = static_cast<size_t>(a.s.imageSizeX) * a.s.imageSizeY;
    const uint32_t bufIdx = a.f.bufferIndex;
    const bool needMis = s.surface && s.selectedTarget > 0.0f;
    const LightSample selected = s.combined.sample;
    MisTerm t; t.target = 0.0f; t.streamLength = 0; t.want = false; t.evaluated = false; t.ro = f3(0.0f); t.rd = f3(0.0f); t.tmax = 0;
    if (k == 0) {
        t.streamLength = s.selfStreamLength; t.evaluated = true;
        if (needMis) {
            const f3 cont = direct_lighting(s.sp.pos, s.sp.vOutLocal, s.sp.frame, s.sp.bsdf, selected);
            t.target = target_weight(cont);
            if (a.f.reuseVisibility && t.target > 0.0f) {
                const ShadowRay sr = shadow_ray(s.sp.pos, selected);
                t.want = true; t.ro = s.sp.pos; t.rd = sr.dir; t.tmax = sr.tmax;
            }
        }
        return t;
    }
    if (needMis) {
        const int x = px.x, y = px.y;
        int nbx, nby;
        spatial_neighbor(a, s.rng, k - 1, x, y, nbx, nby);
        const bool accepted = (nbx >= 0 && nbx < a.s.imageSizeX && nby >= 0 && nby < a.s.imageSizeY) && (nbx != x || nby != y);
        if (accepted) {
            const size_t np = static_cast<size_t>(nby) * a.s.imageSizeX + nbx;
            if (static_cast<const uint4*>(a.s.gbuffer0[bufIdx])[np].x != 0xFFFFFFFFu) {
                ShadingPoint nsp;
                make_shading_point(a, bufIdx, np, prevCam.pos, true, nsp);   // prevCamera as in the reference (:487)
                const Reservoir neighbor = load_reservoir(a.s.reservoirBuffer[a.curRes], numPixels, np);
                const f3 cont = direct_lighting(nsp.pos, nsp.vOutLocal, nsp.frame, nsp.bsdf, selected);
                t.target = target_weight(cont);
                t.streamLength = neighbor.streamLength;
                t.evaluated = true;
                if (a.f.reuseVisibility && t.target > 0.0f) {
                    const ShadowRay sr = shadow_ray(nsp.pos, selected);
                    t.want = true; t.ro = nsp.pos; t.rd = sr.dir; t.tmax = sr.tmax;
                }
            }
        }
    }
    return t;
}
// The MIS weight of the unbiased pass as its terms come in (optix_restir_di_kernels.cu:413-546): term k with its ray's answer
struct MisSum {
    float numWeight, denomWeight; bool visibility;
    GFX_DEV void begin() { numWeight = 0.0f; denomWeight = 0.0f; visibility = true; }
    GFX_DEV void add(const RestirArgs& a, uint32_t k, float target, uint32_t streamLength, bool evaluated, bool occluded, int32_t selectedNeighborIndex) {
        if (k == 0) {
            float targetSelf = target;
            if (occluded) targetSelf = 0.0f;
            if (a.f.reuseVisibility) visibility = targetSelf > 0.0f;
            numWeight = targetSelf;
            denomWeight = targetSelf * streamLength;
            return;
        }
        if (!evaluated) return;                  // out of bounds / self / background: the reference `continue`s
        float nbTarget = target;
        if (occluded) nbTarget = 0.0f;
        denomWeight += nbTarget * streamLength;
        if (static_cast<int32_t>(k - 1) == selectedNeighborIndex) numWeight = nbTarget;
    }
    GFX_DEV float weight(const RestirArgs& a) const {
        float w = numWeight / denomWeight;
        if (a.f.reuseVisibility && !visibility) w = 0.0f;
        return w;
    }
};
template <bool UNBIASED>
GFX_DEV void spatial_reuse(const RestirArgs& a, const PixelId& px) {
    const size_t numPixels = static_cast<size_t>(a.s.imageSizeX) * a.s.imageSizeY;
    const size_t p = px.p;
    const uint32_t numNb = a.f.numSpatialNeighbors;
    const uint32_t dstRes = (a.curRes + 1) % 2;
    const Camera cam = load_camera(a.f.camera);
    uint64_t* rngBuf = static_cast<uint64_t*>(a.s.rngBuffer);
    SpatialSelection sel;
    spatial_select<UNBIASED>(a, px, cam, sel);
    const bool surface = sel.surface;
    Reservoir& combined = sel.combined;

    if (!UNBIASED) {
        if (!surface) return;
        const float weightForEstimate = 1.0f / combined.streamLength;
        float recPDF = weightForEstimate * combined.sumWeights / sel.selectedTarget;
        float target = sel.selectedTarget;
        if (!is_finite(recPDF)) { recPDF = 0.0f; target = 0.0f; }
        rngBuf[p] = sel.rng.state;
        store_reservoir(a.s.reservoirBuffer[dstRes], numPixels, p, combined);
        static_cast<float2*>(a.s.reservoirInfoBuffer[dstRes])[p] = make_float2(recPDF, target);
        return;
    }

    // ---- unbiased: targets of the selected sample at self and at every neighbour, rays where needed
    SpatialSlot* slots = a.spatialScratch + (px.valid ? p * (numNb + 1) : 0);
    const Camera prevCam = load_camera(a.f.prevCamera);
    auto store_term = [&](uint32_t k, const MisTerm& t, uint32_t slot) {
        if (!t.evaluated) slot = kSlotSkipped;   // out of bounds / self / background: the reference `continue`s
        if (px.valid) { SpatialSlot s; s.targetDensity = t.target; s.streamLength = t.streamLength; s.raySlot = slot; slots[k] = s; }
    };
    constexpr int kBatch = 4;                    // self + the reference's three neighbours: one queue reservation
    if (numNb + 1 <= static_cast<uint32_t>(kBatch)) {
        MisTerm terms[kBatch];
        bool want[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            if (static_cast<uint32_t>(k) <= numNb) terms[k] = spatial_mis_term(a, px, sel, prevCam, static_cast<uint32_t>(k));
            else { terms[k].want = false; terms[k].evaluated = false; }
            want[k] = terms[k].want;
        }
        uint32_t raySlots[kBatch];
        queue_reserve<kBatch>(want, a.rayCount, raySlots);
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            if (static_cast<uint32_t>(k) > numNb) continue;
            queue_write(raySlots[k], terms[k].ro, terms[k].rd, 0.0f, terms[k].tmax, a.rayOrg, a.rayDir);
            store_term(static_cast<uint32_t>(k), terms[k], raySlots[k]);
        }
    }
    else {
        {
            const MisTerm t = spatial_mis_term(a, px, sel, prevCam, 0u);
            store_term(0, t, emit_ray(t.want, t.ro, t.rd, 0.0f, t.tmax, a));
        }
        for (uint32_t k = 1; k <= numNb; ++k) {
            const MisTerm t = spatial_mis_term(a, px, sel, prevCam, k);
            store_term(k, t, emit_ray(t.want, t.ro, t.rd, 0.0f, t.tmax, a));
        }
    }
    if (!surface) return;
    rngBuf[p] = sel.rng.state;
    store_reservoir(a.s.reservoirBuffer[dstRes], numPixels, p, combined);
    // stash (selectedNeighborIndex, selectedTarget) for the finishing kernel
    static_cast<float2*>(a.s.reservoirInfoBuffer[dstRes])[p] = make_float2(bits2f(static_cast<uint32_t>(sel.selectedNeighborIndex)), sel.selectedTarget);
}
// (four waves per SIMD for both forms: the unbiased one sits at the 128-register boundary)
template <bool UNBIASED>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4))) void k_spatial(RestirArgs a) { spatial_reuse<UNBIASED>(a, pixel_of_thread(a.px)); }

// MIS weights of the unbiased spatial pass once the rays are back: optix_restir_di_kernels.cu:413-546
__global__ __launch_bounds__(kBlock) void k_spatial_mis_finish(RestirArgs a) {
    const size_t numPixels = static_cast<size_t>(a.s.imageSizeX) * a.s.imageSizeY;
    const PixelId px = pixel_of_thread(a.px);
    if (!px.valid) return;
    const size_t p = px.p;
    const uint32_t bufIdx = a.f.bufferIndex;
    if (static_cast<const uint4*>(a.s.gbuffer0[bufIdx])[p].x == 0xFFFFFFFFu) return;
    const uint32_t numNb = a.f.numSpatialNeighbors;
    const uint32_t dstRes = (a.curRes + 1) % 2;
    float2* infoBuf = static_cast<float2*>(a.s.reservoirInfoBuffer[dstRes]);
    const float2 stash = infoBuf[p];
    const int32_t selectedNeighborIndex = static_cast<int32_t>(f2bits(stash.x));
    const float selectedTarget = stash.y;
    const float sumWeights = static_cast<const float4*>(a.s.reservoirBuffer[dstRes])[2 * numPixels + p].z;
    const SpatialSlot* slots = a.spatialScratch + p * (numNb + 1);

    float weightForEstimate = 0.0f;
    if (selectedTarget > 0.0f) {
        MisSum sum;
        sum.begin();
        for (uint32_t k = 0; k <= numNb; ++k) {
            const SpatialSlot s = slots[k];
            const bool rayed = s.raySlot != GFX_INVALID_SLOT && s.raySlot != kSlotSkipped;
            sum.add(a, k, s.targetDensity, s.streamLength, s.raySlot != kSlotSkipped, rayed && a.occluded[s.raySlot] != 0u, selectedNeighborIndex);
        }
        weightForEstimate = sum.weight(a);
    }
    float recPDF = weightForEstimate * sumWeights / selectedTarget;
    float target = selectedTarget;
    if (!is_finite(recPDF)) { recPDF = 0.0f; target = 0.0f; }
    infoBuf[p] = make_float2(recPDF, target);
}

// ---------------------------------------------------------------- SHADING
// optix_restir_di_kernels.cu:559-629 up to the final shadow ray
// What the first half of the shading pass leaves for the second: the emitted / environment term, the unshadowed direct term with its
// reciprocal PDF estimate, and the final shadow ray (want: there is one).
struct ShadeState { bool want; f3 ro, rd; float tmax; f3 contribution, direct; float recPDF; };
GFX_DEV ShadeState shade_prepare(const RestirArgs& a, const PixelId& px, uint32_t curRes) {
    const size_t numPixels = static_cast<size_t>(a.s.imageSizeX) * a.s.imageSizeY;
    const size_t p = px.p;
    const uint32_t bufIdx = a.f.bufferIndex;
    bool want = false;
    f3 ro(0.0f), rd(0.0f); float tmax = 0;
    f3 contribution(0.01f, 0.01f, 0.01f);
    f3 direct(0.0f);
    float recPDF = 0.0f;
    if (px.valid) {
        const uint32_t instSlot = static_cast<const uint4*>(a.s.gbuffer0[bufIdx])[p].x;
        const uint4 g3 = static_cast<const uint4*>(a.s.gbuffer3[bufIdx])[p];
        if (instSlot != 0xFFFFFFFFu) {
            const Camera cam = load_camera(a.f.camera);
            ShadingPoint sp;
            make_shading_point(a, bufIdx, p, cam.pos, true, sp);
            const gfx_material& mat = a.scene.materials[g3.w];
            const Reservoir reservoir = load_reservoir(a.s.reservoirBuffer[curRes], numPixels, p);
            recPDF = static_cast<const float2*>(a.s.reservoirInfoBuffer[curRes])[p].x;
            contribution = f3(0.0f);
            if (sp.vOutLocal.z > 0) {
                float tu, tv;
                decode_uv(g3.z, tu, tv);
                const f3 e = material_emittance(a.scene, mat, tu, tv);
                contribution = contribution + e / kPi;
            }
            if (recPDF > 0 && is_finite(recPDF)) {
                const bool visDone = a.f.reuseVisibility && (!a.f.enableTemporalReuse || (a.f.enableSpatialReuse && a.f.useUnbiasedEstimator));
                direct = direct_lighting(sp.pos, sp.vOutLocal, sp.frame, sp.bsdf, reservoir.sample);
                if (!visDone) {
                    // performDirectLighting<.., true> traces unconditionally; the result only matters
                    // when the unshadowed term is non-zero
                    if (direct.x != 0.0f || direct.y != 0.0f || direct.z != 0.0f) {
                        const ShadowRay sr = shadow_ray(sp.pos, reservoir.sample);
                        want = true; ro = sp.pos; rd = sr.dir; tmax = sr.tmax;
                    }
                }
            }
        }
        else {
            const EnvMap env = load_env(a.s);
            if (env.present() && a.f.enableEnvLight) {
                const float u = (g3.z & 0xFFFF) / 65535.0f, v = (g3.z >> 16) / 65535.0f;
                contribution = a.f.envLightPowerCoeff * env.fetch(u, v);
            }
        }
    }
    ShadeState st;
    st.want = want; st.ro = ro; st.rd = rd; st.tmax = tmax; st.contribution = contribution; st.direct = direct; st.recPDF = recPDF;
    return st;
}
__global__ __launch_bounds__(kBlock) void k_shade_prepare(RestirArgs a) {
    const PixelId px = pixel_of_thread(a.px);
    const ShadeState st = shade_prepare(a, px, a.curRes);
    const uint32_t slot = emit_ray_at_slot(px, st.want, st.ro, st.rd, 0.0f, st.tmax, a);
    if (px.valid) {
        a.shadeScratch[2 * px.p] = make_float4(st.contribution.x, st.contribution.y, st.contribution.z, bits2f(slot));
        a.shadeScratch[2 * px.p + 1] = make_float4(st.direct.x, st.direct.y, st.direct.z, st.recPDF);
    }
}

// GFX_RESTIR_SPATIAL_BIASED_AND_SHADING at full size: the pixel's last spatial pass and the set-up of its shading in one kernel -- the thread
// that wrote the pixel's entry of the other reservoir forms the shading terms from it, instead of a second per-pixel launch reading the
// pixel's G-buffer, material textures and reservoir again (the shadow ray keeps k_trace, whose refill is worth more than a launch at this
// size: profiles/r04_experiments.txt 19).
__global__ __launch_bounds__(kBlock) void k_spatial_shade_prepare(RestirArgs a) {
    const PixelId px = pixel_of_thread(a.px);
    spatial_reuse<false>(a, px);
    const ShadeState st = shade_prepare(a, px, (a.curRes + 1) % 2);
    const uint32_t slot = emit_ray_at_slot(px, st.want, st.ro, st.rd, 0.0f, st.tmax, a);
    if (px.valid) {
        a.shadeScratch[2 * px.p] = make_float4(st.contribution.x, st.contribution.y, st.contribution.z, bits2f(slot));
        a.shadeScratch[2 * px.p + 1] = make_float4(st.direct.x, st.direct.y, st.direct.z, st.recPDF);
    }
}

// contribution += recPDFEstimate * directCont; running mean (optix_restir_di_kernels.cu:619-636).  `occluded`: the final shadow ray
// found an occluder (false when the pixel had no ray).
GFX_DEV void shade_finish(const RestirArgs& a, const PixelId& px, f3 contribution, f3 direct, float recPDF, bool occluded) {
    const size_t p = px.p;
    const uint32_t bufIdx = a.f.bufferIndex;
    if (static_cast<const uint4*>(a.s.gbuffer0[bufIdx])[p].x != 0xFFFFFFFFu) {
        if (occluded) direct = f3(0.0f);
        contribution = contribution + recPDF * direct;
    }
    float4* beauty = static_cast<float4*>(a.s.beautyAccumBuffer) + p;
    f3 prev(0.0f);
    if (a.f.numAccumFrames > 0) { const float4 b = *beauty; prev = f3(b.x, b.y, b.z); }
    const float curWeight = 1.0f / (1 + a.f.numAccumFrames);
    const f3 result = (1 - curWeight) * prev + curWeight * contribution;
    *beauty = make_float4(result.x, result.y, result.z, 1.0f);
}
__global__ __launch_bounds__(kBlock) void k_shade_finish(RestirArgs a) {
    const PixelId px = pixel_of_thread(a.px);
    if (!px.valid) return;
    const float4 c0 = a.shadeScratch[2 * px.p], c1 = a.shadeScratch[2 * px.p + 1];
    const uint32_t slot = f2bits(c0.w);
    shade_finish(a, px, f3(c0.x, c0.y, c0.z), f3(c1.x, c1.y, c1.z), c1.w, slot != GFX_INVALID_SLOT && a.occluded[slot] != 0u);
}

// ---------------------------------------------------------------- the ray passes as ONE kernel each
// trace_local.hip.h: the kernel that makes a ray traces it and consumes the result.  The G-buffer pass runs this way at every size
// (primary rays are coherent: the wave-local traversal loses little to the missing refill); the candidate and shading passes where the
// launch is a row band of a multi-GPU frame -- its frame is 4 launches instead of 11 and none of them waits for the slowest wave of a
// traversal before the next short kernel may start (restir_launch decides).  Same buffers out as the three-kernel form; the ray queue,
// the per-pixel ray slots, the occlusion words and the shading scratch are skipped: they were only the kernels' way of talking to each
// other.  `spill`: spillCap (trace_local.hip.h local_spill_depth) stack entries per tracing thread of the launch.
__global__ __launch_bounds__(kBlock) void k_gbuffer_fused(RestirArgs a, DevAccel accel, gfx_hit* hits, uint2* spill, int spillCap, int useHint, uint32_t* __restrict__ blockCost) {
    __shared__ uint2 ldsStack[kLdsStackDepth * kBlock];
    __shared__ __attribute__((aligned(16))) uint4 fetchBuf[(kBlock / 64) * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    uint4* waveBuf = fetchBuf + 256 * __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t block = launch_block(a.px);                                     // costliest blocks of one frame ago first (k_order_blocks)
    const PixelId px = pixel_of_block_thread(a.px, block, threadIdx.x);
    const RayPair r = primary_ray(a, px);
    // the triangle this pixel's primary ray hit one frame ago is tested right after the root (trace.hip: temporal hint)
    const uint32_t hint = useHint ? hits[px.slot].triIndex : 0xFFFFFFFFu;
    uint32_t steps = 0;
    const RayHit h = trace_wave_local<false>(accel, true, f3(r.org.x, r.org.y, r.org.z), f3(r.dir.x, r.dir.y, r.dir.z), r.org.w, r.dir.w, ldsStack + tid, kBlock,
                                             spill + (static_cast<size_t>(blockIdx.x) * kBlock + tid) * spillCap, spillCap, waveBuf, lane, hint, &steps);
    if (blockCost && lane == 0) atomicMax(blockCost + block, steps);
    gfx_hit gh; gh.dist = h.t; gh.bcB = h.bcB; gh.bcC = h.bcC; gh.triIndex = h.tri;
    hits[px.slot] = gh;                          // the next frame's hint
    if (px.valid) gbuffer_resolve(a, px, gh, f3(r.dir.x, r.dir.y, r.dir.z));
}

// SPLIT = 4 lanes per pixel in the candidate loop: the 64 rays of a block are handed to its first wave through LDS, the other three
// waves leave (their slots go to the next block), the first one traces and runs the temporal pass of those 64 pixels.  The hand-over
// and the traversal stack live in the record buffers of the waves that have left, so a block holds 16 KB of LDS, not 24: ten blocks
// fit a CU, and the one-wave tails of the blocks that trace do not keep new blocks out.
template <bool EMITTER_TEX, int SPLIT, int MODE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(GFX_INIT_WAVES, GFX_INIT_WAVES))) void k_initial_fused(RestirArgs a, DevAccel accel, uint2* spill, int spillCap, uint32_t* __restrict__ blockCost) {
    static_assert(SPLIT == 1 || SPLIT == 4, "one lane per pixel, or four with the block's rays gathered in its first wave");
    constexpr int kRays = kBlock / SPLIT;
    __shared__ __attribute__((aligned(16))) uint4 fetchBuf[(kBlock / 64) * 256];           // per wave: 256 x 16 B = 64 records
    __shared__ uint2 ownStack[SPLIT == 1 ? kLdsStackDepth * kBlock : 1];
    static_assert(SPLIT == 1 || kLdsStackDepth * kRays * sizeof(uint2) <= 2 * 256 * sizeof(uint4), "the stack of 64 rays fits the buffers of waves 1 and 2");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint4* waveBuf = fetchBuf + 256 * wave;
    // Which pixels this block works for: the blocks of a launch start in index order, a few rounds of them per CU, and the launch ends
    // when the last tracing wave has walked its longest ray -- so the blocks whose rays took longest one frame ago go first
    // (a.px.order: a permutation made by k_order_blocks from the step counts this kernel leaves in blockCost; null: index order)
    const uint32_t block = launch_block(a.px);
    uint32_t sub;
    PixelId px = pixel_of_split_thread<SPLIT>(a.px, sub, block);
    const CandidateRay r = initial_candidates<EMITTER_TEX, SPLIT>(a, waveBuf, lane, px, sub);
    float4 org = make_float4(r.org.x, r.org.y, r.org.z, 0.0f), dir = make_float4(r.dir.x, r.dir.y, r.dir.z, r.want ? r.tmax : -1.0f);
    uint2* stack = ownStack + tid;
    if (SPLIT > 1) {
        // ray k of the block (pixel k of its 64) comes from lanes 4 k .. 4 k + 3 = wave k / 16: it waits in the last 512 bytes of that
        // wave's own buffer, which the wave is done with
        float4* handOver = reinterpret_cast<float4*>(fetchBuf + 256 * (tid >> 6) + 224);
        if (r.writer) { handOver[2 * ((tid & 63) / SPLIT)] = org; handOver[2 * ((tid & 63) / SPLIT) + 1] = dir; }
        __syncthreads();                           // also: the reservoirs the writer lanes stored are visible to the block
        if (tid >= kRays) return;
        const float4* from = reinterpret_cast<const float4*>(fetchBuf + 256 * (tid >> 4) + 224);
        org = from[2 * (tid & 15)]; dir = from[2 * (tid & 15) + 1];
        const uint32_t t = block * kRays + tid;        // thread of the one-lane-per-pixel launch
        px = pixel_of_block_thread(a.px, t / kBlock, t % kBlock);
        stack = reinterpret_cast<uint2*>(fetchBuf + 256) + tid;   // buffers of waves 1 and 2 (their hand-over slots have just been read)
    }
    const bool want = dir.w > org.w;
    uint32_t steps = 0;
    const RayHit h = trace_wave_local<true>(accel, want, f3(org.x, org.y, org.z), f3(dir.x, dir.y, dir.z), org.w, dir.w, stack, kRays,
                                            spill + (static_cast<size_t>(blockIdx.x) * kRays + tid) * spillCap, spillCap, waveBuf, lane, 0xFFFFFFFFu, &steps);
    if (blockCost && lane == 0) atomicMax(blockCost + block, steps);      // (one tracing wave per block with four lanes per pixel, four with one)
    temporal_reuse<MODE>(a, px, want && h.tri != GFX_INVALID_SLOT);
}

// blockOrder = the blocks of the launch by decreasing cost (counting sort over min(cost, 255): the order inside a class is whatever the
// LDS atomics make it -- any permutation gives the same image); the costs are cleared for the next frame.  One block of 1024 threads.
__global__ __launch_bounds__(1024) void k_order_blocks(uint32_t* __restrict__ cost, uint32_t n, uint32_t* __restrict__ order) {
    __shared__ uint32_t start[256];
    if (threadIdx.x < 256) start[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024) atomicAdd(&start[255u - min(cost[i], 255u)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (int c = 0; c < 256; ++c) { const uint32_t k = start[c]; start[c] = sum; sum += k; }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        order[atomicAdd(&start[255u - min(cost[i], 255u)], 1u)] = i;
        cost[i] = 0u;
    }
}

// Cost-ordered block start.  A launch of several rounds of blocks ends when its last block does, and which block that is was left to the
// index order (in which the sky comes first and the street last).  A kernel that takes part leaves a cost per block in `cost` (atomicMax:
// steps of its longest ray, or the clock its slowest wave spent); k_order_blocks turns the costs into the NEXT launch's order on the
// context's side stream, and hardware block b of that launch works as block order[b] (PixelGrid::order) -- a permutation of who works for
// which pixels, on which no result can depend.  `which`: Context::blockOrders slot (one launch shape each, `key`).  order / cost come back
// null when the launch is too small (minBlocks) or "block_order" is 0; order alone is null for the first launch of a shape.
void block_order_begin(Context& ctx, hipStream_t stream, int which, uint32_t blocks, uint64_t key, uint32_t minBlocks, const uint32_t*& order, uint32_t*& cost) {
    order = nullptr; cost = nullptr;
    if (!ctx.tune.blockOrder || blocks <= minBlocks) return;
    Context::BlockOrder& bo = ctx.blockOrders[which];
    if (bo.key != key || bo.blocks != blocks) {
        if (bo.ordered) GFX_HIP(hipStreamWaitEvent(stream, bo.ordered, 0));    // a sort of the old shape may still be running
        bo.cost.reserve(sizeof(uint32_t) * blocks); bo.order.reserve(sizeof(uint32_t) * blocks);
        GFX_HIP(hipMemsetAsync(bo.cost.p, 0, sizeof(uint32_t) * blocks, stream));
        bo.key = key; bo.blocks = blocks; bo.valid = false;
    }
    cost = bo.cost.as<uint32_t>();
    if (bo.valid) { GFX_HIP(hipStreamWaitEvent(stream, bo.ordered, 0)); order = bo.order.as<uint32_t>(); }
}
void block_order_end(Context& ctx, hipStream_t stream, int which, uint32_t blocks, uint32_t* cost) {
    if (!cost) return;
    Context::BlockOrder& bo = ctx.blockOrders[which];
    // On the context's ONE side stream, which the path tracers' NEE traces also use (pathtrace.hip).  A stream of their own for these
    // 10-us sorts was tried in round 5 and cost the NRC frame 0.4 ms: HIP multiplexes its streams onto four hardware queues, a fifth
    // stream moves the side stream onto the caller's queue (whichever streams were created first keep a queue to themselves), and the
    // NEE trace then no longer runs beside the extension trace (profiles/r05_experiments.txt 10).  A sort queued behind a frame's NEE
    // traces is still done long before the next frame's launch waits for it.
    if (!ctx.auxStream) GFX_HIP(hipStreamCreateWithFlags(&ctx.auxStream, hipStreamNonBlocking));
    if (!bo.counted) { GFX_HIP(hipEventCreateWithFlags(&bo.counted, hipEventDisableTiming)); GFX_HIP(hipEventCreateWithFlags(&bo.ordered, hipEventDisableTiming)); }
    GFX_HIP(hipEventRecord(bo.counted, stream));
    GFX_HIP(hipStreamWaitEvent(ctx.auxStream, bo.counted, 0));
    hipLaunchKernelGGL(k_order_blocks, dim3(1), dim3(1024), 0, ctx.auxStream, cost, blocks, bo.order.as<uint32_t>());
    GFX_HIP(hipGetLastError());
    GFX_HIP(hipEventRecord(bo.ordered, ctx.auxStream));
    bo.valid = true;
}

// SPATIAL_FIRST: GFX_RESTIR_SPATIAL_BIASED_AND_SHADING -- the pixel's last spatial pass (it reads neighbours in reservoir a.curRes and
// writes the pixel's own entry of the other one), then the shading of that entry by the same thread.
template <bool SPATIAL_FIRST>
__global__ __launch_bounds__(kBlock) void k_shading_fused(RestirArgs a, DevAccel accel, uint2* spill, int spillCap, uint32_t* __restrict__ blockCost) {
    __shared__ uint2 ldsStack[kLdsStackDepth * kBlock];
    __shared__ __attribute__((aligned(16))) uint4 fetchBuf[(kBlock / 64) * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    uint4* waveBuf = fetchBuf + 256 * __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t block = launch_block(a.px);                                     // costliest blocks of one frame ago first (k_initial_fused)
    const PixelId px = pixel_of_block_thread(a.px, block, threadIdx.x);
    if (SPATIAL_FIRST) spatial_reuse<false>(a, px);
    const ShadeState st = shade_prepare(a, px, SPATIAL_FIRST ? (a.curRes + 1) % 2 : a.curRes);
    uint32_t steps = 0;
    const RayHit h = trace_wave_local<true>(accel, st.want, st.ro, st.rd, 0.0f, st.tmax, ldsStack + tid, kBlock,
                                            spill + (static_cast<size_t>(blockIdx.x) * kBlock + tid) * spillCap, spillCap, waveBuf, lane, 0xFFFFFFFFu, &steps);
    if (blockCost && lane == 0) atomicMax(blockCost + block, steps);
    if (px.valid) shade_finish(a, px, st.contribution, st.direct, st.recPDF, st.want && h.tri != GFX_INVALID_SLOT);
}

#ifdef GFX_LANE_PROFILE   // experiment builds only (gm_math.hip.h GFX_PROF, tools/lane_profile.py)
extern "C" int gfx_debug_lane_profile(unsigned long long* out64, int reset) {
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_laneProfile), sizeof(g_laneProfile)) != hipSuccess) return 1;
    if (reset) { unsigned long long zero[64] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_laneProfile), zero, sizeof(zero)) != hipSuccess) return 1; }
    return 0;
}
#endif

// ---------------------------------------------------------------- host sequencing
static RestirArgs make_args(Context& ctx, uint32_t width, uint32_t height, uint32_t rowBegin, uint32_t rowEnd, bool rearch, uint32_t gapBegin = 0, uint32_t gapEnd = 0) {
    const RestirParams& rp = ctx.restir;
    if (!rp.valid) throw HipError("gfx_restir_launch: gfx_restir_set_params has not been called");
    if (static_cast<uint32_t>(rp.s.imageSizeX) != width || static_cast<uint32_t>(rp.s.imageSizeY) != height)
        throw HipError("gfx_restir_launch: launch size differs from imageSize in the static parameters");
    const uint64_t h = rp.f.travHandle;
    if (h == 0 || h > ctx.accels.size() || !ctx.accels[h - 1]) throw HipError("gfx_restir_launch: invalid travHandle");
    const size_t numPixels = static_cast<size_t>(width) * height;
    // one entry per launch slot for the passes that do not compact their rays (emit_ray_at_slot): the tiled pixel maps pad the frame
    const size_t frameSlots = static_cast<size_t>(make_pixel_grid(ctx, width, 0, height).launchBlocks) * kBlock;
    const size_t maxRays = std::max(frameSlots, numPixels * std::max<size_t>(1 + rp.f.numSpatialNeighbors, rearch ? kRearchRayKinds : 1));
    ctx.rayOrg.reserve(16 * maxRays); ctx.rayDir.reserve(16 * maxRays);
    ctx.rayOut.reserve(4 * maxRays);
    ctx.rayHits.reserve(sizeof(gfx_hit) * numPixels);
    ctx.pixelRaySlot.reserve(4 * numPixels);
    ctx.shadeScratch.reserve(32 * numPixels);
    ctx.spatialScratch.reserve(sizeof(SpatialSlot) * maxRays);
    ctx.smallCounters.reserve(kSmallCountersBytes);
    RestirArgs a;
    a.scene = ctx.devScene();
    a.s = rp.s; a.f = rp.f;
    a.curRes = rp.currentReservoirIndex & 1u;
    a.baseIdx = rp.spatialNeighborBaseIndex & 1023u;   // 10-bit bitfield (restir_di_shared.h:287)
    a.rayOrg = ctx.rayOrg.as<float4>(); a.rayDir = ctx.rayDir.as<float4>();
    a.rayCount = ctx.smallCounters.as<uint32_t>() + 4;
    a.pixelRaySlot = ctx.pixelRaySlot.as<uint32_t>();
    a.occluded = ctx.rayOut.as<uint32_t>();
    a.hits = ctx.rayHits.as<gfx_hit>();
    a.tris = ctx.accels[h - 1]->trisPtr();
    a.shadeScratch = ctx.shadeScratch.as<float4>();
    a.spatialScratch = ctx.spatialScratch.as<SpatialSlot>();
    a.rearchSlots = nullptr;
    if (rearch) {
        ctx.rearchSlots.reserve(sizeof(uint32_t) * kRearchRayKinds * numPixels);
        a.rearchSlots = ctx.rearchSlots.as<uint32_t>();
    }
    if (rowEnd > height || rowBegin > rowEnd) throw HipError("gfx_restir_launch_rows: row range outside the image");
    if (gapEnd > gapBegin && (gapBegin < rowBegin || gapEnd > rowEnd)) throw HipError("gfx_restir_launch_rows_gap: the gap lies outside the rows");
    a.px = make_pixel_grid(ctx, width, rowBegin, rowEnd, gapBegin, gapEnd);
    return a;
}

template <typename K>
static void launch_pixels(Context& ctx, hipStream_t stream, const char* name, K kernel, const RestirArgs& a) {
    if (a.px.rowEnd == a.px.rowBegin) return;
    ScopedKernelTimer timer(ctx, stream, name);
    hipLaunchKernelGGL(kernel, dim3(a.px.launchBlocks), dim3(kBlock), 0, stream, a);
    GFX_HIP(hipGetLastError());
}

static void trace_queue(Context& ctx, hipStream_t stream, const RestirArgs& a, int mode, uint32_t fixedCount, bool useCounter, void* out) {
    TraceLaunch t;
    t.accel = ctx.accels[ctx.restir.f.travHandle - 1]->dev();
    t.rayOrgTmin = a.rayOrg; t.rayDirTmax = a.rayDir;
    t.numRays = fixedCount; t.numRaysPtr = useCounter ? a.rayCount : nullptr;
    t.out = out; t.mode = mode;
    trace_launch(ctx, stream, t);
}

// ---------------------------------------------------------------- output chain (copy_buffers.cu:6-80)
// copyToLinearBuffers: the accumulation buffers and the motion vectors as the linear arrays the denoiser / display
// consume; the normal is normalised unless it is the zero vector.
__global__ __launch_bounds__(kBlock) void k_copy_to_linear(gfx_restir_static_params s, uint32_t bufferIndex, size_t numPixels,
                                                           float4* __restrict__ color, float4* __restrict__ albedo, float4* __restrict__ normal, float2* __restrict__ motion) {
    const size_t p = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (p >= numPixels) return;
    color[p] = static_cast<const float4*>(s.beautyAccumBuffer)[p];
    albedo[p] = static_cast<const float4*>(s.albedoAccumBuffer)[p];
    const float4 nv = static_cast<const float4*>(s.normalAccumBuffer)[p];
    f3 n(nv.x, nv.y, nv.z);
    if (n.x != 0 || n.y != 0 || n.z != 0) n = unit(n);
    normal[p] = make_float4(n.x, n.y, n.z, 1.0f);
    motion[p] = static_cast<const float2*>(s.gbuffer1[bufferIndex])[p];
}
// visualizeToOutputBuffer: bufferType = BufferToDisplay (restir_di_shared.h:292-298)
__global__ __launch_bounds__(kBlock) void k_visualize(const void* __restrict__ linearBuffer, int bufferType, float mvOffset, float mvScale,
                                                      size_t numPixels, float4* __restrict__ out) {
    const size_t p = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (p >= numPixels) return;
    float4 value = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    switch (bufferType) {
    case GFX_DISPLAY_NOISY_BEAUTY: case GFX_DISPLAY_DENOISED_BEAUTY: case GFX_DISPLAY_ALBEDO:
        value = static_cast<const float4*>(linearBuffer)[p];
        break;
    case GFX_DISPLAY_NORMAL:
        value = static_cast<const float4*>(linearBuffer)[p];
        value.x = 0.5f + 0.5f * value.x; value.y = 0.5f + 0.5f * value.y; value.z = 0.5f + 0.5f * value.z;
        break;
    case GFX_DISPLAY_FLOW: {
        const float2 f = static_cast<const float2*>(linearBuffer)[p];
        value = make_float4(fminf(fmaxf(mvScale * f.x + mvOffset, 0.0f), 1.0f), fminf(fmaxf(mvScale * f.y + mvOffset, 0.0f), 1.0f), mvOffset, 1.0f);
        break;
    }
    default: break;
    }
    out[p] = value;
}
void restir_copy_to_linear(Context& ctx, hipStream_t stream, void* color, void* albedo, void* normal, void* motion) {
    if (!ctx.restir.valid) throw HipError("gfx_restir_copy_to_linear: gfx_restir_set_params first");
    const gfx_restir_static_params& s = ctx.restir.s;
    const size_t n = static_cast<size_t>(s.imageSizeX) * s.imageSizeY;
    if (!n) return;
    hipLaunchKernelGGL(k_copy_to_linear, dim3(static_cast<uint32_t>((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, s, ctx.restir.f.bufferIndex, n,
                       static_cast<float4*>(color), static_cast<float4*>(albedo), static_cast<float4*>(normal), static_cast<float2*>(motion));
    GFX_HIP(hipGetLastError());
}
void restir_visualize(Context& ctx, hipStream_t stream, const void* linearBuffer, int bufferType, float mvOffset, float mvScale, uint32_t width, uint32_t height, void* out) {
    (void)ctx;
    const size_t n = static_cast<size_t>(width) * height;
    if (!n) return;
    hipLaunchKernelGGL(k_visualize, dim3(static_cast<uint32_t>((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, linearBuffer, bufferType, mvOffset, mvScale, n,
                       static_cast<float4*>(out));
    GFX_HIP(hipGetLastError());
}

void restir_launch(Context& ctx, hipStream_t stream, int pass, uint32_t width, uint32_t height, uint32_t rowBegin, uint32_t rowEnd, uint32_t gapBegin, uint32_t gapEnd) {
    const bool rearch = pass >= GFX_RESTIR_LIGHT_PRESAMPLING && pass <= GFX_RESTIR_SHADE_AND_RESAMPLE_SPATIOTEMPORAL;
    // a gap (gfx_restir_launch_rows_gap) is for the pass a band renderer splits around its exchange: the biased spatial pass, a plain per-pixel kernel
    if (gapEnd > gapBegin && pass != GFX_RESTIR_SPATIAL_BIASED) throw HipError("gfx_restir_launch_rows_gap: only GFX_RESTIR_SPATIAL_BIASED takes a gap");
    RestirArgs a = make_args(ctx, width, height, rowBegin, rowEnd, rearch, gapBegin, gapEnd);
    if (rowEnd == rowBegin || a.px.launchBlocks == 0) return;
    auto reset_queue = [&]() { GFX_HIP(hipMemsetAsync(a.rayCount, 0, sizeof(uint32_t), stream)); };
    // A launch of up to about half a full-HD frame (a row band of a multi-GPU frame) runs each of the three ray passes as ONE kernel
    // (k_*_fused above); a larger one keeps the persistent k_trace with its refill between two per-pixel kernels (band of 8 / 4 / 2 /
    // whole frame, same box: 0.709 -> 0.653, 1.070 -> 0.974, 1.709 -> 1.647, 3.156 -> 3.199 ms per frame; profiles/r04_experiments.txt 14).
    // "fuse_passes": 0 by launch size, 1 never, 2 always.  Counting launches (gfx_counters_enable) always take the three-kernel form:
    // the counters live in k_trace.
    const uint32_t launchWaves = a.px.launchBlocks * (kBlock / 64), waveSlots = static_cast<uint32_t>(ctx.numCUs) * 4u * GFX_INIT_WAVES;
    const bool smallLaunch = launchWaves <= waveSlots + waveSlots / 2;           // about one round of waves: the candidate loop is split over four lanes
    const bool fusableLaunch = 2u * launchWaves <= 9u * waveSlots;
    const int spillCap = static_cast<int>(local_spill_depth(ctx.accels[ctx.restir.f.travHandle - 1]->maxDepth));   // stack entries per thread behind the LDS part
    const size_t fusedSpillBytes = sizeof(uint2) * static_cast<size_t>(a.px.launchBlocks) * kBlock * spillCap;
    const bool fused = !ctx.countersEnabled && fusedSpillBytes <= (size_t(1) << 30) && (ctx.tune.fusePasses == 2 || (ctx.tune.fusePasses == 0 && fusableLaunch));
    // Cost-ordered block start (block_order_begin / _end below) for the kernels whose launch is several rounds of blocks.
    const uint64_t orderKey = (static_cast<uint64_t>(rowBegin) << 44) ^ (static_cast<uint64_t>(rowEnd) << 24) ^ (static_cast<uint64_t>(width) << 4);
    const uint32_t orderMinBlocks = 8u * static_cast<uint32_t>(ctx.numCUs);     // up to about one round: they all start together
    auto block_order_begin = [&](int which, uint32_t blocks, uint32_t variant, const uint32_t*& order, uint32_t*& cost) {
        gfx::block_order_begin(ctx, stream, which, blocks, orderKey ^ variant, orderMinBlocks, order, cost);
        a.px.order = order;
    };
    auto block_order_end = [&](int which, uint32_t blocks, uint32_t* cost) { gfx::block_order_end(ctx, stream, which, blocks, cost); };
    // (Not for the kernels that gather from neighbouring pixels: started in the candidate kernel's cost order k_spatial took 0.47 ms per
    // frame instead of 0.37 -- blocks of equal cost are scattered over the image, and the XCD-supertile block order is what keeps a tile's
    // neighbours in its XCD's L2.  profiles/r04_experiments.txt 19.)
    switch (pass) {
    case GFX_RESTIR_SETUP_GBUFFERS: {
        // own scratch set (internal.h): this pass may overlap other passes of the previous frame.  One queue entry per
        // launch slot (restir_common.hip.h): the tiled pixel maps pad the frame to whole 16 x 16 blocks / supertiles
        const size_t frameSlots = static_cast<size_t>(make_pixel_grid(ctx, width, 0, height).launchBlocks) * kBlock;
        const uint32_t numSlots = a.px.launchBlocks * kBlock;
        ctx.gbRayOrg.reserve(16 * frameSlots); ctx.gbRayDir.reserve(16 * frameSlots);
        if (sizeof(gfx_hit) * frameSlots > ctx.gbRayHits.bytes) {
            // the first frame finds "no triangle" in every slot's temporal hint, not whatever the allocation held (a stale index
            // costs a triangle test and moves the work counters, never the result)
            ctx.gbRayHits.reserve(sizeof(gfx_hit) * frameSlots);
            GFX_HIP(hipMemsetAsync(ctx.gbRayHits.p, 0xFF, ctx.gbRayHits.bytes, stream));
        }
        a.rayOrg = ctx.gbRayOrg.as<float4>(); a.rayDir = ctx.gbRayDir.as<float4>();
        a.hits = ctx.gbRayHits.as<gfx_hit>();
        // primary rays are coherent (neighbouring lanes walk nearly the same nodes, the temporal hint ends most of them early): the
        // wave-local traversal loses little to the missing refill and saves the ray queue and two launches at every size (rearchitected
        // ReSTIR at 1920x1080: 2.250 -> 2.115 ms per frame, NRC 3.65 -> 3.53) -- fused unless "fuse_passes" says never
        if (!ctx.countersEnabled && fusedSpillBytes <= (size_t(1) << 30) && ctx.tune.fusePasses != 1) {
            ctx.gbSpill.reserve(fusedSpillBytes);
            const uint32_t* order = nullptr;
            uint32_t* cost = nullptr;
            block_order_begin(2, a.px.launchBlocks, 0u, order, cost);
            {
                ScopedKernelTimer timer(ctx, stream, "gbuffer_fused");
                hipLaunchKernelGGL(k_gbuffer_fused, dim3(a.px.launchBlocks), dim3(kBlock), 0, stream, a, ctx.accels[ctx.restir.f.travHandle - 1]->dev(),
                                   ctx.gbRayHits.as<gfx_hit>(), ctx.gbSpill.as<uint2>(), spillCap, ctx.tune.temporalHints ? 1 : 0, cost);
                GFX_HIP(hipGetLastError());
            }
            block_order_end(2, a.px.launchBlocks, cost);
            break;
        }
        launch_pixels(ctx, stream, "primary_rays", k_primary_rays, a);
        TraceLaunch t;
        t.accel = ctx.accels[ctx.restir.f.travHandle - 1]->dev();
        t.rayOrgTmin = a.rayOrg; t.rayDirTmax = a.rayDir;
        t.numRays = numSlots; t.numRaysPtr = nullptr;
        t.out = ctx.gbRayHits.p; t.mode = GFX_TRACE_CLOSEST;
        t.spill = &ctx.gbSpill; t.counters = &ctx.gbCounters;
        t.hintFromOut = true;   // gbRayHits keeps the previous frame's hit of every ray slot
        trace_launch(ctx, stream, t);
        launch_pixels(ctx, stream, "gbuffer_resolve", k_gbuffer_resolve, a);
        break;
    }
    case GFX_RESTIR_INITIAL_RIS:
    case GFX_RESTIR_INITIAL_AND_TEMPORAL_BIASED:
    case GFX_RESTIR_INITIAL_AND_TEMPORAL_UNBIASED:
        {
            // lanes per pixel (k_initial_candidates SPLIT): one while the launch is several rounds of waves -- the kernel is then bound by
            // instruction issue and the split only adds to it (full frame + 2 % with two lanes, + 6 % with four; a band of a 4-way split:
            // no difference) -- four when the whole launch fits the GPU's wave slots about once (a band of an 8-way split: 4 080 waves
            // on 4 096 slots, - 6 % of the band's frame); "candidate_split" overrides (profiles/r04_experiments.txt 12)
            const uint32_t numCandidates = 1u << a.f.log2NumCandidateSamples;
            uint32_t split = ctx.tune.candidateSplit > 0 ? static_cast<uint32_t>(ctx.tune.candidateSplit) : smallLaunch ? 4u : 1u;
            split = std::min(split, numCandidates);
            if (fused && split == 2) split = 1;      // the fused form exists for one and four lanes per pixel
            const uint32_t grid = a.px.launchBlocks * split;
            if (fused) {
                ctx.spill.reserve(fusedSpillBytes);
                const DevAccel accel = ctx.accels[ctx.restir.f.travHandle - 1]->dev();
                const int mode = pass == GFX_RESTIR_INITIAL_RIS ? 0 : pass == GFX_RESTIR_INITIAL_AND_TEMPORAL_BIASED ? 1 : 2;
                const bool tex = a.scene.emitterTexRefs != nullptr;
                void (*kernel)(RestirArgs, DevAccel, uint2*, int, uint32_t*) = nullptr;
#define GFX_PICK(TEX, SPLIT) (mode == 0 ? k_initial_fused<TEX, SPLIT, 0> : mode == 1 ? k_initial_fused<TEX, SPLIT, 1> : k_initial_fused<TEX, SPLIT, 2>)
                if (tex) kernel = split == 4 ? GFX_PICK(true, 4) : GFX_PICK(true, 1);
                else kernel = split == 4 ? GFX_PICK(false, 4) : GFX_PICK(false, 1);
#undef GFX_PICK
                const uint32_t* order = nullptr;
                uint32_t* cost = nullptr;
                block_order_begin(0, grid, split, order, cost);
                {
                    ScopedKernelTimer timer(ctx, stream, "initial_fused");
                    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, stream, a, accel, ctx.spill.as<uint2>(), spillCap, cost);
                    GFX_HIP(hipGetLastError());
                }
                block_order_end(0, grid, cost);
                break;
            }
            void (*kernel)(RestirArgs, uint32_t*) = a.scene.emitterTexRefs
                ? (split == 4 ? k_initial_candidates<true, 4> : split == 2 ? k_initial_candidates<true, 2> : k_initial_candidates<true, 1>)
                : (split == 4 ? k_initial_candidates<false, 4> : split == 2 ? k_initial_candidates<false, 2> : k_initial_candidates<false, 1>);
            const uint32_t* order = nullptr;
            uint32_t* cost = nullptr;
            block_order_begin(3, grid, split, order, cost);
            {
                ScopedKernelTimer timer(ctx, stream, "initial_candidates");
                hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, stream, a, cost);
                GFX_HIP(hipGetLastError());
            }
            block_order_end(3, grid, cost);
            a.px.order = nullptr;                      // the trace and the temporal kernel below run in index order
        }
        trace_queue(ctx, stream, a, GFX_TRACE_ANY, a.px.launchBlocks * kBlock, false, ctx.rayOut.p);   // one entry per launch slot (emit_ray_at_slot)
        if (pass == GFX_RESTIR_INITIAL_RIS) launch_pixels(ctx, stream, "temporal_none", k_temporal<0>, a);
        else if (pass == GFX_RESTIR_INITIAL_AND_TEMPORAL_BIASED) launch_pixels(ctx, stream, "temporal_biased", k_temporal<1>, a);
        else launch_pixels(ctx, stream, "temporal_unbiased", k_temporal<2>, a);
        break;
    case GFX_RESTIR_SPATIAL_BIASED:
        launch_pixels(ctx, stream, "spatial_biased", k_spatial<false>, a);
        break;
    case GFX_RESTIR_SPATIAL_UNBIASED:
        // (one kernel per band -- combine, then form / trace / add the MIS terms one after the other inside the wave -- was built and measured
        // slower at every band size: up to four rays per pixel are what k_trace's refill is good at; profiles/r05_experiments.txt 9)
        reset_queue();
        launch_pixels(ctx, stream, "spatial_unbiased_select", k_spatial<true>, a);
        trace_queue(ctx, stream, a, GFX_TRACE_ANY, 0, true, ctx.rayOut.p);
        launch_pixels(ctx, stream, "spatial_unbiased_finish", k_spatial_mis_finish, a);
        break;
    case GFX_RESTIR_SPATIAL_BIASED_AND_SHADING:
    case GFX_RESTIR_SHADING:
        if (fused) {
            ctx.spill.reserve(fusedSpillBytes);
            const bool both = pass == GFX_RESTIR_SPATIAL_BIASED_AND_SHADING;
            const uint32_t* order = nullptr;
            uint32_t* cost = nullptr;
            block_order_begin(1, a.px.launchBlocks, both ? 1u : 0u, order, cost);
            {
                ScopedKernelTimer timer(ctx, stream, both ? "spatial_shading_fused" : "shading_fused");
                hipLaunchKernelGGL(both ? k_shading_fused<true> : k_shading_fused<false>, dim3(a.px.launchBlocks), dim3(kBlock), 0, stream, a,
                                   ctx.accels[ctx.restir.f.travHandle - 1]->dev(), ctx.spill.as<uint2>(), spillCap, cost);
                GFX_HIP(hipGetLastError());
            }
            block_order_end(1, a.px.launchBlocks, cost);
            break;
        }
        if (pass == GFX_RESTIR_SPATIAL_BIASED_AND_SHADING) launch_pixels(ctx, stream, "spatial_shade_prepare", k_spatial_shade_prepare, a);
        else launch_pixels(ctx, stream, "shade_prepare", k_shade_prepare, a);
        trace_queue(ctx, stream, a, GFX_TRACE_ANY, a.px.launchBlocks * kBlock, false, ctx.rayOut.p);   // one entry per launch slot (emit_ray_at_slot)
        launch_pixels(ctx, stream, "shade_finish", k_shade_finish, a);
        break;
    case GFX_RESTIR_LIGHT_PRESAMPLING: {
        if (!a.s.lightPreSamplingRngs || !a.s.preSampledLights) throw HipError("gfx_restir_launch: light pre-sampling buffers are not set");
        ScopedKernelTimer timer(ctx, stream, "light_presample");
        hipLaunchKernelGGL(k_light_presample, dim3(kNumPreSampledLights / kBlock), dim3(kBlock), 0, stream, a);
        GFX_HIP(hipGetLastError());
        break;
    }
    case GFX_RESTIR_PER_PIXEL_RIS: {
        if (rowBegin % 8 != 0) throw HipError("gfx_restir_launch_rows: per-pixel RIS bands must start on an 8-row tile boundary");
        const uint32_t tilesX = (width + 7) / 8, tileRows = (rowEnd + 7) / 8 - rowBegin / 8;
        const uint32_t numTiles = tilesX * tileRows;
        ScopedKernelTimer timer(ctx, stream, "per_pixel_ris");
        hipLaunchKernelGGL(k_per_pixel_ris, dim3((numTiles + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, stream, a);
        GFX_HIP(hipGetLastError());
        break;
    }
    case GFX_RESTIR_TRACE_SHADOW_RAYS:
    case GFX_RESTIR_TRACE_SHADOW_RAYS_TEMPORAL_BIASED:
    case GFX_RESTIR_TRACE_SHADOW_RAYS_SPATIAL_BIASED:
    case GFX_RESTIR_TRACE_SHADOW_RAYS_SPATIOTEMPORAL_BIASED:
    case GFX_RESTIR_TRACE_SHADOW_RAYS_TEMPORAL_UNBIASED:
    case GFX_RESTIR_TRACE_SHADOW_RAYS_SPATIAL_UNBIASED:
    case GFX_RESTIR_TRACE_SHADOW_RAYS_SPATIOTEMPORAL_UNBIASED: {
        reset_queue();
        void (*emit)(RestirArgs) = nullptr; void (*finish)(RestirArgs) = nullptr;
        switch (pass) {
#define GFX_REARCH_CASE(PASS, T, S, U) case PASS: emit = k_rearch_emit<T, S, U>; finish = k_rearch_vis_finish<T, S, U>; break;
        GFX_REARCH_CASE(GFX_RESTIR_TRACE_SHADOW_RAYS, false, false, false)
        GFX_REARCH_CASE(GFX_RESTIR_TRACE_SHADOW_RAYS_TEMPORAL_BIASED, true, false, false)
        GFX_REARCH_CASE(GFX_RESTIR_TRACE_SHADOW_RAYS_SPATIAL_BIASED, false, true, false)
        GFX_REARCH_CASE(GFX_RESTIR_TRACE_SHADOW_RAYS_SPATIOTEMPORAL_BIASED, true, true, false)
        GFX_REARCH_CASE(GFX_RESTIR_TRACE_SHADOW_RAYS_TEMPORAL_UNBIASED, true, false, true)
        GFX_REARCH_CASE(GFX_RESTIR_TRACE_SHADOW_RAYS_SPATIAL_UNBIASED, false, true, true)
        GFX_REARCH_CASE(GFX_RESTIR_TRACE_SHADOW_RAYS_SPATIOTEMPORAL_UNBIASED, true, true, true)
#undef GFX_REARCH_CASE
        }
        launch_pixels(ctx, stream, "rearch_emit", emit, a);
        trace_queue(ctx, stream, a, GFX_TRACE_ANY, 0, true, ctx.rayOut.p);
        launch_pixels(ctx, stream, "rearch_vis_finish", finish, a);
        break;
    }
    case GFX_RESTIR_SHADE_AND_RESAMPLE: launch_pixels(ctx, stream, "rearch_shade", k_rearch_shade<false, false>, a); break;
    case GFX_RESTIR_SHADE_AND_RESAMPLE_TEMPORAL: launch_pixels(ctx, stream, "rearch_shade_t", k_rearch_shade<true, false>, a); break;
    case GFX_RESTIR_SHADE_AND_RESAMPLE_SPATIAL: launch_pixels(ctx, stream, "rearch_shade_s", k_rearch_shade<false, true>, a); break;
    case GFX_RESTIR_SHADE_AND_RESAMPLE_SPATIOTEMPORAL: launch_pixels(ctx, stream, "rearch_shade_st", k_rearch_shade<true, true>, a); break;
    default:
        throw HipError("gfx_restir_launch: unknown pass");
    }
}

} // namespace gfx
