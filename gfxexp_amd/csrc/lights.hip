// lights.hip -- emitter importance and the three-level light distribution.
//
// Reference: common/gpu_kernels/compute_light_probs.cu:22-46 (triangle importance),
// :86-93 (geomInst importance), :134-142 (instance importance), :206-212 (finalize) and the cubd
// exclusive scans sequenced by common/common_host.h:1102-1359.
//
// Contract: a CDF is the SERIAL left-to-right exclusive prefix sum of its weights (the order a CPU
// loop produces).  cub::DeviceScan leaves the association order unspecified, so the reference's
// own CDFs are only defined up to fp32 rounding; fixing the serial order makes the sampled
// triangle indices reproducible.  One thread scans one distribution; the scene has many small
// distributions (one per emitter geomInst / instance), so the work is still parallel.
#include <cstdlib>
#include "internal.h"
#include "shading.hip.h"

namespace gfx {

__global__ void k_triangle_importance(DevScene sc, uint32_t numGeomInsts, float* __restrict__ weights) {
    // one block row per geomInst (blockIdx.y), threads over its triangles
    const uint32_t gi = blockIdx.y;
    if (gi >= numGeomInsts) return;
    const DevGeomInst g = sc.geomInsts[gi];
    if (g.distOffset == 0xFFFFFFFFu) return;
    const gfx_material& mat = sc.materials[g.materialSlot];
    const f3 e(mat.emittance[0], mat.emittance[1], mat.emittance[2]);
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < g.numTriangles; t += gridDim.x * blockDim.x) {
        const uint32_t* tri = sc.triangles + 3ull * (g.triangleOffset + t);
        const DevVertex v0 = load_vertex(sc.vertices + g.vertexOffset + tri[0]);
        const DevVertex v1 = load_vertex(sc.vertices + g.vertexOffset + tri[1]);
        const DevVertex v2 = load_vertex(sc.vertices + g.vertexOffset + tri[2]);
        const f3 p0(v0.px, v0.py, v0.pz), p1(v1.px, v1.py, v1.pz), p2(v2.px, v2.py, v2.pz);
        const f3 n = cross(p1 - p0, p2 - p0);
        const float area = 0.5f * len(n);
        // mean of the emittance texture at the three vertices (a constant reads the same value thrice)
        f3 e0 = e, e1 = e, e2 = e;
        if (mat.texEmittance) {
            const float4 t0 = tex2d(sc, mat.texEmittance, v0.u, v0.v), t1 = tex2d(sc, mat.texEmittance, v1.u, v1.v), t2 = tex2d(sc, mat.texEmittance, v2.u, v2.v);
            e0 = f3(t0.x, t0.y, t0.z); e1 = f3(t1.x, t1.y, t1.z); e2 = f3(t2.x, t2.y, t2.z);
        }
        f3 est = f3(0.0f) + e0;
        est = est + e1;
        est = est + e2;
        est = est / 3.0f;
        weights[g.distOffset + t] = luminance_srgb(est) * area;
    }
}

// Serial exclusive scan of one distribution per thread; writes the integral into the owning table.
__global__ void k_scan_geom_dists(DevGeomInst* __restrict__ geomInsts, uint32_t numGeomInsts,
                                  const float* __restrict__ weights, float* __restrict__ cdf, float* __restrict__ probs) {
    const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= numGeomInsts) return;
    DevGeomInst g = geomInsts[gi];
    if (g.distOffset == 0xFFFFFFFFu) return;
    float acc = 0.0f, last = 0.0f, lastW = 0.0f;
    for (uint32_t i = 0; i < g.distCount; ++i) {
        const float w = weights[g.distOffset + i];
        cdf[g.distOffset + i] = acc;
        last = acc; lastW = w;
        acc += w;
    }
    const float integral = g.distCount ? last + lastW : 0.0f;   // CDF[n-1] + w[n-1]
    geomInsts[gi].distIntegral = integral;
    // the division DiscreteDistribution1D::sample/evaluatePMF performs per call, done once
    for (uint32_t i = 0; i < g.distCount; ++i) probs[g.distOffset + i] = weights[g.distOffset + i] / integral;
}

__global__ void k_inst_geom_dists(DevInstance* __restrict__ insts, uint32_t numInsts,
                                  const DevGeomInst* __restrict__ geomInsts, const uint32_t* __restrict__ slotPool,
                                  float* __restrict__ weights, float* __restrict__ cdf, float* __restrict__ probs) {
    const uint32_t ii = blockIdx.x * blockDim.x + threadIdx.x;
    if (ii >= numInsts) return;
    DevInstance* inst = insts + ii;
    if (inst->distOffset == 0xFFFFFFFFu) { inst->distIntegral = 0.0f; return; }
    float acc = 0.0f, last = 0.0f, lastW = 0.0f;
    for (uint32_t i = 0; i < inst->numGeomInsts; ++i) {
        const float w = geomInsts[slotPool[inst->slotsOffset + i]].distIntegral;   // compute_light_probs.cu:86-93
        weights[inst->distOffset + i] = w;
        cdf[inst->distOffset + i] = acc;
        last = acc; lastW = w;
        acc += w;
    }
    const float integral = inst->numGeomInsts ? last + lastW : 0.0f;
    inst->distIntegral = integral;
    for (uint32_t i = 0; i < inst->numGeomInsts; ++i) probs[inst->distOffset + i] = weights[inst->distOffset + i] / integral;
}

// Per (instance, geomInst): copy the emitter distribution's integral next to its offsets and
// pre-transform the emitter triangles (EmitterRec, device_types.h).  One block per instance.
__global__ void k_emitter_records(DevScene sc, uint32_t numInsts, LightGeomRef* __restrict__ refs, EmitterRec* __restrict__ recs,
                                  EmitterRecExtra* __restrict__ extras, const uint32_t* __restrict__ instMatrixIndex,
                                  EmitterTexRef* __restrict__ texRefs /* null: no emittance textures in the scene */) {
    const uint32_t ii = blockIdx.x;
    if (ii >= numInsts) return;
    const DevInstance* inst = sc.insts + ii;
    if (inst->distOffset == 0xFFFFFFFFu) return;
    const uint32_t matrixBits = (instMatrixIndex[ii] & kEmitterMatrixMask) << kEmitterMatrixShift;
    const m34 xfm = load_m34(inst->transform);
    for (uint32_t k = 0; k < inst->numGeomInsts; ++k) {
        const DevGeomInst g = sc.geomInsts[sc.geomInstSlotPool[inst->slotsOffset + k]];
        LightGeomRef* ref = refs + inst->distOffset + k;
        if (threadIdx.x == 0) ref->distIntegral = g.distIntegral;
        const uint32_t recBase = ref->recBase;
        if (recBase == 0xFFFFFFFFu) continue;
        const gfx_material& mat = sc.materials[g.materialSlot];
        for (uint32_t t = threadIdx.x; t < g.numTriangles; t += blockDim.x) {
            const uint32_t* tri = sc.triangles + 3ull * (g.triangleOffset + t);
            const DevVertex vA = load_vertex(sc.vertices + g.vertexOffset + tri[0]);
            const DevVertex vB = load_vertex(sc.vertices + g.vertexOffset + tri[1]);
            const DevVertex vC = load_vertex(sc.vertices + g.vertexOffset + tri[2]);
            const f3 pA = xfm_point(xfm, f3(vA.px, vA.py, vA.pz));
            const f3 pB = xfm_point(xfm, f3(vB.px, vB.py, vB.pz));
            const f3 pC = xfm_point(xfm, f3(vC.px, vC.py, vC.pz));
            EmitterRec r;
            r.pA[0] = pA.x; r.pA[1] = pA.y; r.pA[2] = pA.z;
            r.pB[0] = pB.x; r.pB[1] = pB.y; r.pB[2] = pB.z;
            r.pC[0] = pC.x; r.pC[1] = pC.y; r.pC[2] = pC.z;
            r.nA[0] = vA.nx; r.nA[1] = vA.ny; r.nA[2] = vA.nz;
            EmitterRecExtra x;
            x.nB[0] = vB.nx; x.nB[1] = vB.ny; x.nB[2] = vB.nz;
            x.nC[0] = vC.nx; x.nC[1] = vC.ny; x.nC[2] = vC.nz;
            // flat = the three normals have the same bits: interpolating three copies of nA then gives the same result
            const bool smooth = f2bits(vB.nx) != f2bits(vA.nx) || f2bits(vB.ny) != f2bits(vA.ny) || f2bits(vB.nz) != f2bits(vA.nz) ||
                                f2bits(vC.nx) != f2bits(vA.nx) || f2bits(vC.ny) != f2bits(vA.ny) || f2bits(vC.nz) != f2bits(vA.nz);
            // emittance = RGB(1) * texel for an emitter material (restir_di_shared.h:504-514)
            const f3 e = mat.hasEmittance ? f3(1.0f) * f3(mat.emittance[0], mat.emittance[1], mat.emittance[2]) : f3(0.0f);
            r.emittance[0] = e.x; r.emittance[1] = e.y; r.emittance[2] = e.z;
            r.flags = ((texRefs && mat.hasEmittance) ? (mat.texEmittance & kEmitterTexMask) : 0u) | matrixBits | (smooth ? kEmitterSmooth : 0u);
            x.twoOverLenNg = 2.0f / len(cross(pB - pA, pC - pA));
            x.primProb = sc.lightWeights[g.distOffset + t] / g.distIntegral;
            float4* dst = reinterpret_cast<float4*>(recs + recBase + t);
            const float4* src = reinterpret_cast<const float4*>(&r);
            for (int q = 0; q < 4; ++q) dst[q] = src[q];
            float4* dstX = reinterpret_cast<float4*>(extras + recBase + t);
            const float4* srcX = reinterpret_cast<const float4*>(&x);
            dstX[0] = srcX[0]; dstX[1] = srcX[1];
            if (texRefs) {
                EmitterTexRef tr;
                tr.uvA[0] = vA.u; tr.uvA[1] = vA.v; tr.uvB[0] = vB.u; tr.uvB[1] = vB.v; tr.uvC[0] = vC.u; tr.uvC[1] = vC.v;
                const DevTexture desc = sc.textures[mat.hasEmittance ? mat.texEmittance : 0u];   // slot 0 is a zeroed entry
                tr.texelOffset = desc.offset;
                tr.dims = desc.width ? (((desc.width - 1u) & 0x3FFFu) | (((desc.height - 1u) & 0x3FFFu) << 14) | (desc.format << 28)) : 0u;
                texRefs[recBase + t] = tr;
            }
        }
    }
}

__global__ void k_inst_importance(const DevInstance* __restrict__ insts, uint32_t numInsts, uint32_t off, float* __restrict__ weights) {
    const uint32_t ii = blockIdx.x * blockDim.x + threadIdx.x;
    if (ii >= numInsts) return;
    const DevInstance* inst = insts + ii;
    // Matrix4x4::decompose: scale.x = length of column 0 (common/basic_types.h:4643-4646)
    const float sx = sqrtf(inst->transform[0] * inst->transform[0] + inst->transform[4] * inst->transform[4] +
                           inst->transform[8] * inst->transform[8]);
    weights[off + ii] = sq(sx) * inst->distIntegral;
}

// Serial-order exclusive scan of the instance-level distribution (runs every frame,
// restir_di_main.cpp:2303-2309).  The summation order is the contract, so one lane adds -- but out of
// LDS: the block stages the weights with coalesced loads, lane 0 scans chunk by chunk in LDS
// (~10 cycles per element instead of a dependent global load + store), and the block writes the
// CDF back coalesced.
constexpr int kScanChunk = 4096;
__global__ __launch_bounds__(256) void k_scan_inst_dist(uint32_t numInsts, uint32_t off, const float* __restrict__ weights,
                                                        float* __restrict__ cdf, float* __restrict__ probs,
                                                        float* __restrict__ integralOut, uint32_t guideCells) {
    __shared__ float buf[kScanChunk];
    __shared__ float carry[2];   // running sum, last weight
    if (blockIdx.x != 0) return;
    if (threadIdx.x == 0) { carry[0] = 0.0f; carry[1] = 0.0f; }
    float lastCdf = 0.0f;
    for (uint32_t base = 0; base < numInsts; base += kScanChunk) {
        const uint32_t m = min(static_cast<uint32_t>(kScanChunk), numInsts - base);
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) buf[i] = weights[off + base + i];
        __syncthreads();
        if (threadIdx.x == 0) {
            float acc = carry[0];
            for (uint32_t i = 0; i < m; ++i) {
                const float w = buf[i];
                buf[i] = acc;
                lastCdf = acc; carry[1] = w;
                acc += w;
            }
            carry[0] = acc;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) cdf[off + base + i] = buf[i];
        __syncthreads();
    }
    if (threadIdx.x == 0) carry[0] = numInsts ? lastCdf + carry[1] : 0.0f;   // CDF[n-1] + w[n-1]
    __syncthreads();
    const float integral = carry[0];
    for (uint32_t i = threadIdx.x; i < numInsts; i += blockDim.x) probs[off + i] = weights[off + i] / integral;
    if (threadIdx.x == 0) {
        integralOut[0] = integral;
        // guide header: scale and "usable" flag (k_inst_guide clears the flag if the CDF is not monotone)
        const float scale = static_cast<float>(guideCells) / integral;
        const bool usable = numInsts > 0 && numInsts <= 65536 && integral > 0.0f && integral < INFINITY && scale > 0.0f && scale < INFINITY;
        integralOut[1] = usable ? scale : 0.0f;
        reinterpret_cast<uint32_t*>(integralOut)[2] = usable ? 1u : 0u;
    }
}

// Guide table over the instance-level CDF.  cell(x) = min(cells - 1, uint(x * scale)) is monotone in x,
// so with guide[k] = largest i with cell(CDF[i]) <= k a sample u in cell k has its answer (the largest i
// with CDF[i] <= u, what the reference's 12-step binary search finds, common_shared.h:209-247) inside
// [guide[k-1], guide[k]]: CDF[guide[k-1]] lies in an earlier cell, hence below u, and nothing past guide[k]
// can be <= u.  The bracket is typically one or two entries wide.  All of this leans on the CDF being
// monotone, which a sum of non-negative weights is; the kernel checks it anyway and withdraws the table
// (samplers fall back to the plain search) if it ever is not.
__global__ __launch_bounds__(256) void k_inst_guide(uint32_t numInsts, uint32_t off, const float* __restrict__ cdfPool,
                                                    float* __restrict__ header, uint16_t* __restrict__ guide, uint32_t cells) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const float* cdf = cdfPool + off;
    if (reinterpret_cast<const uint32_t*>(header)[2] == 0u) return;
    const float scale = header[1];
    if (t + 1 < numInsts && !(cdf[t] <= cdf[t + 1])) atomicAnd(reinterpret_cast<uint32_t*>(header) + 2, 0u);
    if (t == 0 && !(cdf[0] == 0.0f)) atomicAnd(reinterpret_cast<uint32_t*>(header) + 2, 0u);
    if (t >= cells) return;
    uint32_t idx = 0;
    for (uint32_t d = next_pow2(numInsts) >> 1; d >= 1; d >>= 1) {
        if (idx + d >= numInsts) continue;
        if (guide_cell(cdf[idx + d], scale, cells) <= t) idx += d;
    }
    guide[t] = static_cast<uint16_t>(idx);
}

// ---------------------------------------------------------------- emitter interval table (emitter_spans.h)
// 1. k_span_inst_begin: per instance i, the smallest ul whose instance search returns an index >= i
//    (pure arithmetic bisection over the bit patterns of ul).  Instance i is selected on [begin_i, begin_{i+1}).
// 2. k_span_records: per emitter record (instance i, geometry instance k, primitive t), the smallest ul of
//    that range whose levels 2 and 3 reach (k, t) or further; the last record of a geometry instance also
//    finds where the level-2 search moves past k.  Records behind an early out (probability zero at level
//    1 or 2) get an empty interval.
// 3. k_span_finish: end_e = begin_{e+1} inside a geometry instance; checks that the intervals ascend and,
//    with the reference's own three searches (light_locate_3level), that both ends of every interval select
//    exactly that record and the values next to them do not.  Any failure withdraws the table (header[0] = 0)
//    and sample_light runs the three searches instead.
// 4. k_span_guide: guide table over ul.
__global__ void k_span_init(const float* __restrict__ instHeader, uint32_t* __restrict__ spanHeader) {
    const float integral = instHeader[0];
    spanHeader[0] = (integral > 0.0f && integral < INFINITY) ? 1u : 0u;
    spanHeader[1] = 0u;
}

__global__ void k_span_inst_begin(uint32_t numInsts, const float* __restrict__ cdf1, const float* __restrict__ header,
                                  uint32_t* __restrict__ instBegin) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > numInsts) return;
    if (i == numInsts) { instBegin[i] = kSpanBitsEnd; return; }
    SpanInstPred pred; pred.integral = header[0]; pred.i = i; pred.cdfAtI = cdf1[i];
    instBegin[i] = span_bisect(0u, kSpanBitsEnd, pred);
}

__global__ void k_span_records(DevScene sc, uint32_t numInsts, const uint32_t* __restrict__ instBegin,
                               EmitterSpan* __restrict__ spans, uint32_t* __restrict__ spanHeader) {
    const uint32_t ii = blockIdx.x;
    if (ii >= numInsts) return;
    const DevInstance* inst = sc.insts + ii;
    if (inst->distOffset == 0xFFFFFFFFu) return;
    const float* cdf1 = sc.lightCDF + sc.lightInstDistOffset;
    SpanRecordKey key;
    key.integral1 = *sc.lightInstIntegral;
    key.lo1 = cdf1[ii];
    key.hi1 = ii + 1 < numInsts ? cdf1[ii + 1] : key.integral1;
    key.integral2 = inst->distIntegral;
    key.n2 = inst->numGeomInsts;
    const float instProb = sc.lightProbs[sc.lightInstDistOffset + ii];
    const uint32_t rangeLo = instBegin[ii], rangeHi = instBegin[ii + 1];
    bool monotone = key.lo1 <= key.hi1;
    for (uint32_t k = 0; k < key.n2; ++k) {
        const LightGeomRef ref = sc.lightGeomRefs[inst->distOffset + k];
        if (ref.recBase == 0xFFFFFFFFu) continue;
        key.k = k;
        key.lo2 = sc.lightCDF[inst->distOffset + k];
        key.hi2 = k + 1 < key.n2 ? sc.lightCDF[inst->distOffset + k + 1] : key.integral2;
        key.integral3 = ref.distIntegral;
        monotone = monotone && key.lo2 <= key.hi2;
        const float geomProb = sc.lightProbs[inst->distOffset + k];
        const bool earlyOut = instProb == 0.0f || geomProb == 0.0f;
        for (uint32_t t = threadIdx.x; t < ref.distCount; t += blockDim.x) {
            key.t = t;
            key.cdf3AtT = sc.lightCDF[ref.distOffset + t];
            if (t + 1 < ref.distCount) monotone = monotone && key.cdf3AtT <= sc.lightCDF[ref.distOffset + t + 1];
            uint32_t b, e;
            span_record_interval(key, rangeLo, rangeHi, earlyOut, t + 1 == ref.distCount, b, e);
            const EmitterRecExtra* rec = sc.emitterRecExtras + ref.recBase + t;
            EmitterSpan s;
            s.begin = span_float(b);
            s.end = span_float(e);
            // lightProb = 1; lightProb *= instProb; *= geomInstProb; *= primProb; density = lightProb * (2 / |ng|)
            s.density = ((instProb * geomProb) * rec->primProb) * rec->twoOverLenNg;
            s.instSlot = ii;
            *reinterpret_cast<SpanWords*>(spans + ref.recBase + t) = __builtin_bit_cast(SpanWords, s);
        }
    }
    if (!monotone) atomicAnd(spanHeader, 0u);
}

__global__ void k_span_finish(DevScene sc, EmitterSpan* __restrict__ spans, uint32_t* __restrict__ spanHeader) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= sc.numSpans) return;
    EmitterSpan s = spans[e];
    const uint32_t nextBegin = e + 1 < sc.numSpans ? span_bits(spans[e + 1].begin) : kSpanBitsEnd;
    if (span_bits(s.end) == kSpanPending) s.end = span_float(nextBegin);
    const uint32_t b = span_bits(s.begin), en = span_bits(s.end);
    bool ok = b <= en && en <= nextBegin && en <= kSpanBitsEnd;
    // the reference's own searches at and next to both ends
    const InstDist dist = inst_dist_global_unguided(sc);
    auto selects = [&](uint32_t ulBits) {
        uint32_t rec, instSlot; float partial;
        return light_locate_3level(sc, dist, span_float(ulBits), rec, instSlot, partial) && rec == e;
    };
    if (b < en) {
        ok = ok && selects(b) && selects(en - 1u);
        if (b > 0u) ok = ok && !selects(b - 1u);
        if (en < kSpanBitsEnd) ok = ok && !selects(en);
    }
    else if (b < kSpanBitsEnd) ok = ok && !selects(b);
    spans[e].end = s.end;   // only this thread writes spans[e].end; neighbours read spans[e].begin
    if (!ok) atomicAnd(spanHeader, 0u);
    else atomicAdd(spanHeader + 1, 1u);
}

__global__ void k_span_guide(const EmitterSpan* __restrict__ spans, uint32_t numSpans, SpanGuide* __restrict__ guide, uint32_t cells) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cells) return;
    guide[c] = span_guide_entry(spans, numSpans, cells, c);
}

static void span_table_build(Context& ctx, hipStream_t stream) {
    const uint32_t ni = static_cast<uint32_t>(ctx.insts.size());
    const uint32_t ne = ctx.numEmitterRecs;
    uint32_t* header = ctx.dSpanHeader.as<uint32_t>();
    GFX_HIP(hipMemsetAsync(header, 0, 16, stream));
    static int disabled = -1;
    if (disabled < 0) { const char* e = getenv("GFX_LIGHT_TABLE"); disabled = (e && e[0] == '0') ? 1 : 0; }
    if (!ni || !ne || disabled) return;
    // usable until a check fails; the integral must be a positive finite number (k_scan_inst_dist's "usable" covers that)
    hipLaunchKernelGGL(k_span_init, dim3(1), dim3(1), 0, stream, ctx.dLightInstIntegral.as<float>(), header);
    hipLaunchKernelGGL(k_span_inst_begin, dim3((ni + 1 + 255) / 256), dim3(256), 0, stream, ni,
                       ctx.dLightCDF.as<float>() + ctx.lightInstDistOffset, ctx.dLightInstIntegral.as<float>(), ctx.dSpanInstBegin.as<uint32_t>());
    hipLaunchKernelGGL(k_span_records, dim3(ni), dim3(64), 0, stream, ctx.devScene(), ni, ctx.dSpanInstBegin.as<uint32_t>(),
                       ctx.dSpans.as<EmitterSpan>(), header);
    hipLaunchKernelGGL(k_span_finish, dim3((ne + 255) / 256), dim3(256), 0, stream, ctx.devScene(), ctx.dSpans.as<EmitterSpan>(), header);
    hipLaunchKernelGGL(k_span_guide, dim3((ctx.spanGuideCells + 255) / 256), dim3(256), 0, stream, ctx.dSpans.as<EmitterSpan>(), ne,
                       ctx.dSpanGuide.as<SpanGuide>(), ctx.spanGuideCells);
    GFX_HIP(hipGetLastError());
}

void lights_build_static(Context& ctx, hipStream_t stream) {
    scene_upload(ctx, stream);
    const uint32_t ng = static_cast<uint32_t>(ctx.geoms.size());
    const uint32_t ni = static_cast<uint32_t>(ctx.insts.size());
    if (ng) {
        uint32_t maxTris = 1;
        for (const DevGeomInst& g : ctx.hGeomInsts) if (g.distOffset != 0xFFFFFFFFu) maxTris = std::max(maxTris, g.numTriangles);
        const dim3 grid(std::min<uint32_t>((maxTris + 255) / 256, 64), ng);
        hipLaunchKernelGGL(k_triangle_importance, grid, dim3(256), 0, stream, ctx.devScene(), ng, ctx.dLightW.as<float>());
        hipLaunchKernelGGL(k_scan_geom_dists, dim3((ng + 63) / 64), dim3(64), 0, stream,
                           ctx.dGeomInsts.as<DevGeomInst>(), ng, ctx.dLightW.as<float>(), ctx.dLightCDF.as<float>(), ctx.dLightP.as<float>());
    }
    if (ni)
        hipLaunchKernelGGL(k_inst_geom_dists, dim3((ni + 63) / 64), dim3(64), 0, stream,
                           ctx.dInsts.as<DevInstance>(), ni, ctx.dGeomInsts.as<DevGeomInst>(), ctx.dSlotPool.as<uint32_t>(),
                           ctx.dLightW.as<float>(), ctx.dLightCDF.as<float>(), ctx.dLightP.as<float>());
    if (ni)
        hipLaunchKernelGGL(k_emitter_records, dim3(ni), dim3(64), 0, stream, ctx.devScene(), ni,
                           ctx.dLightRefs.as<LightGeomRef>(), ctx.dEmitterRecs.as<EmitterRec>(), ctx.dEmitterRecExtras.as<EmitterRecExtra>(), ctx.dInstMatrixIndex.as<uint32_t>(),
                               ctx.anyEmittanceTexture ? ctx.dEmitterTexRefs.as<EmitterTexRef>() : nullptr);
    GFX_HIP(hipGetLastError());
    // keep the host mirrors of the integrals current (read by gfx_lights_read and the launch params)
    GFX_HIP(hipMemcpyAsync(ctx.hGeomInsts.data(), ctx.dGeomInsts.p, sizeof(DevGeomInst) * ng, hipMemcpyDeviceToHost, stream));
    GFX_HIP(hipMemcpyAsync(ctx.hInsts.data(), ctx.dInsts.p, sizeof(DevInstance) * ni, hipMemcpyDeviceToHost, stream));
    GFX_HIP(hipStreamSynchronize(stream));
    ctx.lightsStaticBuilt = true;
}

void lights_build_instances(Context& ctx, hipStream_t stream, uint32_t /*bufferIndex*/) {
    if (!ctx.lightsStaticBuilt) lights_build_static(ctx, stream);
    transforms_upload(ctx, stream);
    if (ctx.emitterRecsDirty) {
        // animated instances moved: their emitter triangles are stored in world space, their normal matrices changed
        light_matrices_upload(ctx, stream);
        const uint32_t numInsts = static_cast<uint32_t>(ctx.insts.size());
        if (numInsts)
            hipLaunchKernelGGL(k_emitter_records, dim3(numInsts), dim3(64), 0, stream, ctx.devScene(), numInsts,
                               ctx.dLightRefs.as<LightGeomRef>(), ctx.dEmitterRecs.as<EmitterRec>(), ctx.dEmitterRecExtras.as<EmitterRecExtra>(), ctx.dInstMatrixIndex.as<uint32_t>(),
                               ctx.anyEmittanceTexture ? ctx.dEmitterTexRefs.as<EmitterTexRef>() : nullptr);
        GFX_HIP(hipGetLastError());
        ctx.emitterRecsDirty = false;
    }
    // scene.setupLightInstDistribution runs every frame in the reference (restir_di_main.cpp:2303-2309); its inputs
    // -- the instances' scale and emitter integrals -- only change with the scene, so an unchanged scene keeps the
    // distribution (same values, a serial-order scan and two small kernels less per frame)
    if (ctx.instDistValid) return;
    ctx.instDistValid = true;
    const uint32_t ni = static_cast<uint32_t>(ctx.insts.size());
    // The integral stays device resident (like the DiscreteDistribution1D inside the reference's
    // static launch parameters, restir_di_main.cpp:2303-2309): no host round trip per frame.
    float* dIntegral = ctx.dLightInstIntegral.as<float>();
    if (ni) {
        hipLaunchKernelGGL(k_inst_importance, dim3((ni + 63) / 64), dim3(64), 0, stream,
                           ctx.dInsts.as<DevInstance>(), ni, ctx.lightInstDistOffset, ctx.dLightW.as<float>());
        hipLaunchKernelGGL(k_scan_inst_dist, dim3(1), dim3(256), 0, stream, ni, ctx.lightInstDistOffset,
                           ctx.dLightW.as<float>(), ctx.dLightCDF.as<float>(), ctx.dLightP.as<float>(), dIntegral, ctx.lightInstGuideCells);
        const uint32_t guideThreads = std::max(ni, ctx.lightInstGuideCells);
        hipLaunchKernelGGL(k_inst_guide, dim3((guideThreads + 255) / 256), dim3(256), 0, stream, ni, ctx.lightInstDistOffset,
                           ctx.dLightCDF.as<float>(), dIntegral, ctx.dLightInstGuide.as<uint16_t>(), ctx.lightInstGuideCells);
        GFX_HIP(hipGetLastError());
    }
    else GFX_HIP(hipMemsetAsync(dIntegral, 0, 4 * sizeof(float), stream));
    span_table_build(ctx, stream);
}

} // namespace gfx
