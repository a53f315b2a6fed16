// pass_common.hip.h -- pieces shared by the per-pixel pass kernels (restir.hip, pathtrace.hip).
#pragma once
#include "internal.h"
#include "shading.hip.h"

namespace gfx {

struct Camera { f3 pos; m33 ori; float aspect, fovY; };
GFX_DEV Camera load_camera(const gfx_camera& c) {
    Camera r;
    r.pos = f3(c.position[0], c.position[1], c.position[2]);
    r.ori.r0 = f3(c.orientation[0], c.orientation[1], c.orientation[2]);
    r.ori.r1 = f3(c.orientation[3], c.orientation[4], c.orientation[5]);
    r.ori.r2 = f3(c.orientation[6], c.orientation[7], c.orientation[8]);
    r.aspect = c.aspect; r.fovY = c.fovY;
    return r;
}
GFX_DEV EnvMap load_env(const gfx_restir_static_params& s) {
    EnvMap e;
    e.texels = static_cast<const float4*>(s.envLightTexture);
    e.rowPDF = static_cast<const float*>(s.envRowPDF); e.rowCDF = static_cast<const float*>(s.envRowCDF);
    e.topPDF = static_cast<const float*>(s.envTopPDF); e.topCDF = static_cast<const float*>(s.envTopCDF);
    e.rowGuide = static_cast<const uint16_t*>(s.envRowGuide); e.topGuide = static_cast<const uint16_t*>(s.envTopGuide);
    e.rowTable = static_cast<const EnvRowRec*>(s.envRowTable);
    e.rowSketch = e.rowTable ? static_cast<const uint32_t*>(s.envRowSketch) : nullptr;
    e.w = s.envWidth; e.h = s.envHeight;
    return e;
}

// Dense ray-queue append for the threads with want == true.  EVERY thread of the block must call it
// (block-uniform control flow): the waves' counts are combined in LDS and the block takes ONE atomic
// on the queue head -- with one atomic per wave the 32 k waves of a full-HD launch serialise on that
// single address (profiles/r01f: k_shade_prepare 0.38 -> 0.1x ms).  Returns the slot or GFX_INVALID_SLOT.
GFX_DEV uint32_t queue_append(bool want, f3 org, f3 dir, float tmin, float tmax,
                              float4* rayOrg, float4* rayDir, uint32_t* rayCount) {
    __shared__ uint32_t qa[1 + 16];                       // [0] block base, [1 + w] count of wave w
    const unsigned long long mask = __ballot(want);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, numWaves = (blockDim.x + 63) >> 6;
    if (lane == 0) qa[1 + wave] = static_cast<uint32_t>(__popcll(mask));
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < numWaves; ++w) total += qa[1 + w];
        qa[0] = total ? atomicAdd(rayCount, total) : 0u;
    }
    __syncthreads();
    uint32_t base = qa[0];
    for (int w = 0; w < wave; ++w) base += qa[1 + w];
    __syncthreads();                                      // qa is reused by the next append of this block
    if (!want) return GFX_INVALID_SLOT;
    const uint32_t slot = base + __popcll(mask & ((1ull << lane) - 1ull));
    rayOrg[slot] = make_float4(org.x, org.y, org.z, tmin);
    rayDir[slot] = make_float4(dir.x, dir.y, dir.z, tmax);
    return slot;
}

// Several appends of one block at once: every thread asks for up to K slots (want[k]); the block takes ONE atomic
// for all of them and two barriers instead of three per kind.  The block's chunk of the queue is laid out kind by
// kind (all rays of kind 0, then kind 1, ...), so consecutive entries are still the same kind of ray from
// neighbouring pixels.  EVERY thread of the block must call it.  slot[k] = GFX_INVALID_SLOT where !want[k]; the
// caller writes the rays (queue_write).
template <int K>
GFX_DEV void queue_reserve(const bool (&want)[K], uint32_t* rayCount, uint32_t (&slot)[K]) {
    __shared__ uint32_t qr[1 + K + 16 * K];               // [0] block base, [1 + k] block total of kind k, then [wave][kind]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, numWaves = (blockDim.x + 63) >> 6;
    unsigned long long mask[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        mask[k] = __ballot(want[k]);
        if (lane == 0) qr[1 + K + wave * K + k] = static_cast<uint32_t>(__popcll(mask[k]));
    }
    __syncthreads();
    if (threadIdx.x < K) {
        uint32_t total = 0;
        for (int w = 0; w < numWaves; ++w) total += qr[1 + K + w * K + threadIdx.x];
        qr[1 + threadIdx.x] = total;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int k = 0; k < K; ++k) total += qr[1 + k];
        qr[0] = total ? atomicAdd(rayCount, total) : 0u;
    }
    __syncthreads();
    uint32_t kindBase = qr[0];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        uint32_t base = kindBase;
        for (int w = 0; w < wave; ++w) base += qr[1 + K + w * K + k];
        slot[k] = want[k] ? base + static_cast<uint32_t>(__popcll(mask[k] & ((1ull << lane) - 1ull))) : GFX_INVALID_SLOT;
        kindBase += qr[1 + k];
    }
    __syncthreads();                                      // qr is reused by the next call of this block
}
// Same idea for K different queues (one counter each): K threads take the K atomics side by side, two barriers.
template <int K>
GFX_DEV void queue_reserve_each(const bool (&want)[K], uint32_t* const (&counters)[K], uint32_t (&slot)[K]) {
    __shared__ uint32_t qe[K + 16 * K];                   // [k] base of queue k for this block, then [wave][kind]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, numWaves = (blockDim.x + 63) >> 6;
    unsigned long long mask[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        mask[k] = __ballot(want[k]);
        if (lane == 0) qe[K + wave * K + k] = static_cast<uint32_t>(__popcll(mask[k]));
    }
    __syncthreads();
    if (threadIdx.x < K) {
        uint32_t total = 0;
        for (int w = 0; w < numWaves; ++w) total += qe[K + w * K + threadIdx.x];
        qe[threadIdx.x] = total ? atomicAdd(counters[threadIdx.x], total) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        uint32_t base = qe[k];
        for (int w = 0; w < wave; ++w) base += qe[K + w * K + k];
        slot[k] = want[k] ? base + static_cast<uint32_t>(__popcll(mask[k] & ((1ull << lane) - 1ull))) : GFX_INVALID_SLOT;
    }
    __syncthreads();                                      // qe is reused by the next call of this block
}
GFX_DEV void queue_write(uint32_t slot, f3 org, f3 dir, float tmin, float tmax, float4* rayOrg, float4* rayDir) {
    if (slot == GFX_INVALID_SLOT) return;
    rayOrg[slot] = make_float4(org.x, org.y, org.z, tmin);
    rayDir[slot] = make_float4(dir.x, dir.y, dir.z, tmax);
}

// Wave-level variant (one atomic per wave, no barrier): for long kernels whose waves finish at very
// different times, where holding a block back at a barrier costs more than the spread-out atomics.
GFX_DEV uint32_t queue_append_wave(bool want, f3 org, f3 dir, float tmin, float tmax,
                                   float4* rayOrg, float4* rayDir, uint32_t* rayCount) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return GFX_INVALID_SLOT;
    const int lane = threadIdx.x & 63;
    const int leader = __builtin_ctzll(mask);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(rayCount, static_cast<uint32_t>(__popcll(mask)));
    base = __shfl(base, leader);
    if (!want) return GFX_INVALID_SLOT;
    const uint32_t slot = base + __popcll(mask & ((1ull << lane) - 1ull));
    rayOrg[slot] = make_float4(org.x, org.y, org.z, tmin);
    rayDir[slot] = make_float4(dir.x, dir.y, dir.z, tmax);
    return slot;
}

} // namespace gfx
