// trace.hip -- the wavefront ray-query kernel: persistent waves pull rays from a queue, traverse
// the BVH8 with an LDS stack, and refill finished lanes in place.
//
// Replaces optixTrace on a W x H launch (utils/optix_util.h:557-603, 2149-2151).  Rays arrive as a
// dense SoA queue (org.xyz|tmin, dir.xyz|tmax) written by the producing pass; results go to a dense
// array indexed like the queue.  Scheduling: a wave owns 64 lanes; whenever at least
// kRefillThreshold lanes are idle and the queue is not exhausted, the wave takes one atomic ticket
// for exactly popcount(idle) rays (ballot + mbcnt compaction) and the idle lanes start new rays
// while the others keep traversing -- the SIMT analogue of OptiX's hardware ray scheduling.
#include "bvh8.hip.h"
#include "internal.h"

namespace gfx {

constexpr int kTraceBlock = 256;
constexpr int kRefillThreshold = 16;

struct TraceArgs {
    DevAccel accel;
    const float4* __restrict__ rayOrgTmin;
    const float4* __restrict__ rayDirTmax;
    const uint32_t* numRaysPtr;
    uint32_t numRays;
    void* out;
    uint32_t* ticket;           // queue head (zeroed before the launch)
    uint2* spill;               // kSpillStackDepth entries per thread of the grid
    unsigned long long* counters; // optional: node fetches, triangle fetches, rays, spills
};

template <bool ANY_HIT, bool COUNT>
__global__ __launch_bounds__(kTraceBlock) void k_trace(TraceArgs a) {
    __shared__ uint2 ldsStack[kLdsStackDepth * kTraceBlock];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    LaneStack stack;
    stack.lds = ldsStack + tid;
    stack.ldsStride = kTraceBlock;
    stack.spill = a.spill + (static_cast<size_t>(blockIdx.x) * kTraceBlock + tid) * kSpillStackDepth;
    stack.sp = 0;
    const uint32_t n = a.numRaysPtr ? *a.numRaysPtr : a.numRays;
    const bool hasNodes = a.accel.numNodes != 0;

    Traversal tr;
    tr.active = false;
    uint32_t rayIdx = 0;
    bool exhausted = false;           // wave-uniform: the queue has no more rays
    TraceCounters cnt = { 0, 0, 0 };
    uint32_t raysDone = 0;

    while (true) {
        const unsigned long long idleMask = __ballot(!tr.active);
        const int numIdle = __popcll(idleMask);
        if (!exhausted && numIdle >= kRefillThreshold) {
            uint32_t base = 0;
            if (lane == __builtin_ctzll(idleMask)) base = atomicAdd(a.ticket, static_cast<uint32_t>(numIdle));
            base = __shfl(base, __builtin_ctzll(idleMask));
            if (base + numIdle >= n) exhausted = true;
            if (!tr.active) {
                const uint32_t rank = __popcll(idleMask & ((1ull << lane) - 1ull));
                const uint32_t i = base + rank;
                if (i < n) {
                    const float4 o = a.rayOrgTmin[i];
                    const float4 d = a.rayDirTmax[i];
                    rayIdx = i;
                    tr.begin(f3(o.x, o.y, o.z), f3(d.x, d.y, d.z), o.w, d.w, stack, hasNodes);
                    if (!hasNodes || !(d.w > o.w)) {   // empty interval or empty scene: immediate miss
                        tr.active = false;
                        if (ANY_HIT) static_cast<uint32_t*>(a.out)[i] = 0u;
                        else { gfx_hit h; h.dist = d.w; h.bcB = 0; h.bcC = 0; h.triIndex = GFX_INVALID_SLOT; static_cast<gfx_hit*>(a.out)[i] = h; }
                        if (COUNT) ++raysDone;
                    }
                }
            }
        }
        if (__ballot(tr.active) == 0ull) {
            if (exhausted) break;
            continue;
        }
        if (tr.active) {
            const bool more = tr.template step<ANY_HIT, COUNT>(a.accel, stack, cnt);
            if (!more) {
                if (ANY_HIT) static_cast<uint32_t*>(a.out)[rayIdx] = tr.hit.tri != GFX_INVALID_SLOT ? 1u : 0u;
                else {
                    gfx_hit h; h.dist = tr.hit.t; h.bcB = tr.hit.bcB; h.bcC = tr.hit.bcC; h.triIndex = tr.hit.tri;
                    static_cast<gfx_hit*>(a.out)[rayIdx] = h;
                }
                if (COUNT) ++raysDone;
            }
        }
    }
    if (COUNT && a.counters) {
        // wave-level reduction, one atomic per wave and counter
        unsigned long long v[4] = { cnt.nodes, cnt.tris, raysDone, cnt.spills };
        for (int k = 0; k < 4; ++k) {
            unsigned long long x = v[k];
            for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
            if (lane == 0 && x) atomicAdd(a.counters + k, x);
        }
    }
}

static uint32_t persistent_grid(Context& ctx) {
    static int numCUs = 0;
    if (!numCUs) {
        hipDeviceProp_t prop;
        GFX_HIP(hipGetDeviceProperties(&prop, ctx.device));
        numCUs = prop.multiProcessorCount;
    }
    return static_cast<uint32_t>(numCUs) * 4u;   // 4 blocks of 256 per CU (LDS: 4 x 24 KiB)
}

void trace_launch(Context& ctx, hipStream_t stream, const TraceLaunch& t) {
    const uint32_t grid = persistent_grid(ctx);
    ctx.spill.reserve(sizeof(uint2) * static_cast<size_t>(grid) * kTraceBlock * kSpillStackDepth);
    ctx.smallCounters.reserve(256);
    uint32_t* ticket = ctx.smallCounters.as<uint32_t>();
    GFX_HIP(hipMemsetAsync(ticket, 0, sizeof(uint32_t), stream));
    TraceArgs a;
    a.accel = t.accel;
    a.rayOrgTmin = t.rayOrgTmin; a.rayDirTmax = t.rayDirTmax;
    a.numRaysPtr = t.numRaysPtr; a.numRays = t.numRays;
    a.out = t.out; a.ticket = ticket; a.spill = ctx.spill.as<uint2>();
    a.counters = ctx.countersEnabled ? ctx.dTraceCounters.as<unsigned long long>() : nullptr;
    const bool any = t.mode == GFX_TRACE_ANY;
    ScopedKernelTimer timer(ctx, stream, any ? "trace_any" : "trace_closest");
    if (ctx.countersEnabled) {
        if (any) hipLaunchKernelGGL((k_trace<true, true>), dim3(grid), dim3(kTraceBlock), 0, stream, a);
        else hipLaunchKernelGGL((k_trace<false, true>), dim3(grid), dim3(kTraceBlock), 0, stream, a);
    }
    else {
        if (any) hipLaunchKernelGGL((k_trace<true, false>), dim3(grid), dim3(kTraceBlock), 0, stream, a);
        else hipLaunchKernelGGL((k_trace<false, false>), dim3(grid), dim3(kTraceBlock), 0, stream, a);
    }
    GFX_HIP(hipGetLastError());
}

} // namespace gfx
