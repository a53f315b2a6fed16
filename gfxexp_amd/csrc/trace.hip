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
#include "coop_fetch.hip.h"
#include "internal.h"

namespace gfx {

constexpr int kTraceBlock = 256;
// The ray queue is handed out in chunks of `ticketBatch` rays.  ONE device-scope counter for all chunks is a serial resource: atomics
// on one address retire at ~13 ns each on MI355X whatever the grid does, so 2.07 M primary rays in chunks of 64 cost 0.42 ms of
// atomics alone -- the floor every launch of rounds 1-3a sat on (tools/trace_tail.py: rays with an EMPTY interval took 0.418 ms,
// the full closest-hit traversal 0.489).  Chunk j now belongs to counter j % kTicketCounters (each on its own 128-byte line): a
// wave draws from "its" counter and moves on to the next one when that runs dry (a plain load first: only a wave that can
// still get a chunk issues the atomic).
constexpr uint32_t kTicketCounters = 32;
constexpr uint32_t kTicketStride = 32;      // words between counters

struct TraceArgs {
    DevAccel accel;
    const float4* __restrict__ rayOrgTmin;
    const float4* __restrict__ rayDirTmax;
    const uint32_t* numRaysPtr;
    uint32_t numRays;
    void* out;
    uint32_t* ticket;           // kTicketCounters queue heads, kTicketStride words apart (zero at the start of the launch)
    uint32_t* ticketNext;       // the area the NEXT launch on this buffer draws from: block 0 zeroes it
    uint32_t* zeroWords[2];     // optional: words block 0 sets to zero (TraceLaunch::zeroWords)
    uint2* spill;               // kSpillStackDepth entries per thread of the grid
    unsigned long long* counters; // optional: node fetches, triangle fetches, rays, spills
    uint32_t* perRayItems;        // optional (counting launches): items (nodes + triangle records) each ray fetched, indexed like the queue
    unsigned long long* diag;     // optional (counting launches): wave iterations, item-lanes, drain iterations, drain item-lanes
    int refillThreshold;        // refill when at least this many lanes are idle
    int ticketBatch;            // rays bought per device atomic
    int hintFromOut;            // closest-hit launches: out[i].triIndex of the previous launch is ray i's first triangle to test
};

// fetch_items: coop_fetch.hip.h (the cooperative 64-byte gather; the candidate kernel of restir.hip uses it for emitter records).
#ifndef GFX_TRACE_MIN_WAVES
#define GFX_TRACE_MIN_WAVES 1
#endif
template <bool ANY_HIT, bool COUNT>
__global__ __launch_bounds__(kTraceBlock, GFX_TRACE_MIN_WAVES) void k_trace(TraceArgs a) {
    __shared__ uint2 ldsStack[kTraceLdsStackDepth * kTraceBlock];
    __shared__ __attribute__((aligned(16))) uint4 fetchBuf[kTraceBlock * 4];   // 4 KiB per wave
    __shared__ uint8_t octPerm[8 * 256];              // bvh8.hip.h process_node: hit masks in (slot ^ octant) order by table
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    for (int i = tid; i < 8 * 256; i += kTraceBlock) {
        const uint32_t o = static_cast<uint32_t>(i) >> 8;
        uint32_t m = static_cast<uint32_t>(i) & 0xFFu;
        if (o & 1u) m = ((m & 0x55u) << 1) | ((m & 0xAAu) >> 1);
        if (o & 2u) m = ((m & 0x33u) << 2) | ((m & 0xCCu) >> 2);
        if (o & 4u) m = ((m & 0x0Fu) << 4) | ((m & 0xF0u) >> 4);
        octPerm[i] = static_cast<uint8_t>(m);
    }
    __syncthreads();
    uint4* waveBuf = fetchBuf + 256 * __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, kept scalar
    LaneStack stack;
    stack.lds = ldsStack + tid;
    stack.ldsStride = kTraceBlock;
    stack.spill = a.spill + (static_cast<size_t>(blockIdx.x) * kTraceBlock + tid) * kSpillStackDepth;
    stack.sp = 0;
    stack.spillCap = kSpillStackDepth;
    stack.ldsDepth = kTraceLdsStackDepth;
    // the other ticket area belongs to the next launch (stream order: nobody reads it while this kernel runs)
    if (blockIdx.x == 0 && tid < static_cast<int>(kTicketCounters)) a.ticketNext[tid * kTicketStride] = 0u;
    if (blockIdx.x == 0 && tid < 2 && a.zeroWords[tid]) *a.zeroWords[tid] = 0u;
    const uint32_t n = a.numRaysPtr ? *a.numRaysPtr : a.numRays;
    const bool hasNodes = a.accel.numNodes != 0;
    const f3 sceneMaxAbs = scene_max_abs(a.accel);

    Traversal tr;
    tr.active = false;
    uint32_t rayIdx = 0;
    bool exhausted = false;           // wave-uniform: the queue has no more rays
    uint32_t waveNext = 0, waveEnd = 0; // wave-uniform: rays [waveNext, waveEnd) already ticketed for this wave
    uint32_t myCounter = (blockIdx.x * (kTraceBlock / 64) + (tid >> 6)) % kTicketCounters;   // wave-uniform: the counter this wave draws from
    uint32_t dryCounters = 0;         // wave-uniform: counters this wave has seen run dry
    const uint32_t numChunks = (n + static_cast<uint32_t>(a.ticketBatch) - 1u) / static_cast<uint32_t>(a.ticketBatch);
    TraceCounters cnt = { 0, 0, 0 };
    uint32_t raysDone = 0, rayItems = 0;
    uint32_t diagIter = 0, diagLanes = 0, diagDrainIter = 0, diagDrainLanes = 0;   // wave-uniform (COUNT only)
    // COUNT only: where a wave's clock cycles go (s_memtime): ray refill (ticket + ray loads + setup), item fetch (issue to data in
    // registers), item processing (slab / triangle tests); the rest is item selection and loop overhead
    unsigned long long cycRefill = 0, cycFetch = 0, cycProcess = 0;
    const unsigned long long cycStart = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;

    auto write_result = [&]() {
        if (ANY_HIT) static_cast<uint32_t*>(a.out)[rayIdx] = tr.hit.tri != GFX_INVALID_SLOT ? 1u : 0u;
        else {
            gfx_hit h; h.dist = tr.hit.t; h.bcB = tr.hit.bcB; h.bcC = tr.hit.bcC; h.triIndex = tr.hit.tri;
            static_cast<gfx_hit*>(a.out)[rayIdx] = h;
        }
        if (COUNT) { ++raysDone; if (a.perRayItems) a.perRayItems[rayIdx] = rayItems; }
    };

    while (true) {
        const unsigned long long cyc0 = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
        const unsigned long long idleMask = __ballot(!tr.active);
        const int numIdle = __popcll(idleMask);
        bool newRay = false;
        float4 rayO, rayD;             // read only under newRay (no default: eight v_mov per wave iteration, a ray refilled or not)
        uint32_t hint = 0xFFFFFFFFu;
        if (!exhausted && numIdle >= a.refillThreshold) {
            // wave-local ticket range: one device atomic buys a batch of rays, bought on demand
            // (buying ahead of need strands rays in waves that finish late; measured slower)
            if (waveNext == waveEnd) {
                // counter c hands out chunks c, c + K, c + 2 K, ...
                uint32_t chunk = 0xFFFFFFFFu, c = myCounter, dry = dryCounters;
                if (lane == 0) {
                    for (uint32_t attempt = 0; attempt < kTicketCounters && chunk == 0xFFFFFFFFu; ++attempt) {
                        const uint32_t cc = (myCounter + attempt) % kTicketCounters;
                        if (dry & (1u << cc)) continue;
                        uint32_t* counter = a.ticket + cc * kTicketStride;
                        const uint32_t seen = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (static_cast<uint64_t>(seen) * kTicketCounters + cc < numChunks) {
                            const uint32_t t = atomicAdd(counter, 1u);
                            const uint64_t j = static_cast<uint64_t>(t) * kTicketCounters + cc;
                            if (j < numChunks) { chunk = static_cast<uint32_t>(j); c = cc; continue; }
                        }
                        dry |= 1u << cc;
                    }
                }
                chunk = __shfl(chunk, 0); myCounter = __shfl(c, 0); dryCounters = __shfl(dry, 0);
                if (chunk == 0xFFFFFFFFu) exhausted = true;
                else { waveNext = chunk * static_cast<uint32_t>(a.ticketBatch); waveEnd = min(waveNext + static_cast<uint32_t>(a.ticketBatch), n); }
            }
            const uint32_t take = min(static_cast<uint32_t>(numIdle), waveEnd - waveNext);
            if (!tr.active) {
                const uint32_t rank = __popcll(idleMask & ((1ull << lane) - 1ull));
                if (rank < take) {
                    // only the loads are issued here: a new ray's first item is the root node whatever the ray is, so it asks for it
                    // in this very iteration and its origin / direction arrive together with the items (one wait for both)
                    const uint32_t i = waveNext + rank;
                    rayO = a.rayOrgTmin[i];
                    rayD = a.rayDirTmax[i];
                    // Temporal hint: the triangle this ray slot hit in the previous launch (the same pixel's primary ray one frame
                    // ago) is tested right after the root -- as the one pending "leaf" of a node that does not exist.  When it is hit
                    // again the traversal descends with the right upper bound and skips what lies behind it; the answer cannot change
                    // (closest hit with the order-independent tie rule), a stale or garbage index costs one triangle test.
                    if (!ANY_HIT && a.hintFromOut) hint = static_cast<const gfx_hit*>(a.out)[i].triIndex;
                    rayIdx = i;
                    newRay = true;
                }
            }
            waveNext += take;
        }
        if (COUNT) cycRefill += __builtin_amdgcn_s_memtime() - cyc0;
        if (__ballot(tr.active || newRay) == 0ull) {
            if (exhausted) break;
            continue;
        }
        uint32_t code = kItemNone;
        if (tr.active) {
            code = tr.next_item(stack, a.accel.triItemOffset);
            if (code == kItemNone) write_result();          // traversal finished
        }
        if (newRay && hasNodes) code = 0u;                      // the root node
        if (COUNT) {
            if (newRay) rayItems = 0;
            if (code != kItemNone) ++rayItems;
            const int held = __popcll(__ballot(code != kItemNone));
            ++diagIter; diagLanes += held;
            if (exhausted) { ++diagDrainIter; diagDrainLanes += held; }
        }
        uint4 link;                    // read only by process_node, i.e. under the condition of its load
        if (code != kItemNone && !(code & kItemTri)) link = reinterpret_cast<const uint4*>(a.accel.links)[code];   // in flight with the item fetch
        uint4 q0, q1, q2, q3;
        const unsigned long long cyc1 = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
        fetch_items(code, a.accel, waveBuf, lane, q0, q1, q2, q3);
        if (COUNT) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); cycFetch += __builtin_amdgcn_s_memtime() - cyc1; }
        const unsigned long long cyc2 = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
        if (newRay) {                                           // its origin and direction have arrived with the items
            tr.begin(f3(rayO.x, rayO.y, rayO.z), f3(rayD.x, rayD.y, rayD.z), rayO.w, rayD.w, stack, hasNodes, sceneMaxAbs);
            tr.grp.y = 0u;                                      // the root (begin's one-child group) is this iteration's item
            if (!hasNodes || !(rayD.w > rayO.w)) {              // empty interval or empty scene: immediate miss
                tr.active = false;
                code = kItemNone;
                write_result();
                if (COUNT) --raysDone;                           // the counters report rays that were traversed: a queue entry without a
            }                                                   // ray (emit_ray_at_slot, padding slots) is not one
        }
        if (code != kItemNone) {
            if (code & kItemTri) {
                if (!tr.template process_triangle<ANY_HIT, COUNT>((code & 0x7FFFFFFFu) - a.accel.triItemOffset, q0, q1, q2, q3, a.accel.tris, cnt))
                    write_result();                         // any-hit ray found its occluder
            }
            else tr.template process_node<COUNT>(q0, q1, q2, q3, link, stack, cnt, (const __attribute__((address_space(3))) uint8_t*)octPerm);
        }
        if (!ANY_HIT && newRay && tr.active && hint < a.accel.numTris && tr.triMask == 0u) { tr.triBase = hint; tr.triMask = 0x0101u; }
        if (COUNT) cycProcess += __builtin_amdgcn_s_memtime() - cyc2;
    }
    if (COUNT && a.diag && lane == 0) {
        atomicAdd(a.diag + 0, static_cast<unsigned long long>(diagIter)); atomicAdd(a.diag + 1, static_cast<unsigned long long>(diagLanes));
        atomicAdd(a.diag + 2, static_cast<unsigned long long>(diagDrainIter)); atomicAdd(a.diag + 3, static_cast<unsigned long long>(diagDrainLanes));
        atomicAdd(a.diag + 4, __builtin_amdgcn_s_memtime() - cycStart); atomicAdd(a.diag + 5, cycRefill);
        atomicAdd(a.diag + 6, cycFetch); atomicAdd(a.diag + 7, cycProcess);
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
    return static_cast<uint32_t>(ctx.numCUs) * static_cast<uint32_t>(ctx.tune.traceBlocksPerCU);
}

void trace_launch(Context& ctx, hipStream_t stream, const TraceLaunch& t) {
    const uint32_t grid = persistent_grid(ctx);
    DevBuf& spill = t.spill ? *t.spill : ctx.spill;
    DevBuf& small = t.counters ? *t.counters : ctx.smallCounters;
    spill.reserve(sizeof(uint2) * static_cast<size_t>(grid) * kTraceBlock * kSpillStackDepth);
    constexpr size_t kTicketAreaBytes = sizeof(uint32_t) * kTicketCounters * kTicketStride;
    static_assert(2 * kTicketAreaBytes <= kSmallCountersBytes - kSmallCountersTicketOffset, "two ticket areas must fit");
    small.reserve(kSmallCountersBytes);
    // Two ticket areas per counter buffer: a launch draws from one and its block 0 zeroes the other, which the next launch on this
    // buffer (same stream by construction: the pipelined G-buffer pass has its own buffer) then finds zero -- no memset per launch.
    Context::TicketState& ts = ctx.ticketState[&small == &ctx.smallCounters ? 0 : &small == &ctx.auxCounters ? 2 : 1];
    char* areas = static_cast<char*>(small.p) + kSmallCountersTicketOffset;
    if (ts.zeroed && ts.buffer == small.p && ts.stream != stream) {
        // the "previous launch zeroed my area" hand-over is stream order; a launch on another stream first waits for the last launch on
        // this buffer (an event recorded behind it: the stream it ran on may be gone by now -- a renderer's private stream), then starts
        // over with two zeroed areas (a caller that alternates streams pays a wait and a memset per switch)
        if (ts.lastLaunch) GFX_HIP(hipStreamWaitEvent(stream, ts.lastLaunch, 0));
        ts.zeroed = false;
    }
    if (!ts.zeroed || ts.buffer != small.p) {
        GFX_HIP(hipMemsetAsync(areas, 0, 2 * kTicketAreaBytes, stream));
        ts.zeroed = true; ts.next = 0; ts.buffer = small.p;
    }
    ts.stream = stream;
    uint32_t* ticket = reinterpret_cast<uint32_t*>(areas + ts.next * kTicketAreaBytes);
    uint32_t* ticketNext = reinterpret_cast<uint32_t*>(areas + (ts.next ^ 1u) * kTicketAreaBytes);
    ts.next ^= 1u;
    TraceArgs a;
    a.accel = t.accel;
    a.rayOrgTmin = t.rayOrgTmin; a.rayDirTmax = t.rayDirTmax;
    a.numRaysPtr = t.numRaysPtr; a.numRays = t.numRays;
    a.out = t.out; a.ticket = ticket; a.ticketNext = ticketNext; a.spill = spill.as<uint2>();
    a.zeroWords[0] = t.zeroWords[0]; a.zeroWords[1] = t.zeroWords[1];
    // the context's own counters keep any-hit launches in [0..3] and closest-hit launches in [4..7]
    a.counters = ctx.countersEnabled ? ctx.dTraceCounters.as<unsigned long long>() + ((ctx.countersSplit && t.mode != GFX_TRACE_ANY) ? 4 : 0) : nullptr;
    a.diag = nullptr;
    a.perRayItems = ctx.countersEnabled ? t.perRayItems : nullptr;
    if (ctx.countersEnabled) {
        if (!ctx.dTraceDiag.p) { ctx.dTraceDiag.reserve(64); GFX_HIP(hipMemsetAsync(ctx.dTraceDiag.p, 0, 64, stream)); }
        a.diag = ctx.dTraceDiag.as<unsigned long long>();
    }
    a.hintFromOut = (t.hintFromOut && t.mode != GFX_TRACE_ANY && ctx.tune.temporalHints) ? 1 : 0;
    a.refillThreshold = ctx.tune.traceRefill;
    a.ticketBatch = ctx.tune.traceBatch;
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
    if (!ts.lastLaunch) GFX_HIP(hipEventCreateWithFlags(&ts.lastLaunch, hipEventDisableTiming));
    GFX_HIP(hipEventRecord(ts.lastLaunch, stream));
}

} // namespace gfx
