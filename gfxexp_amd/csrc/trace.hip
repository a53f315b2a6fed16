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
    uint32_t* anyHint;          // optional (any-hit): per ray slot, the occluder the slot's ray found in the previous launch as triangle + 1
                                // (0: none); read when the ray starts, rewritten with the result (TraceLaunch::anyHint)
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
    int keepAnyHint;            // any-hit launches with anyHint: an unoccluded ray leaves its slot's hint as it is
};

// ---------------------------------------------------------------- ray segments (SEG > 1)
// A launch that does not fill the machine lasts as long as its longest ray: one ray per lane, nothing to refill, up to 86 dependent
// 64-byte fetches at 1.5-3 us each for a primary ray of the bench street while the mean ray needs 14 (profiles/r02_band_notes.txt) --
// the wall a rank's band of an 8-way split frame runs into.  Such a launch cuts every ray into SEG pieces along t and gives each
// piece its own lane: SEG adjacent lanes hold the SEG segments (lo_k, hi_k) of one ray, each a traversal of its own with the
// segment as its interval, so the dependent chain of a long ray shrinks to about 1/SEG of its leaf-level items plus the common
// ancestors.  The segments PARTITION the ray's interval exactly -- whatever the split points are, every t with tmin < t < tmax
// falls into exactly one of them (segment k accepts m_k < t <= m_k+1, the first one tmin < t, the last one t < tmax) -- and the
// triangle test does not read the interval (ray_triangle computes t, b, c from origin and direction alone), so:
//   any-hit      the ray is occluded iff one of its segments is;
//   closest-hit  the hit is the hit of the first segment that has one (equal t -> same segment -> the tie rule applies inside it).
// Same answers, bit for bit; the split points only decide how evenly the work is shared.  They are taken uniformly over the
// part of the ray inside the root's box (read from node 0 once per wave).  A segment that finds a hit retires the segments it
// makes pointless (any-hit: all others; closest-hit: the ones behind it) through one ballot per iteration; the group's leader
// lane writes the merged result once all SEG lanes are idle, and only then is the group refilled.
template <int SEG> struct SegConst {
    static constexpr unsigned long long leaderBits = SEG == 2 ? 0x5555555555555555ull : SEG == 4 ? 0x1111111111111111ull : SEG == 8 ? 0x0101010101010101ull : ~0ull;
    static constexpr uint32_t groupMask = (1u << SEG) - 1u;
};

struct SceneBox { f3 lo, hi; };
// World box of the root's children in the traversal's own decode (origin + q * scale): conservative enough for choosing split points,
// which is all it is used for.  Every lane computes the same values (the loads are broadcasts).
GFX_DEV SceneBox root_box(const DevAccel& acc) {
    const uint4* n = reinterpret_cast<const uint4*>(acc.nodes);
    const uint4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
    const uint32_t valid = reinterpret_cast<const uint4*>(acc.links)[0].z;
    const f3 origin(bits2f(n0.x), bits2f(n0.y), bits2f(n0.z));
    const f3 scale(bits2f(bfe(n0.w, 0, 8) << 23), bits2f(bfe(n0.w, 8, 8) << 23), bits2f(bfe(n0.w, 16, 8) << 23));
    const uint32_t lx[2] = { n1.x, n1.y }, ly[2] = { n1.z, n1.w }, lz[2] = { n2.x, n2.y };
    const uint32_t hx[2] = { n2.z, n2.w }, hy[2] = { n3.x, n3.y }, hz[2] = { n3.z, n3.w };
    SceneBox b; b.lo = f3(3.0e38f); b.hi = f3(-3.0e38f);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (!((valid >> s) & 1u)) continue;
        const int w = s >> 2, sh = (s & 3) * 8;
        b.lo.x = fminf(b.lo.x, fmaf(static_cast<float>((lx[w] >> sh) & 0xFFu), scale.x, origin.x));
        b.lo.y = fminf(b.lo.y, fmaf(static_cast<float>((ly[w] >> sh) & 0xFFu), scale.y, origin.y));
        b.lo.z = fminf(b.lo.z, fmaf(static_cast<float>((lz[w] >> sh) & 0xFFu), scale.z, origin.z));
        b.hi.x = fmaxf(b.hi.x, fmaf(static_cast<float>((hx[w] >> sh) & 0xFFu), scale.x, origin.x));
        b.hi.y = fmaxf(b.hi.y, fmaf(static_cast<float>((hy[w] >> sh) & 0xFFu), scale.y, origin.y));
        b.hi.z = fmaxf(b.hi.z, fmaf(static_cast<float>((hz[w] >> sh) & 0xFFu), scale.z, origin.z));
    }
    return b;
}
// the next float above m (m finite, not NaN): t < next_up(m)  <=>  t <= m
GFX_DEV float next_up(float m) {
    const float x = m + 0.0f;                       // -0 -> +0
    const uint32_t b = f2bits(x);
    return bits2f(x >= 0.0f ? b + 1u : b - 1u);
}
// Interval of segment `seg` of SEG of the ray (o, d, tmin, tmax), tmax > tmin: see the partition argument above.
template <int SEG>
GFX_DEV void segment_interval(f3 o, f3 d, float tmin, float tmax, const SceneBox& box, int seg, float& lo, float& hi) {
    const float dx = fabsf(d.x) < 1e-20f ? copysignf(1e-20f, d.x) : d.x;
    const float dy = fabsf(d.y) < 1e-20f ? copysignf(1e-20f, d.y) : d.y;
    const float dz = fabsf(d.z) < 1e-20f ? copysignf(1e-20f, d.z) : d.z;
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    const float ax = (box.lo.x - o.x) * ix, bx = (box.hi.x - o.x) * ix;
    const float ay = (box.lo.y - o.y) * iy, by = (box.hi.y - o.y) * iy;
    const float az = (box.lo.z - o.z) * iz, bz = (box.hi.z - o.z) * iz;
    float a = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
    float b = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    if (!(b > a)) { a = tmin; b = tmin; }           // the ray misses the box (or NaNs): the last segment gets everything
    // split point k (1 .. SEG - 1), the same expression in both lanes that use it; NaN / out of range -> clamped into [tmin, tmax]
    auto split = [&](int k) { return fminf(fmaxf(fmaf(b - a, static_cast<float>(k) * (1.0f / SEG), a), tmin), tmax); };
    lo = seg == 0 ? tmin : split(seg);
    hi = seg == SEG - 1 ? tmax : fminf(next_up(split(seg + 1)), tmax);
}

// fetch_items: coop_fetch.hip.h (the cooperative 64-byte gather; the candidate kernel of restir.hip uses it for emitter records).
#ifndef GFX_TRACE_MIN_WAVES
#define GFX_TRACE_MIN_WAVES 1
#endif
template <bool ANY_HIT, bool COUNT, int SEG>
__global__ __launch_bounds__(kTraceBlock, GFX_TRACE_MIN_WAVES) void k_trace(TraceArgs a) {
    __shared__ uint2 ldsStack[kLdsStackDepth * kTraceBlock];
    __shared__ __attribute__((aligned(16))) uint4 fetchBuf[kTraceBlock * 4];   // 4 KiB per wave
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int seg = lane & (SEG - 1);                  // which segment of its ray this lane traverses (SEG == 1: the ray)
    const int leader = lane & ~(SEG - 1);              // first lane of the group of SEG lanes that share a ray
    uint4* waveBuf = fetchBuf + 256 * __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, kept scalar
    LaneStack stack;
    stack.lds = ldsStack + tid;
    stack.ldsStride = kTraceBlock;
    stack.spill = a.spill + (static_cast<size_t>(blockIdx.x) * kTraceBlock + tid) * kSpillStackDepth;
    stack.sp = 0;
    // the other ticket area belongs to the next launch (stream order: nobody reads it while this kernel runs)
    if (blockIdx.x == 0 && tid < static_cast<int>(kTicketCounters)) a.ticketNext[tid * kTicketStride] = 0u;
    if (blockIdx.x == 0 && tid < 2 && a.zeroWords[tid]) *a.zeroWords[tid] = 0u;
    const uint32_t n = a.numRaysPtr ? *a.numRaysPtr : a.numRays;
    const bool hasNodes = a.accel.numNodes != 0;
    SceneBox box; box.lo = f3(0.0f); box.hi = f3(0.0f);
    if (SEG > 1 && hasNodes) box = root_box(a.accel);

    Traversal tr;
    tr.active = false;
    tr.hit.tri = GFX_INVALID_SLOT;
    uint32_t rayIdx = 0;
    bool hasRay = false;              // SEG > 1: the group holds a ray whose merged result is not written yet (same in all its lanes)
    bool exhausted = false;           // wave-uniform: the queue has no more rays
    uint32_t waveNext = 0, waveEnd = 0; // wave-uniform: rays [waveNext, waveEnd) already ticketed for this wave
    uint32_t myCounter = (blockIdx.x * (kTraceBlock / 64) + (tid >> 6)) % kTicketCounters;   // wave-uniform: the counter this wave draws from
    uint32_t dryCounters = 0;         // wave-uniform: counters this wave has seen run dry
    const uint32_t numChunks = (n + static_cast<uint32_t>(a.ticketBatch) - 1u) / static_cast<uint32_t>(a.ticketBatch);
    TraceCounters cnt = { 0, 0, 0 };
    uint32_t raysDone = 0, rayItems = 0;
    bool traversed = false;           // COUNT: this lane's ray (segment) entered the tree
    uint32_t diagIter = 0, diagLanes = 0, diagDrainIter = 0, diagDrainLanes = 0;   // wave-uniform (COUNT only)
    // COUNT only: where a wave's clock cycles go (s_memtime): ray refill (ticket + ray loads + setup), item fetch (issue to data in
    // registers), item processing (slab / triangle tests); the rest is item selection and loop overhead
    unsigned long long cycRefill = 0, cycFetch = 0, cycProcess = 0;
    const unsigned long long cycStart = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;

    // SEG == 1: the lane's own result, written the moment its ray finishes.  SEG > 1: nothing here -- the group's leader writes the
    // merged result at the top of the loop once all its lanes are idle.
    auto write_result = [&]() {
        if (SEG > 1) return;
        if (ANY_HIT) {
            const bool occluded = tr.hit.tri != GFX_INVALID_SLOT;
            static_cast<uint32_t*>(a.out)[rayIdx] = occluded ? 1u : 0u;
            if (a.anyHint && (occluded || !a.keepAnyHint)) a.anyHint[rayIdx] = occluded ? tr.hit.tri + 1u : 0u;
        }
        else {
            gfx_hit h; h.dist = tr.hit.t; h.bcB = tr.hit.bcB; h.bcC = tr.hit.bcC; h.triIndex = tr.hit.tri;
            static_cast<gfx_hit*>(a.out)[rayIdx] = h;
        }
        if (COUNT) { ++raysDone; if (a.perRayItems) a.perRayItems[rayIdx] = rayItems; }
    };

    while (true) {
        const unsigned long long cyc0 = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
        unsigned long long idleMask = __ballot(!tr.active);     // SEG == 1: idle lanes; SEG > 1: leader bits of refillable groups (below)
        if (SEG > 1) {
            const bool groupIdle = ((idleMask >> leader) & SegConst<SEG>::groupMask) == SegConst<SEG>::groupMask;
            const bool finish = hasRay && groupIdle;
            if (__ballot(finish) != 0ull) {
                // the first segment with a hit holds the ray's result (any-hit: any of them will do)
                const unsigned long long found = __ballot(tr.hit.tri != GFX_INVALID_SLOT);
                const uint32_t gbits = static_cast<uint32_t>(found >> leader) & SegConst<SEG>::groupMask;
                const int src = leader + (gbits ? __builtin_ctz(gbits) : 0);
                const uint32_t tri = __shfl(tr.hit.tri, src);
                if (ANY_HIT) {
                    if (finish && seg == 0) {
                        static_cast<uint32_t*>(a.out)[rayIdx] = gbits ? 1u : 0u;
                        if (a.anyHint && (gbits || !a.keepAnyHint)) a.anyHint[rayIdx] = gbits ? tri + 1u : 0u;
                    }
                }
                else {
                    gfx_hit h;
                    h.dist = __shfl(tr.hit.t, src); h.bcB = __shfl(tr.hit.bcB, src); h.bcC = __shfl(tr.hit.bcC, src); h.triIndex = tri;
                    // no segment has a hit: every lane still holds bcB = bcC = 0 and its own upper bound; the ray's is the last segment's
                    const float tmaxRay = __shfl(tr.hit.t, leader + SEG - 1);
                    if (!gbits) h.dist = tmaxRay;
                    if (finish && seg == 0) static_cast<gfx_hit*>(a.out)[rayIdx] = h;
                }
                if (COUNT) {
                    uint32_t items = rayItems;
                    bool any = traversed;
#pragma unroll
                    for (int off = 1; off < SEG; off <<= 1) { items += __shfl_xor(items, off); const int other = __shfl_xor(any ? 1 : 0, off); any = any || other != 0; }
                    if (finish && seg == 0) { if (any) ++raysDone; if (a.perRayItems) a.perRayItems[rayIdx] = items; }
                }
                if (finish) hasRay = false;
            }
            idleMask = __ballot(!hasRay) & SegConst<SEG>::leaderBits;
        }
        const int numIdle = __popcll(idleMask);                 // idle lanes (SEG == 1) / idle groups
        bool newRay = false;
        float4 rayO = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rayD = rayO;
        uint32_t hint = 0xFFFFFFFFu;
        if (!exhausted && numIdle * SEG >= a.refillThreshold) {
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
            if (SEG > 1 ? !hasRay : !tr.active) {
                const uint32_t rank = __popcll(idleMask & ((1ull << leader) - 1ull));   // SEG == 1: leader == lane
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
                    // Any-hit: the occluder this slot's ray found in the previous launch (the same pixel's shadow ray one frame ago) is
                    // the FIRST item, before the root: whichever triangle stops an any-hit ray, the answer is "occluded"
                    // (optix_restir_di_kernels.cu:5-8 sets visibility 0 on any hit), so the order of the tests cannot change it.
                    if (ANY_HIT && a.anyHint) hint = a.anyHint[i] - 1u;
                    rayIdx = i;
                    newRay = true;
                    hasRay = true;
                }
            }
            waveNext += take;
        }
        if (COUNT) cycRefill += __builtin_amdgcn_s_memtime() - cyc0;
        if (__ballot(tr.active || newRay) == 0ull) {
            if (exhausted && (SEG == 1 || __ballot(hasRay) == 0ull)) break;
            continue;
        }
        uint32_t code = kItemNone;
        if (tr.active) {
            code = tr.next_item(stack, a.accel.triItemOffset);
            if (code == kItemNone) write_result();          // traversal finished
        }
        const bool hintFirst = ANY_HIT && newRay && hasNodes && hint < a.accel.numTris;
        if (newRay && hasNodes) code = hintFirst ? (kItemTri | (a.accel.triItemOffset + hint)) : 0u;   // the hinted occluder, else the root node
        if (COUNT) {
            if (newRay) rayItems = 0;
            if (code != kItemNone) ++rayItems;
            const int held = __popcll(__ballot(code != kItemNone));
            ++diagIter; diagLanes += held;
            if (exhausted) { ++diagDrainIter; diagDrainLanes += held; }
        }
        uint4 link = make_uint4(0u, 0u, 0u, 0u);
        if (code != kItemNone && !(code & kItemTri)) link = reinterpret_cast<const uint4*>(a.accel.links)[code];   // in flight with the item fetch
        uint4 q0, q1, q2, q3;
        const unsigned long long cyc1 = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
        fetch_items(code, a.accel, waveBuf, lane, q0, q1, q2, q3);
        if (COUNT) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); cycFetch += __builtin_amdgcn_s_memtime() - cyc1; }
        const unsigned long long cyc2 = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
        if (newRay) {                                           // its origin and direction have arrived with the items
            float lo = rayO.w, hi = rayD.w;
            const bool empty = !hasNodes || !(rayD.w > rayO.w);  // empty interval or empty scene: immediate miss
            if (SEG > 1 && !empty) segment_interval<SEG>(f3(rayO.x, rayO.y, rayO.z), f3(rayD.x, rayD.y, rayD.z), rayO.w, rayD.w, box, seg, lo, hi);
            tr.begin(f3(rayO.x, rayO.y, rayO.z), f3(rayD.x, rayD.y, rayD.z), lo, hi, stack, hasNodes);
            if (!hintFirst) tr.grp.y = 0u;                      // the root (begin's one-child group) is this iteration's item
            if (COUNT) traversed = !empty;
            if (empty || !(hi > lo)) {                          // (an empty segment of a non-empty ray: nothing to find in it)
                tr.active = false;
                code = kItemNone;
                write_result();
                if (COUNT && SEG == 1) --raysDone;               // the counters report rays that were traversed: a queue entry without a
            }                                                   // ray (emit_ray_at_slot, padding slots) is not one
        }
        if (code != kItemNone) {
            if (code & kItemTri) {
                if (!tr.template process_triangle<ANY_HIT, COUNT>((code & 0x7FFFFFFFu) - a.accel.triItemOffset, q0, q1, q2, q3, a.accel.tris, cnt))
                    write_result();                         // any-hit ray found its occluder
            }
            else tr.template process_node<COUNT>(q0, q1, q2, q3, link, stack, cnt);
        }
        if (!ANY_HIT && newRay && tr.active && hint < a.accel.numTris && tr.triMask == 0u) { tr.triBase = hint; tr.triMask = 0x0101u; }
        if (SEG > 1) {
            // a segment with a hit retires the segments it makes pointless: all others (any-hit), the ones behind it (closest-hit).
            // Lanes of a group start their ray in the same iteration (begin() resets hit.tri in all of them), so the bits are the ray's.
            const unsigned long long found = __ballot(hasRay && tr.hit.tri != GFX_INVALID_SLOT);
            const uint32_t gbits = static_cast<uint32_t>(found >> leader) & SegConst<SEG>::groupMask;
            if (ANY_HIT ? gbits != 0u : (gbits & ((1u << seg) - 1u)) != 0u) tr.active = false;
        }
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
    Context::TicketState& ts = ctx.ticketState[&small == &ctx.smallCounters ? 0 : 1];
    char* areas = static_cast<char*>(small.p) + kSmallCountersTicketOffset;
    if (!ts.zeroed || ts.buffer != small.p) {
        GFX_HIP(hipMemsetAsync(areas, 0, 2 * kTicketAreaBytes, stream));
        ts.zeroed = true; ts.next = 0; ts.buffer = small.p;
    }
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
    const bool any = t.mode == GFX_TRACE_ANY;
    a.anyHint = (any && ctx.tune.anyHints) ? t.anyHint : nullptr;
    a.keepAnyHint = ctx.tune.anyHints == 2 ? 1 : 0;
    a.refillThreshold = ctx.tune.traceRefill;
    a.ticketBatch = ctx.tune.traceBatch;
    // Ray segments (see k_trace): a launch with about one ray per lane of the persistent grid or fewer is latency bound -- it lasts as
    // long as its longest ray -- and is cut into segments; a launch that fills the machine several times over is throughput bound
    // and is not (segments repeat the walk through the common ancestors).  Device-counted queues pass their capacity as maxRays.
    int seg = ctx.tune.traceSegments;
    if (seg == 0) {
        const uint64_t lanes = static_cast<uint64_t>(grid) * kTraceBlock;
        const uint64_t rays = t.numRaysPtr ? t.maxRays : t.numRays;
        seg = 1;
        if (rays != 0 && !ctx.countersEnabled) {
            if (rays * 4 <= lanes * static_cast<uint64_t>(ctx.tune.traceSegFill)) seg = 4;
            else if (rays * 2 <= lanes * static_cast<uint64_t>(ctx.tune.traceSegFill)) seg = 2;
        }
    }
    ScopedKernelTimer timer(ctx, stream, any ? "trace_any" : "trace_closest");
#define GFX_TRACE_LAUNCH(ANY, COUNT, SEG) hipLaunchKernelGGL((k_trace<ANY, COUNT, SEG>), dim3(grid), dim3(kTraceBlock), 0, stream, a)
#define GFX_TRACE_SEGS(ANY, COUNT) do { switch (seg) { case 8: GFX_TRACE_LAUNCH(ANY, COUNT, 8); break; case 4: GFX_TRACE_LAUNCH(ANY, COUNT, 4); break; \
                                         case 2: GFX_TRACE_LAUNCH(ANY, COUNT, 2); break; default: GFX_TRACE_LAUNCH(ANY, COUNT, 1); break; } } while (0)
    if (ctx.countersEnabled) { if (any) GFX_TRACE_SEGS(true, true); else GFX_TRACE_SEGS(false, true); }
    else { if (any) GFX_TRACE_SEGS(true, false); else GFX_TRACE_SEGS(false, false); }
#undef GFX_TRACE_SEGS
#undef GFX_TRACE_LAUNCH
    GFX_HIP(hipGetLastError());
}

} // namespace gfx
