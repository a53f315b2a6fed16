// bvh8.hip.h -- software BVH8 traversal for CDNA4 wavefronts.
//
// Replaces optixTrace / RT cores (utils/optix_util.h:557-603) and the scalar CPU traversal
// common/bvh_builder.cpp:1272-1514.  What is kept from the reference: the ray/triangle test
// (common/bvh_builder.cpp:1251-1270, bit-identical arithmetic so hits agree with the CPU oracle),
// exclusive (tmin, tmax) intervals, and the idea of 2-dword "group" stack entries.  What is
// different, by design for wave64 SIMT:
//   * 64-byte nodes with 8-bit child boxes, one byte per plane value (device_types.h) + a 16-byte
//     link record -> five aligned dwordx4 loads per visit, one v_cvt_f32_ubyteN per plane;
//   * children visited in (slot XOR ray-octant) order -- no per-node sorting network;
//   * the stack lives in LDS, one 8-byte column per lane ([depth][lane] => conflict-free
//     ds_read/write_b64), spilling to a per-lane HBM area past kLdsStackDepth entries;
//   * closest-hit ties (equal t) resolve to the triangle that comes first in the flattened triangle list
//     (instance slot ascending, group list order, primitive index), so the result does not depend on
//     traversal order.
#pragma once
#include "device_types.h"
#include "gm_math.hip.h"

namespace gfx {

#ifndef GFX_TRACE_LDS_STACK
#define GFX_TRACE_LDS_STACK 12         // experiment builds trade stack depth for resident waves (tools/microbench/README)
#endif
constexpr int kLdsStackDepth = GFX_TRACE_LDS_STACK;     // entries per lane held in LDS (the kernels that trace inside a per-pixel kernel)
// k_trace holds one entry less and, in the 2 KB that frees, the octant table of process_node: 11 x 8 B x 256 + 16 KB of item buffers + 2 KB
// = 40 KB, the same as a block of k_gbuffer_fused (12 x 8 B x 256 + 16 KB) -- four blocks per CU, and, while the pipelined G-buffer pass
// shares the GPU with a k_trace launch, a slot either kernel leaves fits a block of the other (with 38.9-KB G-buffer blocks the slots
// stayed with the G-buffer pass and k_trace ran a quarter short until it ended: +0.12 ms per frame, profiles/r06_experiments.txt 12)
constexpr int kTraceLdsStackDepth = kLdsStackDepth - 1;
constexpr int kSpillStackDepth = 64;   // entries per lane in the HBM spill area

struct TraceCounters { uint32_t nodes, tris, spills; };

struct RayHit {
    float t;           // current upper bound / hit distance
    float bcB, bcC;
    uint32_t tri;      // triangle record index or GFX_INVALID_SLOT
};

GFX_DEV uint32_t bfe(uint32_t v, uint32_t off, uint32_t bits) { return (v >> off) & ((1u << bits) - 1u); }

// Ray vs triangle, common/bvh_builder.cpp:1251-1270 (same operations, same order).  The edge
// vectors eAB = pB - pA, eCA = pA - pC and the normal n = cross(eCA, eAB) are ray independent and
// are stored in the 64-byte record by the builder (same fp32 operations, so the same bits); the
// distMax comparison is left to the caller (it also handles exact ties).
GFX_DEV bool ray_triangle(f3 org, f3 dir, float tmin, f3 pA, f3 eAB, f3 eCA, f3 n, float& t, float& bcB, float& bcC) {
    const f3 e = (1.0f / dot(n, dir)) * (pA - org);
    const f3 i = cross(dir, e);
    bcB = dot(i, eCA);
    bcC = dot(i, eAB);
    t = dot(n, e);
    return (t > tmin) && (bcB >= 0.0f) && (bcC >= 0.0f) && (bcB + bcC <= 1);
}

// Stack of 8-byte group entries: LDS column first, HBM spill behind it.
struct LaneStack {
    uint2* lds; int ldsStride; uint2* spill; int sp;
    int spillCap;                      // entries of this lane's spill area (kSpillStackDepth in k_trace; sized by the tree's depth in trace_local.hip.h)
    int ldsDepth;                      // entries of the LDS column (kLdsStackDepth; kTraceLdsStackDepth in k_trace)
    GFX_DEV void push(uint2 e, TraceCounters& cnt, bool count) {
        if (sp < ldsDepth) lds[sp * ldsStride] = e;
        else { if (sp - ldsDepth < spillCap) spill[sp - ldsDepth] = e; if (count) ++cnt.spills; }
        ++sp;
    }
    GFX_DEV uint2 pop() {
        --sp;
        return sp < ldsDepth ? lds[sp * ldsStride] : spill[sp - ldsDepth];
    }
};

// max |coordinate| per axis over the tree's planes, from the root node (item 0): |origin_k| + 257 scale_k -- the root's 255 cells and one
// more either side for the outward rounding of a descendant's own, finer grid.  The same for every ray of a launch (scalar loads).
GFX_DEV f3 scene_max_abs(const DevAccel& acc) {
    if (acc.numNodes == 0u) return f3(0.0f);
    const uint4 n0 = *reinterpret_cast<const uint4*>(acc.nodes);
    const f3 origin(bits2f(n0.x), bits2f(n0.y), bits2f(n0.z));
    const f3 scale(bits2f(((n0.w >> 0) & 0xFFu) << 23), bits2f(((n0.w >> 8) & 0xFFu) << 23), bits2f(((n0.w >> 16) & 0xFFu) << 23));
    return f3(fmaf(257.0f, scale.x, fabsf(origin.x)), fmaf(257.0f, scale.y, fabsf(origin.y)), fmaf(257.0f, scale.z, fabsf(origin.z)));
}

constexpr uint32_t kItemNone = 0xFFFFFFFFu;   // nothing to fetch
constexpr uint32_t kItemTri = 0x80000000u;    // item code: bit 31 = triangle record, low bits = index

// Per-lane traversal state: one ray in flight.  Each iteration of the wave loop a lane asks for ONE
// 64-byte item -- the next node of its current group, or the next pending triangle of the node it
// visited last -- the wave fetches all 64 items cooperatively (trace.hip), and the lane then
// processes its item.  A lane never waits inside another lane's per-child or per-triangle loop.
//
// Slab test in the node's quantised frame: plane t = A_k + q * B_k with A_k = (origin_k - org_k) /
// dir_k, B_k = scale_k / dir_k (one fmaf per plane).  The builder guarantees that the fp32 DECODED
// box origin + q * scale contains the child; the fmaf form differs from that decode by a few
// roundings, bounded by 1.5 * 2^-22 * |1/dir_k| * (max |plane coordinate| + |org_k|); slabs are
// widened by 2^-21 of that magnitude, so no box the exact test keeps is ever culled.  The magnitude is taken from the ROOT's box
// (scene_max_abs: every plane of every node lies inside it, give or take a cell of outward rounding): one widening per ray, set up in
// begin(), instead of six instructions per visited node for the node's own, smaller bound -- 2^-21 of the scene's extent in world units,
// far below any box that matters.
struct Traversal {
    f3 org, dir, inv;
    f3 slack;                              // 2^-21 |1/dir_k| (max |plane coordinate_k| over the tree + |org_k|): the slab widening of this ray
    f3 orgInv;                             // org_k / dir_k: A_k = fma(origin_k, 1 / dir_k, -orgInv_k), one instruction per axis and node
    float tmin;
    RayHit hit;
    uint2 grp;
    uint32_t oct;
    uint32_t triBase, triMask;
    bool xNeg, yNeg, zNeg;                 // the ray travels toward -k: near plane = hi, far plane = lo
    bool active;

    GFX_DEV void begin(f3 o, f3 d, float t0, float t1, LaneStack& stack, bool hasNodes, f3 sceneMaxAbs) {
        org = o; dir = d; tmin = t0;
        // |dir_k| below 1e-20 behaves like an axis-parallel ray without producing inf / NaN
        const float dx = fabsf(d.x) < 1e-20f ? copysignf(1e-20f, d.x) : d.x;
        const float dy = fabsf(d.y) < 1e-20f ? copysignf(1e-20f, d.y) : d.y;
        const float dz = fabsf(d.z) < 1e-20f ? copysignf(1e-20f, d.z) : d.z;
        inv = f3(1.0f / dx, 1.0f / dy, 1.0f / dz);
        slack = f3((sceneMaxAbs.x + fabsf(o.x)) * (fabsf(inv.x) * 4.76837158203125e-07f), (sceneMaxAbs.y + fabsf(o.y)) * (fabsf(inv.y) * 4.76837158203125e-07f),
                   (sceneMaxAbs.z + fabsf(o.z)) * (fabsf(inv.z) * 4.76837158203125e-07f));
        orgInv = f3(o.x * inv.x, o.y * inv.y, o.z * inv.z);
        hit.t = t1; hit.bcB = 0; hit.bcC = 0; hit.tri = GFX_INVALID_SLOT;
        // octant mask: bit k set when the ray travels toward -k, so (slot ^ oct) ascending = near to far
        oct = (dx < 0 ? 1u : 0u) | (dy < 0 ? 2u : 0u) | (dz < 0 ? 4u : 0u);
        xNeg = dx < 0; yNeg = dy < 0; zNeg = dz < 0;
        // current group: x = index of the first internal child; y = hit bits | imask << 8.
        // Hit bit p stands for child slot (p ^ oct).  The root is a one-child group: slot 0
        // (bit 0 ^ oct), empty imask -> node index 0.
        grp = make_uint2(0u, 1u << oct);
        triBase = 0; triMask = 0;
        stack.sp = 0;
        active = hasNodes;
    }

    // Which 64-byte item does this lane need next?  kItemNone = the ray has finished.
    // Triangle items are numbered behind the nodes (item = triItemOffset + triangle index): one base
    // address serves both kinds.
    GFX_DEV uint32_t next_item(LaneStack& stack, uint32_t triItemOffset) {
        if (triMask & 0xFFu) {
            // triMask: bits 0-7 hit leaf slots still to test, bits 8-15 all leaf slots of that node;
            // the triangles of a node's leaf children sit behind triBase in slot order
            const uint32_t slot = __builtin_ctz(triMask);
            triMask &= triMask - 1u;
            return kItemTri | (triItemOffset + triBase + __builtin_popcount((triMask >> 8) & ((1u << slot) - 1u)));
        }
        triMask = 0;
        uint32_t hits = grp.y & 0xFFu;
        if (hits == 0) {
            if (stack.sp == 0) { active = false; return kItemNone; }
            grp = stack.pop();
            hits = grp.y & 0xFFu;
        }
        const uint32_t pos = __builtin_ctz(hits);
        grp.y &= ~(1u << pos);
        const uint32_t slot = pos ^ oct;
        const uint32_t imaskG = (grp.y >> 8) & 0xFFu;
        return grp.x + __builtin_popcount(imaskG & ((1u << slot) - 1u));
    }

    // One triangle record (device_types.h Bvh8Tri).  Returns false when an any-hit ray is done.
    template <bool ANY_HIT, bool COUNT>
    GFX_DEV bool process_triangle(uint32_t ti, uint4 q0, uint4 q1, uint4 q2, uint4 q3, const Bvh8Tri* __restrict__ tris, TraceCounters& cnt) {
        if (COUNT) ++cnt.tris;
        const f3 pA(bits2f(q0.x), bits2f(q0.y), bits2f(q0.z));
        const f3 eAB(bits2f(q0.w), bits2f(q1.x), bits2f(q1.y));
        const f3 eCA(bits2f(q1.z), bits2f(q1.w), bits2f(q2.x));
        const f3 n(bits2f(q2.y), bits2f(q2.z), bits2f(q2.w));
        float t, bb, cc;
        if (!ray_triangle(org, dir, tmin, pA, eAB, eCA, n, t, bb, cc)) return true;
        bool take = t < hit.t;
        if (!ANY_HIT && !take && t == hit.t && hit.tri != GFX_INVALID_SLOT) {
            // exact tie: the triangle that comes first in the flattened triangle list wins
            take = q3.w < tris[hit.tri].flatIndex;
        }
        if (take) {
            hit.t = t; hit.bcB = bb; hit.bcC = cc; hit.tri = ti;
            if (ANY_HIT) { active = false; return false; }
        }
        return true;
    }

    // One node (device_types.h Bvh8Node + Bvh8Link): 8 slab tests, leaf hits -> triangle mask, node hits -> group.
    // octPerm (k_trace): 8 x 256 bytes of LDS, octPerm[oct * 256 + m] = the 8-bit mask m with bit s moved to bit s ^ oct -- one ds_read_u8
    // instead of the three conditional swaps (15 instructions); null: the swaps.
    template <bool COUNT>
    GFX_DEV void process_node(uint4 n0, uint4 n1, uint4 n2, uint4 n3, uint4 link, LaneStack& stack, TraceCounters& cnt,
                              const __attribute__((address_space(3))) uint8_t* octPerm = nullptr) {
        if (COUNT) ++cnt.nodes;
        const f3 origin(bits2f(n0.x), bits2f(n0.y), bits2f(n0.z));
        const f3 scale(bits2f(bfe(n0.w, 0, 8) << 23), bits2f(bfe(n0.w, 8, 8) << 23), bits2f(bfe(n0.w, 16, 8) << 23));
        const uint32_t imask = n0.w >> 24;
        const uint32_t valid = link.z;
        // plane bytes: lo.x = n1.xy, lo.y = n1.zw, lo.z = n2.xy, hi.x = n2.zw, hi.y = n3.xy, hi.z = n3.zw
        const uint32_t nx[2] = { xNeg ? n2.z : n1.x, xNeg ? n2.w : n1.y }, fx[2] = { xNeg ? n1.x : n2.z, xNeg ? n1.y : n2.w };
        const uint32_t ny[2] = { yNeg ? n3.x : n1.z, yNeg ? n3.y : n1.w }, fy[2] = { yNeg ? n1.z : n3.x, yNeg ? n1.w : n3.y };
        const uint32_t nz[2] = { zNeg ? n3.z : n2.x, zNeg ? n3.w : n2.y }, fz[2] = { zNeg ? n2.x : n3.z, zNeg ? n2.y : n3.w };

        const f3 B = scale * inv;
        // (origin - org) / dir as origin / dir - org / dir: the second rounding is of magnitude 2^-24 |org_k / dir_k|, inside the bound above
        const f3 A(fmaf(origin.x, inv.x, -orgInv.x), fmaf(origin.y, inv.y, -orgInv.y), fmaf(origin.z, inv.z, -orgInv.z));
        const f3 An = A - slack, Af = A + slack;     // (the ray's widening: begin())

        // branch-free: one bit per SLOT, classified after the loop.  Child s is missed iff tf < tn, i.e. iff the sign bit of
        // tf - tn is set (no NaNs here: every input is finite; tf is never -0: the far planes carry a positive slack and hit.t
        // is a positive bound, so tf - tn = -0 would need tf = tn = -0); the sign bits are shifted into one word with a funnel
        // shift per child -- subtract + v_alignbit instead of compare + select + or.
        uint32_t missSlots = 0;
#pragma unroll
        for (int s = 7; s >= 0; --s) {
            const int w = s >> 2, sh = (s & 3) * 8;
            const float tnx = fmaf(static_cast<float>((nx[w] >> sh) & 0xFFu), B.x, An.x);
            const float tny = fmaf(static_cast<float>((ny[w] >> sh) & 0xFFu), B.y, An.y);
            const float tnz = fmaf(static_cast<float>((nz[w] >> sh) & 0xFFu), B.z, An.z);
            const float tfx = fmaf(static_cast<float>((fx[w] >> sh) & 0xFFu), B.x, Af.x);
            const float tfy = fmaf(static_cast<float>((fy[w] >> sh) & 0xFFu), B.y, Af.y);
            const float tfz = fmaf(static_cast<float>((fz[w] >> sh) & 0xFFu), B.z, Af.z);
            const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
            const float tf = fminf(fminf(tfx, tfy), fminf(tfz, hit.t));
            missSlots = __builtin_amdgcn_alignbit(missSlots, f2bits(tf - tn), 31);   // (missSlots << 1) | sign(tf - tn)
        }
        uint32_t hitSlots = ~missSlots & 0xFFu;                 // children were taken 7 .. 0: bit s belongs to child s
        hitSlots &= valid;
        // internal children in (slot ^ oct) order: XOR-permute the 8 bit positions with three conditional swaps
        uint32_t nodeHits = hitSlots & imask;
        // (xNeg / yNeg / zNeg are the bits of oct as lane masks the near / far selection above already holds in scalar registers)
        if (octPerm) nodeHits = octPerm[(oct << 8) + nodeHits];
        else {
            nodeHits = xNeg ? (((nodeHits & 0x55u) << 1) | ((nodeHits & 0xAAu) >> 1)) : nodeHits;
            nodeHits = yNeg ? (((nodeHits & 0x33u) << 2) | ((nodeHits & 0xCCu) >> 2)) : nodeHits;
            nodeHits = zNeg ? (((nodeHits & 0x0Fu) << 4) | ((nodeHits & 0xF0u) >> 4)) : nodeHits;
        }
        // leaf children: keep the hit slots and the node's leaf-slot set; the rank (= triangle offset) is
        // taken when the triangle is fetched (next_item)
        const uint32_t leafBits = valid & ~imask;
        const uint32_t leafMask = (hitSlots & leafBits) ? ((hitSlots & leafBits) | (leafBits << 8)) : 0u;
        if (leafMask) { triMask = leafMask; triBase = link.y; }
        if (nodeHits) {
            if (grp.y & 0xFFu) stack.push(grp, cnt, COUNT);
            grp = make_uint2(link.x, nodeHits | (imask << 8));
        }
    }
};

} // namespace gfx
