// bvh8.hip.h -- software BVH8 traversal for CDNA4 wavefronts.
//
// Replaces optixTrace / RT cores (utils/optix_util.h:557-603) and the scalar CPU traversal
// common/bvh_builder.cpp:1272-1514.  What is kept from the reference: the ray/triangle test
// (common/bvh_builder.cpp:1251-1270, bit-identical arithmetic so hits agree with the CPU oracle),
// exclusive (tmin, tmax) intervals, and the idea of 2-dword "group" stack entries.  What is
// different, by design for wave64 SIMT:
//   * 64-byte nodes with 6-bit child boxes (device_types.h) -> 4 aligned dwordx4 loads per visit;
//   * children visited in (slot XOR ray-octant) order -- no per-node sorting network;
//   * the stack lives in LDS, one 8-byte column per lane ([depth][lane] => conflict-free
//     ds_read/write_b64), spilling to a per-lane HBM area past kLdsStackDepth entries;
//   * closest-hit ties (equal t) resolve to the lowest (instSlot, geomInstSlot, primIndex), so the
//     result does not depend on traversal order.
#pragma once
#include "device_types.h"
#include "gm_math.hip.h"

namespace gfx {

constexpr int kLdsStackDepth = 12;     // entries per lane held in LDS
constexpr int kSpillStackDepth = 64;   // entries per lane in the HBM spill area

struct TraceCounters { uint32_t nodes, tris, spills; };

struct RayHit {
    float t;           // current upper bound / hit distance
    float bcB, bcC;
    uint32_t tri;      // triangle record index or GFX_INVALID_SLOT
};

GFX_DEV uint32_t bfe(uint32_t v, uint32_t off, uint32_t bits) { return (v >> off) & ((1u << bits) - 1u); }

// Ray vs triangle, common/bvh_builder.cpp:1251-1270 (same operations, same order); the distMax
// comparison is left to the caller (it also handles exact ties).
GFX_DEV bool ray_triangle(f3 org, f3 dir, float tmin, f3 pA, f3 pB, f3 pC, float& t, float& bcB, float& bcC) {
    const f3 eAB = pB - pA;
    const f3 eCA = pA - pC;
    const f3 n = cross(eCA, eAB);
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
    GFX_DEV void push(uint2 e, TraceCounters& cnt, bool count) {
        if (sp < kLdsStackDepth) lds[sp * ldsStride] = e;
        else { if (sp - kLdsStackDepth < kSpillStackDepth) spill[sp - kLdsStackDepth] = e; if (count) ++cnt.spills; }
        ++sp;
    }
    GFX_DEV uint2 pop() {
        --sp;
        return sp < kLdsStackDepth ? lds[sp * ldsStride] : spill[sp - kLdsStackDepth];
    }
};

// Per-lane traversal state: one ray in flight.  step() performs one group pop + node visit
// (including the triangle tests of the node's hit leaf children) and returns false once the ray
// has finished, so a persistent wave can refill finished lanes between steps.
struct Traversal {
    f3 org, dir, inv;
    float tmin;
    RayHit hit;
    uint2 grp;
    uint32_t oct;
    bool active;

    GFX_DEV void begin(f3 o, f3 d, float t0, float t1, LaneStack& stack, bool hasNodes) {
        org = o; dir = d; tmin = t0;
        inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        hit.t = t1; hit.bcB = 0; hit.bcC = 0; hit.tri = GFX_INVALID_SLOT;
        // octant mask: bit k set when the ray travels toward -k, so (slot ^ oct) ascending = near to far
        oct = (d.x < 0 ? 1u : 0u) | (d.y < 0 ? 2u : 0u) | (d.z < 0 ? 4u : 0u);
        // current group: x = index of the first internal child; y = hit bits | imask << 8.
        // Hit bit p stands for child slot (p ^ oct).  The root is a one-child group: slot 0
        // (bit 0 ^ oct), empty imask -> node index 0.
        grp = make_uint2(0u, 1u << oct);
        stack.sp = 0;
        active = hasNodes;
    }

    template <bool ANY_HIT, bool COUNT>
    GFX_DEV bool step(const DevAccel& acc, LaneStack& stack, TraceCounters& cnt) {
        const Bvh8Node* __restrict__ nodes = acc.nodes;
        const Bvh8Tri* __restrict__ tris = acc.tris;
        uint32_t hits = grp.y & 0xFFu;
        if (hits == 0) {
            if (stack.sp == 0) { active = false; return false; }
            grp = stack.pop();
            hits = grp.y & 0xFFu;
        }
        const uint32_t pos = __builtin_ctz(hits);
        grp.y &= ~(1u << pos);
        const uint32_t slot = pos ^ oct;
        const uint32_t imaskG = (grp.y >> 8) & 0xFFu;
        const uint32_t nodeIdx = grp.x + __builtin_popcount(imaskG & ((1u << slot) - 1u));

        const uint4* np = reinterpret_cast<const uint4*>(nodes + nodeIdx);
        const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
        if (COUNT) ++cnt.nodes;
        const f3 origin(bits2f(n0.x), bits2f(n0.y), bits2f(n0.z));
        const f3 scale(bits2f(bfe(n0.w, 0, 8) << 23), bits2f(bfe(n0.w, 8, 8) << 23), bits2f(bfe(n0.w, 16, 8) << 23));
        const uint32_t imask = n0.w >> 24;
        const uint32_t triBase = n1.y;
        const uint32_t cw[8] = { n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, n3.x, n3.y };
        const uint32_t zb[2] = { n3.z, n3.w };

        uint32_t nodeHits = 0;     // hit internal children, bit (slot ^ oct)
        uint32_t triOff = 0;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const uint32_t w = cw[s];
            const uint32_t zc = bfe(zb[s >> 2], (s & 3) * 8, 8);
            const uint32_t count = (w >> 30) | ((zc >> 6) << 2);
            if (count == 0) continue;
            const bool internal = (imask >> s) & 1u;
            // dequantise: origin + q * scale (q * scale is exact, one rounding per coordinate)
            const f3 lo(origin.x + static_cast<float>(bfe(w, 0, 6)) * scale.x,
                        origin.y + static_cast<float>(bfe(w, 6, 6)) * scale.y,
                        origin.z + static_cast<float>(bfe(w, 12, 6)) * scale.z);
            const f3 hi(origin.x + static_cast<float>(bfe(w, 18, 6)) * scale.x,
                        origin.y + static_cast<float>(bfe(w, 24, 6)) * scale.y,
                        origin.z + static_cast<float>(zc & 63u) * scale.z);
            const f3 t0 = (lo - org) * inv, t1 = (hi - org) * inv;
            // fminf/fmaxf drop NaNs (0 * inf when the ray origin lies in a slab plane)
            float tn = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fminf(t0.z, t1.z));
            float tf = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fmaxf(t0.z, t1.z));
            // conservative slabs: widen by a few ulp so rounding never culls a box the exact test keeps
            tn = tn * (tn > 0 ? 0.9999995f : 1.0000005f);
            tf = tf * (tf > 0 ? 1.0000005f : 0.9999995f);
            tn = fmaxf(tn, tmin);
            tf = fminf(tf, hit.t);
            const bool boxHit = tn <= tf;
            if (internal) {
                if (boxHit) nodeHits |= 1u << (s ^ oct);
                continue;
            }
            if (boxHit) {
                for (uint32_t k = 0; k < count; ++k) {
                    const uint32_t ti = triBase + triOff + k;
                    const float4* tp = reinterpret_cast<const float4*>(tris + ti);
                    const float4 a = tp[0], b = tp[1], c = tp[2];
                    if (COUNT) ++cnt.tris;
                    float t, bb, cc;
                    if (!ray_triangle(org, dir, tmin, f3(a.x, a.y, a.z), f3(a.w, b.x, b.y), f3(b.z, b.w, c.x), t, bb, cc))
                        continue;
                    bool take = t < hit.t;
                    if (!ANY_HIT && !take && t == hit.t && hit.tri != GFX_INVALID_SLOT) {
                        // exact tie: lowest (instSlot, geomInstSlot, primIndex) wins
                        const Bvh8Tri* o = tris + hit.tri;
                        const uint32_t ni = __float_as_uint(c.y), ng = __float_as_uint(c.z), npm = __float_as_uint(c.w);
                        take = ni < o->instSlot || (ni == o->instSlot && (ng < o->geomInstSlot ||
                               (ng == o->geomInstSlot && npm < o->primIndex)));
                    }
                    if (take) {
                        hit.t = t; hit.bcB = bb; hit.bcC = cc; hit.tri = ti;
                        if (ANY_HIT) { active = false; return false; }
                    }
                }
            }
            triOff += count;
        }
        if (nodeHits) {
            if (grp.y & 0xFFu) stack.push(grp, cnt, COUNT);
            grp = make_uint2(n1.x, nodeHits | (imask << 8));
        }
        return true;
    }
};

} // namespace gfx
