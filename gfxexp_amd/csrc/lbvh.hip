// lbvh.hip -- HIP LBVH -> BVH8 builder (replaces the CPU top-down SAH builder
// common/bvh_builder.cpp:656-1125 and OptiX's GAS/IAS build, common/common_host.h:1027-1100).
//
// Pipeline (all on the GPU, one stream):
//   1. flatten     every (instance, geomInst, triangle) -> world-space 48-byte record (same
//                  arithmetic as calcTriangleVertices, bvh_builder.cpp:178-209) + scene bounds
//   2. morton      63-bit Morton code of the triangle-box centre; rocPRIM radix sort (key, index)
//   3. karras      binary radix tree over the sorted codes (Karras 2012), index tie-break
//   4. fit         bottom-up AABBs; second arriver at a node continues (agent-scope fences)
//   5. collapse    top-down, level by level: a wide node repeatedly opens its largest-area child
//                  until it has 8 (the reference's "split the child with the maximum surface
//                  area", bvh_builder.cpp:785-888); children with <= maxLeafTris triangles become
//                  leaves; child boxes are quantised to a 6-bit power-of-two grid, conservatively
//                  (the reference grid is 8-bit, common_shared.h:814-851); children are placed in
//                  octant slots by a greedy auction so traversal needs no sorting.
//   6. triangles of a node's leaf children are copied contiguously behind the node's triBase.
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdexcept>
#include "bvh8.hip.h"
#include "internal.h"
#include "shading.hip.h"

namespace gfx {

GFX_DEV uint32_t ordered_from_float(float f) { const uint32_t u = f2bits(f); return u ^ (u < 0x80000000u ? 0x80000000u : 0xFFFFFFFFu); }
GFX_DEV float float_from_ordered(uint32_t o) { const uint32_t u = o ^ (o >= 0x80000000u ? 0x80000000u : 0xFFFFFFFFu); return bits2f(u); }

struct Box { f3 lo, hi; };
GFX_DEV Box tri_box(const BuildTri& t) {
    Box b;
    b.lo = f3(fminf(fminf(t.ax, t.bx), t.cx), fminf(fminf(t.ay, t.by), t.cy), fminf(fminf(t.az, t.bz), t.cz));
    b.hi = f3(fmaxf(fmaxf(t.ax, t.bx), t.cx), fmaxf(fmaxf(t.ay, t.by), t.cy), fmaxf(fmaxf(t.az, t.bz), t.cz));
    return b;
}
GFX_DEV Box box_union(const Box& a, const Box& b) {
    Box r;
    r.lo = f3(fminf(a.lo.x, b.lo.x), fminf(a.lo.y, b.lo.y), fminf(a.lo.z, b.lo.z));
    r.hi = f3(fmaxf(a.hi.x, b.hi.x), fmaxf(a.hi.y, b.hi.y), fmaxf(a.hi.z, b.hi.z));
    return r;
}
GFX_DEV float half_area(const Box& b) { const f3 d = b.hi - b.lo; return d.x * d.y + d.y * d.z + d.z * d.x; }

GFX_DEV BuildTri load_tri(const BuildTri* p) {
    const float4* q = reinterpret_cast<const float4*>(p);
    const float4 a = q[0], b = q[1], c = q[2];
    BuildTri t;
    t.ax = a.x; t.ay = a.y; t.az = a.z; t.bx = a.w; t.by = b.x; t.bz = b.y; t.cx = b.z; t.cy = b.w; t.cz = c.x;
    t.instSlot = f2bits(c.y); t.geomInstSlot = f2bits(c.z); t.primIndex = f2bits(c.w);
    return t;
}
GFX_DEV void store_tri(BuildTri* p, const BuildTri& t) {
    float4* q = reinterpret_cast<float4*>(p);
    q[0] = make_float4(t.ax, t.ay, t.az, t.bx);
    q[1] = make_float4(t.by, t.bz, t.cx, t.cy);
    q[2] = make_float4(t.cz, bits2f(t.instSlot), bits2f(t.geomInstSlot), bits2f(t.primIndex));
}
// Final 64-byte traversal record: pA and the ray-independent terms of testRayVsTriangle
// (common/bvh_builder.cpp:1256-1258), computed with exactly those fp32 operations.
GFX_DEV void store_final_tri(Bvh8Tri* p, const BuildTri& t, uint32_t flatIndex) {
    const f3 pA(t.ax, t.ay, t.az), pB(t.bx, t.by, t.bz), pC(t.cx, t.cy, t.cz);
    const f3 eAB = pB - pA;
    const f3 eCA = pA - pC;
    const f3 n = cross(eCA, eAB);
    float4* q = reinterpret_cast<float4*>(p);
    q[0] = make_float4(pA.x, pA.y, pA.z, eAB.x);
    q[1] = make_float4(eAB.y, eAB.z, eCA.x, eCA.y);
    q[2] = make_float4(eCA.z, n.x, n.y, n.z);
    q[3] = make_float4(bits2f(t.instSlot), bits2f(t.geomInstSlot), bits2f(t.primIndex), bits2f(flatIndex));
}

// ---------------------------------------------------------------- 1. flatten
__global__ void k_flatten(DevScene sc, const SubsetGeom* __restrict__ flat, uint32_t numFlat, uint32_t n,
                          BuildTri* __restrict__ out, uint32_t* __restrict__ flatIndexOut, uint32_t* __restrict__ bounds /* ordered lo xyz, hi xyz */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    f3 lo(INFINITY), hi(-INFINITY);
    if (i < n) {
        // binary search the subtree's geometry list (localBegin ascending)
        uint32_t a = 0, b = numFlat;
        while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (flat[m].localBegin <= i) a = m; else b = m; }
        const SubsetGeom fg = flat[a];
        const uint32_t prim = i - fg.localBegin;
        flatIndexOut[i] = fg.globalBegin + prim;   // position in the whole scene's flattened triangle list (closest-hit tie rule)
        const DevGeomInst g = sc.geomInsts[fg.geomInstSlot];
        const uint32_t* tri = sc.triangles + 3ull * (g.triangleOffset + prim);
        const DevVertex vA = load_vertex(sc.vertices + g.vertexOffset + tri[0]);
        const DevVertex vB = load_vertex(sc.vertices + g.vertexOffset + tri[1]);
        const DevVertex vC = load_vertex(sc.vertices + g.vertexOffset + tri[2]);
        const m34 xfm = load_m34(sc.insts[fg.instSlot].transform);
        const f3 pA = xfm_point(xfm, f3(vA.px, vA.py, vA.pz));
        const f3 pB = xfm_point(xfm, f3(vB.px, vB.py, vB.pz));
        const f3 pC = xfm_point(xfm, f3(vC.px, vC.py, vC.pz));
        BuildTri t;
        t.ax = pA.x; t.ay = pA.y; t.az = pA.z; t.bx = pB.x; t.by = pB.y; t.bz = pB.z; t.cx = pC.x; t.cy = pC.y; t.cz = pC.z;
        t.instSlot = fg.instSlot; t.geomInstSlot = fg.geomInstSlot; t.primIndex = prim;
        store_tri(out + i, t);
        const Box bx = tri_box(t);
        lo = bx.lo; hi = bx.hi;
    }
    // wave reduction, then one atomic per wave and component
    for (int off = 32; off >= 1; off >>= 1) {
        lo.x = fminf(lo.x, __shfl_xor(lo.x, off)); lo.y = fminf(lo.y, __shfl_xor(lo.y, off)); lo.z = fminf(lo.z, __shfl_xor(lo.z, off));
        hi.x = fmaxf(hi.x, __shfl_xor(hi.x, off)); hi.y = fmaxf(hi.y, __shfl_xor(hi.y, off)); hi.z = fmaxf(hi.z, __shfl_xor(hi.z, off));
    }
    if ((threadIdx.x & 63) == 0 && lo.x <= hi.x) {
        atomicMin(bounds + 0, ordered_from_float(lo.x)); atomicMin(bounds + 1, ordered_from_float(lo.y)); atomicMin(bounds + 2, ordered_from_float(lo.z));
        atomicMax(bounds + 3, ordered_from_float(hi.x)); atomicMax(bounds + 4, ordered_from_float(hi.y)); atomicMax(bounds + 5, ordered_from_float(hi.z));
    }
}

// ---------------------------------------------------------------- 2. morton
GFX_DEV uint64_t spread21(uint32_t v) {
    uint64_t x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__global__ void k_morton(const BuildTri* __restrict__ tris, uint32_t n, const uint32_t* __restrict__ bounds,
                         uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 lo(float_from_ordered(bounds[0]), float_from_ordered(bounds[1]), float_from_ordered(bounds[2]));
    const f3 hi(float_from_ordered(bounds[3]), float_from_ordered(bounds[4]), float_from_ordered(bounds[5]));
    const Box b = tri_box(load_tri(tris + i));
    const f3 c = 0.5f * (b.lo + b.hi);
    const f3 e = hi - lo;
    const float sx = e.x > 0 ? (c.x - lo.x) / e.x : 0.0f, sy = e.y > 0 ? (c.y - lo.y) / e.y : 0.0f, sz = e.z > 0 ? (c.z - lo.z) / e.z : 0.0f;
    const uint32_t qx = min(f2u_sat(sx * 2097152.0f), 2097151u), qy = min(f2u_sat(sy * 2097152.0f), 2097151u), qz = min(f2u_sat(sz * 2097152.0f), 2097151u);
    keys[i] = (spread21(qx) << 2) | (spread21(qy) << 1) | spread21(qz);
    vals[i] = i;
}

// ---------------------------------------------------------------- 3. karras
// child reference: >= 0 internal node, < 0 leaf ~ref
GFX_DEV int delta_fn(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t x = keys[i] ^ keys[j];
    if (x == 0) return 64 + __clz(static_cast<uint32_t>(i ^ j));
    return __clzll(x);
}
__global__ void k_karras(const uint64_t* __restrict__ keys, int n, int2* __restrict__ lr, uint32_t* __restrict__ parentInt,
                         uint32_t* __restrict__ parentLeaf, uint2* __restrict__ ranges) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta_fn(keys, n, i, i + 1) - delta_fn(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta_fn(keys, n, i, i - d);
    int lmax = 2;
    while (delta_fn(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta_fn(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta_fn(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta_fn(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int first = min(i, j), last = max(i, j);
    const int left = (first == gamma) ? ~gamma : gamma;
    const int right = (last == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    lr[i] = make_int2(left, right);
    ranges[i] = make_uint2(static_cast<uint32_t>(first), static_cast<uint32_t>(last));
    if (left >= 0) parentInt[left] = i; else parentLeaf[~left] = i;
    if (right >= 0) parentInt[right] = i; else parentLeaf[~right] = i;
    if (i == 0) parentInt[0] = 0xFFFFFFFFu;
}

// ---------------------------------------------------------------- 4. fit + wide-node DP
// Bottom-up AABBs, and in the same pass the dynamic program of Ylitie, Karras & Laine 2017 (sec. 3)
// that decides how the binary tree is cut into 8-wide nodes: cost(n, i) = cheapest SAH cost of
// representing the subtree of n with at most i roots (i = 1..7),
//   cost(n, 1) = min(leaf: A(n) * cPrim * tris(n) if tris(n) <= maxLeafTris,
//                    node: A(n) * cNode + distribute(n, 8))
//   cost(n, i) = min(distribute(n, i), cost(n, i - 1)),   distribute(n, j) = min_k cost(l, k) + cost(r, j - k).
// dec[n]: nibble i (1..7) = k chosen for distribute(n, i) or 0 = "same as i - 1"; nibble 0 = k of
// distribute(n, 8); bit 31 = n is cheaper as a leaf.  The collapse follows these decisions top-down.
constexpr float kCostNode = 1.0f;
constexpr float kCostPrimDefault = 1.0f;     // a triangle test occupies a lane for one wave iteration, like a node visit

GFX_DEV void dp_leaf_table(float area, float kCostPrim, float c[8]) {
#pragma unroll
    for (int i = 1; i <= 7; ++i) c[i] = area * kCostPrim;
}
// combine the tables of the two children of a binary node
GFX_DEV uint32_t dp_combine(const float l[8], const float r[8], float area, uint32_t numTris, uint32_t maxLeafTris, float kCostPrim, float out[8]) {
    uint32_t dec = 0;
    float dist[9];
#pragma unroll
    for (int j = 2; j <= 8; ++j) {
        float best = INFINITY; int bk = 1;
#pragma unroll
        for (int k = 1; k < j; ++k) {
            if (k > 7 || j - k > 7) continue;
            const float v = l[k] + r[j - k];
            if (v < best) { best = v; bk = k; }
        }
        dist[j] = best;
        if (j == 8) dec |= static_cast<uint32_t>(bk);
        else dec |= static_cast<uint32_t>(bk) << (4 * j);
    }
    const float asNode = area * kCostNode + dist[8];
    const float asLeaf = numTris <= maxLeafTris ? area * kCostPrim * static_cast<float>(numTris) : INFINITY;
    out[1] = fminf(asLeaf, asNode);
    if (asLeaf <= asNode) dec |= 0x80000000u;
#pragma unroll
    for (int i = 2; i <= 7; ++i) {
        if (out[i - 1] <= dist[i]) { out[i] = out[i - 1]; dec &= ~(0xFu << (4 * i)); }
        else out[i] = dist[i];
    }
    return dec;
}

__global__ void k_fit(const BuildTri* __restrict__ tris, const uint32_t* __restrict__ sortedIdx, int n,
                      const int2* __restrict__ lr, const uint32_t* __restrict__ parentInt, const uint32_t* __restrict__ parentLeaf,
                      uint32_t* __restrict__ flags, float* nodeBoxes /* 8 floats per internal node */,
                      const uint2* __restrict__ ranges, uint32_t maxLeafTris, float kCostPrim, float* costs /* 8 floats per internal node */, uint32_t* __restrict__ dec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t node = parentLeaf[i];
    int cameFrom = ~i;
    Box mine = tri_box(load_tri(tris + sortedIdx[i]));
    float myCost[8];
    dp_leaf_table(half_area(mine), kCostPrim, myCost);
    while (true) {
        // release my subtree's box and cost table (internal only; leaf data is recomputed from the triangle)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t old = atomicAdd(flags + node, 1u);
        if (old == 0) return;                       // first arriver: the sibling finishes this node
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int2 c = lr[node];
        const bool fromLeft = c.x == cameFrom;
        const int sib = fromLeft ? c.y : c.x;
        Box sb;
        float sibCost[8];
        if (sib < 0) {
            sb = tri_box(load_tri(tris + sortedIdx[~sib]));
            dp_leaf_table(half_area(sb), kCostPrim, sibCost);
        }
        else {
            const float* p = nodeBoxes + 8ull * sib;
            sb.lo = f3(__hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            sb.hi = f3(__hip_atomic_load(p + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(p + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(p + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const float* q = costs + 8ull * sib;
#pragma unroll
            for (int k = 1; k <= 7; ++k) sibCost[k] = __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        mine = box_union(mine, sb);
        const uint2 rg = ranges[node];
        float merged[8];
        const uint32_t d = fromLeft ? dp_combine(myCost, sibCost, half_area(mine), rg.y - rg.x + 1, maxLeafTris, kCostPrim, merged)
                                    : dp_combine(sibCost, myCost, half_area(mine), rg.y - rg.x + 1, maxLeafTris, kCostPrim, merged);
#pragma unroll
        for (int k = 1; k <= 7; ++k) myCost[k] = merged[k];
        dec[node] = d;
        float* q = nodeBoxes + 8ull * node;
        __hip_atomic_store(q + 0, mine.lo.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, mine.lo.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 2, mine.lo.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 4, mine.hi.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 5, mine.hi.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 6, mine.hi.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float* cq = costs + 8ull * node;
#pragma unroll
        for (int k = 1; k <= 7; ++k) __hip_atomic_store(cq + k, myCost[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (node == 0) return;
        cameFrom = static_cast<int>(node);
        node = parentInt[node];
    }
}

// ---------------------------------------------------------------- 5./6. collapse
struct WideChild {
    int ref;            // binary child reference (>= 0 internal, < 0 leaf)
    uint32_t first, last;
    Box box;
};

GFX_DEV void load_child(int ref, const BuildTri* __restrict__ tris, const uint32_t* __restrict__ sortedIdx,
                        const float* __restrict__ nodeBoxes, const uint2* __restrict__ ranges, WideChild& c) {
    c.ref = ref;
    if (ref < 0) {
        c.first = c.last = static_cast<uint32_t>(~ref);
        c.box = tri_box(load_tri(tris + sortedIdx[~ref]));
    }
    else {
        const uint2 r = ranges[ref];
        c.first = r.x; c.last = r.y;
        const float* p = nodeBoxes + 8ull * ref;
        c.box.lo = f3(p[0], p[1], p[2]);
        c.box.hi = f3(p[4], p[5], p[6]);
    }
}

// counters: [0] wide-node allocator, [1] triangle-record allocator, [2 + l] work items of level l
__global__ void k_collapse_level(uint32_t level, uint32_t maxLeafTris,
                                 const uint2* __restrict__ queueIn, uint2* __restrict__ queueOut, uint32_t* __restrict__ counters,
                                 const int2* __restrict__ lr, const uint2* __restrict__ ranges, const float* __restrict__ nodeBoxes,
                                 const uint32_t* __restrict__ dec, int useDp,
                                 const BuildTri* __restrict__ trisIn, const uint32_t* __restrict__ sortedIdx, const uint32_t* __restrict__ flatIndex,
                                 Bvh8Node* __restrict__ nodesOut, Bvh8Link* __restrict__ linksOut, Bvh8Tri* __restrict__ trisOut) {
    const uint32_t numItems = counters[2 + level];
    for (uint32_t item = blockIdx.x * blockDim.x + threadIdx.x; item < numItems; item += gridDim.x * blockDim.x) {
        const uint2 work = queueIn[item];       // x = binary node, y = wide node index
        WideChild ch[8];
        bool asLeaf[8];
        int n = 0;
        if (useDp) {
            // expand distribute(work.x, 8) along the DP decisions: (ref, i) = "ref gets at most i roots"
            int stRef[8]; int stI[8]; int sp = 0;
            {
                const int2 c = lr[work.x];
                const int k = static_cast<int>(dec[work.x] & 0xFu);
                stRef[sp] = c.y; stI[sp] = 8 - k; ++sp;
                stRef[sp] = c.x; stI[sp] = k; ++sp;
            }
            while (sp > 0) {
                --sp;
                const int ref = stRef[sp]; int i = stI[sp];
                if (ref < 0) { load_child(ref, trisIn, sortedIdx, nodeBoxes, ranges, ch[n]); asLeaf[n] = true; ++n; continue; }
                const uint32_t d = dec[ref];
                int k = 0;
                while (i > 1 && (k = static_cast<int>((d >> (4 * i)) & 0xFu)) == 0) --i;   // "same as i - 1"
                if (i == 1) { load_child(ref, trisIn, sortedIdx, nodeBoxes, ranges, ch[n]); asLeaf[n] = (d >> 31) != 0; ++n; continue; }
                const int2 c = lr[ref];
                stRef[sp] = c.y; stI[sp] = i - k; ++sp;
                stRef[sp] = c.x; stI[sp] = k; ++sp;
            }
        }
        else {
            n = 2;
            const int2 c = lr[work.x];
            load_child(c.x, trisIn, sortedIdx, nodeBoxes, ranges, ch[0]);
            load_child(c.y, trisIn, sortedIdx, nodeBoxes, ranges, ch[1]);
            while (n < 8) {
                int bestIdx = -1; float bestArea = -INFINITY;
                for (int k = 0; k < n; ++k) {
                    if (ch[k].ref < 0) continue;
                    const float a = half_area(ch[k].box);
                    if (a > bestArea) { bestArea = a; bestIdx = k; }
                }
                if (bestIdx < 0) break;
                const int2 c2 = lr[ch[bestIdx].ref];
                load_child(c2.x, trisIn, sortedIdx, nodeBoxes, ranges, ch[bestIdx]);
                load_child(c2.y, trisIn, sortedIdx, nodeBoxes, ranges, ch[n]);
                ++n;
            }
            for (int k = 0; k < n; ++k) asLeaf[k] = ch[k].last == ch[k].first;
        }
        // node frame
        Box nb = ch[0].box;
        for (int k = 1; k < n; ++k) nb = box_union(nb, ch[k].box);
        const f3 origin = nb.lo;
        uint32_t ex[3];
        float scale[3];
        {
            const float ext[3] = { nb.hi.x - nb.lo.x, nb.hi.y - nb.lo.y, nb.hi.z - nb.lo.z };
            const float org[3] = { origin.x, origin.y, origin.z };
            const float top[3] = { nb.hi.x, nb.hi.y, nb.hi.z };
            for (int a = 0; a < 3; ++a) {
                const uint32_t us = f2bits(ext[a] / 255.0f);
                uint32_t e = (us >> 23) + ((us & 0x7FFFFFu) ? 1u : 0u);
                // the decoded far end origin + 255 * scale must not round below the true maximum
                while (e < 254u && org[a] + 255.0f * bits2f(e << 23) < top[a]) ++e;
                ex[a] = e; scale[a] = bits2f(e << 23);
            }
        }
        // octant slots by greedy auction: maximise sum of dot(child centre - node centre, slot sign)
        const f3 nc = 0.5f * (nb.lo + nb.hi);
        int slotOf[8];
        {
            float cost[8][8];
            for (int k = 0; k < n; ++k) {
                const f3 cc = 0.5f * (ch[k].box.lo + ch[k].box.hi) - nc;
                for (int s = 0; s < 8; ++s)
                    cost[k][s] = ((s & 1) ? cc.x : -cc.x) + ((s & 2) ? cc.y : -cc.y) + ((s & 4) ? cc.z : -cc.z);
            }
            uint32_t freeSlots = 0xFFu, freeKids = (1u << n) - 1u;
            for (int it = 0; it < n; ++it) {
                float best = -INFINITY; int bk = 0, bs = 0;
                for (int k = 0; k < n; ++k) {
                    if (!((freeKids >> k) & 1u)) continue;
                    for (int s = 0; s < 8; ++s) {
                        if (!((freeSlots >> s) & 1u)) continue;
                        if (cost[k][s] > best) { best = cost[k][s]; bk = k; bs = s; }
                    }
                }
                slotOf[bk] = bs;
                freeKids &= ~(1u << bk); freeSlots &= ~(1u << bs);
            }
        }
        int kidAt[8];
        for (int s = 0; s < 8; ++s) kidAt[s] = -1;
        for (int k = 0; k < n; ++k) kidAt[slotOf[k]] = k;

        uint32_t imask = 0, numInternal = 0, numLeafTris = 0;
        for (int s = 0; s < 8; ++s) {
            const int k = kidAt[s];
            if (k < 0) continue;
            if (!asLeaf[k]) { imask |= 1u << s; ++numInternal; }
            else numLeafTris += 1;
        }
        const uint32_t childBase = numInternal ? atomicAdd(counters + 0, numInternal) : 0xFFFFFFFFu;
        const uint32_t triBase = numLeafTris ? atomicAdd(counters + 1, numLeafTris) : 0xFFFFFFFFu;
        const uint32_t qBase = numInternal ? atomicAdd(counters + 2 + level + 1, numInternal) : 0u;

        Bvh8Node node;
        node.w[0] = f2bits(origin.x); node.w[1] = f2bits(origin.y); node.w[2] = f2bits(origin.z);
        node.w[3] = ex[0] | (ex[1] << 8) | (ex[2] << 16) | (imask << 24);
        // empty slots: inverted box (lo 255 / hi 0); the valid mask of the link record is what traversal trusts
        for (int k = 4; k < 10; ++k) node.w[k] = 0xFFFFFFFFu;
        for (int k = 10; k < 16; ++k) node.w[k] = 0u;
        uint32_t internalRank = 0, triOff = 0, validMask = 0;
        const float org[3] = { origin.x, origin.y, origin.z };
        for (int s = 0; s < 8; ++s) {
            const int k = kidAt[s];
            if (k < 0) continue;
            validMask |= 1u << s;
            const float lo[3] = { ch[k].box.lo.x, ch[k].box.lo.y, ch[k].box.lo.z };
            const float hi[3] = { ch[k].box.hi.x, ch[k].box.hi.y, ch[k].box.hi.z };
            for (int a = 0; a < 3; ++a) {
                uint32_t l = 0, h = 1;
                if (scale[a] > 0.0f) {
                    l = min(f2u_sat((lo[a] - org[a]) / scale[a]), 254u);
                    h = min(f2u_sat((hi[a] - org[a]) / scale[a]) + 1u, 255u);
                }
                // make the DECODED box (origin + q * scale, as traversal computes it) contain the child
                while (l > 0 && org[a] + static_cast<float>(l) * scale[a] > lo[a]) --l;
                while (h < 255 && org[a] + static_cast<float>(h) * scale[a] < hi[a]) ++h;
                const uint32_t shift = (s & 3) * 8;
                uint32_t& wl = node.w[4 + 2 * a + (s >> 2)];
                uint32_t& wh = node.w[10 + 2 * a + (s >> 2)];
                wl = (wl & ~(0xFFu << shift)) | (l << shift);
                wh = (wh & ~(0xFFu << shift)) | (h << shift);
            }
            const bool internal = (imask >> s) & 1u;
            if (internal) {
                queueOut[qBase + internalRank] = make_uint2(static_cast<uint32_t>(ch[k].ref), childBase + internalRank);
                ++internalRank;
            }
            else {
                store_final_tri(trisOut + triBase + triOff, load_tri(trisIn + sortedIdx[ch[k].first]), flatIndex[sortedIdx[ch[k].first]]);
                ++triOff;
            }
        }
        reinterpret_cast<uint4*>(linksOut)[work.y] = make_uint4(childBase, triBase, validMask, 0u);
        uint4* dst = reinterpret_cast<uint4*>(nodesOut + work.y);
        dst[0] = make_uint4(node.w[0], node.w[1], node.w[2], node.w[3]);
        dst[1] = make_uint4(node.w[4], node.w[5], node.w[6], node.w[7]);
        dst[2] = make_uint4(node.w[8], node.w[9], node.w[10], node.w[11]);
        dst[3] = make_uint4(node.w[12], node.w[13], node.w[14], node.w[15]);
    }
}

// Node frame (origin + power-of-two scale exponents) of a box, and the 8-bit quantisation of a child box inside
// it -- the same rules as k_collapse_level: the DECODED box must contain the child.
GFX_DEV void node_frame(const Box& nb, uint32_t ex[3], float scale[3]) {
    const float ext[3] = { nb.hi.x - nb.lo.x, nb.hi.y - nb.lo.y, nb.hi.z - nb.lo.z };
    const float org[3] = { nb.lo.x, nb.lo.y, nb.lo.z };
    const float top[3] = { nb.hi.x, nb.hi.y, nb.hi.z };
    for (int a = 0; a < 3; ++a) {
        const uint32_t us = f2bits(ext[a] / 255.0f);
        uint32_t e = (us >> 23) + ((us & 0x7FFFFFu) ? 1u : 0u);
        while (e < 254u && org[a] + 255.0f * bits2f(e << 23) < top[a]) ++e;
        ex[a] = e; scale[a] = bits2f(e << 23);
    }
}
GFX_DEV void put_child_box(uint32_t w[16], int slot, const Box& nb, const float scale[3], const Box& child) {
    const float org[3] = { nb.lo.x, nb.lo.y, nb.lo.z };
    const float lo[3] = { child.lo.x, child.lo.y, child.lo.z };
    const float hi[3] = { child.hi.x, child.hi.y, child.hi.z };
    for (int a = 0; a < 3; ++a) {
        uint32_t l = 0, h = 1;
        if (scale[a] > 0.0f) {
            l = min(f2u_sat((lo[a] - org[a]) / scale[a]), 254u);
            h = min(f2u_sat((hi[a] - org[a]) / scale[a]) + 1u, 255u);
        }
        while (l > 0 && org[a] + static_cast<float>(l) * scale[a] > lo[a]) --l;
        while (h < 255 && org[a] + static_cast<float>(h) * scale[a] < hi[a]) ++h;
        const uint32_t shift = (slot & 3) * 8;
        uint32_t& wl = w[4 + 2 * a + (slot >> 2)];
        uint32_t& wh = w[10 + 2 * a + (slot >> 2)];
        wl = (wl & ~(0xFFu << shift)) | (l << shift);
        wh = (wh & ~(0xFFu << shift)) | (h << shift);
    }
}
GFX_DEV void store_node(Bvh8Node* nodes, Bvh8Link* links, uint32_t index, const Box& nb, const uint32_t ex[3], uint32_t imask,
                        const uint32_t w[16], uint32_t childBase, uint32_t triBase, uint32_t validMask) {
    uint4* dst = reinterpret_cast<uint4*>(nodes + index);
    dst[0] = make_uint4(f2bits(nb.lo.x), f2bits(nb.lo.y), f2bits(nb.lo.z), ex[0] | (ex[1] << 8) | (ex[2] << 16) | (imask << 24));
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    dst[2] = make_uint4(w[8], w[9], w[10], w[11]);
    dst[3] = make_uint4(w[12], w[13], w[14], w[15]);
    reinterpret_cast<uint4*>(links)[index] = make_uint4(childBase, triBase, validMask, 0u);
}

// Subtree of at most eight triangles (a lone scene triangle; the usual animated set: a rectangle light or two):
// one wide node whose children are the triangles, written by one thread.  rootBox: {lo, hi} of the subtree.
__global__ void k_small_subtree(const BuildTri* __restrict__ trisIn, const uint32_t* __restrict__ flatIndex, uint32_t n,
                                uint32_t nodeIndex, uint32_t triBase,
                                Bvh8Node* __restrict__ nodesOut, Bvh8Link* __restrict__ linksOut, Bvh8Tri* __restrict__ trisOut,
                                float* __restrict__ rootBox) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Box boxes[8];
    Box nb;
    for (uint32_t k = 0; k < n; ++k) {
        boxes[k] = tri_box(load_tri(trisIn + k));
        nb = k ? box_union(nb, boxes[k]) : boxes[k];
    }
    uint32_t ex[3]; float scale[3];
    node_frame(nb, ex, scale);
    uint32_t w[16];
    for (int k = 4; k < 10; ++k) w[k] = 0xFFFFFFFFu;      // empty slots: inverted box
    for (int k = 10; k < 16; ++k) w[k] = 0u;
    for (uint32_t k = 0; k < n; ++k) {
        put_child_box(w, static_cast<int>(k), nb, scale, boxes[k]);
        store_final_tri(trisOut + triBase + k, load_tri(trisIn + k), flatIndex[k]);
    }
    store_node(nodesOut, linksOut, nodeIndex, nb, ex, 0u, w, 0xFFFFFFFFu, triBase, (1u << n) - 1u);
    if (rootBox) { rootBox[0] = nb.lo.x; rootBox[1] = nb.lo.y; rootBox[2] = nb.lo.z; rootBox[3] = nb.hi.x; rootBox[4] = nb.hi.y; rootBox[5] = nb.hi.z; }
}

// Root of a split tree: node 0 with two internal children, the static subtree's root (node 1, slot 0) and the
// animated subtree's root (node 2, slot 1).  rootBoxes: 2 x {lo, hi}.
__global__ void k_super_root(const float* __restrict__ rootBoxes, Bvh8Node* __restrict__ nodesOut, Bvh8Link* __restrict__ linksOut) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Box b[2];
    for (int k = 0; k < 2; ++k) {
        b[k].lo = f3(rootBoxes[8 * k + 0], rootBoxes[8 * k + 1], rootBoxes[8 * k + 2]);
        b[k].hi = f3(rootBoxes[8 * k + 3], rootBoxes[8 * k + 4], rootBoxes[8 * k + 5]);
    }
    const Box nb = box_union(b[0], b[1]);
    uint32_t ex[3]; float scale[3];
    node_frame(nb, ex, scale);
    uint32_t w[16];
    for (int k = 4; k < 10; ++k) w[k] = 0xFFFFFFFFu;
    for (int k = 10; k < 16; ++k) w[k] = 0u;
    put_child_box(w, 0, nb, scale, b[0]);
    put_child_box(w, 1, nb, scale, b[1]);
    store_node(nodesOut, linksOut, 0u, nb, ex, 0x3u, w, 1u, 0xFFFFFFFFu, 0x3u);
}
// the binary root's box of a general build -> rootBoxes slot
__global__ void k_copy_root_box(const float* __restrict__ nodeBoxes, float* __restrict__ rootBox) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    rootBox[0] = nodeBoxes[0]; rootBox[1] = nodeBoxes[1]; rootBox[2] = nodeBoxes[2];
    rootBox[3] = nodeBoxes[4]; rootBox[4] = nodeBoxes[5]; rootBox[5] = nodeBoxes[6];
}

__global__ void k_tri_ids(const Bvh8Tri* __restrict__ tris, uint32_t first, uint32_t n, gfx_tri_ids* __restrict__ ids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Bvh8Tri* t = tris + first + i;
    gfx_tri_ids o; o.instSlot = t->instSlot; o.geomInstSlot = t->geomInstSlot; o.primIndex = t->primIndex;
    ids[first + i] = o;
}

constexpr uint32_t kMaxCollapseLevels = 96;
constexpr uint32_t kSmallSubtreeTris = 8;

// One subtree over the triangles of ctx.hSubset[subset]: its root wide node at `rootWide`, further nodes
// allocated from `nodeAllocStart`, triangle records from `triAllocStart`.  Returns {next free node, depth}.
struct SubtreeResult { uint32_t nodeEnd; uint32_t depth; };
static SubtreeResult build_subtree(Context& ctx, hipStream_t stream, Accel& out, int subset, uint32_t rootWide,
                                   uint32_t nodeAllocStart, uint32_t triAllocStart, float* rootBoxOut) {
    const uint32_t n = ctx.subsetTris[subset];
    const uint32_t numFlat = static_cast<uint32_t>(ctx.hSubset[subset].size());
    const dim3 blk(256), grd((n + 255) / 256);
    ctx.bTris.reserve(sizeof(BuildTri) * static_cast<size_t>(n));
    ctx.bFlatIdx.reserve(4ull * n);
    ctx.bCounters.reserve(4 * (2 + kMaxCollapseLevels + 2) + 64);
    uint32_t* counters = ctx.bCounters.as<uint32_t>();
    uint32_t* bounds = counters + 2 + kMaxCollapseLevels + 2;
    if (n <= kSmallSubtreeTris) {   // no sort, no allocator, no host round trip (the scene-box atomics of k_flatten go unused)
        hipLaunchKernelGGL(k_flatten, grd, blk, 0, stream, ctx.devScene(), ctx.dSubset[subset].as<SubsetGeom>(), numFlat, n,
                           ctx.bTris.as<BuildTri>(), ctx.bFlatIdx.as<uint32_t>(), bounds);
        hipLaunchKernelGGL(k_small_subtree, dim3(1), dim3(64), 0, stream, ctx.bTris.as<BuildTri>(), ctx.bFlatIdx.as<uint32_t>(), n,
                           rootWide, triAllocStart, out.nodes.as<Bvh8Node>(), out.links.as<Bvh8Link>(), out.trisPtr(), rootBoxOut);
        GFX_HIP(hipGetLastError());
        return { nodeAllocStart, 1u };
    }
    {
        std::vector<uint32_t> init(2 + kMaxCollapseLevels + 2 + 6, 0u);
        init[0] = nodeAllocStart;        // next free wide node
        init[1] = triAllocStart;         // next free triangle record
        init[2] = 1;                     // one work item at level 0
        for (int k = 0; k < 3; ++k) { init[2 + kMaxCollapseLevels + 2 + k] = 0xFFFFFFFFu; init[2 + kMaxCollapseLevels + 2 + 3 + k] = 0u; }
        GFX_HIP(hipMemcpyAsync(counters, init.data(), sizeof(uint32_t) * init.size(), hipMemcpyHostToDevice, stream));
        GFX_HIP(hipStreamSynchronize(stream));   // init goes out of scope
    }
    hipLaunchKernelGGL(k_flatten, grd, blk, 0, stream, ctx.devScene(), ctx.dSubset[subset].as<SubsetGeom>(), numFlat, n,
                       ctx.bTris.as<BuildTri>(), ctx.bFlatIdx.as<uint32_t>(), bounds);
    ctx.bKeys.reserve(8ull * n); ctx.bKeysAlt.reserve(8ull * n);
    ctx.bVals.reserve(4ull * n); ctx.bValsAlt.reserve(4ull * n);
    ctx.bNodesLR.reserve(8ull * n); ctx.bParents.reserve(8ull * n + 16); ctx.bFlags.reserve(4ull * n);
    ctx.bNodeBoxes.reserve(32ull * n); ctx.bRanges.reserve(8ull * n);
    ctx.bQueueA.reserve(8ull * n + 16); ctx.bQueueB.reserve(8ull * n + 16);
    hipLaunchKernelGGL(k_morton, grd, blk, 0, stream, ctx.bTris.as<BuildTri>(), n, bounds, ctx.bKeys.as<uint64_t>(), ctx.bVals.as<uint32_t>());
    size_t tempBytes = 0;
    GFX_HIP(rocprim::radix_sort_pairs(nullptr, tempBytes, ctx.bKeys.as<uint64_t>(), ctx.bKeysAlt.as<uint64_t>(),
                                      ctx.bVals.as<uint32_t>(), ctx.bValsAlt.as<uint32_t>(), n, 0, 63, stream));
    ctx.bSortTemp.reserve(std::max<size_t>(tempBytes, 16));
    GFX_HIP(rocprim::radix_sort_pairs(ctx.bSortTemp.p, tempBytes, ctx.bKeys.as<uint64_t>(), ctx.bKeysAlt.as<uint64_t>(),
                                      ctx.bVals.as<uint32_t>(), ctx.bValsAlt.as<uint32_t>(), n, 0, 63, stream));
    const uint64_t* keys = ctx.bKeysAlt.as<uint64_t>();
    const uint32_t* sortedIdx = ctx.bValsAlt.as<uint32_t>();
    uint32_t* parentInt = ctx.bParents.as<uint32_t>();
    uint32_t* parentLeaf = parentInt + n;
    hipLaunchKernelGGL(k_karras, grd, blk, 0, stream, keys, static_cast<int>(n), ctx.bNodesLR.as<int2>(), parentInt, parentLeaf,
                       ctx.bRanges.as<uint2>());
    GFX_HIP(hipMemsetAsync(ctx.bFlags.p, 0, 4ull * n, stream));
    ctx.bCosts.reserve(32ull * n); ctx.bDec.reserve(4ull * n);
    static float costPrim = -1.0f;
    if (costPrim < 0) { const char* e = getenv("GFX_BVH_CPRIM"); costPrim = e ? static_cast<float>(atof(e)) : kCostPrimDefault; if (!(costPrim > 0)) costPrim = kCostPrimDefault; }
    hipLaunchKernelGGL(k_fit, grd, blk, 0, stream, ctx.bTris.as<BuildTri>(), sortedIdx, static_cast<int>(n), ctx.bNodesLR.as<int2>(),
                       parentInt, parentLeaf, ctx.bFlags.as<uint32_t>(), ctx.bNodeBoxes.as<float>(),
                       ctx.bRanges.as<uint2>(), 1u /* one triangle per leaf slot */, costPrim, ctx.bCosts.as<float>(), ctx.bDec.as<uint32_t>());
    if (rootBoxOut) hipLaunchKernelGGL(k_copy_root_box, dim3(1), dim3(64), 0, stream, ctx.bNodeBoxes.as<float>(), rootBoxOut);
    static int useDp = -1;   // GFX_BVH_COLLAPSE=greedy: open the largest-area child until 8 (the reference's rule)
    if (useDp < 0) { const char* e = getenv("GFX_BVH_COLLAPSE"); useDp = (e && std::strcmp(e, "greedy") == 0) ? 0 : 1; }
    // level 0 work item: binary root 0 -> wide node rootWide
    const uint2 rootItem = make_uint2(0u, rootWide);
    GFX_HIP(hipMemcpyAsync(ctx.bQueueA.p, &rootItem, sizeof(rootItem), hipMemcpyHostToDevice, stream));
    const uint32_t gridC = std::min<uint32_t>((n + 255) / 256, 2048u);
    for (uint32_t level = 0; level < kMaxCollapseLevels; ++level) {
        uint2* qin = (level & 1) ? ctx.bQueueB.as<uint2>() : ctx.bQueueA.as<uint2>();
        uint2* qout = (level & 1) ? ctx.bQueueA.as<uint2>() : ctx.bQueueB.as<uint2>();
        hipLaunchKernelGGL(k_collapse_level, dim3(gridC), dim3(256), 0, stream, level, ctx.maxLeafTris, qin, qout, counters,
                           ctx.bNodesLR.as<int2>(), ctx.bRanges.as<uint2>(), ctx.bNodeBoxes.as<float>(),
                           ctx.bDec.as<uint32_t>(), useDp,
                           ctx.bTris.as<BuildTri>(), sortedIdx, ctx.bFlatIdx.as<uint32_t>(), out.nodes.as<Bvh8Node>(), out.links.as<Bvh8Link>(), out.trisPtr());
    }
    GFX_HIP(hipGetLastError());
    std::vector<uint32_t> h(2 + kMaxCollapseLevels + 2);
    GFX_HIP(hipMemcpyAsync(h.data(), counters, sizeof(uint32_t) * h.size(), hipMemcpyDeviceToHost, stream));
    GFX_HIP(hipStreamSynchronize(stream));
    uint32_t depth = 0;
    for (uint32_t l = 0; l < kMaxCollapseLevels; ++l) if (h[2 + l]) depth = l + 1;
    if (h[2 + kMaxCollapseLevels] != 0) throw HipError("lbvh_build: tree deeper than kMaxCollapseLevels");
    if (h[1] - triAllocStart != n) throw HipError("lbvh_build: triangle count mismatch after collapse");
    return { h[0], depth };
}

// Tree layouts.  No animated instances (or only animated ones): one tree, root = node 0.  Both kinds: node 0 is a
// two-child root over the static subtree (root node 1, nodes from 3, triangle records [0, nStatic)) and the animated
// subtree (root node 2, nodes behind the static ones, triangle records [nStatic, n)); a transform update of the
// animated instances then rebuilds only the second subtree and node 0 (lbvh_update_dynamic).
void lbvh_build(Context& ctx, hipStream_t stream, Accel& out) {
    scene_upload(ctx, stream);
    transforms_upload(ctx, stream);
    const uint32_t n = ctx.totalTriangles;
    out.numInputTris = n;
    out.numNodes = 0; out.numTris = 0; out.maxDepth = 0; out.split = false;
    if (n == 0) return;
    // the traversal kernel addresses items (node slots + triangle records, 64 B each) with 32-bit byte offsets
    if (n >= (1u << 25) - 4u) throw std::runtime_error("gfx: acceleration structure limited to 2^25 triangles");
    const uint32_t nodeSlots = n + 4;
    out.nodes.reserve(sizeof(Bvh8Node) * (static_cast<size_t>(nodeSlots) + n));   // node slots + n triangle records
    out.triItemOffset = nodeSlots;
    out.links.reserve(sizeof(Bvh8Link) * static_cast<size_t>(nodeSlots));
    out.triIds.reserve(sizeof(gfx_tri_ids) * static_cast<size_t>(n));
    out.rootBoxes.reserve(sizeof(float) * 16);
    const uint32_t nStatic = ctx.subsetTris[0], nDynamic = ctx.subsetTris[1];
    if (nStatic == 0 || nDynamic == 0) {
        const SubtreeResult r = build_subtree(ctx, stream, out, nStatic ? 0 : 1, 0u, 1u, 0u, nullptr);
        out.numNodes = r.nodeEnd; out.maxDepth = r.depth;
    }
    else {
        out.split = true;
        out.numStaticTris = nStatic;
        const SubtreeResult rs = build_subtree(ctx, stream, out, 0, 1u, 3u, 0u, out.rootBoxes.as<float>());
        out.staticNodeEnd = std::max(rs.nodeEnd, 3u);
        const SubtreeResult rd = build_subtree(ctx, stream, out, 1, 2u, out.staticNodeEnd, nStatic, out.rootBoxes.as<float>() + 8);
        hipLaunchKernelGGL(k_super_root, dim3(1), dim3(64), 0, stream, out.rootBoxes.as<float>(), out.nodes.as<Bvh8Node>(), out.links.as<Bvh8Link>());
        out.numNodes = rd.nodeEnd; out.staticDepth = rs.depth; out.maxDepth = std::max(rs.depth, rd.depth) + 1;
    }
    // the traversal keeps one stack entry per level it has descended past (bvh8.hip.h LaneStack: LDS + spill area)
    if (out.maxDepth + 1 > static_cast<uint32_t>(kTraceLdsStackDepth + kSpillStackDepth))
        throw std::runtime_error("gfx: acceleration structure is " + std::to_string(out.maxDepth) + " levels deep; the traversal stack holds " +
                                 std::to_string(kTraceLdsStackDepth + kSpillStackDepth) + " entries");
    out.numTris = n;
    hipLaunchKernelGGL(k_tri_ids, dim3((n + 255) / 256), dim3(256), 0, stream, out.trisPtr(), 0u, n, out.triIds.as<gfx_tri_ids>());
    GFX_HIP(hipGetLastError());
    GFX_HIP(hipStreamSynchronize(stream));
}

// Animated instances moved, nothing else changed: rebuild their subtree and the two-child root.  For the usual
// handful of triangles this is a few small kernels and no host round trip.
bool lbvh_update_dynamic(Context& ctx, hipStream_t stream, Accel& out) {
    if (ctx.sceneDirty || !out.split || out.numInputTris != ctx.totalTriangles || out.numStaticTris != ctx.subsetTris[0]) return false;
    transforms_upload(ctx, stream);
    const uint32_t nStatic = ctx.subsetTris[0], nDynamic = ctx.subsetTris[1];
    const SubtreeResult rd = build_subtree(ctx, stream, out, 1, 2u, out.staticNodeEnd, nStatic, out.rootBoxes.as<float>() + 8);
    hipLaunchKernelGGL(k_super_root, dim3(1), dim3(64), 0, stream, out.rootBoxes.as<float>(), out.nodes.as<Bvh8Node>(), out.links.as<Bvh8Link>());
    hipLaunchKernelGGL(k_tri_ids, dim3((nDynamic + 255) / 256), dim3(256), 0, stream, out.trisPtr(), nStatic, nDynamic, out.triIds.as<gfx_tri_ids>());
    GFX_HIP(hipGetLastError());
    out.numNodes = rd.nodeEnd;
    // The animated subtree may have grown deeper than it was at build time: the kernels that trace inside the producing kernel size their
    // per-lane spill area by the tree's depth (trace_local.hip.h local_spill_depth; a push past it would be dropped).  The depth never
    // shrinks here, so spill areas sized for an earlier frame stay large enough.
    out.maxDepth = std::max(out.maxDepth, std::max(out.staticDepth, rd.depth) + 1);
    if (out.maxDepth + 1 > static_cast<uint32_t>(kTraceLdsStackDepth + kSpillStackDepth))
        throw std::runtime_error("gfx: acceleration structure is " + std::to_string(out.maxDepth) + " levels deep after the update; the traversal stack holds " +
                                 std::to_string(kTraceLdsStackDepth + kSpillStackDepth) + " entries");
    return true;
}

} // namespace gfx
