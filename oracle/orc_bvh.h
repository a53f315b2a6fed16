// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// orc_bvh.h: CPU restatement of common/bvh_builder.{h,cpp} (binned-SAH + spatial-split BVH8
// builder, 80-byte quantised nodes, compressed-stack scalar traversal) plus a brute-force
// closest-hit / any-hit used to pin it.
//   node layout            common/common_shared.h:756-917  (CompressedInternalNode_T<8>)
//   build                  common/bvh_builder.cpp:213-311 (object split), :313-415 (spatial split),
//                          :419-486 (partition), :506-652 (spatial split execution), :656-1125
//   traversal              common/bvh_builder.cpp:1227-1247 (sortOrder), :1251-1270 (triangle test),
//                          :1272-1514 (compressed stack)
//   AABB slab test         common/basic_types.h:3450-3465
#pragma once
#include <algorithm>
#include <limits>
#include <vector>
#include "orc_math.h"

namespace orc {
namespace bvh {

constexpr uint32_t arity = 8;

struct AABB { // common/basic_types.h AABB_T
    V3 minP, maxP;
    AABB() : minP(INFINITY), maxP(-INFINITY) {}
    AABB(V3 a, V3 b) : minP(a), maxP(b) {}
    AABB& unify(V3 p) { minP = vmin(minP, p); maxP = vmax(maxP, p); return *this; }
    AABB& unify(const AABB& b) { minP = vmin(minP, b.minP); maxP = vmax(maxP, b.maxP); return *this; }
    AABB& intersect(const AABB& b) { minP = vmax(minP, b.minP); maxP = vmin(maxP, b.maxP); return *this; }
    V3 getCenter() const { return 0.5f * (minP + maxP); }
    float calcHalfSurfaceArea() const { const V3 d = maxP - minP; return d.x * d.y + d.y * d.z + d.z * d.x; }
    V3 normalize(V3 p) const { // safeDivide(p - minP, maxP - minP)
        const V3 n = p - minP, d = maxP - minP;
        return V3(d.x != 0 ? n.x / d.x : 0.0f, d.y != 0 ? n.y / d.y : 0.0f, d.z != 0 ? n.z / d.z : 0.0f);
    }
    bool isValid() const { const V3 d = maxP - minP; return d.x >= 0 && d.y >= 0 && d.z >= 0; }
    // basic_types.h:3450-3465.  widen = false is the reference's test as written.  Its slab distances carry the rounding of
    // (plane - org) * (1 / dir) while the triangle test computes its distance another way, so a triangle lying IN a box face
    // (flat, axis-aligned geometry: every sign and wall of the street scenes) can be accepted at a distance one ulp below the
    // entry distance computed for its own box -- and is then culled once a coincident triangle of another instance was hit
    // first (found at 1920x1080 on the textured street: two overlapping sign instances).  The reference only runs this
    // traversal in a builder test harness; the oracle renders with it, so its ray queries widen every slab by
    // 2^-20 |1/dir_k| (|org_k| + |plane_k|), twice the bound of that rounding.  A widened test only ever visits MORE boxes.
    bool intersect(V3 org, V3 dir, float distMin, float distMax, float* hitDistMin, float* hitDistMax, bool widen = false) const {
        if (!isValid()) return false;
        const V3 invRayDir(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
        const V3 tNear = (minP - org) * invRayDir;
        const V3 tFar = (maxP - org) * invRayDir;
        V3 near_ = vminNaN(tNear, tFar);
        V3 far_ = vmaxNaN(tNear, tFar);
        if (widen) {
            constexpr float k = 9.5367431640625e-07f;   // 2^-20
            const V3 e(k * std::fabs(invRayDir.x) * (std::fabs(org.x) + std::fmax(std::fabs(minP.x), std::fabs(maxP.x))),
                       k * std::fabs(invRayDir.y) * (std::fabs(org.y) + std::fmax(std::fabs(minP.y), std::fabs(maxP.y))),
                       k * std::fabs(invRayDir.z) * (std::fabs(org.z) + std::fmax(std::fabs(minP.z), std::fabs(maxP.z))));
            // an axis the ray runs parallel to has e = inf and near / far = -+inf already (or NaN, which fmin / fmax drop)
            if (std::isfinite(e.x)) { near_.x -= e.x; far_.x += e.x; }
            if (std::isfinite(e.y)) { near_.y -= e.y; far_.y += e.y; }
            if (std::isfinite(e.z)) { near_.z -= e.z; far_.z += e.z; }
        }
        *hitDistMin = std::fmax(std::fmax(near_.x, near_.y), near_.z);
        *hitDistMax = std::fmin(std::fmin(far_.x, far_.y), far_.z);
        *hitDistMin = std::fmax(*hitDistMin, distMin);
        *hitDistMax = std::fmin(*hitDistMax, distMax);
        return *hitDistMin <= *hitDistMax && *hitDistMax > 0.0f;
    }
    static V3 vminNaN(V3 a, V3 b) { return V3(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)); }
    static V3 vmaxNaN(V3 a, V3 b) { return V3(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)); }
};
static inline AABB unify(AABB a, const AABB& b) { return a.unify(b); }
static inline AABB unify(AABB a, V3 p) { return a.unify(p); }
static inline AABB intersect(AABB a, const AABB& b) { return a.intersect(b); }

static inline uint32_t floatToOrderedUInt(float f) { // basic_types.h:429-436
    const uint32_t u = f2bits(f);
    return u ^ (u < 0x80000000u ? 0x80000000u : 0xFFFFFFFFu);
}
static inline uint32_t popcnt(uint32_t x) { return static_cast<uint32_t>(__builtin_popcount(x)); }

#pragma pack(push, 1)
struct InternalNode { // common/common_shared.h:756-917, 80 bytes
    V3 quantBoxOrigin;
    uint8_t quantBoxExpScaleX, quantBoxExpScaleY, quantBoxExpScaleZ;
    uint8_t internalMask;
    uint32_t intNodeChildBaseIndex;
    uint32_t leafBaseIndex;
    uint8_t childMetas[arity]; // leafOffset
    uint8_t childQMinXs[arity], childQMinYs[arity], childQMinZs[arity];
    uint8_t childQMaxXs[arity], childQMaxYs[arity], childQMaxZs[arity];

    V3 decodeQuantBoxScale() const {
        return V3(bits2f(static_cast<uint32_t>(quantBoxExpScaleX) << 23),
                  bits2f(static_cast<uint32_t>(quantBoxExpScaleY) << 23),
                  bits2f(static_cast<uint32_t>(quantBoxExpScaleZ) << 23));
    }
    void setQuantizationAabb(const AABB& box) { // :814-830
        quantBoxOrigin = box.minP;
        const V3 d = (box.maxP - box.minP) / 255;
        auto calcExpScale = [](float s) {
            const uint32_t us = f2bits(s);
            return static_cast<uint8_t>((us >> 23) + ((us & 0x7FFFFF) ? 1 : 0));
        };
        quantBoxExpScaleX = calcExpScale(d.x);
        quantBoxExpScaleY = calcExpScale(d.y);
        quantBoxExpScaleZ = calcExpScale(d.z);
    }
    void setChildAabb(uint32_t slot, const AABB& box) { // :839-851
        const V3 s = decodeQuantBoxScale();
        const V3 recD(s.x != 0 ? 1.0f / s.x : 0.0f, s.y != 0 ? 1.0f / s.y : 0.0f, s.z != 0 ? 1.0f / s.z : 0.0f);
        const V3 qMinPf = (box.minP - quantBoxOrigin) * recD;
        const V3 qMaxPf = (box.maxP - quantBoxOrigin) * recD;
        auto u = [](float f) { return f2u(f); };
        auto mx = [](uint32_t v) { return v + 1 < 255u ? v + 1 : 255u; };
        childQMinXs[slot] = static_cast<uint8_t>(u(qMinPf.x));
        childQMinYs[slot] = static_cast<uint8_t>(u(qMinPf.y));
        childQMinZs[slot] = static_cast<uint8_t>(u(qMinPf.z));
        childQMaxXs[slot] = static_cast<uint8_t>(mx(u(qMaxPf.x)));
        childQMaxYs[slot] = static_cast<uint8_t>(mx(u(qMaxPf.y)));
        childQMaxZs[slot] = static_cast<uint8_t>(mx(u(qMaxPf.z)));
    }
    void setInvalidChildBox(uint32_t slot) {
        childQMinXs[slot] = childQMinYs[slot] = childQMinZs[slot] = 255;
        childQMaxXs[slot] = childQMaxYs[slot] = childQMaxZs[slot] = 0;
    }
    bool getChildIsValid(uint32_t slot) const { return childQMinXs[slot] != 255 || childQMaxXs[slot] != 0; }
    AABB getChildAabb(uint32_t slot) const { // :802-812, 867-871
        const V3 d = decodeQuantBoxScale();
        const V3 qMinPf(childQMinXs[slot], childQMinYs[slot], childQMinZs[slot]);
        const V3 qMaxPf(childQMaxXs[slot], childQMaxYs[slot], childQMaxZs[slot]);
        return AABB(quantBoxOrigin + qMinPf * d, quantBoxOrigin + qMaxPf * d);
    }
    bool getChildIsLeaf(uint32_t slot) const { return ((internalMask >> slot) & 1) == 0; }
    uint32_t getInternalChildNumber(uint32_t slot) const { return popcnt(internalMask & ((1u << slot) - 1)); }
};
#pragma pack(pop)
static_assert(sizeof(InternalNode) == 80, "CompressedInternalNode_T<8> is 80 bytes");

struct PrimitiveReference { uint32_t storageIndex : 31; uint32_t isLeafEnd : 1; }; // :1012-1015
struct TriangleStorage { V3 pA, pB, pC; uint32_t geomIndex, primIndex, padding; };  // :1017-1024
static_assert(sizeof(TriangleStorage) == 48, "TriangleStorage is 48 bytes");

struct HitObject { // :1065-1078
    float dist;
    uint32_t geomIndex, primIndex;
    float bcA, bcB, bcC;
    bool isHit() const { return primIndex != 0xFFFFFFFFu; }
};

struct Geometry { // common/bvh_builder.h:26-36 (Fp32x3 / UI32x3 only)
    const uint8_t* vertices; uint32_t vertexStride; uint32_t numVertices;
    const uint8_t* triangles; uint32_t triangleStride; uint32_t numTriangles;
    M34 preTransform;
};
struct BuildConfig { // common/bvh_builder.h:38-44
    float splittingBudget = 0.3f;
    float intNodeTravCost = 1.2f;
    float primIntersectCost = 1.0f;
    uint32_t minNumPrimsPerLeaf = 1;
    uint32_t maxNumPrimsPerLeaf = 128;
};
struct TraversalStatistics { // common/bvh_builder.h:79-86 (+ node fetch counter)
    uint64_t numAabbTests = 0, numTriTests = 0, numNodeFetches = 0;
    int32_t maxStackDepth = -1;
};

struct GeometryBVH {
    std::vector<InternalNode> intNodes;
    std::vector<TriangleStorage> triStorages;
    std::vector<PrimitiveReference> primRefs;
    uint32_t numGeoms = 0, totalNumPrims = 0;
};

// ---------------------------------------------------------------- builder internals
constexpr int32_t numObjBins = 16, numObjPlanes = 15, numSpaBins = 32, numSpaPlanes = 31;

struct BPrimRef { AABB box; uint32_t geomIndex = 0, primIndex = 0; };
struct PrimSplitInfo { uint8_t bin[3]; uint8_t isRight; };
struct SplitTask {
    AABB geomAabb, centAabb;
    uint32_t begin = 0, reserved = 0;   // span into the primRef arrays
    uint32_t numActualElems = 0;
    uint32_t parentIndex = 0;
    uint32_t slotInParent = 0;
    bool isSplittable = false;
};
struct SplitInfo {
    uint32_t leftPrimCount, rightPrimCount;
    AABB leftAabb, rightAabb;
    float cost;
    uint32_t dim, planeIndex;
    bool isSpecialSplit;
};

static inline void calcTriangleVertices(const Geometry* geoms, uint32_t g, uint32_t p, V3* pA, V3* pB, V3* pC) { // :178-209
    const Geometry& geom = geoms[g];
    const uint32_t* tri = reinterpret_cast<const uint32_t*>(geom.triangles + static_cast<size_t>(geom.triangleStride) * p);
    V3 ps[3];
    for (int i = 0; i < 3; ++i) {
        const float* v = reinterpret_cast<const float*>(geom.vertices + static_cast<size_t>(geom.vertexStride) * tri[i]);
        ps[i] = V3(v[0], v[1], v[2]);
    }
    *pA = xfmPoint(geom.preTransform, ps[0]);
    *pB = xfmPoint(geom.preTransform, ps[1]);
    *pC = xfmPoint(geom.preTransform, ps[2]);
}

class Builder {
    const Geometry* geoms;
    uint32_t numGeoms;
    BuildConfig cfg;
    std::vector<BPrimRef> primRefs;
    std::vector<PrimSplitInfo> infos;

    void findBestObjectSplit(const SplitTask& t, SplitInfo* out) { // :213-311
        AABB binAabbs[numObjBins][3];
        uint32_t binCounts[numObjBins][3] = {};
        for (uint32_t i = 0; i < t.numActualElems; ++i) {
            const BPrimRef& r = primRefs[t.begin + i];
            const V3 np = t.centAabb.normalize(r.box.getCenter());
            for (int dim = 0; dim < 3; ++dim) {
                uint32_t b = f2u(numObjBins * np[dim]);
                if (b > numObjBins - 1) b = numObjBins - 1;
                binAabbs[b][dim].unify(r.box);
                ++binCounts[b][dim];
                infos[t.begin + i].bin[dim] = static_cast<uint8_t>(b);
            }
        }
        AABB rightAabbs[numObjPlanes][3];
        uint32_t rightCounts[numObjPlanes][3];
        {
            AABB acc[3]; uint32_t cnt[3] = { 0, 0, 0 };
            for (int32_t pl = numObjPlanes - 1; pl >= 0; --pl)
                for (int dim = 0; dim < 3; ++dim) {
                    acc[dim].unify(binAabbs[pl + 1][dim]);
                    cnt[dim] += binCounts[pl + 1][dim];
                    rightAabbs[pl][dim] = acc[dim];
                    rightCounts[pl][dim] = cnt[dim];
                }
        }
        int32_t bestPlane[3] = { -1, -1, -1 };
        float bestCost[3] = { INFINITY, INFINITY, INFINITY };
        uint32_t bestL[3] = {}, bestR[3] = {};
        AABB bestLA[3], bestRA[3];
        {
            AABB acc[3]; uint32_t cnt[3] = { 0, 0, 0 };
            for (int32_t pl = 0; pl < numObjPlanes; ++pl)
                for (int dim = 0; dim < 3; ++dim) {
                    acc[dim].unify(binAabbs[pl][dim]);
                    cnt[dim] += binCounts[pl][dim];
                    const float leftArea = acc[dim].calcHalfSurfaceArea();
                    const float rightArea = rightAabbs[pl][dim].calcHalfSurfaceArea();
                    const float cost = leftArea * cnt[dim] + rightArea * rightCounts[pl][dim];
                    if (cost < bestCost[dim]) {
                        bestPlane[dim] = pl; bestCost[dim] = cost;
                        bestL[dim] = cnt[dim]; bestR[dim] = rightCounts[pl][dim];
                        bestLA[dim] = acc[dim]; bestRA[dim] = rightAabbs[pl][dim];
                    }
                }
        }
        const uint32_t bd = static_cast<uint32_t>(std::min_element(bestCost, bestCost + 3) - bestCost);
        out->dim = bd; out->planeIndex = static_cast<uint32_t>(bestPlane[bd]); out->cost = bestCost[bd];
        out->leftPrimCount = bestL[bd]; out->rightPrimCount = bestR[bd];
        out->leftAabb = bestLA[bd]; out->rightAabb = bestRA[bd];
        out->isSpecialSplit = false;
    }

    void findBestSpatialSplit(const SplitTask& t, SplitInfo* out) { // :313-415
        AABB binAabbs[numSpaBins][3];
        uint32_t entryCounts[numSpaBins][3] = {}, exitCounts[numSpaBins][3] = {};
        const V3 planePosCoeff = (t.geomAabb.maxP - t.geomAabb.minP) / numSpaBins;
        for (uint32_t i = 0; i < t.numActualElems; ++i) {
            const BPrimRef& r = primRefs[t.begin + i];
            const V3 entryNp = t.geomAabb.normalize(r.box.minP);
            const V3 exitNp = t.geomAabb.normalize(r.box.maxP);
            for (int dim = 0; dim < 3; ++dim) {
                uint32_t e0 = f2u(numSpaBins * entryNp[dim]); if (e0 > numSpaBins - 1) e0 = numSpaBins - 1;
                uint32_t e1 = f2u(numSpaBins * exitNp[dim]); if (e1 > numSpaBins - 1) e1 = numSpaBins - 1;
                for (int32_t b = static_cast<int32_t>(e0); b <= static_cast<int32_t>(e1); ++b)
                    binAabbs[b][dim].unify(r.box);
                ++entryCounts[e0][dim];
                ++exitCounts[e1][dim];
            }
        }
        AABB rightAabbs[numSpaPlanes][3];
        uint32_t rightCounts[numSpaPlanes][3];
        {
            AABB acc[3]; uint32_t cnt[3] = { 0, 0, 0 };
            for (int32_t pl = numSpaPlanes - 1; pl >= 0; --pl)
                for (int dim = 0; dim < 3; ++dim) {
                    acc[dim].unify(binAabbs[pl + 1][dim]);
                    cnt[dim] += exitCounts[pl + 1][dim];
                    AABB ra = acc[dim];
                    ra.minP[dim] = t.geomAabb.minP[dim] + (pl + 1) * planePosCoeff[dim];
                    rightAabbs[pl][dim] = ra;
                    rightCounts[pl][dim] = cnt[dim];
                }
        }
        int32_t bestPlane[3] = { -1, -1, -1 };
        float bestCost[3] = { INFINITY, INFINITY, INFINITY };
        uint32_t bestL[3] = {}, bestR[3] = {};
        AABB bestLA[3], bestRA[3];
        {
            AABB acc[3]; uint32_t cnt[3] = { 0, 0, 0 };
            for (int32_t pl = 0; pl < numSpaPlanes; ++pl)
                for (int dim = 0; dim < 3; ++dim) {
                    acc[dim].unify(binAabbs[pl][dim]);
                    cnt[dim] += entryCounts[pl][dim];
                    AABB la = acc[dim];
                    la.maxP[dim] = t.geomAabb.minP[dim] + (pl + 1) * planePosCoeff[dim];
                    const float leftArea = la.calcHalfSurfaceArea();
                    const float rightArea = rightAabbs[pl][dim].calcHalfSurfaceArea();
                    const float cost = leftArea * cnt[dim] + rightArea * rightCounts[pl][dim];
                    if (cost < bestCost[dim]) {
                        bestPlane[dim] = pl; bestCost[dim] = cost;
                        bestL[dim] = cnt[dim]; bestR[dim] = rightCounts[pl][dim];
                        bestLA[dim] = la; bestRA[dim] = rightAabbs[pl][dim];
                    }
                }
        }
        const uint32_t bd = static_cast<uint32_t>(std::min_element(bestCost, bestCost + 3) - bestCost);
        out->dim = bd; out->planeIndex = static_cast<uint32_t>(bestPlane[bd]); out->cost = bestCost[bd];
        out->leftPrimCount = bestL[bd]; out->rightPrimCount = bestR[bd];
        out->leftAabb = bestLA[bd]; out->rightAabb = bestRA[bd];
        out->isSpecialSplit = true;
    }

    template <typename Pred>
    void performPartition(const SplitTask& t, Pred pred, uint32_t leftCount, uint32_t rightCount,
                          SplitTask* L, SplitTask* R) { // :419-486
        *L = SplitTask(); *R = SplitTask();
        const uint32_t numActual = leftCount + rightCount;
        uint32_t li = 0, ri = numActual - 1;
        BPrimRef* refs = primRefs.data() + t.begin;
        PrimSplitInfo* inf = infos.data() + t.begin;
        while (li < ri) {
            while (li < ri && pred(li)) {
                L->geomAabb.unify(refs[li].box); L->centAabb.unify(refs[li].box.getCenter()); ++li;
            }
            while (li < ri && !pred(ri)) {
                R->geomAabb.unify(refs[ri].box); R->centAabb.unify(refs[ri].box.getCenter()); --ri;
            }
            if (li < ri) { std::swap(refs[li], refs[ri]); std::swap(inf[li], inf[ri]); }
            else { R->geomAabb.unify(refs[ri].box); R->centAabb.unify(refs[ri].box.getCenter()); }
        }
        const uint32_t reserved = t.reserved;
        const uint32_t leftReserved = std::max(
            static_cast<uint32_t>(reserved * static_cast<float>(leftCount) / (leftCount + rightCount)), leftCount);
        const uint32_t rightReserved = reserved - leftReserved;
        if (leftCount < leftReserved) {
            std::copy_backward(refs + leftCount, refs + numActual, refs + leftReserved + rightCount);
            for (uint32_t i = leftCount; i < leftReserved; ++i) refs[i] = BPrimRef();
        }
        L->begin = t.begin; L->reserved = leftReserved; L->numActualElems = leftCount;
        L->isSplittable = leftCount > cfg.minNumPrimsPerLeaf;
        R->begin = t.begin + leftReserved; R->reserved = rightReserved; R->numActualElems = rightCount;
        R->isSplittable = rightCount > cfg.minNumPrimsPerLeaf;
    }

    static void splitTriangle(V3 pA, V3 pB, V3 pC, float splitPlane, uint32_t axis, AABB* bbA, AABB* bbB) { // :506-545
        uint32_t mask = ((pC[axis] >= splitPlane) << 2) | ((pB[axis] >= splitPlane) << 1) | ((pA[axis] >= splitPlane) << 0);
        bool lrSwap = false;
        if (pA[axis] >= splitPlane) { mask = ~mask & 0b111; lrSwap = true; }
        if (popcnt(mask) == 1) {
            const V3 temp = pA;
            if (mask == 0b010) { pA = pB; pB = temp; }
            else { pA = pC; pC = temp; }
            lrSwap ^= true;
        }
        const float tAB = (splitPlane - pA[axis]) / (pB[axis] - pA[axis]);
        const V3 pAB = pA + tAB * (pB - pA);
        const float tAC = (splitPlane - pA[axis]) / (pC[axis] - pA[axis]);
        const V3 pAC = pA + tAC * (pC - pA);
        AABB aabb;
        aabb.unify(pAB).unify(pAC);
        *bbA = unify(aabb, pA);
        *bbB = unify(aabb, pB).unify(pC);
        if (lrSwap) std::swap(*bbA, *bbB);
    }

    void performSpatialSplit(const SplitTask& t, const SplitInfo& s, SplitTask* L, SplitTask* R) { // :547-652
        const uint32_t dim = s.dim, planeIdx = s.planeIndex;
        const float binCoeff = (t.geomAabb.maxP[dim] - t.geomAabb.minP[dim]) / numSpaBins;
        const float splitPlane = t.geomAabb.minP[dim] + (planeIdx + 1) * binCoeff;
        const float addToLeftPartialCost = s.rightAabb.calcHalfSurfaceArea() * (s.rightPrimCount - 1);
        const float addToRightPartialCost = s.leftAabb.calcHalfSurfaceArea() * (s.leftPrimCount - 1);
        uint32_t leftCount = 0, rightCount = 0, cur = t.numActualElems;
        for (uint32_t i = 0; i < t.numActualElems; ++i) {
            BPrimRef& r = primRefs[t.begin + i];
            PrimSplitInfo& inf = infos[t.begin + i];
            const float fEntry = (r.box.minP[dim] - t.geomAabb.minP[dim]) / binCoeff;
            const uint32_t entryBin = std::min(f2u(fEntry), static_cast<uint32_t>(numSpaBins - 1));
            const float fExit = (r.box.maxP[dim] - t.geomAabb.minP[dim]) / binCoeff;
            const uint32_t exitBin = std::min(f2u(fExit), static_cast<uint32_t>(numSpaBins - 1));
            if (entryBin <= planeIdx && exitBin > planeIdx) {
                const float splitCost = s.cost;
                const float addToLeftCost =
                    unify(s.leftAabb, r.box).calcHalfSurfaceArea() * s.leftPrimCount + addToLeftPartialCost;
                const float addToRightCost =
                    addToRightPartialCost + unify(s.rightAabb, r.box).calcHalfSurfaceArea() * s.rightPrimCount;
                if (splitCost < addToLeftCost && splitCost < addToRightCost && cur < t.reserved) {
                    inf.isRight = 0;
                    V3 pA, pB, pC;
                    calcTriangleVertices(geoms, r.geomIndex, r.primIndex, &pA, &pB, &pC);
                    BPrimRef& nr = primRefs[t.begin + cur];
                    PrimSplitInfo& ninf = infos[t.begin + cur];
                    AABB la, ra;
                    splitTriangle(pA, pB, pC, splitPlane, dim, &la, &ra);
                    la.intersect(r.box); ra.intersect(r.box);
                    r.box = la; nr.box = ra;
                    nr.geomIndex = r.geomIndex; nr.primIndex = r.primIndex;
                    ninf.isRight = 1;
                    ++leftCount; ++rightCount; ++cur;
                }
                else if (addToLeftCost < addToRightCost) { inf.isRight = 0; ++leftCount; }
                else { inf.isRight = 1; ++rightCount; }
            }
            else {
                if (entryBin <= planeIdx) { inf.isRight = 0; ++leftCount; }
                else { inf.isRight = 1; ++rightCount; }
            }
        }
        const PrimSplitInfo* base = infos.data() + t.begin;
        performPartition(t, [base](uint32_t idx) { return !base[idx].isRight; }, leftCount, rightCount, L, R);
    }

public:
    Builder(const Geometry* g, uint32_t n, const BuildConfig& c) : geoms(g), numGeoms(n), cfg(c) {}

    void build(GeometryBVH* bvhOut) { // :656-1125
        std::vector<uint32_t> inputPrimOffsets(numGeoms);
        uint32_t numInput = 0;
        for (uint32_t g = 0; g < numGeoms; ++g) { inputPrimOffsets[g] = numInput; numInput += geoms[g].numTriangles; }
        const uint32_t allocated = std::max(numInput, static_cast<uint32_t>((1.0f + cfg.splittingBudget) * numInput));
        primRefs.assign(allocated, BPrimRef());
        infos.assign(allocated, PrimSplitInfo());
        {
            uint32_t idx = 0;
            for (uint32_t g = 0; g < numGeoms; ++g)
                for (uint32_t p = 0; p < geoms[g].numTriangles; ++p, ++idx) {
                    V3 pA, pB, pC;
                    calcTriangleVertices(geoms, g, p, &pA, &pB, &pC);
                    BPrimRef r; r.box.unify(pA).unify(pB).unify(pC); r.geomIndex = g; r.primIndex = p;
                    primRefs[idx] = r;
                }
        }
        struct TempChild { AABB aabb; uint32_t index = 0; uint32_t numLeaves = 0; };
        struct TempNode { TempChild children[arity]; };
        std::vector<SplitTask> stack;
        {
            SplitTask root;
            for (uint32_t i = 0; i < numInput; ++i) {
                root.geomAabb.unify(primRefs[i].box);
                root.centAabb.unify(primRefs[i].box.getCenter());
            }
            root.begin = 0; root.reserved = allocated; root.numActualElems = numInput;
            root.parentIndex = 0xFFFFFFFFu; root.slotInParent = 0; root.isSplittable = numInput > 1;
            stack.push_back(root);
        }
        const bool allowPrimRefIncrease = allocated > numInput;
        const float rootSA = stack.back().geomAabb.calcHalfSurfaceArea();
        std::vector<TempNode> tempNodes;

        while (!stack.empty()) {
            const SplitTask task = stack.back();
            stack.pop_back();
            SplitTask children[arity];
            children[0] = task;
            uint32_t numChildren = 1;
            while (numChildren < arity) {
                float maxArea = -INFINITY;
                uint32_t slotToSplit = 0xFFFFFFFFu;
                for (uint32_t slot = 0; slot < numChildren; ++slot) {
                    if (!children[slot].isSplittable) continue;
                    const float area = children[slot].geomAabb.calcHalfSurfaceArea();
                    if (area > maxArea) { maxArea = area; slotToSplit = slot; }
                }
                if (slotToSplit == 0xFFFFFFFFu) break;
                const SplitTask taskToSplit = children[slotToSplit];
                const uint32_t n = taskToSplit.numActualElems;
                const float geomSA = taskToSplit.geomAabb.calcHalfSurfaceArea();
                const float leafCost = geomSA * n * cfg.primIntersectCost;
                SplitInfo splitInfo;
                findBestObjectSplit(taskToSplit, &splitInfo);
                float splitCost = geomSA * cfg.intNodeTravCost + splitInfo.cost * cfg.primIntersectCost;
                const bool objSplitSuccess = !std::isinf(splitInfo.cost);
                if (allowPrimRefIncrease && objSplitSuccess && n < taskToSplit.reserved) {
                    const AABB overlapped = intersect(splitInfo.leftAabb, splitInfo.rightAabb);
                    const float overlappedSA = overlapped.isValid() ? overlapped.calcHalfSurfaceArea() : 0.0f;
                    constexpr float splittingThreshold = 1e-5f;
                    if (overlappedSA / rootSA > splittingThreshold) {
                        SplitInfo spa;
                        findBestSpatialSplit(taskToSplit, &spa);
                        const float spaCost = geomSA * cfg.intNodeTravCost + spa.cost * cfg.primIntersectCost;
                        if (spaCost < splitCost) { splitInfo = spa; splitCost = spaCost; }
                    }
                }
                if (leafCost < splitCost && n <= cfg.maxNumPrimsPerLeaf) {
                    children[slotToSplit].isSplittable = false;
                    continue;
                }
                SplitTask L, R;
                if (objSplitSuccess) {
                    if (splitInfo.isSpecialSplit) performSpatialSplit(taskToSplit, splitInfo, &L, &R);
                    else {
                        const PrimSplitInfo* base = infos.data() + taskToSplit.begin;
                        const uint32_t dim = splitInfo.dim, plane = splitInfo.planeIndex;
                        performPartition(taskToSplit,
                                         [base, dim, plane](uint32_t idx) { return base[idx].bin[dim] <= plane; },
                                         splitInfo.leftPrimCount, splitInfo.rightPrimCount, &L, &R);
                    }
                }
                else {
                    const uint32_t lc = n / 2, rc = n - lc;
                    performPartition(taskToSplit, [lc](uint32_t idx) { return idx < lc; }, lc, rc, &L, &R);
                }
                children[slotToSplit] = L;
                children[numChildren] = R;
                ++numChildren;
            }
            if (numChildren == 1 && task.parentIndex != 0xFFFFFFFFu) {
                TempChild& self = tempNodes[task.parentIndex].children[task.slotInParent];
                self.index = task.begin;
                self.numLeaves = task.numActualElems;
                continue;
            }
            std::stable_sort(children, children + numChildren,
                             [](const SplitTask& a, const SplitTask& b) { return a.numActualElems > b.numActualElems; });
            const uint32_t nodeIdx = static_cast<uint32_t>(tempNodes.size());
            if (task.parentIndex != 0xFFFFFFFFu)
                tempNodes[task.parentIndex].children[task.slotInParent].index = nodeIdx;
            tempNodes.resize(tempNodes.size() + 1);
            for (uint32_t slot = 0; slot < numChildren; ++slot) {
                SplitTask& ct = children[slot];
                TempChild& child = tempNodes[nodeIdx].children[slot];
                child.aabb = ct.geomAabb;
                if (ct.isSplittable) {
                    ct.parentIndex = nodeIdx; ct.slotInParent = slot;
                    stack.push_back(ct);
                    child.numLeaves = 0;
                }
                else { child.index = ct.begin; child.numLeaves = ct.numActualElems; }
            }
            for (uint32_t slot = numChildren; slot < arity; ++slot) {
                TempChild& child = tempNodes[nodeIdx].children[slot];
                child.aabb = AABB(); child.index = 0xFFFFFFFFu; child.numLeaves = 0;
            }
        }

        std::vector<TriangleStorage> triStorages(numInput);
        {
            uint32_t idx = 0;
            for (uint32_t g = 0; g < numGeoms; ++g)
                for (uint32_t p = 0; p < geoms[g].numTriangles; ++p, ++idx) {
                    TriangleStorage& ts = triStorages[idx];
                    calcTriangleVertices(geoms, g, p, &ts.pA, &ts.pB, &ts.pC);
                    ts.geomIndex = g; ts.primIndex = p; ts.padding = 0;
                }
        }
        const uint32_t numIntNodes = static_cast<uint32_t>(tempNodes.size());
        std::vector<uint32_t> dstIdx(numIntNodes), leafBlock(numIntNodes);
        dstIdx[0] = 0;
        uint32_t intChildBlockIdx = 1, leafChildBlockIdx = 0;
        for (uint32_t i = 0; i < numIntNodes; ++i) { // :974-997
            leafBlock[i] = leafChildBlockIdx;
            uint32_t intChildCount = 0;
            for (uint32_t slot = 0; slot < arity; ++slot) {
                const TempChild& c = tempNodes[i].children[slot];
                if (c.index == 0xFFFFFFFFu) break;
                if (c.numLeaves > 0) leafChildBlockIdx += c.numLeaves;
                else { dstIdx[c.index] = intChildBlockIdx + intChildCount; ++intChildCount; }
            }
            intChildBlockIdx += intChildCount;
        }
        std::vector<InternalNode> dstNodes(numIntNodes);
        std::vector<PrimitiveReference> dstPrimRefs(leafChildBlockIdx);
        for (uint32_t si = 0; si < numIntNodes; ++si) { // :1009-1079
            const TempNode& src = tempNodes[si];
            InternalNode& dst = dstNodes[dstIdx[si]];
            AABB quantAabb;
            uint32_t internalMask = 0, firstIntChildSlot = 0xFFFFFFFFu;
            uint32_t primRefOffset = leafBlock[si], numValid = 0;
            for (uint32_t slot = 0; slot < arity; ++slot) {
                const TempChild& c = src.children[slot];
                if (c.index == 0xFFFFFFFFu) break;
                ++numValid;
                quantAabb.unify(c.aabb);
                if (c.numLeaves > 0) {
                    for (uint32_t k = 0; k < c.numLeaves; ++k) {
                        const BPrimRef& sp = primRefs[c.index + k];
                        PrimitiveReference& dp = dstPrimRefs[primRefOffset + k];
                        dp.storageIndex = inputPrimOffsets[sp.geomIndex] + sp.primIndex;
                        dp.isLeafEnd = k == c.numLeaves - 1;
                    }
                    primRefOffset += c.numLeaves;
                }
                else {
                    internalMask |= 1u << slot;
                    if (firstIntChildSlot == 0xFFFFFFFFu) firstIntChildSlot = slot;
                }
            }
            dst.setQuantizationAabb(quantAabb);
            dst.internalMask = static_cast<uint8_t>(internalMask);
            dst.intNodeChildBaseIndex = firstIntChildSlot != 0xFFFFFFFFu ? dstIdx[src.children[firstIntChildSlot].index] : 0xFFFFFFFFu;
            dst.leafBaseIndex = (~internalMask & ((1u << numValid) - 1)) ? leafBlock[si] : 0xFFFFFFFFu;
            uint32_t leafOffset = 0;
            for (uint32_t slot = 0; slot < arity; ++slot) {
                const TempChild& c = src.children[slot];
                if (c.index != 0xFFFFFFFFu) {
                    dst.setChildAabb(slot, c.aabb);
                    uint8_t meta = 0;
                    if (c.numLeaves > 0) { meta = static_cast<uint8_t>(leafOffset); leafOffset += c.numLeaves; }
                    dst.childMetas[slot] = meta;
                }
                else { dst.setInvalidChildBox(slot); dst.childMetas[slot] = 0; }
            }
        }
        bvhOut->intNodes = std::move(dstNodes);
        bvhOut->primRefs = std::move(dstPrimRefs);
        bvhOut->triStorages = std::move(triStorages);
        bvhOut->numGeoms = numGeoms;
        bvhOut->totalNumPrims = numInput;
    }
};

static inline void buildGeometryBVH(const Geometry* geoms, uint32_t numGeoms, const BuildConfig& cfg, GeometryBVH* bvh) {
    Builder b(geoms, numGeoms, cfg);
    b.build(bvh);
}

// ---------------------------------------------------------------- traversal
// common/bvh_builder.cpp:1251-1270
static inline bool testRayVsTriangle(V3 rayOrg, V3 rayDir, float distMin, float distMax,
                                     V3 pA, V3 pB, V3 pC, float* hitDist, float* bcB, float* bcC) {
    const V3 eAB = pB - pA;
    const V3 eCA = pA - pC;
    const V3 hitNormal = cross(eCA, eAB);
    const V3 e = (1.0f / dot(hitNormal, rayDir)) * (pA - rayOrg);
    const V3 i = cross(rayDir, e);
    *bcB = dot(i, eCA);
    *bcC = dot(i, eAB);
    *hitDist = dot(hitNormal, e);
    return (*hitDist < distMax) && (*hitDist > distMin) && (*bcB >= 0.0f) && (*bcC >= 0.0f) && (*bcB + *bcC <= 1);
}

static inline void sortOrder8(uint32_t (&keys)[8], uint32_t* values) { // :1239-1247
#define ORC_SWAP_ORDER(A, B) \
    if (keys[A] > keys[B]) { \
        std::swap(keys[A], keys[B]); \
        const uint32_t oa = A * 3, ob = B * 3; \
        const uint32_t vA = (*values >> oa) & 7u, vB = (*values >> ob) & 7u; \
        *values &= ~(7u << oa); *values |= (vB << oa); \
        *values &= ~(7u << ob); *values |= (vA << ob); \
    }
    ORC_SWAP_ORDER(0, 2) ORC_SWAP_ORDER(1, 3) ORC_SWAP_ORDER(4, 6) ORC_SWAP_ORDER(5, 7)
    ORC_SWAP_ORDER(0, 4) ORC_SWAP_ORDER(1, 5) ORC_SWAP_ORDER(2, 6) ORC_SWAP_ORDER(3, 7)
    ORC_SWAP_ORDER(0, 1) ORC_SWAP_ORDER(2, 3) ORC_SWAP_ORDER(4, 5) ORC_SWAP_ORDER(6, 7)
    ORC_SWAP_ORDER(2, 4) ORC_SWAP_ORDER(3, 5)
    ORC_SWAP_ORDER(1, 4) ORC_SWAP_ORDER(3, 6)
    ORC_SWAP_ORDER(1, 2) ORC_SWAP_ORDER(3, 4) ORC_SWAP_ORDER(5, 6)
#undef ORC_SWAP_ORDER
}

// common/bvh_builder.cpp:1272-1514 (USE_COMPRESSED_STACK path).  `anyHit` = stop at the first
// accepted triangle (the reference has closest-hit only; occlusion is order independent).
// `widenBoxes`: the renderer's ray queries (AABB::intersect).
static inline HitObject traverse(const GeometryBVH& bvh, V3 rayOrg, V3 rayDir, float distMin, float distMax,
                                 TraversalStatistics* stats = nullptr, bool anyHit = false, bool widenBoxes = false) {
    HitObject ret;
    ret.dist = distMax; ret.geomIndex = 0xFFFFFFFFu; ret.primIndex = 0xFFFFFFFFu;
    ret.bcA = ret.bcB = ret.bcC = NAN;
    if (bvh.intNodes.empty()) return ret;
    constexpr uint32_t orderBitWidth = 3, orderMask = 7;
    struct Entry { uint32_t baseIndex; uint32_t isLeafGroup; uint32_t orderInfo; uint32_t numItems; };
    Entry stack[64];
    uint8_t leafOffsets[arity] = {};
    int32_t stackIdx = 0;
    Entry curGroup = { 0, 0, 0, 1 };
    while (true) {
        if (curGroup.numItems == 0) {
            if (stackIdx == 0) break;
            curGroup = stack[--stackIdx];
        }
        Entry curTriGroup = { 0, 0, 0, 0 };
        if (curGroup.isLeafGroup) {
            curTriGroup = curGroup;
            curGroup.numItems = 0;
        }
        else {
            const uint32_t nodeIdx = curGroup.baseIndex + (curGroup.orderInfo & orderMask);
            curGroup.orderInfo >>= orderBitWidth;
            --curGroup.numItems;
            const InternalNode& intNode = bvh.intNodes[nodeIdx];
            if (stats) ++stats->numNodeFetches;
            uint32_t keys[arity];
            uint32_t orderInfo = 0, numIntHits = 0, numLeafHits = 0;
            for (uint32_t slot = 0; slot < arity; ++slot) {
                if (!intNode.getChildIsValid(slot)) {
                    for (; slot < arity; ++slot) keys[slot] = floatToOrderedUInt(INFINITY);
                    break;
                }
                if (stats) ++stats->numAabbTests;
                const AABB aabb = intNode.getChildAabb(slot);
                float hitDistMin, hitDistMax;
                if (aabb.intersect(rayOrg, rayDir, distMin, ret.dist, &hitDistMin, &hitDistMax, widenBoxes)) {
                    const bool isLeaf = intNode.getChildIsLeaf(slot);
                    const float dist = 0.5f * (hitDistMin + hitDistMax);
                    keys[slot] = (floatToOrderedUInt(dist) >> 1) | (static_cast<uint32_t>(!isLeaf) << 31);
                    if (isLeaf) { orderInfo |= (slot << (orderBitWidth * slot)); ++numLeafHits; }
                    else { orderInfo |= (intNode.getInternalChildNumber(slot) << (orderBitWidth * slot)); ++numIntHits; }
                }
                else keys[slot] = floatToOrderedUInt(INFINITY);
            }
            if (numIntHits + numLeafHits > 0) sortOrder8(keys, &orderInfo);
            if (numLeafHits > 0) {
                curTriGroup.numItems = numLeafHits;
                curTriGroup.baseIndex = intNode.leafBaseIndex;
                curTriGroup.isLeafGroup = 1;
                curTriGroup.orderInfo = orderInfo & 0x0FFFFFFFu;
                for (uint32_t slot = 0; slot < arity; ++slot) leafOffsets[slot] = intNode.childMetas[slot];
            }
            if (numIntHits > 0) {
                if (curGroup.numItems > 0) stack[stackIdx++] = curGroup;
                curGroup.numItems = numIntHits;
                curGroup.baseIndex = intNode.intNodeChildBaseIndex;
                curGroup.isLeafGroup = 0;
                curGroup.orderInfo = (orderInfo >> (orderBitWidth * numLeafHits)) & 0x0FFFFFFFu;
            }
            if (stats) stats->maxStackDepth = std::max(stats->maxStackDepth, stackIdx);
        }
        if (curTriGroup.numItems > 0) {
            const uint32_t slot = curTriGroup.orderInfo & orderMask;
            const uint32_t primRefIdx = curTriGroup.baseIndex + leafOffsets[slot]++;
            if (stats) ++stats->numTriTests;
            const PrimitiveReference primRef = bvh.primRefs[primRefIdx];
            const TriangleStorage& ts = bvh.triStorages[primRef.storageIndex];
            float hitDist, hitBcB, hitBcC;
            const bool hit = testRayVsTriangle(rayOrg, rayDir, distMin, ret.dist, ts.pA, ts.pB, ts.pC, &hitDist, &hitBcB, &hitBcC);
            if (hit) {
                ret.dist = hitDist; ret.geomIndex = ts.geomIndex; ret.primIndex = ts.primIndex;
                ret.bcA = 1.0f - (hitBcB + hitBcC); ret.bcB = hitBcB; ret.bcC = hitBcC;
                if (anyHit) return ret;
            }
            if (primRef.isLeafEnd) { curTriGroup.orderInfo >>= orderBitWidth; --curTriGroup.numItems; }
            if (curTriGroup.numItems > 0) {
                if (curGroup.numItems > 0) stack[stackIdx++] = curGroup;
                curGroup = curTriGroup;
            }
        }
    }
    return ret;
}

// Brute force over every triangle.  Tie rule (equal dist): lowest (geomIndex, primIndex) wins --
// this is the product's documented tie-break (the reference's result on exact ties depends on
// traversal order, bvh_builder.cpp:1486-1497).
static inline HitObject bruteForce(const std::vector<TriangleStorage>& tris, V3 rayOrg, V3 rayDir,
                                   float distMin, float distMax, bool anyHit = false) {
    HitObject ret;
    ret.dist = distMax; ret.geomIndex = 0xFFFFFFFFu; ret.primIndex = 0xFFFFFFFFu;
    ret.bcA = ret.bcB = ret.bcC = NAN;
    for (const TriangleStorage& ts : tris) {
        float t, b, c;
        // accept t <= current best; resolve ties by index
        if (!testRayVsTriangle(rayOrg, rayDir, distMin, distMax, ts.pA, ts.pB, ts.pC, &t, &b, &c)) continue;
        const bool closer = t < ret.dist ||
            (t == ret.dist && ret.isHit() &&
             (ts.geomIndex < ret.geomIndex || (ts.geomIndex == ret.geomIndex && ts.primIndex < ret.primIndex)));
        if (!ret.isHit() || closer) {
            ret.dist = t; ret.geomIndex = ts.geomIndex; ret.primIndex = ts.primIndex;
            ret.bcA = 1.0f - (b + c); ret.bcB = b; ret.bcC = c;
            if (anyHit) return ret;
        }
    }
    return ret;
}

} // namespace bvh
} // namespace orc
