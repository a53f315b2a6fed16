// scene.cpp -- flattens the host-side scene mirror into the HBM tables the kernels read.
// Reference: Scene slot buffers and createInstance (common/common_host.h:862-969,
// common/common_host.cpp:1817-1905, 2582-2656).
#include <cmath>
#include <cstring>
#include "internal.h"
#include <cstdlib>
#include <map>
#include <array>

namespace gfx {

Context::~Context() {
    for (Accel* a : accels) { if (a) { a->nodes.release(); a->links.release(); a->triIds.release(); a->rootBoxes.release(); delete a; } }
    DevBuf* all[] = { &dMaterials, &dGeomInsts, &dInsts, &dVertices, &dTriangles, &dSlotPool, &dFlatGeoms, &dSubset[0], &dSubset[1], &dLightW, &dLightP, &dLightCDF, &dLightRefs, &dEmitterRecs, &dEmitterRecExtras, &dLightNormalMatrices, &dInstMatrixIndex, &dTextures, &dTexelPool, &dSrgbLut, &dEmitterTexRefs,
                      &rayOrg, &rayDir, &rayOut, &rayHits, &spill, &gbRayOrg, &gbRayDir, &gbRayHits, &gbSpill, &gbCounters, &auxSpill, &auxCounters, &blockOrders[0].cost, &blockOrders[0].order, &blockOrders[1].cost, &blockOrders[1].order, &blockOrders[2].cost, &blockOrders[2].order, &blockOrders[3].cost, &blockOrders[3].order, &blockOrders[4].cost, &blockOrders[4].order, &pixelRaySlot, &shadeScratch, &spatialScratch, &smallCounters,
                      &bTris, &bBoxes, &bKeys, &bKeysAlt, &bVals, &bValsAlt, &bSortTemp, &bNodesLR, &bParents, &bFlags,
                      &bNodeBoxes, &bRanges, &bQueueA, &bQueueB, &bCounters, &dTraceCounters, &dLightInstIntegral, &dLightInstGuide, &dSpans, &dSpanGuide, &dSpanHeader, &dSpanInstBegin,
                      &dTraceDiag, &ptDiag, &bCosts, &bDec, &bFlatIdx, &ptPending, &ptExtOrg, &ptExtDir, &ptExtOwner, &ptState,
                      &rearchSlots, &nrcState, &neeTrainIdx };
    for (NrcNet* net : nrcNets) if (net) nrc_destroy(net);
    for (DevBuf* b : all) b->release();
    for (auto& e : pendingEvents) { (void)hipEventDestroy(e.second.first); (void)hipEventDestroy(e.second.second); }
    for (TicketState& ts : ticketState) if (ts.lastLaunch) (void)hipEventDestroy(ts.lastLaunch);
    for (PinnedStage& st : lightStage) { if (st.done) (void)hipEventDestroy(st.done); if (st.p) (void)hipHostFree(st.p); }
    for (BlockOrder& b : blockOrders) { if (b.counted) (void)hipEventDestroy(b.counted); if (b.ordered) (void)hipEventDestroy(b.ordered); }
    if (auxFork) (void)hipEventDestroy(auxFork);
    if (auxJoin) (void)hipEventDestroy(auxJoin);
    if (auxStream) (void)hipStreamDestroy(auxStream);
}

DevScene Context::devScene() const {
    DevScene s;
    s.materials = dMaterials.as<gfx_material>();
    s.geomInsts = dGeomInsts.as<DevGeomInst>();
    s.insts = dInsts.as<DevInstance>();
    s.vertices = dVertices.as<DevVertex>();
    s.triangles = dTriangles.as<uint32_t>();
    s.geomInstSlotPool = dSlotPool.as<uint32_t>();
    s.lightWeights = dLightW.as<float>();
    s.lightProbs = dLightP.as<float>();
    s.lightCDF = dLightCDF.as<float>();
    s.lightGeomRefs = dLightRefs.as<LightGeomRef>();
    s.emitterRecs = dEmitterRecs.as<EmitterRec>();
    s.emitterRecExtras = dEmitterRecExtras.as<EmitterRecExtra>();
    s.lightNormalMatrices = dLightNormalMatrices.as<float>();
    s.numLightMatrices = numLightMatrices;
    s.lightInstIntegral = dLightInstIntegral.as<float>();
    s.lightInstGuide = dLightInstGuide.as<uint16_t>();
    s.lightInstGuideCells = lightInstGuideCells;
    s.spans = dSpans.as<EmitterSpan>();
    s.spanGuide = dSpanGuide.as<SpanGuide>();
    s.spanHeader = dSpanHeader.as<uint32_t>();
    s.numSpans = numEmitterRecs;
    s.spanGuideCells = spanGuideCells;
    s.textures = dTextures.as<DevTexture>();
    s.texelPool = dTexelPool.as<uint32_t>();
    s.srgbLut = dSrgbLut.as<float>();
    s.emitterTexRefs = anyEmittanceTexture ? dEmitterTexRefs.as<EmitterTexRef>() : nullptr;
    s.lightInstDistOffset = lightInstDistOffset;
    s.numInsts = static_cast<uint32_t>(insts.size());
    return s;
}

// normalMatrix = transpose(invert(upper-left 3x3)); Matrix3x3::invert is determinant + adjugate
// scaled by 1/det (common/basic_types.h:4118-4157).  Written out so every product/sum happens in
// the same order as in the kernels' gfx::inverse.
static void normal_matrix(const float x[12], float out[12]) {
    const float a = x[0], b = x[1], c = x[2], d = x[4], e = x[5], f = x[6], g = x[8], h = x[9], i = x[10];
    const float det = a * e * i + b * f * g + c * d * h - c * e * g - b * d * i - a * f * h;
    const float r = 1 / det;
    const float inv[9] = {
        (e * i - f * h) * r, -(b * i - c * h) * r, (b * f - c * e) * r,
        -(d * i - f * g) * r, (a * i - c * g) * r, -(a * f - c * d) * r,
        (d * h - e * g) * r, -(a * h - b * g) * r, (a * e - b * d) * r };
    for (int rr = 0; rr < 3; ++rr) {
        for (int cc = 0; cc < 3; ++cc) out[rr * 4 + cc] = inv[cc * 3 + rr];
        out[rr * 4 + 3] = 0.0f;
    }
}

// curToPrevTransform = prevTransform * invert(matM2W) (common/common_host.h:851), rows 0..2 of the 4x4 product.
// Matrix4x4::invert (basic_types.h:4597-4626) is the cofactor expansion: entry (i, j) of the inverse = +/- the
// 3x3 minor without row j and column i, its six triple products summed in the order
//   (r0c0 r1c1 r2c2) - (r2c0 r1c1 r0c2) + (r1c0 r2c1 r0c2) - (r0c0 r2c1 r1c2) + (r2c0 r0c1 r1c2) - (r1c0 r0c1 r2c2)
// over the remaining rows r0 < r1 < r2 / columns c0 < c1 < c2, times 1 / det with det expanded along column 0;
// the matrix product is row-of-left . column-of-right, x y z w in that order (:4552-4559).
void instance_cur_to_prev(const float prev[12], const float cur[12], float out[12]) {
    float a[16], inv[16];   // row-major 4x4
    for (int k = 0; k < 12; ++k) a[k] = cur[k];
    a[12] = 0.0f; a[13] = 0.0f; a[14] = 0.0f; a[15] = 1.0f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            int rows[3], cols[3], nr = 0, nc = 0;
            for (int k = 0; k < 4; ++k) { if (k != j) rows[nr++] = k; if (k != i) cols[nc++] = k; }
            auto e = [&](int r, int c) { return a[4 * rows[r] + cols[c]]; };
            const float minor = (e(0, 0) * e(1, 1) * e(2, 2)) - (e(2, 0) * e(1, 1) * e(0, 2)) + (e(1, 0) * e(2, 1) * e(0, 2)) -
                                (e(0, 0) * e(2, 1) * e(1, 2)) + (e(2, 0) * e(0, 1) * e(1, 2)) - (e(1, 0) * e(0, 1) * e(2, 2));
            inv[4 * i + j] = ((i + j) % 2) ? -minor : minor;
        }
    const float recDet = 1.0f / (a[0] * inv[0] + a[4] * inv[1] + a[8] * inv[2] + a[12] * inv[3]);
    for (int k = 0; k < 16; ++k) inv[k] *= recDet;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j)
            out[4 * i + j] = prev[4 * i + 0] * inv[0 + j] + prev[4 * i + 1] * inv[4 + j] + prev[4 * i + 2] * inv[8 + j] + prev[4 * i + 3] * inv[12 + j];
}

// transform, curToPrevTransform, normal matrix and uniform scale of a DevInstance from the host instance
static void fill_instance_transform(const HostInstance& hi, DevInstance& d) {
    std::memcpy(d.transform, hi.transform, sizeof(float) * 12);
    // curToPrevTransform: identity for static instances (common_host.cpp:2631), prev * invert(cur) once moved
    const float ident[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    std::memcpy(d.curToPrevTransform, hi.animated ? hi.curToPrev : ident, sizeof(ident));
    if (hi.hasNormalMatrix)
        for (int rr = 0; rr < 3; ++rr) {
            for (int cc = 0; cc < 3; ++cc) d.normalMatrix[rr * 4 + cc] = hi.normalMatrix[rr * 3 + cc];
            d.normalMatrix[rr * 4 + 3] = 0.0f;
        }
    else normal_matrix(hi.transform, d.normalMatrix);
    d.uniformScale = std::sqrt(hi.transform[0] * hi.transform[0] + hi.transform[4] * hi.transform[4] + hi.transform[8] * hi.transform[8]);
}

// Transform-only update (gfx_instance_set_transform on instances that already live in the animated subtree): the
// pools, the distributions' layout and the static subtree stay; only the moved DevInstance entries go to the device.
// The emitter records (world-space triangles) are refreshed by the next lights_build_instances.
// The normal matrices of the emitter instances, deduplicated by bit pattern (instances of one prototype placed by translation
// share theirs): EmitterRec::flags carries the index, light_from_record reads the rows.  A table of few entries stays in L1 /
// LDS; the rows themselves are the ones fill_instance_transform computed, so nothing changes numerically.
void light_matrices_upload(Context& ctx, hipStream_t stream) {
    std::map<std::array<uint32_t, 9>, uint32_t> seen;
    std::vector<float> rows;              // 16 floats per matrix: three rows padded to float4 + one zero quarter (64-byte items)
    std::vector<uint32_t> index(ctx.hInsts.size(), 0u);
    for (size_t ii = 0; ii < ctx.hInsts.size(); ++ii) {
        const DevInstance& d = ctx.hInsts[ii];
        if (d.distOffset == 0xFFFFFFFFu) continue;
        std::array<uint32_t, 9> key;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) std::memcpy(&key[3 * r + c], &d.normalMatrix[4 * r + c], 4);
        auto it = seen.find(key);
        if (it == seen.end()) {
            it = seen.emplace(key, static_cast<uint32_t>(seen.size())).first;
            for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) rows.push_back(d.normalMatrix[4 * r + c]); rows.push_back(0.0f); }
            rows.insert(rows.end(), 4, 0.0f);
        }
        index[ii] = it->second;
    }
    if (seen.size() > kEmitterMatrixMask + 1u)
        throw HipError("gfx: more than 65536 distinct normal matrices among the emitter instances");
    ctx.numLightMatrices = static_cast<uint32_t>(seen.size());
    // (a little slack: an animated emitter that gains a distinct matrix must not make reserve() free a buffer kernels in flight read)
    ctx.dLightNormalMatrices.reserve(std::max<size_t>(rows.size() * sizeof(float) + 64 * 64, 64));
    ctx.dInstMatrixIndex.reserve(std::max<size_t>(index.size() * sizeof(uint32_t), 16));
    const size_t rowBytes = rows.size() * sizeof(float), indexBytes = index.size() * sizeof(uint32_t);
    if (rowBytes + indexBytes == 0) return;
    Context::PinnedStage& st = ctx.lightStage[ctx.lightStageNext];
    ctx.lightStageNext ^= 1u;
    if (st.done) GFX_HIP(hipEventSynchronize(st.done));     // the copy issued out of this buffer two uploads ago: long finished
    else GFX_HIP(hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    if (st.bytes < rowBytes + indexBytes) {
        if (st.p) GFX_HIP(hipHostFree(st.p));
        st.p = nullptr; st.bytes = 0;
        GFX_HIP(hipHostMalloc(&st.p, (rowBytes + indexBytes) * 2, hipHostMallocDefault));
        st.bytes = (rowBytes + indexBytes) * 2;
    }
    char* base = static_cast<char*>(st.p);
    if (rowBytes) { std::memcpy(base, rows.data(), rowBytes); GFX_HIP(hipMemcpyAsync(ctx.dLightNormalMatrices.p, base, rowBytes, hipMemcpyHostToDevice, stream)); }
    if (indexBytes) { std::memcpy(base + rowBytes, index.data(), indexBytes); GFX_HIP(hipMemcpyAsync(ctx.dInstMatrixIndex.p, base + rowBytes, indexBytes, hipMemcpyHostToDevice, stream)); }
    GFX_HIP(hipEventRecord(st.done, stream));
}

void transforms_upload(Context& ctx, hipStream_t stream) {
    if (!ctx.transformsDirty) return;
    for (uint32_t slot : ctx.movedInsts) {
        DevInstance& d = ctx.hInsts[slot];
        fill_instance_transform(ctx.insts[slot], d);
        GFX_HIP(hipMemcpyAsync(ctx.dInsts.as<DevInstance>() + slot, &d, sizeof(DevInstance), hipMemcpyHostToDevice, stream));
    }
    ctx.movedInsts.clear();
    ctx.transformsDirty = false;
    ctx.emitterRecsDirty = true;
    ctx.instDistValid = false;
}

template <typename T>
static void upload(DevBuf& b, const std::vector<T>& v, hipStream_t stream) {
    b.reserve(std::max<size_t>(sizeof(T) * v.size(), 16));
    if (!v.empty()) GFX_HIP(hipMemcpyAsync(b.p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, stream));
}

void scene_upload(Context& ctx, hipStream_t stream) {
    if (!ctx.sceneDirty) return;
    // pools
    std::vector<DevVertex> vertices;
    std::vector<uint32_t> triangles;
    std::vector<uint32_t> slotPool;
    ctx.hGeomInsts.assign(ctx.geoms.size(), DevGeomInst());
    uint32_t lightPool = 0;
    for (size_t gi = 0; gi < ctx.geoms.size(); ++gi) {
        const HostGeom& g = ctx.geoms[gi];
        DevGeomInst& d = ctx.hGeomInsts[gi];
        d.vertexOffset = static_cast<uint32_t>(vertices.size());
        d.triangleOffset = static_cast<uint32_t>(triangles.size() / 3);
        d.numVertices = static_cast<uint32_t>(g.vertices.size());
        d.numTriangles = static_cast<uint32_t>(g.triangles.size() / 3);
        if (g.materialSlot >= ctx.materials.size())
            throw std::runtime_error("gfx: a geometry refers to material slot " + std::to_string(g.materialSlot) + " but only " +
                                     std::to_string(ctx.materials.size()) + " materials are set (gfx_material_set)");
        d.materialSlot = g.materialSlot;
        const bool emitter = g.materialSlot < ctx.materials.size() && ctx.materials[g.materialSlot].hasEmittance;
        d.distOffset = emitter ? lightPool : 0xFFFFFFFFu;   // only emitters own a distribution
        d.distCount = emitter ? d.numTriangles : 0;
        d.distIntegral = 0;
        if (emitter) lightPool += d.numTriangles;
        vertices.insert(vertices.end(), g.vertices.begin(), g.vertices.end());
        triangles.insert(triangles.end(), g.triangles.begin(), g.triangles.end());
    }
    ctx.hInsts.assign(ctx.insts.size(), DevInstance());
    ctx.hFlatGeoms.clear();
    ctx.hSubset[0].clear(); ctx.hSubset[1].clear();
    ctx.subsetTris[0] = ctx.subsetTris[1] = 0;
    uint32_t triCursor = 0;
    for (size_t ii = 0; ii < ctx.insts.size(); ++ii) {
        const HostInstance& hi = ctx.insts[ii];
        DevInstance& d = ctx.hInsts[ii];
        fill_instance_transform(hi, d);
        const std::vector<uint32_t>& slots = ctx.groups[hi.group];
        d.slotsOffset = static_cast<uint32_t>(slotPool.size());
        d.numGeomInsts = static_cast<uint32_t>(slots.size());
        bool hasEmitter = false;
        for (uint32_t s : slots) {
            slotPool.push_back(s);
            if (ctx.hGeomInsts[s].distOffset != 0xFFFFFFFFu) hasEmitter = true;
            DevFlatGeom fg; fg.instSlot = static_cast<uint32_t>(ii); fg.geomInstSlot = s; fg.triBegin = triCursor;
            fg.numTriangles = ctx.hGeomInsts[s].numTriangles;
            triCursor += fg.numTriangles;
            ctx.hFlatGeoms.push_back(fg);
            const int sub = hi.dynamic ? 1 : 0;
            ctx.hSubset[sub].push_back(SubsetGeom{ fg.instSlot, fg.geomInstSlot, ctx.subsetTris[sub], fg.triBegin });
            ctx.subsetTris[sub] += fg.numTriangles;
        }
        d.distOffset = hasEmitter ? lightPool : 0xFFFFFFFFu;
        d.distIntegral = 0;
        if (hasEmitter) lightPool += d.numGeomInsts;
        d.pad[0] = d.pad[1] = d.pad[2] = 0;
    }
    ctx.totalTriangles = triCursor;
    // per (instance, geomInst) light references + pre-transformed emitter triangle records
    ctx.hLightRefs.assign(lightPool + ctx.insts.size(), LightGeomRef{ 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0.0f });
    uint32_t recCursor = 0;
    for (size_t ii = 0; ii < ctx.insts.size(); ++ii) {
        const DevInstance& d = ctx.hInsts[ii];
        if (d.distOffset == 0xFFFFFFFFu) continue;
        for (uint32_t k = 0; k < d.numGeomInsts; ++k) {
            const DevGeomInst& g = ctx.hGeomInsts[slotPool[d.slotsOffset + k]];
            LightGeomRef& r = ctx.hLightRefs[d.distOffset + k];
            r.distOffset = g.distOffset; r.distCount = g.distCount; r.distIntegral = 0.0f;
            if (g.distOffset != 0xFFFFFFFFu) { r.recBase = recCursor; recCursor += g.numTriangles; }
        }
    }
    ctx.numEmitterRecs = recCursor;
    ctx.lightInstDistOffset = lightPool;
    lightPool += static_cast<uint32_t>(ctx.insts.size());
    ctx.lightPoolSize = lightPool;
    ctx.lightsStaticBuilt = false;
    ctx.dLightInstIntegral.reserve(16);
    {   // guide table: about one cell per instance, power of two, small enough to sit in LDS next to the CDF
        uint32_t cells = 256;
        while (cells < ctx.insts.size() && cells < 8192) cells *= 2;
        ctx.lightInstGuideCells = cells;
        ctx.dLightInstGuide.reserve(sizeof(uint16_t) * cells);
    }
    GFX_HIP(hipMemsetAsync(ctx.dLightInstIntegral.p, 0, 16, stream));
    {   // emitter interval table: GFX_SPAN_CELLS_PER_REC guide cells per record, rounded up to a power of two (ul * cells is exact);
        // finer cells = more of the ul range in interior cells (one load per lookup), bigger table
        uint32_t perRec = GFX_SPAN_CELLS_PER_REC;
        if (const char* e = std::getenv("GFX_SPAN_CELLS_PER_REC")) { const long v = std::atol(e); if (v >= 1 && v <= 64) perRec = static_cast<uint32_t>(v); }
        uint32_t cells = 256;
        while (cells < static_cast<uint64_t>(perRec) * ctx.numEmitterRecs && cells < (1u << 22)) cells *= 2;
        ctx.spanGuideCells = cells;
        ctx.dSpans.reserve(std::max<size_t>(sizeof(EmitterSpan) * ctx.numEmitterRecs, 16));
        ctx.dSpanGuide.reserve(sizeof(SpanGuide) * cells);
        ctx.dSpanHeader.reserve(16);
        ctx.dSpanInstBegin.reserve(sizeof(uint32_t) * (ctx.insts.size() + 1));
        GFX_HIP(hipMemsetAsync(ctx.dSpanHeader.p, 0, 16, stream));
    }

    {   // textures: descriptor table + one texel pool (every texture 16-byte aligned) + the sRGB decode table
        std::vector<DevTexture> descs(std::max<size_t>(ctx.textures.size(), 1));
        std::vector<uint32_t> pool;
        for (size_t t = 0; t < ctx.textures.size(); ++t) {
            const HostTexture& ht = ctx.textures[t];
            DevTexture& d = descs[t];
            d.offset = static_cast<uint32_t>(pool.size()); d.width = ht.width; d.height = ht.height; d.format = ht.format;
            const size_t words = (ht.texels.size() + 3) / 4;
            const size_t at = pool.size();
            pool.resize(at + ((words + 3) & ~size_t(3)), 0u);
            if (!ht.texels.empty()) std::memcpy(pool.data() + at, ht.texels.data(), ht.texels.size());
        }
        if (pool.empty()) pool.resize(4, 0u);
        if (pool.size() >= (1ull << 32)) throw HipError("gfx: texel pool exceeds 16 GiB");
        float lut[256];
        for (int c = 0; c < 256; ++c) {   // sampler_sRGB: degamma of c / 255 (basic_types.h:5396-5402), fp32
            const float v = static_cast<float>(c) / 255.0f;
            lut[c] = v <= 0.04045f ? v / 12.92f : std::pow((v + 0.055f) / 1.055f, 2.4f);
        }
        upload(ctx.dTextures, descs, stream);
        upload(ctx.dTexelPool, pool, stream);
        ctx.dSrgbLut.reserve(sizeof(lut));
        GFX_HIP(hipMemcpyAsync(ctx.dSrgbLut.p, lut, sizeof(lut), hipMemcpyHostToDevice, stream));
        GFX_HIP(hipStreamSynchronize(stream));
        for (const gfx_material& m : ctx.materials) {
            const uint32_t slots[5] = { m.texA, m.texB, m.texSmoothness, m.texNormal, m.texEmittance };
            for (uint32_t sl : slots)
                if (sl != 0 && (sl >= ctx.textures.size() || ctx.textures[sl].width == 0)) throw HipError("gfx: material references a texture slot that was never set");
            if (m.hasEmittance && m.texEmittance > kEmitterTexMask) throw HipError("gfx: an emittance texture must sit in a texture slot below 32768");
        }
        ctx.anyEmittanceTexture = false;
        for (const HostGeom& g : ctx.geoms)
            if (g.materialSlot < ctx.materials.size() && ctx.materials[g.materialSlot].hasEmittance && ctx.materials[g.materialSlot].texEmittance != 0) ctx.anyEmittanceTexture = true;
    }
    upload(ctx.dMaterials, ctx.materials, stream);
    upload(ctx.dGeomInsts, ctx.hGeomInsts, stream);
    upload(ctx.dInsts, ctx.hInsts, stream);
    upload(ctx.dVertices, vertices, stream);
    upload(ctx.dTriangles, triangles, stream);
    upload(ctx.dSlotPool, slotPool, stream);
    upload(ctx.dFlatGeoms, ctx.hFlatGeoms, stream);
    upload(ctx.dSubset[0], ctx.hSubset[0], stream);
    upload(ctx.dSubset[1], ctx.hSubset[1], stream);
    ctx.transformsDirty = false; ctx.movedInsts.clear(); ctx.emitterRecsDirty = false; ctx.instDistValid = false;
    upload(ctx.dLightRefs, ctx.hLightRefs, stream);
    ctx.dEmitterRecs.reserve(std::max<size_t>(sizeof(EmitterRec) * ctx.numEmitterRecs, 16));
    ctx.dEmitterRecExtras.reserve(std::max<size_t>(sizeof(EmitterRecExtra) * ctx.numEmitterRecs, 16));
    light_matrices_upload(ctx, stream);
    ctx.dEmitterTexRefs.reserve(std::max<size_t>(ctx.anyEmittanceTexture ? sizeof(EmitterTexRef) * ctx.numEmitterRecs : 0, 16));
    ctx.dLightW.reserve(std::max<size_t>(sizeof(float) * lightPool, 16));
    ctx.dLightCDF.reserve(std::max<size_t>(sizeof(float) * lightPool, 16));
    ctx.dLightP.reserve(std::max<size_t>(sizeof(float) * lightPool, 16));
    GFX_HIP(hipMemsetAsync(ctx.dLightW.p, 0, ctx.dLightW.bytes, stream));
    GFX_HIP(hipMemsetAsync(ctx.dLightP.p, 0, ctx.dLightP.bytes, stream));
    GFX_HIP(hipMemsetAsync(ctx.dLightCDF.p, 0, ctx.dLightCDF.bytes, stream));
    GFX_HIP(hipStreamSynchronize(stream));   // host vectors above go out of scope
    ctx.sceneDirty = false;
}

} // namespace gfx
