// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// orc_capi.cpp: extern "C" surface of the CPU restatement, loaded with ctypes by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg.  It mirrors include/gfxexp.h entry for
// entry (same structs, HOST pointers instead of device pointers) so that a parity test reads
// "run the product, run the oracle, compare buffers".
#include <omp.h>
#include <chrono>
#include <memory>
#include <random>
#include <string>
#include "orc_restir.h"
#include "orc_pathtrace.h"
#include "orc_restir_rearch.h"
#include "orc_nrc.h"

using namespace orc;

struct orc_scene {
    Scene scene;
    std::vector<std::vector<uint32_t>> groups;
    WorldAccel accel;
    std::vector<std::vector<uint8_t>> geomVertexBytes; // keeps bvh::Geometry pointers alive
    int numThreads = 1;
};

static M34 toM34(const float x[12]) { M34 m; for (int i = 0; i < 12; ++i) m.m[i] = x[i]; return m; }
static M34 identity34() { const float x[12] = { 1,0,0,0, 0,1,0,0, 0,0,1,0 }; return toM34(x); }

extern "C" {

orc_scene* orc_scene_create() { return new orc_scene(); }
void orc_scene_destroy(orc_scene* s) { delete s; }
void orc_set_num_threads(orc_scene* s, int n) { s->numThreads = n < 1 ? 1 : n; }
int orc_max_threads() { return omp_get_max_threads(); }

int orc_material_set(orc_scene* s, uint32_t slot, const gfx_material* m) {
    if (s->scene.materials.size() <= slot) s->scene.materials.resize(slot + 1);
    MaterialData& d = s->scene.materials[slot];
    d.bsdfType = m->bsdfType;
    for (int i = 0; i < 3; ++i) { d.a[i] = m->a[i]; d.b[i] = m->b[i]; d.emittance[i] = m->emittance[i]; }
    d.smoothness = m->smoothness;
    d.hasEmittance = m->hasEmittance;
    d.texA = m->texA; d.texB = m->texB; d.texSmoothness = m->texSmoothness; d.texNormal = m->texNormal; d.texEmittance = m->texEmittance;
    d.bumpMapType = m->bumpMapType;
    return 0;
}

int orc_texture_set(orc_scene* s, uint32_t slot, uint32_t width, uint32_t height, uint32_t format, const void* texels) {
    if (slot == 0) return 1;
    if (s->scene.textures.size() <= slot) s->scene.textures.resize(slot + 1);
    Texture& t = s->scene.textures[slot];
    const size_t bpp = format == TexRGBA32F ? 16 : format == TexR8_UNorm ? 1 : format == TexRG8_UNorm ? 2 : 4;
    t.width = width; t.height = height; t.format = format; t.lutReady = false;
    t.texels.assign(static_cast<const uint8_t*>(texels), static_cast<const uint8_t*>(texels) + bpp * width * height);
    return 0;
}

// tex2DLod / tex2Dgather of one texture at n coordinates (tests of the sampler contract)
void orc_texture_sample(orc_scene* s, uint32_t slot, const float* uv, uint32_t n, float* out4, int gather) {
    const Texture& t = s->scene.textures[slot];
    for (uint32_t i = 0; i < n; ++i) {
        const Texel4 r = gather ? t.gatherR(uv[2 * i], uv[2 * i + 1]) : t.sample(uv[2 * i], uv[2 * i + 1]);
        out4[4 * i] = r.x; out4[4 * i + 1] = r.y; out4[4 * i + 2] = r.z; out4[4 * i + 3] = r.w;
    }
}

int orc_geom_create(orc_scene* s, const void* vertices, uint32_t stride, uint32_t numVertices,
                    const uint32_t* triangles, uint32_t numTriangles, uint32_t matSlot, uint32_t* slotOut) {
    GeometryInstanceData g;
    g.vertexBuffer.resize(numVertices);
    for (uint32_t i = 0; i < numVertices; ++i)
        std::memcpy(&g.vertexBuffer[i], static_cast<const uint8_t*>(vertices) + static_cast<size_t>(stride) * i, sizeof(Vertex));
    g.triangleBuffer.resize(numTriangles);
    std::memcpy(g.triangleBuffer.data(), triangles, sizeof(Triangle) * numTriangles);
    g.materialSlot = matSlot;
    g.geomInstSlot = static_cast<uint32_t>(s->scene.geomInsts.size());
    *slotOut = g.geomInstSlot;
    s->scene.geomInsts.push_back(std::move(g));
    return 0;
}

int orc_group_create(orc_scene* s, const uint32_t* slots, uint32_t n, uint32_t* group) {
    *group = static_cast<uint32_t>(s->groups.size());
    s->groups.emplace_back(slots, slots + n);
    return 0;
}

// common/common_host.cpp:2582-2656 createInstance
int orc_instance_create(orc_scene* s, uint32_t group, const float xfm[12], uint32_t* instSlot) {
    InstanceData inst;
    inst.transform = toM34(xfm);
    inst.curToPrevTransform = identity34();
    inst.normalMatrix = transpose(invert(upperLeft(inst.transform)));
    inst.uniformScale = length(V3(xfm[0], xfm[4], xfm[8]));
    inst.geomInstSlots = s->groups[group];
    *instSlot = static_cast<uint32_t>(s->scene.insts.size());
    s->scene.insts.push_back(std::move(inst));
    return 0;
}

// common/common_host.h:837-855 InstanceController::update, the InstanceData part: the new object-to-world
// matrix replaces the old one, curToPrevTransform = prevTransform * invert(matM2W).  normalMatrix9: the
// controller's matRot / curScale (row-major 3x3) or NULL for transpose(invert(upper-left 3x3)) as at creation.
// orc_scene_commit must run again afterwards (updateASs + the light distributions, restir_di_main.cpp:2263-2264).
int orc_instance_set_transform(orc_scene* s, uint32_t instSlot, const float xfm[12], const float* normalMatrix9) {
    if (instSlot >= s->scene.insts.size()) return 1;
    InstanceData& inst = s->scene.insts[instSlot];
    const M34 cur = toM34(xfm);
    inst.curToPrevTransform = curToPrev(inst.transform, cur);
    inst.transform = cur;
    if (normalMatrix9) {
        inst.normalMatrix.r0 = V3(normalMatrix9[0], normalMatrix9[1], normalMatrix9[2]);
        inst.normalMatrix.r1 = V3(normalMatrix9[3], normalMatrix9[4], normalMatrix9[5]);
        inst.normalMatrix.r2 = V3(normalMatrix9[6], normalMatrix9[7], normalMatrix9[8]);
    }
    else inst.normalMatrix = transpose(invert(upperLeft(inst.transform)));
    inst.uniformScale = length(V3(xfm[0], xfm[4], xfm[8]));
    return 0;
}

// Build the emitter distributions and the world-space BVH (SAH builder restatement).
// config: {splittingBudget, intNodeTravCost, primIntersectCost, minLeaf, maxLeaf} or NULL for the
// nrtdsm defaults {0.3, 1.2, 1.0, 1, 128} (nrtdsm/nrtdsm_main.cpp:811-816).
int orc_scene_commit(orc_scene* s, int useBruteForce, const float* config, double* buildSeconds) {
    Scene& sc = s->scene;
    sc.setupLightGeomDistributions();
    sc.setupLightInstDistribution();
    WorldAccel& a = s->accel;
    a.geoms.clear(); a.geomToInst.clear(); a.geomToGeomInst.clear(); a.primOffsets.clear();
    uint32_t off = 0;
    for (uint32_t i = 0; i < sc.insts.size(); ++i)
        for (uint32_t slot : sc.insts[i].geomInstSlots) {
            const GeometryInstanceData& g = sc.geomInsts[slot];
            bvh::Geometry bg;
            bg.vertices = reinterpret_cast<const uint8_t*>(g.vertexBuffer.data());
            bg.vertexStride = sizeof(Vertex);
            bg.numVertices = static_cast<uint32_t>(g.vertexBuffer.size());
            bg.triangles = reinterpret_cast<const uint8_t*>(g.triangleBuffer.data());
            bg.triangleStride = sizeof(Triangle);
            bg.numTriangles = static_cast<uint32_t>(g.triangleBuffer.size());
            bg.preTransform = sc.insts[i].transform;
            a.geoms.push_back(bg);
            a.geomToInst.push_back(i);
            a.geomToGeomInst.push_back(slot);
            a.primOffsets.push_back(off);
            off += bg.numTriangles;
        }
    a.useBruteForce = useBruteForce != 0;
    bvh::BuildConfig cfg;
    if (config) {
        cfg.splittingBudget = config[0]; cfg.intNodeTravCost = config[1]; cfg.primIntersectCost = config[2];
        cfg.minNumPrimsPerLeaf = static_cast<uint32_t>(config[3]); cfg.maxNumPrimsPerLeaf = static_cast<uint32_t>(config[4]);
    }
    const auto t0 = std::chrono::steady_clock::now();
    a.bvh = bvh::GeometryBVH();
    if (!a.geoms.empty())
        bvh::buildGeometryBVH(a.geoms.data(), static_cast<uint32_t>(a.geoms.size()), cfg, &a.bvh);
    const auto t1 = std::chrono::steady_clock::now();
    if (buildSeconds) *buildSeconds = std::chrono::duration<double>(t1 - t0).count();
    return 0;
}

// {numTriangles, numIntNodes, numPrimRefs, 0}
int orc_accel_stats(orc_scene* s, uint32_t stats[4]) {
    stats[0] = s->accel.bvh.totalNumPrims;
    stats[1] = static_cast<uint32_t>(s->accel.bvh.intNodes.size());
    stats[2] = static_cast<uint32_t>(s->accel.bvh.primRefs.size());
    stats[3] = 0;
    return 0;
}

// Node invariants used by tests: every child box (dequantised) contains the boxes of the
// triangles below it; every triangle is referenced at least once; leaf chains end in isLeafEnd.
// Returns 0 when all hold, else a bitmask of failures.
int orc_accel_validate(orc_scene* s) {
    const bvh::GeometryBVH& b = s->accel.bvh;
    if (b.intNodes.empty()) return 0;
    int fail = 0;
    std::vector<uint32_t> refCount(b.triStorages.size(), 0);
    struct Item { uint32_t node; bvh::AABB bound; bool hasBound; };
    std::vector<Item> st; st.push_back({ 0, bvh::AABB(), false });
    while (!st.empty()) {
        const Item it = st.back(); st.pop_back();
        const bvh::InternalNode& n = b.intNodes[it.node];
        for (uint32_t slot = 0; slot < bvh::arity; ++slot) {
            if (!n.getChildIsValid(slot)) break;
            const bvh::AABB cb = n.getChildAabb(slot);
            if (!n.getChildIsLeaf(slot)) {
                st.push_back({ n.intNodeChildBaseIndex + n.getInternalChildNumber(slot), cb, true });
                continue;
            }
            uint32_t idx = n.leafBaseIndex + n.childMetas[slot];
            uint32_t guard = 0;
            while (true) {
                if (idx >= b.primRefs.size()) { fail |= 4; break; }
                const bvh::PrimitiveReference pr = b.primRefs[idx];
                ++refCount[pr.storageIndex];
                // with spatial splits a reference covers only the clipped part of its triangle, so
                // the containment check is on the overlap: the child box must intersect the triangle box.
                const bvh::TriangleStorage& ts = b.triStorages[pr.storageIndex];
                bvh::AABB tb; tb.unify(ts.pA).unify(ts.pB).unify(ts.pC);
                bvh::AABB ov = bvh::intersect(tb, cb);
                if (!ov.isValid()) fail |= 1;
                if (pr.isLeafEnd) break;
                ++idx;
                if (++guard > 100000) { fail |= 4; break; }
            }
        }
    }
    for (uint32_t c : refCount) if (c == 0) { fail |= 2; break; }
    return fail;
}

int orc_lights_read(orc_scene* s, uint32_t level, uint32_t index, float* weights, float* cdf,
                    uint32_t capacity, uint32_t* n, float* integral) {
    const std::vector<float>* w; const std::vector<float>* c; const DiscreteDistribution1D* d;
    if (level == 0) { w = &s->scene.lightInstWeights; c = &s->scene.lightInstCDF; d = &s->scene.lightInstDist; }
    else if (level == 1) { const InstanceData& i = s->scene.insts[index]; w = &i.lightGeomInstWeights; c = &i.lightGeomInstCDF; d = &i.lightGeomInstDist; }
    else { const GeometryInstanceData& g = s->scene.geomInsts[index]; w = &g.emitterPrimWeights; c = &g.emitterPrimCDF; d = &g.emitterPrimDist; }
    *n = static_cast<uint32_t>(w->size());
    *integral = d->integral();
    const uint32_t m = std::min<uint32_t>(capacity, *n);
    if (weights) std::memcpy(weights, w->data(), sizeof(float) * m);
    if (cdf) std::memcpy(cdf, c->data(), sizeof(float) * m);
    return 0;
}

// mode 0: closest (reference traversal + canonical tie-break) -> gfx_hit, triIndex = flattened
//         triangle index (enumeration order of orc_scene_commit)
// mode 1: any hit -> uint32 occluded flags
// mode 2: closest by brute force -> gfx_hit
// mode 3: closest, reference traversal verbatim (no tie canonicalisation) -> gfx_hit
// stats (optional): u64[4] = {node fetches, triangle tests, rays, child box tests}
int orc_trace(orc_scene* s, int mode, const float* rayOrgTmin, const float* rayDirTmax, uint32_t numRays,
              void* out, uint64_t* stats) {
    const WorldAccel& a = s->accel;
    uint64_t nodeFetches = 0, triTests = 0, boxTests = 0;
#pragma omp parallel for schedule(dynamic, 256) num_threads(s->numThreads) reduction(+ : nodeFetches, triTests, boxTests)
    for (int64_t i = 0; i < static_cast<int64_t>(numRays); ++i) {
        const V3 o(rayOrgTmin[4 * i + 0], rayOrgTmin[4 * i + 1], rayOrgTmin[4 * i + 2]);
        const float tmin = rayOrgTmin[4 * i + 3];
        const V3 d(rayDirTmax[4 * i + 0], rayDirTmax[4 * i + 1], rayDirTmax[4 * i + 2]);
        const float tmax = rayDirTmax[4 * i + 3];
        if (mode == 1) {
            static_cast<uint32_t*>(out)[i] = occluded(a, o, d, tmin, tmax) ? 1u : 0u;
            continue;
        }
        bvh::HitObject h;
        if (mode == 2) h = bvh::bruteForce(a.bvh.triStorages, o, d, tmin, tmax, false);
        else if (mode == 3) {
            bvh::TraversalStatistics st;
            h = bvh::traverse(a.bvh, o, d, tmin, tmax, &st);
            nodeFetches += st.numNodeFetches; triTests += st.numTriTests; boxTests += st.numAabbTests;
        }
        else h = closestHitCanonical(a, o, d, tmin, tmax);
        gfx_hit gh;
        gh.dist = h.dist; gh.bcB = h.bcB; gh.bcC = h.bcC;
        gh.triIndex = h.isHit() ? a.primOffsets[h.geomIndex] + h.primIndex : 0xFFFFFFFFu;
        if (!h.isHit()) { gh.bcB = 0; gh.bcC = 0; }
        static_cast<gfx_hit*>(out)[i] = gh;
    }
    if (stats) { stats[0] = nodeFetches; stats[1] = triTests; stats[2] = numRays; stats[3] = boxTests; }
    return 0;
}

// flattened triangle index -> (instSlot, geomInstSlot, primIndex)
int orc_tri_ids(orc_scene* s, gfx_tri_ids* ids, uint32_t capacity, uint32_t* count) {
    const WorldAccel& a = s->accel;
    uint32_t n = 0;
    for (size_t g = 0; g < a.geoms.size(); ++g)
        for (uint32_t p = 0; p < a.geoms[g].numTriangles; ++p, ++n)
            if (n < capacity) { ids[n].instSlot = a.geomToInst[g]; ids[n].geomInstSlot = a.geomToGeomInst[g]; ids[n].primIndex = p; }
    *count = n;
    return 0;
}

// World-space triangle soup {pA,pB,pC} x n in flattened order (for brute-force cross-checks).
int orc_world_triangles(orc_scene* s, float* out9, uint32_t capacity, uint32_t* count) {
    const auto& ts = s->accel.bvh.triStorages;
    *count = static_cast<uint32_t>(ts.size());
    for (uint32_t i = 0; i < ts.size() && i < capacity; ++i) {
        const float v[9] = { ts[i].pA.x, ts[i].pA.y, ts[i].pA.z, ts[i].pB.x, ts[i].pB.y, ts[i].pB.z, ts[i].pC.x, ts[i].pC.y, ts[i].pC.z };
        std::memcpy(out9 + 9 * i, v, sizeof(v));
    }
    return 0;
}

int orc_env_set(orc_scene* s, const gfx_restir_static_params* sp) {
    EnvLight& e = s->scene.env;
    e = EnvLight();
    if (!sp->envLightTexture) return 0;
    e.texels = static_cast<const float*>(sp->envLightTexture);
    e.w = static_cast<uint32_t>(sp->envWidth); e.h = static_cast<uint32_t>(sp->envHeight);
    e.importanceMap.rowPDF = static_cast<const float*>(sp->envRowPDF);
    e.importanceMap.rowCDF = static_cast<const float*>(sp->envRowCDF);
    e.importanceMap.rowIntegrals = static_cast<const float*>(sp->envRowIntegrals);
    e.importanceMap.w = e.w; e.importanceMap.h = e.h;
    e.importanceMap.top.PDF = static_cast<const float*>(sp->envTopPDF);
    e.importanceMap.top.CDF = static_cast<const float*>(sp->envTopCDF);
    e.importanceMap.top.integralValue = sp->envTopIntegral;
    e.importanceMap.top.numValues = e.h;
    return 0;
}

// gfx_restir_set_params + gfx_restir_launch in one call (host pointers inside `sp`).
int orc_restir_launch(orc_scene* s, const gfx_restir_static_params* sp, const gfx_restir_frame_params* fp,
                      uint32_t currentReservoirIndex, uint32_t spatialNeighborBaseIndex, int pass,
                      int x0, int y0, int x1, int y1) {
    if (pass == GFX_RESTIR_SPATIAL_BIASED_AND_SHADING) {   // include/gfxexp.h: the two passes back to back (restir_di_main.cpp:2393-2420)
        if (int rc = orc_restir_launch(s, sp, fp, currentReservoirIndex, spatialNeighborBaseIndex, GFX_RESTIR_SPATIAL_BIASED, x0, y0, x1, y1)) return rc;
        return orc_restir_launch(s, sp, fp, currentReservoirIndex + 1, spatialNeighborBaseIndex, GFX_RESTIR_SHADING, x0, y0, x1, y1);
    }
    orc_env_set(s, sp);
    Params p;
    p.scene = &s->scene; p.accel = &s->accel; p.s = sp; p.f = fp;
    p.currentReservoirIndex = currentReservoirIndex & 1u;
    p.spatialNeighborBaseIndex = spatialNeighborBaseIndex & 1023u; // 10-bit bitfield, restir_di_shared.h:287
    p.camera = toCamera(fp->camera);
    p.prevCamera = toCamera(fp->prevCamera);
    if (x1 <= 0) x1 = sp->imageSizeX;
    if (y1 <= 0) y1 = sp->imageSizeY;
    if (pass == GFX_RESTIR_LIGHT_PRESAMPLING) {
#pragma omp parallel for schedule(static) num_threads(s->numThreads)
        for (int i = 0; i < static_cast<int>(kNumLightSubsets * kLightSubsetSize); ++i) lightPreSamplingThread(p, static_cast<uint32_t>(i));
        return 0;
    }
    if (pass == GFX_RESTIR_PER_PIXEL_RIS) {
        const int tx0 = x0 / kTileSizeX, ty0 = y0 / kTileSizeY;
        const int tx1 = (x1 + kTileSizeX - 1) / kTileSizeX, ty1 = (y1 + kTileSizeY - 1) / kTileSizeY;
#pragma omp parallel for schedule(dynamic, 1) num_threads(s->numThreads)
        for (int ty = ty0; ty < ty1; ++ty)
            for (int tx = tx0; tx < tx1; ++tx) perPixelRISTile(p, tx, ty);
        return 0;
    }
    if (pass >= GFX_RESTIR_TRACE_SHADOW_RAYS && pass <= GFX_RESTIR_SHADE_AND_RESAMPLE_SPATIOTEMPORAL) {
        // RearchitectedReSTIREntryPoint order (restir_di_main.cpp:83-95)
        static const bool kT[11] = { false, true, false, true, true, false, true, false, true, false, true };
        static const bool kS[11] = { false, false, true, true, false, true, true, false, false, true, true };
        static const bool kU[11] = { false, false, false, false, true, true, true, false, false, false, false };
        const int e = pass - GFX_RESTIR_TRACE_SHADOW_RAYS;
#pragma omp parallel for schedule(dynamic, 4) num_threads(s->numThreads)
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                if (e < 7) traceShadowRaysPixel(p, kT[e], kS[e], kU[e], x, y);
                else shadeAndResamplePixel(p, kT[e], kS[e], x, y);
            }
        return 0;
    }
#pragma omp parallel for schedule(dynamic, 4) num_threads(s->numThreads)
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            switch (pass) {
            case GFX_RESTIR_SETUP_GBUFFERS: setupGBuffersPixel(p, x, y); break;
            case GFX_RESTIR_INITIAL_RIS: initialAndTemporalRISPixel(p, false, false, x, y); break;
            case GFX_RESTIR_INITIAL_AND_TEMPORAL_BIASED: initialAndTemporalRISPixel(p, true, false, x, y); break;
            case GFX_RESTIR_INITIAL_AND_TEMPORAL_UNBIASED: initialAndTemporalRISPixel(p, true, true, x, y); break;
            case GFX_RESTIR_SPATIAL_BIASED: spatialRISPixel(p, false, x, y); break;
            case GFX_RESTIR_SPATIAL_UNBIASED: spatialRISPixel(p, true, x, y); break;
            case GFX_RESTIR_SHADING: shadingPixel(p, x, y); break;
            default: break;
            }
        }
    return 0;
}

// Path tracers (path_tracing/path_tracing_main.cpp:2068-2093, regir/regir_main.cpp:2021-2066).
// pass = gfx_pt_pass.
static gfx_regir_params g_regirParams;
static bool g_regirValid = false;
int orc_regir_set_params(orc_scene*, const gfx_regir_params* p) { g_regirParams = *p; g_regirValid = true; return 0; }

// reservoir index of the ReSTIR passes that ran before GFX_PT_PATH_TRACE_NRC_RESTIR (the C ABI takes it through gfx_restir_set_params)
static uint32_t g_ptReservoirIndex = 0;
int orc_pt_set_reservoir_index(orc_scene*, uint32_t index) { g_ptReservoirIndex = index & 1u; return 0; }

static gfx_nrc_params g_nrcParams;
static bool g_nrcValid = false;
int orc_nrc_set_render_params(orc_scene*, const gfx_nrc_params* p) { g_nrcParams = *p; g_nrcValid = true; return 0; }

int orc_pt_launch(orc_scene* s, const gfx_restir_static_params* sp, const gfx_restir_frame_params* fp,
                  int pass, uint32_t maxPathLength, int x0, int y0, int x1, int y1) {
    orc_env_set(s, sp);
    if (pass == GFX_PT_SETUP_GBUFFERS)
        return orc_restir_launch(s, sp, fp, 0, 0, GFX_RESTIR_SETUP_GBUFFERS, x0, y0, x1, y1);
    RegirState rs; rs.g = &g_regirParams;
    Params rp; rp.scene = &s->scene; rp.accel = &s->accel; rp.s = sp; rp.f = fp;
    rp.currentReservoirIndex = 0; rp.spatialNeighborBaseIndex = 0;
    rp.camera = toCamera(fp->camera); rp.prevCamera = toCamera(fp->prevCamera);
    if (pass == GFX_PT_REGIR_BUILD_CELL_RESERVOIRS || pass == GFX_PT_REGIR_BUILD_CELL_RESERVOIRS_TEMPORAL) {
        if (!g_regirValid) return 1;
        const uint32_t numCells = rs.numCells();
        *static_cast<uint32_t*>(g_regirParams.numActiveCells[fp->bufferIndex]) = 0;
        // the kernel zeroes the access counter of every cell before the activity test (:76-78)
        std::memset(g_regirParams.perCellNumAccesses, 0, sizeof(uint32_t) * numCells);
        const bool temporal = pass == GFX_PT_REGIR_BUILD_CELL_RESERVOIRS_TEMPORAL;
#pragma omp parallel for schedule(dynamic, 512) num_threads(s->numThreads)
        for (long long i = 0; i < static_cast<long long>(rs.numLightSlots()); ++i)
            buildCellReservoirThread(rp, rs, temporal, static_cast<uint32_t>(i));
        return 0;
    }
    if (pass == GFX_PT_REGIR_UPDATE_LAST_ACCESS) {
        if (!g_regirValid) return 1;
        const uint32_t numCells = rs.numCells();
        const uint32_t* acc = static_cast<const uint32_t*>(g_regirParams.perCellNumAccesses);
        uint32_t* last = static_cast<uint32_t*>(g_regirParams.lastAccessFrameIndices);
        uint32_t active = 0;
        for (uint32_t c = 0; c < numCells; ++c)
            if (acc[c] > 0) { last[c] = fp->frameIndex; ++active; }
        *static_cast<uint32_t*>(g_regirParams.numActiveCells[fp->bufferIndex]) += active;
        return 0;
    }
    PathTraceParams p;
    p.scene = &s->scene; p.accel = &s->accel; p.s = sp; p.f = fp;
    p.camera = toCamera(fp->camera);
    p.maxPathLength = maxPathLength & 15u; // 4-bit bitfield, path_tracing_shared.h:165
    if ((pass >= GFX_PT_NRC_PREPROCESS && pass <= GFX_PT_NRC_VISUALIZE_PREDICTION) || pass == GFX_PT_PATH_TRACE_NRC_REGIR || pass == GFX_PT_PATH_TRACE_NRC_RESTIR) {
        if (!g_nrcValid) return 1;
        if (pass == GFX_PT_PATH_TRACE_NRC_REGIR) {
            if (!g_regirValid) return 1;
            p.regir = &rs;
        }
        if (pass == GFX_PT_PATH_TRACE_NRC_RESTIR) {
            rp.currentReservoirIndex = g_ptReservoirIndex;
            p.restir = &rp;
        }
        NrcState ns; ns.n = &g_nrcParams;
        const int W = sp->imageSizeX, H = sp->imageSizeY;
        switch (pass) {
        case GFX_PT_NRC_PREPROCESS: preprocessNRC(ns, *fp); break;
        case GFX_PT_PATH_TRACE_NRC_REGIR:
        case GFX_PT_PATH_TRACE_NRC_RESTIR:
        case GFX_PT_PATH_TRACE_NRC: {
            // one thread, row-major: the training-record order (an atomicAdd race in the reference) is defined
            // a window (x0, y0, x1, y1) restricts the pass to those pixels (full-size parity tests): per-pixel results
            // do not depend on other pixels; the training-record indices do and are not comparable then
            uint32_t* counter = ns.numTrainingData(fp->bufferIndex);
            const int wx1 = x1 > 0 ? x1 : W, wy1 = y1 > 0 ? y1 : H;
            for (int y = y0; y < wy1; ++y) for (int x = x0; x < wx1; ++x) nrcPathTracePixel(p, ns, counter, x, y);
            break;
        }
        case GFX_PT_NRC_ACCUMULATE: {   // per pixel: honours the window like the path tracing (band renderers)
            const int wx1 = x1 > 0 ? x1 : W, wy1 = y1 > 0 ? y1 : H;
            for (int y = y0; y < wy1; ++y) for (int x = x0; x < wx1; ++x) accumulateInferredRadiancePixel(ns, *sp, *fp, static_cast<size_t>(y) * W + x);
            break;
        }
        case GFX_PT_NRC_PROPAGATE:
            for (uint32_t i = 0; i < g_nrcParams.maxNumTrainingSuffixes; ++i) propagateRadianceSuffix(ns, *sp, i);
            break;
        case GFX_PT_NRC_SHUFFLE: shuffleTrainingData(ns, *fp); break;
        case GFX_PT_NRC_VISUALIZE_PREDICTION: {
            const int wx1 = x1 > 0 ? x1 : W, wy1 = y1 > 0 ? y1 : H;
            for (int y = y0; y < wy1; ++y) for (int x = x0; x < wx1; ++x) visualizePredictionPixel(p, ns, x, y);
            break;
        }
        }
        return 0;
    }
    if (pass == GFX_PT_PATH_TRACE_REGIR) {
        if (!g_regirValid) return 1;
        p.regir = &rs;
    }
    else if (pass != GFX_PT_PATH_TRACE_BASELINE) return 1;
    if (x1 <= 0) x1 = sp->imageSizeX;
    if (y1 <= 0) y1 = sp->imageSizeY;
#pragma omp parallel for schedule(dynamic, 4) num_threads(s->numThreads)
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) pathTracePixel(p, x, y);
    return 0;
}

// ---------------------------------------------------------------- host-side tables / seeds
// restir_di/restir_di_main.cpp:1316-1321: rng.setState(mt19937_64(seed)()) row-major.
void orc_seed_rngs(uint64_t* states, uint64_t count, uint64_t seed) {
    std::mt19937_64 rngSeed(seed);
    for (uint64_t i = 0; i < count; ++i) states[i] = rngSeed();
}
void orc_spatial_neighbor_deltas(float* out2x1024) {
    const std::vector<V2> t = makeSpatialNeighborDeltas();
    for (int i = 0; i < 1024; ++i) { out2x1024[2 * i] = t[i].x; out2x1024[2 * i + 1] = t[i].y; }
}

// ---------------------------------------------------------------- unit-level exports (tests)
void orc_pcg32_floats(uint64_t* state, float* out, uint32_t n) {
    PCG32RNG r; r.setState(*state);
    for (uint32_t i = 0; i < n; ++i) out[i] = r.getFloat0cTo1o();
    *state = r.state;
}
void orc_pcg32_uints(uint64_t* state, uint32_t* out, uint32_t n) {
    PCG32RNG r; r.setState(*state);
    for (uint32_t i = 0; i < n; ++i) out[i] = r();
    *state = r.state;
}
void orc_math_sincos(const float* x, float* s, float* c, uint32_t n) { for (uint32_t i = 0; i < n; ++i) gm_sincos(x[i], &s[i], &c[i]); }
void orc_math_acos(const float* x, float* y, uint32_t n) { for (uint32_t i = 0; i < n; ++i) y[i] = gm_acos(x[i]); }
void orc_math_atan2(const float* y, const float* x, float* r, uint32_t n) { for (uint32_t i = 0; i < n; ++i) r[i] = gm_atan2(y[i], x[i]); }
void orc_encode_normal(const float* v3, uint32_t* q, uint32_t n) { for (uint32_t i = 0; i < n; ++i) q[i] = encodeNormal(V3(v3[3 * i], v3[3 * i + 1], v3[3 * i + 2])); }
void orc_decode_normal(const uint32_t* q, float* v3, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) { const V3 v = decodeNormal(q[i]); v3[3 * i] = v.x; v3[3 * i + 1] = v.y; v3[3 * i + 2] = v.z; }
}
void orc_offset_ray_origin(const float* p3, const float* n3, float* out3, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) {
        const V3 r = offsetRayOrigin(V3(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]), V3(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]));
        out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
    }
}
// DiscreteDistribution1D::sample on host arrays
void orc_discrete_sample(const float* weights, uint32_t numValues, const float* us, uint32_t n,
                         uint32_t* idx, float* prob, float* remapped, float* integralOut) {
    std::vector<float> w(weights, weights + numValues), cdf;
    DiscreteDistribution1D d;
    Scene::buildCDF(w, cdf, &d);
    *integralOut = d.integral();
    for (uint32_t i = 0; i < n; ++i) idx[i] = d.sample(us[i], &prob[i], &remapped[i]);
}
// BSDF unit evaluation: mode 0 evaluate(vGiven,vSampled)->rgb ; 1 evaluatePDF -> out[0];
// 2 sampleThroughput(vGiven,u0=vSampled.x,u1=vSampled.y) -> rgb, dir (3), pdf ; 3 DH reflectance
void orc_bsdf_eval(const gfx_material* m, int mode, const float* vGiven3, const float* vSampled3, float* out7, uint32_t n) {
    MaterialData d; d.bsdfType = m->bsdfType;
    for (int i = 0; i < 3; ++i) { d.a[i] = m->a[i]; d.b[i] = m->b[i]; d.emittance[i] = m->emittance[i]; }
    d.smoothness = m->smoothness; d.hasEmittance = m->hasEmittance;
    const TextureTable noTextures; BSDF b; b.setup(noTextures, d, V2{ 0.0f, 0.0f });   // constants only
    for (uint32_t i = 0; i < n; ++i) {
        const V3 vg(vGiven3[3 * i], vGiven3[3 * i + 1], vGiven3[3 * i + 2]);
        const V3 vs(vSampled3[3 * i], vSampled3[3 * i + 1], vSampled3[3 * i + 2]);
        float* o = out7 + 7 * i;
        for (int k = 0; k < 7; ++k) o[k] = 0;
        if (mode == 0) { const RGB r = b.evaluate(vg, vs); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
        else if (mode == 1) o[0] = b.evaluatePDF(vg, vs);
        else if (mode == 2) {
            V3 dir; float pdf;
            const RGB r = b.sampleThroughput(vg, vs.x, vs.y, &dir, &pdf);
            o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = dir.x; o[4] = dir.y; o[5] = dir.z; o[6] = pdf;
        }
        else { const RGB r = b.evaluateDHReflectanceEstimate(vg); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
    }
}
// sampleLight<false> + unshadowed performDirectLighting from one shading point (config #1 plumbing)
void orc_sample_light(orc_scene* s, const float* shadingPoint3, const float* u3, uint32_t n,
                      float* lightSample10, float* areaPDensity) {
    const V3 sp(shadingPoint3[0], shadingPoint3[1], shadingPoint3[2]);
    for (uint32_t i = 0; i < n; ++i) {
        LightSample ls; float pd = 0;
        sampleLight(s->scene, 0.0f, 0.0f, sp, u3[3 * i], false, u3[3 * i + 1], u3[3 * i + 2], &ls, &pd);
        float* o = lightSample10 + 10 * i;
        o[0] = ls.emittance.x; o[1] = ls.emittance.y; o[2] = ls.emittance.z;
        o[3] = ls.position.x; o[4] = ls.position.y; o[5] = ls.position.z;
        o[6] = ls.normal.x; o[7] = ls.normal.y; o[8] = ls.normal.z; o[9] = static_cast<float>(ls.atInfinity);
        areaPDensity[i] = pd;
    }
}


// Streaming weighted-reservoir selection (Reservoir::update, restir_di_shared.h:118-125) over K
// independent streams of M candidates: weights[M*K] and us[M*K] are (M, K) row-major.
// Returns the selected candidate index per stream (-1 if none), sumWeights and streamLength.
void orc_reservoir_stream(const float* weights, const float* us, uint32_t M, uint32_t K,
                          int32_t* selected, float* sumWeights, uint32_t* streamLength) {
    for (uint32_t k = 0; k < K; ++k) {
        Reservoir r; r.initialize(LightSample());
        int32_t sel = -1;
        for (uint32_t m = 0; m < M; ++m) {
            LightSample ls; ls.position = V3(static_cast<float>(m), 0, 0);
            if (r.update(ls, weights[m * K + k], us[m * K + k])) sel = static_cast<int32_t>(m);
        }
        selected[k] = sel; sumWeights[k] = r.sumWeights; streamLength[k] = r.streamLength;
    }
}

// Environment importance map: loadEnvironmentalTexture (common/common_host.cpp:2675-2691) +
// RegularConstantContinuousDistribution1D/2D::initialize (:292-357), CompensatedSum_T
// (common/basic_types.h:5428-5452).  sin(theta) goes through the math contract (gm_sin).
static float orc_rccd1d(const float* values, uint32_t n, float* PDF, float* CDF) {
    struct Kahan { float result = 0, comp = 0; void add(float v) { const float c = v - comp; const float t = result + c; comp = (t - result) - c; result = t; } } sum;
    for (uint32_t i = 0; i < n; ++i) { PDF[i] = values[i]; CDF[i] = sum.result; sum.add(PDF[i] / n); }
    const float integral = sum.result;
    for (uint32_t i = 0; i < n; ++i) { PDF[i] /= integral; CDF[i] /= integral; }
    CDF[n] = 1.0f;
    return integral;
}
void orc_env_build(float* texels, uint32_t w, uint32_t h, float* rowPDF, float* rowCDF, float* rowIntegrals,
                   float* topPDF, float* topCDF, float* topIntegral) {
    std::vector<float> importance(static_cast<size_t>(w) * h);
    for (uint32_t y = 0; y < h; ++y) {
        const float theta = kPi * (y + 0.5f) / h;
        const float sinTheta = gm_sin(theta);
        for (uint32_t x = 0; x < w; ++x) {
            float* t = texels + 4 * (static_cast<size_t>(y) * w + x);
            for (int c = 0; c < 3; ++c) t[c] = fmin2(fmax2(t[c], 0.0f), 65504.0f);
            importance[static_cast<size_t>(y) * w + x] = sRGB_calcLuminance(RGB(t[0], t[1], t[2])) * sinTheta;
        }
    }
    for (uint32_t y = 0; y < h; ++y)
        rowIntegrals[y] = orc_rccd1d(importance.data() + static_cast<size_t>(y) * w, w, rowPDF + static_cast<size_t>(y) * w,
                                     rowCDF + static_cast<size_t>(y) * (w + 1));
    *topIntegral = orc_rccd1d(rowIntegrals, h, topPDF, topCDF);
}
// RegularConstantContinuousDistribution2D::sample on host arrays (common_shared.h:372-379)
void orc_env_sample(const float* rowPDF, const float* rowCDF, const float* rowIntegrals, const float* topPDF, const float* topCDF,
                    float topIntegral, uint32_t w, uint32_t h, const float* u2, uint32_t n, float* out3) {
    RegularConstantContinuousDistribution2D d;
    d.rowPDF = rowPDF; d.rowCDF = rowCDF; d.rowIntegrals = rowIntegrals; d.w = w; d.h = h;
    d.top.PDF = topPDF; d.top.CDF = topCDF; d.top.integralValue = topIntegral; d.top.numValues = h;
    for (uint32_t i = 0; i < n; ++i) d.sample(u2[2 * i], u2[2 * i + 1], &out3[3 * i], &out3[3 * i + 1], &out3[3 * i + 2]);
}

const char* orc_version() { return "gfxexp oracle (CPU restatement) 1"; }

} // extern "C"
