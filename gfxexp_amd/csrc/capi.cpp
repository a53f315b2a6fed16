// capi.cpp -- extern "C" entry points declared in include/gfxexp.h.
// Error model: every HIP failure becomes a non-zero return + gfx_last_error() text (the reference
// throws std::runtime_error from CUDADRV_CHECK, utils/cuda_util.cpp:58-69; a C++ shim above this
// ABI can re-throw).
#include <cstdlib>
#include <cstring>
#include <memory>
#include "internal.h"

using namespace gfx;

struct gfx_ctx { Context c; };

static thread_local std::string g_createError;

// Every entry point runs on the context's device and leaves the calling thread's current device as it found it
// (a process may drive several GPUs: torch, several contexts).
namespace {
struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(int device) {
        GFX_HIP(hipGetDevice(&prev));
        if (prev != device) { GFX_HIP(hipSetDevice(device)); switched = true; }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
int env_int(const char* name, int fallback, int lo, int hi) {
    const char* e = getenv(name);
    if (!e || !*e) return fallback;
    const int v = atoi(e);
    return (v < lo || v > hi) ? fallback : v;
}
}
#define GFX_TRY(ctx) if (!(ctx)) return 1; try { DeviceGuard deviceGuard__((ctx)->c.device);
#define GFX_CATCH(ctx) \
    return 0; } \
    catch (const std::exception& e) { (ctx)->c.lastError = e.what(); return 1; } \
    catch (...) { (ctx)->c.lastError = "unknown error"; return 1; }

extern "C" {

const char* gfx_version(void) { return "gfxexp_amd 0.1 gfx950"; }

int gfx_ctx_create(int device, gfx_ctx** out) {
    *out = nullptr;
    try {
        int count = 0;
        GFX_HIP(hipGetDeviceCount(&count));
        if (device < 0 || device >= count) throw HipError("gfx_ctx_create: no such HIP device");
        GFX_HIP(hipSetDevice(device));
        std::unique_ptr<gfx_ctx> ctx(new gfx_ctx());
        ctx->c.device = device;
        hipDeviceProp_t prop;
        GFX_HIP(hipGetDeviceProperties(&prop, device));
        ctx->c.numCUs = prop.multiProcessorCount;
        Tunables& t = ctx->c.tune;
        t.pixelMap = env_int("GFX_PIXEL_MAP", t.pixelMap, 0, 2);
        t.superShiftX = env_int("GFX_SUPER_X", t.superShiftX, 0, 6);
        t.superShiftY = env_int("GFX_SUPER_Y", t.superShiftY, 0, 6);
        t.traceBlocksPerCU = env_int("GFX_TRACE_BLOCKS_PER_CU", t.traceBlocksPerCU, 1, 8);
        t.traceRefill = env_int("GFX_TRACE_REFILL", t.traceRefill, 1, 64);
        t.traceBatch = env_int("GFX_TRACE_BATCH", t.traceBatch, 1, 65536);
        t.temporalHints = env_int("GFX_TEMPORAL_HINTS", t.temporalHints, 0, 1);
        t.ptOverlap = env_int("GFX_PT_OVERLAP", t.ptOverlap, 0, 1);
        t.ptRegen = env_int("GFX_PT_REGEN", t.ptRegen, 0, 8);
        t.ptRegenMin = env_int("GFX_PT_REGEN_MIN", t.ptRegenMin, 1, 64);
        t.candidateSplit = env_int("GFX_CANDIDATE_SPLIT", t.candidateSplit, 0, 4);
        t.fusePasses = env_int("GFX_FUSE_PASSES", t.fusePasses, 0, 2);
        t.blockOrder = env_int("GFX_BLOCK_ORDER", t.blockOrder, 0, 1);
        t.nrcStagedInfer = env_int("GFX_NRC_STAGED_INFER", t.nrcStagedInfer, 0, 3);
        if (t.candidateSplit == 3) t.candidateSplit = 2;
        ctx->c.dTraceCounters.reserve(64);
        GFX_HIP(hipMemset(ctx->c.dTraceCounters.p, 0, 64));
        *out = ctx.release();
        return 0;
    }
    catch (const std::exception& e) { g_createError = e.what(); return 1; }
}

void gfx_ctx_destroy(gfx_ctx* ctx) {
    if (!ctx) return;
    (void)hipDeviceSynchronize();
    delete ctx;
}

const char* gfx_last_error(gfx_ctx* ctx) { return ctx ? ctx->c.lastError.c_str() : g_createError.c_str(); }

int gfx_material_set(gfx_ctx* ctx, uint32_t matSlot, const gfx_material* mat) {
    GFX_TRY(ctx)
    if (!mat) throw HipError("gfx_material_set: null material");
    if (matSlot >= (1u << 24)) throw HipError("gfx_material_set: material slot out of range");   // also keeps matSlot + 1 from wrapping
    if (ctx->c.materials.size() <= matSlot) {
        gfx_material zero; std::memset(&zero, 0, sizeof(zero));
        ctx->c.materials.resize(matSlot + 1, zero);
    }
    ctx->c.materials[matSlot] = *mat;
    ctx->c.sceneDirty = true;
    GFX_CATCH(ctx)
}

int gfx_texture_set(gfx_ctx* ctx, uint32_t texSlot, uint32_t width, uint32_t height, uint32_t format, const void* texels) {
    GFX_TRY(ctx)
    if (texSlot == 0 || texSlot > (1u << 20)) throw HipError("gfx_texture_set: texture slots are 1-based");
    if (width == 0 || height == 0 || width > 16384 || height > 16384) throw HipError("gfx_texture_set: bad size");   // TexDimInfo: 14 bits
    size_t bpp = 0;
    switch (format) {
    case GFX_TEX_RGBA8_SRGB: case GFX_TEX_RGBA8_UNORM: bpp = 4; break;
    case GFX_TEX_R8_UNORM: bpp = 1; break;
    case GFX_TEX_RG8_UNORM: bpp = 2; break;
    case GFX_TEX_RGBA32F: bpp = 16; break;
    default: throw HipError("gfx_texture_set: unknown format");
    }
    if (!texels) throw HipError("gfx_texture_set: null texels");
    if (ctx->c.textures.size() <= texSlot) ctx->c.textures.resize(texSlot + 1);
    HostTexture& t = ctx->c.textures[texSlot];
    t.width = width; t.height = height; t.format = format;
    t.texels.assign(static_cast<const uint8_t*>(texels), static_cast<const uint8_t*>(texels) + bpp * width * height);
    ctx->c.sceneDirty = true;
    GFX_CATCH(ctx)
}

int gfx_texture_sample(gfx_ctx* ctx, void* stream, uint32_t texSlot, const void* dUv, uint32_t n, void* dOut, int gather) {
    GFX_TRY(ctx)
    texture_sample(ctx->c, static_cast<hipStream_t>(stream), texSlot, dUv, n, dOut, gather);
    GFX_CATCH(ctx)
}

int gfx_geom_create(gfx_ctx* ctx, const void* vertices, uint32_t vertexStride, uint32_t numVertices,
                    const uint32_t* triangles, uint32_t numTriangles, uint32_t matSlot, uint32_t* geomInstSlot) {
    GFX_TRY(ctx)
    if (vertexStride < sizeof(gfx_vertex)) throw HipError("gfx_geom_create: vertexStride smaller than gfx_vertex");
    HostGeom g;
    g.vertices.resize(numVertices);
    for (uint32_t i = 0; i < numVertices; ++i) {
        gfx_vertex v;
        std::memcpy(&v, static_cast<const uint8_t*>(vertices) + static_cast<size_t>(vertexStride) * i, sizeof(v));
        DevVertex& d = g.vertices[i];
        d.px = v.position[0]; d.py = v.position[1]; d.pz = v.position[2];
        d.nx = v.normal[0]; d.ny = v.normal[1]; d.nz = v.normal[2];
        d.tx = v.texCoord0Dir[0]; d.ty = v.texCoord0Dir[1]; d.tz = v.texCoord0Dir[2];
        d.u = v.texCoord[0]; d.v = v.texCoord[1]; d.pad = 0;
    }
    g.triangles.assign(triangles, triangles + 3ull * numTriangles);
    for (uint32_t idx : g.triangles) if (idx >= numVertices) throw HipError("gfx_geom_create: triangle index out of range");
    g.materialSlot = matSlot;
    *geomInstSlot = static_cast<uint32_t>(ctx->c.geoms.size());
    ctx->c.geoms.push_back(std::move(g));
    ctx->c.sceneDirty = true;
    GFX_CATCH(ctx)
}

int gfx_group_create(gfx_ctx* ctx, const uint32_t* geomInstSlots, uint32_t n, uint32_t* group) {
    GFX_TRY(ctx)
    for (uint32_t i = 0; i < n; ++i) if (geomInstSlots[i] >= ctx->c.geoms.size()) throw HipError("gfx_group_create: unknown geomInstSlot");
    *group = static_cast<uint32_t>(ctx->c.groups.size());
    ctx->c.groups.emplace_back(geomInstSlots, geomInstSlots + n);
    ctx->c.sceneDirty = true;
    GFX_CATCH(ctx)
}

int gfx_instance_create(gfx_ctx* ctx, uint32_t group, const float xfm[12], uint32_t* instSlot) {
    GFX_TRY(ctx)
    if (group >= ctx->c.groups.size()) throw HipError("gfx_instance_create: unknown group");
    HostInstance inst;
    inst.group = group;
    std::memcpy(inst.transform, xfm, sizeof(float) * 12);
    std::memcpy(inst.prevTransform, xfm, sizeof(float) * 12);
    *instSlot = static_cast<uint32_t>(ctx->c.insts.size());
    ctx->c.insts.push_back(inst);
    ctx->c.sceneDirty = true;
    GFX_CATCH(ctx)
}

static void instance_update(gfx_ctx* ctx, uint32_t instSlot, const float xfm[12], const float* normalMatrix) {
    if (instSlot >= ctx->c.insts.size()) throw HipError("gfx_instance_set_transform: unknown instSlot");
    HostInstance& inst = ctx->c.insts[instSlot];
    std::memcpy(inst.prevTransform, inst.transform, sizeof(float) * 12);
    std::memcpy(inst.transform, xfm, sizeof(float) * 12);
    instance_cur_to_prev(inst.prevTransform, inst.transform, inst.curToPrev);
    inst.animated = true;
    inst.hasNormalMatrix = normalMatrix != nullptr;
    if (normalMatrix) std::memcpy(inst.normalMatrix, normalMatrix, sizeof(float) * 9);
    if (inst.dynamic && !ctx->c.sceneDirty) {       // already in the animated subtree: transform-only update
        ctx->c.transformsDirty = true;
        ctx->c.movedInsts.push_back(instSlot);
    }
    else {                                          // first move: the instance changes subtree, full rebuild once
        inst.dynamic = true;
        ctx->c.sceneDirty = true;
    }
}

int gfx_instance_set_dynamic(gfx_ctx* ctx, uint32_t instSlot, int dynamic) {
    GFX_TRY(ctx)
    if (instSlot >= ctx->c.insts.size()) throw HipError("gfx_instance_set_dynamic: unknown instSlot");
    if (ctx->c.insts[instSlot].dynamic != (dynamic != 0)) {
        ctx->c.insts[instSlot].dynamic = dynamic != 0;
        ctx->c.sceneDirty = true;
    }
    GFX_CATCH(ctx)
}

int gfx_instance_set_transform(gfx_ctx* ctx, uint32_t instSlot, const float xfm[12]) {
    GFX_TRY(ctx)
    instance_update(ctx, instSlot, xfm, nullptr);
    GFX_CATCH(ctx)
}

int gfx_instance_set_transform_and_normal_matrix(gfx_ctx* ctx, uint32_t instSlot, const float xfm[12], const float normalMatrix[9]) {
    GFX_TRY(ctx)
    instance_update(ctx, instSlot, xfm, normalMatrix);
    GFX_CATCH(ctx)
}

int gfx_accel_build(gfx_ctx* ctx, void* stream, uint64_t* handle) {
    GFX_TRY(ctx)
    Accel* a = nullptr;
    if (*handle != 0 && *handle <= ctx->c.accels.size() && ctx->c.accels[*handle - 1]) a = ctx->c.accels[*handle - 1]; // rebuild in place
    else { a = new Accel(); ctx->c.accels.push_back(a); *handle = ctx->c.accels.size(); }
    if (!lbvh_update_dynamic(ctx->c, static_cast<hipStream_t>(stream), *a)) lbvh_build(ctx->c, static_cast<hipStream_t>(stream), *a);
    GFX_CATCH(ctx)
}

static Accel* find_accel(gfx_ctx* ctx, uint64_t handle) {
    if (handle == 0 || handle > ctx->c.accels.size() || !ctx->c.accels[handle - 1]) throw HipError("invalid accel handle");
    return ctx->c.accels[handle - 1];
}

int gfx_accel_stats(gfx_ctx* ctx, uint64_t handle, uint32_t stats[4]) {
    GFX_TRY(ctx)
    const Accel* a = find_accel(ctx, handle);
    stats[0] = a->numInputTris; stats[1] = a->numNodes; stats[2] = a->numTris; stats[3] = a->maxDepth;
    GFX_CATCH(ctx)
}

int gfx_accel_tri_ids(gfx_ctx* ctx, uint64_t handle, const void** dTriIds, uint32_t* count) {
    GFX_TRY(ctx)
    const Accel* a = find_accel(ctx, handle);
    *dTriIds = a->triIds.p; *count = a->numTris;
    GFX_CATCH(ctx)
}

int gfx_accel_set_max_leaf(gfx_ctx* ctx, uint32_t maxLeafTris) {
    GFX_TRY(ctx)
    ctx->c.maxLeafTris = maxLeafTris < 1 ? 1 : (maxLeafTris > 4 ? 4 : maxLeafTris);   // 8 children x 4 = one 32-bit triangle mask
    GFX_CATCH(ctx)
}

int gfx_lights_build_static(gfx_ctx* ctx, void* stream) {
    GFX_TRY(ctx)
    lights_build_static(ctx->c, static_cast<hipStream_t>(stream));
    GFX_CATCH(ctx)
}

int gfx_lights_build_instances(gfx_ctx* ctx, void* stream, uint32_t bufferIndex) {
    GFX_TRY(ctx)
    lights_build_instances(ctx->c, static_cast<hipStream_t>(stream), bufferIndex);
    GFX_CATCH(ctx)
}

int gfx_lights_read(gfx_ctx* ctx, uint32_t level, uint32_t index, float* weights, float* cdf,
                    uint32_t capacity, uint32_t* n, float* integral) {
    GFX_TRY(ctx)
    Context& c = ctx->c;
    GFX_HIP(hipDeviceSynchronize());
    uint32_t off = 0xFFFFFFFFu, count = 0;
    float integ = 0.0f;
    if (level == 0) {
        off = c.lightInstDistOffset; count = static_cast<uint32_t>(c.insts.size());
        GFX_HIP(hipMemcpy(&integ, c.dLightInstIntegral.p, sizeof(float), hipMemcpyDeviceToHost));
    }
    else if (level == 1) {
        if (index >= c.hInsts.size()) throw HipError("gfx_lights_read: bad instance index");
        DevInstance d;
        GFX_HIP(hipMemcpy(&d, c.dInsts.as<DevInstance>() + index, sizeof(d), hipMemcpyDeviceToHost));
        off = d.distOffset; count = off == 0xFFFFFFFFu ? 0 : d.numGeomInsts; integ = d.distIntegral;
    }
    else {
        if (index >= c.hGeomInsts.size()) throw HipError("gfx_lights_read: bad geomInst index");
        DevGeomInst d;
        GFX_HIP(hipMemcpy(&d, c.dGeomInsts.as<DevGeomInst>() + index, sizeof(d), hipMemcpyDeviceToHost));
        off = d.distOffset; count = off == 0xFFFFFFFFu ? 0 : d.distCount; integ = d.distIntegral;
    }
    *n = count; *integral = integ;
    const uint32_t m = count < capacity ? count : capacity;
    if (m && weights) GFX_HIP(hipMemcpy(weights, c.dLightW.as<float>() + off, sizeof(float) * m, hipMemcpyDeviceToHost));
    if (m && cdf) GFX_HIP(hipMemcpy(cdf, c.dLightCDF.as<float>() + off, sizeof(float) * m, hipMemcpyDeviceToHost));
    GFX_CATCH(ctx)
}

int gfx_lights_table_info(gfx_ctx* ctx, uint32_t info[8]) {
    GFX_TRY(ctx)
    Context& c = ctx->c;
    GFX_HIP(hipDeviceSynchronize());
    uint32_t header[4] = { 0, 0, 0, 0 };
    if (c.dSpanHeader.p) GFX_HIP(hipMemcpy(header, c.dSpanHeader.p, sizeof(header), hipMemcpyDeviceToHost));
    info[0] = header[0]; info[1] = header[1]; info[2] = c.numEmitterRecs; info[3] = c.spanGuideCells;
    info[4] = c.numLightMatrices; info[5] = 0; info[6] = 0; info[7] = 0;
    if (header[0] && c.spanGuideCells && c.dSpanGuide.p) {
        std::vector<SpanGuide> guide(c.spanGuideCells);
        GFX_HIP(hipMemcpy(guide.data(), c.dSpanGuide.p, sizeof(SpanGuide) * guide.size(), hipMemcpyDeviceToHost));
        for (const SpanGuide& g : guide) if (g.a & kGuideInterior) ++info[5];
    }
    GFX_CATCH(ctx)
}

int gfx_trace(gfx_ctx* ctx, void* stream, uint64_t accel, int mode, const void* dRayOrgTmin, const void* dRayDirTmax,
              uint32_t numRays, void* dOut, void* dCounters) {
    return gfx_trace_counted(ctx, stream, accel, mode, dRayOrgTmin, dRayDirTmax, numRays, dOut, dCounters, nullptr);
}

int gfx_trace_counted(gfx_ctx* ctx, void* stream, uint64_t accel, int mode, const void* dRayOrgTmin, const void* dRayDirTmax,
                      uint32_t numRays, void* dOut, void* dCounters, void* dPerRayItems) {
    GFX_TRY(ctx)
    if (dPerRayItems && !dCounters) throw HipError("gfx_trace_counted: per-ray item counts need the counter buffer (the counting kernel)");
    const Accel* a = find_accel(ctx, accel);
    TraceLaunch t;
    t.accel = a->dev();
    t.rayOrgTmin = static_cast<const float4*>(dRayOrgTmin);
    t.rayDirTmax = static_cast<const float4*>(dRayDirTmax);
    t.numRays = numRays; t.numRaysPtr = nullptr; t.out = dOut; t.mode = mode;
    t.perRayItems = static_cast<uint32_t*>(dPerRayItems);
    // explicit counter buffer: count into the caller's u64[4]
    const bool savedEnabled = ctx->c.countersEnabled;
    DevBuf saved = ctx->c.dTraceCounters;
    if (dCounters) { ctx->c.countersEnabled = true; ctx->c.dTraceCounters.p = dCounters; ctx->c.countersSplit = false; }
    try { trace_launch(ctx->c, static_cast<hipStream_t>(stream), t); }
    catch (...) { ctx->c.dTraceCounters = saved; ctx->c.countersEnabled = savedEnabled; ctx->c.countersSplit = true; throw; }
    ctx->c.dTraceCounters = saved; ctx->c.countersEnabled = savedEnabled; ctx->c.countersSplit = true;
    GFX_CATCH(ctx)
}

int gfx_restir_copy_to_linear(gfx_ctx* ctx, void* stream, void* dLinearColor, void* dLinearAlbedo, void* dLinearNormal, void* dLinearMotionVector) {
    GFX_TRY(ctx)
    restir_copy_to_linear(ctx->c, static_cast<hipStream_t>(stream), dLinearColor, dLinearAlbedo, dLinearNormal, dLinearMotionVector);
    GFX_CATCH(ctx)
}
int gfx_visualize(gfx_ctx* ctx, void* stream, const void* dLinearBuffer, int bufferTypeToDisplay, float motionVectorOffset, float motionVectorScale,
                  uint32_t width, uint32_t height, void* dOutputFloat4) {
    GFX_TRY(ctx)
    restir_visualize(ctx->c, static_cast<hipStream_t>(stream), dLinearBuffer, bufferTypeToDisplay, motionVectorOffset, motionVectorScale, width, height, dOutputFloat4);
    GFX_CATCH(ctx)
}

int gfx_restir_set_params(gfx_ctx* ctx, void* /*stream*/, const gfx_restir_static_params* s, const gfx_restir_frame_params* f,
                          uint32_t currentReservoirIndex, uint32_t spatialNeighborBaseIndex) {
    GFX_TRY(ctx)
    // Parameters travel by value in the kernel arguments (the reference copies three structs to
    // the device with cuMemcpyHtoDAsync per frame, restir_di_main.cpp:2350-2359).
    if (s) ctx->c.restir.s = *s;
    if (f) ctx->c.restir.f = *f;
    ctx->c.restir.currentReservoirIndex = currentReservoirIndex & 1u;
    ctx->c.restir.spatialNeighborBaseIndex = spatialNeighborBaseIndex & 1023u;
    ctx->c.restir.valid = true;
    GFX_CATCH(ctx)
}

int gfx_restir_launch(gfx_ctx* ctx, void* stream, int pass, uint32_t width, uint32_t height) {
    GFX_TRY(ctx)
    restir_launch(ctx->c, static_cast<hipStream_t>(stream), pass, width, height, 0, height);
    GFX_CATCH(ctx)
}

int gfx_restir_launch_rows_gap(gfx_ctx* ctx, void* stream, int pass, uint32_t width, uint32_t height, uint32_t rowBegin, uint32_t rowEnd,
                               uint32_t gapBegin, uint32_t gapEnd) {
    GFX_TRY(ctx)
    if (rowBegin == 0 && rowEnd == 0) rowEnd = height;
    restir_launch(ctx->c, static_cast<hipStream_t>(stream), pass, width, height, rowBegin, rowEnd, gapBegin, gapEnd);
    GFX_CATCH(ctx)
}

int gfx_restir_launch_rows(gfx_ctx* ctx, void* stream, int pass, uint32_t width, uint32_t height, uint32_t rowBegin, uint32_t rowEnd) {
    GFX_TRY(ctx)
    if (rowBegin == 0 && rowEnd == 0) rowEnd = height;   // 0, 0 = every row, as in gfx_pt_launch
    restir_launch(ctx->c, static_cast<hipStream_t>(stream), pass, width, height, rowBegin, rowEnd);
    GFX_CATCH(ctx)
}

int gfx_regir_set_params(gfx_ctx* ctx, const gfx_regir_params* p) {
    GFX_TRY(ctx)
    if (!p) throw HipError("gfx_regir_set_params: null parameters");
    if (!p->gridDimension[0] || !p->gridDimension[1] || !p->gridDimension[2]) throw HipError("gfx_regir_set_params: empty grid");
    ctx->c.regir = *p;
    ctx->c.regirValid = true;
    GFX_CATCH(ctx)
}

int gfx_nrc_set_render_params(gfx_ctx* ctx, const gfx_nrc_params* p) {
    GFX_TRY(ctx)
    if (!p) throw HipError("gfx_nrc_set_render_params: null parameters");
    ctx->c.nrcRender = *p;
    ctx->c.nrcRenderValid = true;
    GFX_CATCH(ctx)
}

int gfx_pt_launch(gfx_ctx* ctx, void* stream, int pass, uint32_t width, uint32_t height, uint32_t maxPathLength,
                  uint32_t rowBegin, uint32_t rowEnd) {
    GFX_TRY(ctx)
    if (rowEnd == 0 && rowBegin == 0) rowEnd = height;
    pathtrace_launch(ctx->c, static_cast<hipStream_t>(stream), pass, width, height, maxPathLength, rowBegin, rowEnd);
    GFX_CATCH(ctx)
}

static NrcNet* nrc_of(gfx_ctx* ctx, uint64_t handle) {
    if (handle == 0 || handle > ctx->c.nrcNets.size() || !ctx->c.nrcNets[handle - 1]) throw HipError("gfx_nrc: invalid network handle");
    return ctx->c.nrcNets[handle - 1];
}
int gfx_nrc_create(gfx_ctx* ctx, int positionEncoding, uint32_t numHiddenLayers, float learningRate, uint64_t* outHandle) {
    GFX_TRY(ctx)
    ctx->c.nrcNets.push_back(nrc_create(ctx->c, positionEncoding, numHiddenLayers, learningRate));
    *outHandle = ctx->c.nrcNets.size();
    GFX_CATCH(ctx)
}
int gfx_nrc_destroy(gfx_ctx* ctx, uint64_t handle) {
    GFX_TRY(ctx)
    NrcNet* net = nrc_of(ctx, handle);
    GFX_HIP(hipDeviceSynchronize());
    nrc_destroy(net);
    ctx->c.nrcNets[handle - 1] = nullptr;
    GFX_CATCH(ctx)
}
int gfx_nrc_infer(gfx_ctx* ctx, void* stream, uint64_t handle, const void* dInputData, uint32_t numData, void* dPredictionData) {
    GFX_TRY(ctx)
    nrc_infer(ctx->c, static_cast<hipStream_t>(stream), nrc_of(ctx, handle), static_cast<const float*>(dInputData), numData, static_cast<float*>(dPredictionData));
    GFX_CATCH(ctx)
}
int gfx_nrc_infer_indirect(gfx_ctx* ctx, void* stream, uint64_t handle, const void* dInputData, const void* dNumData, uint32_t maxNumData, void* dPredictionData) {
    GFX_TRY(ctx)
    if (!dNumData) throw HipError("gfx_nrc_infer_indirect: null batch-size pointer");
    nrc_infer(ctx->c, static_cast<hipStream_t>(stream), nrc_of(ctx, handle), static_cast<const float*>(dInputData), maxNumData, static_cast<float*>(dPredictionData),
              static_cast<const uint32_t*>(dNumData));
    GFX_CATCH(ctx)
}
int gfx_nrc_query_count_ptr(gfx_ctx* ctx, void** dNumData) {
    GFX_TRY(ctx)
    if (!ctx->c.nrcQueryCount.p) { ctx->c.nrcQueryCount.reserve(256); GFX_HIP(hipMemset(ctx->c.nrcQueryCount.p, 0, 256)); }
    *dNumData = ctx->c.nrcQueryCount.p;
    GFX_CATCH(ctx)
}
int gfx_nrc_train(gfx_ctx* ctx, void* stream, uint64_t handle, const void* dInputData, const void* dTargetData, uint32_t numData, float* lossOnCPU) {
    GFX_TRY(ctx)
    nrc_train(ctx->c, static_cast<hipStream_t>(stream), nrc_of(ctx, handle), static_cast<const float*>(dInputData), static_cast<const float*>(dTargetData), numData, lossOnCPU);
    GFX_CATCH(ctx)
}
int gfx_nrc_num_params(gfx_ctx* ctx, uint64_t handle, uint32_t* outCount) {
    GFX_TRY(ctx)
    *outCount = nrc_num_params(nrc_of(ctx, handle));
    GFX_CATCH(ctx)
}
int gfx_nrc_set_params(gfx_ctx* ctx, uint64_t handle, const float* hostParams, uint32_t count) {
    GFX_TRY(ctx)
    nrc_set_params(ctx->c, nullptr, nrc_of(ctx, handle), hostParams, count);
    GFX_CATCH(ctx)
}
int gfx_nrc_get_params(gfx_ctx* ctx, uint64_t handle, int which, float* hostOut, uint32_t count) {
    GFX_TRY(ctx)
    nrc_get_params(nrc_of(ctx, handle), which, hostOut, count);
    GFX_CATCH(ctx)
}

int gfx_nrc_inference_image(gfx_ctx* ctx, uint64_t handle, int which, void** dPtr, uint64_t* bytes) {
    GFX_TRY(ctx)
    if (!dPtr || !bytes) throw HipError("gfx_nrc_inference_image: null output");
    nrc_inference_image(ctx->c, nrc_of(ctx, handle), which, dPtr, bytes);
    GFX_CATCH(ctx)
}

int gfx_nrc_inference_image_async(gfx_ctx* ctx, void* stream, uint64_t handle, int which, void** dPtr, uint64_t* bytes) {
    GFX_TRY(ctx)
    if (!dPtr || !bytes) throw HipError("gfx_nrc_inference_image_async: null output");
    nrc_inference_image(ctx->c, nrc_of(ctx, handle), which, dPtr, bytes, static_cast<hipStream_t>(stream), true);
    GFX_CATCH(ctx)
}

int gfx_nrc_params_checksum(gfx_ctx* ctx, void* stream, uint64_t handle, void* dOutU32) {
    GFX_TRY(ctx)
    if (!dOutU32) throw HipError("gfx_nrc_params_checksum: null output");
    nrc_params_checksum(ctx->c, static_cast<hipStream_t>(stream), nrc_of(ctx, handle), static_cast<uint32_t*>(dOutU32));
    GFX_CATCH(ctx)
}

int gfx_read_device(gfx_ctx* ctx, const void* dSrc, void* hostDst, size_t bytes) {
    GFX_TRY(ctx)
    GFX_HIP(hipDeviceSynchronize());
    GFX_HIP(hipMemcpy(hostDst, dSrc, bytes, hipMemcpyDeviceToHost));
    GFX_CATCH(ctx)
}

int gfx_timing_enable(gfx_ctx* ctx, int enable) {
    GFX_TRY(ctx)
    ctx->c.timingEnabled = enable != 0;
    GFX_CATCH(ctx)
}

int gfx_timing_collect(gfx_ctx* ctx, char names[][48], float* totalMs, uint32_t* calls, uint32_t capacity, uint32_t* n) {
    GFX_TRY(ctx)
    Context& c = ctx->c;
    for (auto& e : c.pendingEvents) {
        GFX_HIP(hipEventSynchronize(e.second.second));
        float ms = 0;
        GFX_HIP(hipEventElapsedTime(&ms, e.second.first, e.second.second));
        KernelTiming& t = c.timings[e.first];
        t.ms += ms; ++t.calls;
        (void)hipEventDestroy(e.second.first); (void)hipEventDestroy(e.second.second);
    }
    c.pendingEvents.clear();
    uint32_t i = 0;
    for (const auto& kv : c.timings) {
        if (i < capacity) {
            std::memset(names[i], 0, 48);
            std::strncpy(names[i], kv.first.c_str(), 47);
            totalMs[i] = static_cast<float>(kv.second.ms); calls[i] = kv.second.calls;
        }
        ++i;
    }
    *n = i;
    c.timings.clear();
    GFX_CATCH(ctx)
}

int gfx_tunable_set(gfx_ctx* ctx, const char* name, int value) {
    GFX_TRY(ctx)
    if (!name) throw HipError("gfx_tunable_set: null name");
    Tunables& t = ctx->c.tune;
    const std::string n(name);
    auto in = [&](int lo, int hi) { if (value < lo || value > hi) throw HipError("gfx_tunable_set: value out of range for " + n); return value; };
    if (n == "pixel_map") t.pixelMap = in(0, 2);
    else if (n == "super_x") t.superShiftX = in(0, 6);
    else if (n == "super_y") t.superShiftY = in(0, 6);
    else if (n == "trace_blocks_per_cu") t.traceBlocksPerCU = in(1, 8);
    else if (n == "trace_refill") t.traceRefill = in(1, 64);
    else if (n == "temporal_hints") t.temporalHints = in(0, 1);
    else if (n == "trace_batch") t.traceBatch = in(1, 65536);
    else if (n == "pt_overlap") t.ptOverlap = in(0, 1);
    else if (n == "pt_regen") t.ptRegen = in(0, 8);
    else if (n == "pt_diag") t.ptDiag = in(0, 1);
    else if (n == "pt_regen_min") t.ptRegenMin = in(1, 64);
    else if (n == "fuse_passes") t.fusePasses = in(0, 2);
    else if (n == "block_order") t.blockOrder = in(0, 1);
    else if (n == "nrc_staged_infer") t.nrcStagedInfer = in(0, 3);
    else if (n == "candidate_split") { if (value == 3) throw HipError("gfx_tunable_set: candidate_split is 0 (automatic), 1, 2 or 4"); t.candidateSplit = in(0, 4); }
    else throw HipError("gfx_tunable_set: unknown tunable " + n);
    GFX_CATCH(ctx)
}

int gfx_stream_copy(gfx_ctx* ctx, void* dDst, const void* dSrc, size_t bytes, void* stream) {
    GFX_TRY(ctx)
    stream_copy(ctx->c, static_cast<hipStream_t>(stream), dDst, dSrc, bytes);
    GFX_CATCH(ctx)
}

int gfx_counters_enable(gfx_ctx* ctx, int enable) {
    GFX_TRY(ctx)
    ctx->c.countersEnabled = enable != 0;
    GFX_CATCH(ctx)
}

int gfx_trace_diag_read(gfx_ctx* ctx, uint64_t diag[8], int reset) {
    GFX_TRY(ctx)
    GFX_HIP(hipDeviceSynchronize());
    for (int i = 0; i < 8; ++i) diag[i] = 0;
    if (ctx->c.dTraceDiag.p) {
        GFX_HIP(hipMemcpy(diag, ctx->c.dTraceDiag.p, 64, hipMemcpyDeviceToHost));
        if (reset) GFX_HIP(hipMemset(ctx->c.dTraceDiag.p, 0, 64));
    }
    GFX_CATCH(ctx)
}

int gfx_pt_diag_read(gfx_ctx* ctx, uint64_t diag[8], int reset) {
    GFX_TRY(ctx)
    GFX_HIP(hipDeviceSynchronize());
    for (int i = 0; i < 8; ++i) diag[i] = 0;
    if (ctx->c.ptDiag.p) {
        GFX_HIP(hipMemcpy(diag, ctx->c.ptDiag.p, 64, hipMemcpyDeviceToHost));
        if (reset) GFX_HIP(hipMemset(ctx->c.ptDiag.p, 0, 64));
    }
    GFX_CATCH(ctx)
}

int gfx_counters_read(gfx_ctx* ctx, uint64_t counters[8], int reset) {
    GFX_TRY(ctx)
    GFX_HIP(hipDeviceSynchronize());
    GFX_HIP(hipMemcpy(counters, ctx->c.dTraceCounters.p, 64, hipMemcpyDeviceToHost));
    if (reset) GFX_HIP(hipMemset(ctx->c.dTraceCounters.p, 0, 64));
    GFX_CATCH(ctx)
}

} // extern "C"
