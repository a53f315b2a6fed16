// restir_driver.cpp -- headless ReSTIR DI frame driver (include/gfxexp_host.h).
//
// Re-creates, on top of the C ABI, the part of restir_di/restir_di_main.cpp that surrounds the hot
// path: buffer allocation and seeding (:1210-1325), the Halton neighbour table (:1487-1542),
// launch-parameter defaults (:1560-1631, :1938-1986) and the per-frame sequencing with its index
// bookkeeping (:1694-1700, :2303-2493): bufferIndex = frameIndex % 2, prevCamera latch, reservoir
// ping-pong (lastReservoirIndex starts at 1), spatialNeighborBaseIndex growth, newSequence.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../../include/gfxexp_host.h"

namespace {
thread_local std::string g_driverError;
bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    g_driverError = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}
#define DRV_HIP(call) do { if (!hip_ok((call), #call)) return 1; } while (0)
}

struct gfxh_restir {
    gfx_ctx* ctx = nullptr;
    gfxh_restir_config cfg;
    gfx_restir_static_params sp;
    gfx_restir_frame_params fp;
    std::vector<void*> allocations;
    uint64_t accel = 0;
    uint32_t frameIndex = 0;
    uint32_t lastReservoirIndex = 1;            // restir_di_main.cpp:1686
    uint32_t lastSpatialNeighborBaseIndex = 0;
    uint32_t numAccumFrames = 0;
    bool resetRequested = false;
    gfx_camera camera, prevCamera;
    float envPowerCoeff = 1.0f, envRotation = 0.0f;
    gfx_regir_params regir;
    // Frame pipelining: the G-buffer pass of frame N + 1 (primary rays, closest-hit traversal, resolve) depends on
    // nothing frame N computes, only on frame N being done with the G-buffer it overwrites (the "previous" one,
    // last read by the temporal pass).  It runs on gbStream underneath the rest of frame N; the context keeps
    // a separate scratch set for it.  evPrevRead: frame N no longer reads the previous G-buffer;
    // evGbuffer: the G-buffer of the frame is complete.
    hipStream_t gbStream = nullptr;
    hipEvent_t evPrevRead = nullptr, evGbuffer = nullptr;
    bool prevReadPending = false, pipelineFrames = true;
};

extern "C" {

void gfxh_restir_default_config(gfxh_restir_config* cfg, uint32_t width, uint32_t height, int renderer) {
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->width = width; cfg->height = height; cfg->renderer = renderer;
    cfg->log2NumCandidateSamples = 5;                     // ReSTIRConfigs(5, 2, 5) / (5, 1, 3), :1966-1967
    cfg->enableTemporalReuse = 1; cfg->enableSpatialReuse = 1;
    cfg->numSpatialReusePasses = renderer == GFXH_ORIGINAL_RESTIR_BIASED ? 2 : 1;
    cfg->numSpatialNeighbors = renderer == GFXH_ORIGINAL_RESTIR_BIASED ? 5 : renderer == GFXH_ORIGINAL_RESTIR_UNBIASED ? 3 : 1;
    cfg->spatialNeighborRadius = 20.0f;
    cfg->useLowDiscrepancyNeighbors = 1;
    cfg->reuseVisibility = 1;
    cfg->enableAccumulation = 0;
    cfg->log2MaxNumAccums = 16;
    cfg->maxPathLength = 5;                                // path_tracing_main.cpp:1519
    cfg->regirGridDimension[0] = 32; cfg->regirGridDimension[1] = 8; cfg->regirGridDimension[2] = 32;   // regir_main.cpp:1112
    cfg->regirLog2CandidatesPerLightSlot = 3; cfg->regirLog2CandidatesPerCell = 2;                       // :1733-1734
    cfg->regirEnableTemporalReuse = 1; cfg->regirEnableCellRandomization = 1;                             // :1735-1736
    cfg->enableJittering = 0;
    cfg->enableBumpMapping = 0;
    cfg->camera.aspect = static_cast<float>(width) / height;
    cfg->camera.fovY = 50 * 3.14159265358979323846f / 180;   // :1613
    const float ident[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    std::memcpy(cfg->camera.orientation, ident, sizeof(ident));
}

void gfxh_band_plan_compute(uint32_t height, uint32_t bandBegin, uint32_t bandEnd, uint32_t radiusRows,
                            uint32_t numSpatialPasses, uint32_t maxMotionRows, gfxh_band_plan* out) {
    std::memset(out, 0, sizeof(*out));
    if (numSpatialPasses > 8) numSpatialPasses = 8;
    const uint32_t b = bandBegin, e = bandEnd;
    const uint32_t halo = radiusRows * numSpatialPasses + maxMotionRows;
    auto lo = [&](uint32_t rows) { return b > rows ? b - rows : 0u; };
    auto hi = [&](uint32_t rows) { return std::min(height, e + rows); };
    out->bandBegin = b; out->bandEnd = e; out->haloRows = halo;
    out->gbufferRows[0] = lo(halo); out->gbufferRows[1] = hi(halo);
    out->initialRows[0] = lo(halo); out->initialRows[1] = hi(halo);
    for (uint32_t i = 0; i < numSpatialPasses; ++i) {
        const uint32_t rows = radiusRows * (numSpatialPasses - 1 - i);
        out->spatialRows[i][0] = lo(rows); out->spatialRows[i][1] = hi(rows);
    }
    out->shadingRows[0] = b; out->shadingRows[1] = e;
    out->recvAbove[0] = lo(halo); out->recvAbove[1] = b;
    out->recvBelow[0] = e; out->recvBelow[1] = hi(halo);
    out->sendAbove[0] = b; out->sendAbove[1] = std::min(e, b + halo);     // what rank-1 receives "below" its band
    out->sendBelow[0] = e > halo ? std::max(b, e - halo) : b; out->sendBelow[1] = e;
    if (b == 0) { out->sendAbove[0] = out->sendAbove[1] = 0; }
    if (e >= height) { out->sendBelow[0] = out->sendBelow[1] = e; }
}

static int alloc_dev(gfxh_restir* r, void** p, size_t bytes, bool zero) {
    DRV_HIP(hipMalloc(p, bytes));
    r->allocations.push_back(*p);
    if (zero) DRV_HIP(hipMemset(*p, 0, bytes));
    return 0;
}

int gfxh_restir_create(gfx_ctx* ctx, const gfxh_restir_config* cfg, gfxh_restir** out) {
    *out = nullptr;
    gfxh_restir* r = new gfxh_restir();
    r->ctx = ctx; r->cfg = *cfg;
    const size_t n = static_cast<size_t>(cfg->width) * cfg->height;
    gfx_restir_static_params& sp = r->sp;
    std::memset(&sp, 0, sizeof(sp));
    std::memset(&r->fp, 0, sizeof(r->fp));
    sp.imageSizeX = static_cast<int32_t>(cfg->width); sp.imageSizeY = static_cast<int32_t>(cfg->height);
    int err = 0;
    err |= alloc_dev(r, &sp.rngBuffer, 8 * n, false);
    for (int i = 0; i < 2; ++i) {
        err |= alloc_dev(r, &sp.gbuffer0[i], sizeof(gfx_gbuffer0) * n, true);
        err |= alloc_dev(r, &sp.gbuffer1[i], sizeof(gfx_gbuffer1) * n, true);
        err |= alloc_dev(r, &sp.gbuffer2[i], sizeof(gfx_gbuffer2) * n, true);
        err |= alloc_dev(r, &sp.gbuffer3[i], sizeof(gfx_gbuffer3) * n, true);
        err |= alloc_dev(r, &sp.reservoirBuffer[i], 48 * n, true);
        err |= alloc_dev(r, &sp.reservoirInfoBuffer[i], sizeof(gfx_reservoir_info) * n, true);
        err |= alloc_dev(r, &sp.sampleVisibilityBuffer[i], 4 * n, true);
    }
    err |= alloc_dev(r, &sp.beautyAccumBuffer, 16 * n, true);
    err |= alloc_dev(r, &sp.albedoAccumBuffer, 16 * n, true);
    err |= alloc_dev(r, &sp.normalAccumBuffer, 16 * n, true);
    void* deltas = nullptr;
    err |= alloc_dev(r, &deltas, 8 * 1024, false);
    const bool rearch = cfg->renderer == GFXH_REARCHITECTED_RESTIR_BIASED || cfg->renderer == GFXH_REARCHITECTED_RESTIR_UNBIASED;
    constexpr size_t numPreSampledLights = 128 * 1024;     // numLightSubsets * lightSubsetSize, restir_di_shared.h:8-9
    if (rearch) {
        err |= alloc_dev(r, &sp.lightPreSamplingRngs, 8 * numPreSampledLights, false);
        err |= alloc_dev(r, &sp.preSampledLights, 48 * numPreSampledLights, true);
    }
    if (cfg->renderer == GFXH_PATH_TRACE_REGIR) {          // regir_main.cpp:1071-1097
        gfx_regir_params& g = r->regir;
        std::memset(&g, 0, sizeof(g));
        const uint32_t* d = cfg->regirGridDimension;
        const size_t numCells = static_cast<size_t>(d[0]) * d[1] * d[2], numSlots = numCells * 512;
        if (numCells == 0) { g_driverError = "gfxh_restir_create: empty ReGIR grid"; gfxh_restir_destroy(r); return 1; }
        for (int i = 0; i < 2; ++i) {
            err |= alloc_dev(r, &g.reservoirs[i], 48 * numSlots, true);
            err |= alloc_dev(r, &g.reservoirInfos[i], 8 * numSlots, true);
            err |= alloc_dev(r, &g.numActiveCells[i], 4, true);
        }
        err |= alloc_dev(r, &g.lightSlotRngs, 8 * numSlots, false);
        err |= alloc_dev(r, &g.perCellNumAccesses, 4 * numCells, true);
        err |= alloc_dev(r, &g.lastAccessFrameIndices, 4 * numCells, false);
        if (!err) {
            std::vector<uint64_t> states(numSlots);
            gfxh_seed_rng_states(states.data(), numSlots, 591842031321323413ull);
            if (!hip_ok(hipMemcpy(g.lightSlotRngs, states.data(), 8 * numSlots, hipMemcpyHostToDevice), "upload light-slot rng states") ||
                !hip_ok(hipMemset(g.lastAccessFrameIndices, 0xFF, 4 * numCells), "fill lastAccessFrameIndices")) err = 1;
        }
        for (int k = 0; k < 3; ++k) {
            g.gridOrigin[k] = cfg->regirAabbMin[k];
            g.gridCellSize[k] = (cfg->regirAabbMax[k] - cfg->regirAabbMin[k]) / static_cast<float>(d[k]);
            g.gridDimension[k] = d[k];
        }
        g.log2NumCandidatesPerLightSlot = cfg->regirLog2CandidatesPerLightSlot;
        g.log2NumCandidatesPerCell = cfg->regirLog2CandidatesPerCell;
        g.enableCellRandomization = cfg->regirEnableCellRandomization;
    }
    if (err) { gfxh_restir_destroy(r); return 1; }
    sp.spatialNeighborDeltas = deltas;
    sp.numTilesX = (cfg->width + 7) / 8; sp.numTilesY = (cfg->height + 7) / 8;
    {
        // pixel RNGs: row-major mt19937_64(591842031321323413) (restir_di_main.cpp:1316-1321)
        std::vector<uint64_t> states(n);
        gfxh_seed_rng_states(states.data(), n, 591842031321323413ull);
        if (!hip_ok(hipMemcpy(sp.rngBuffer, states.data(), 8 * n, hipMemcpyHostToDevice), "upload rng states")) { gfxh_restir_destroy(r); return 1; }
        if (rearch) {   // restir_di_main.cpp:1216-1219
            std::vector<uint64_t> pre(numPreSampledLights);
            gfxh_seed_rng_states(pre.data(), numPreSampledLights, 894213312210ull);
            if (!hip_ok(hipMemcpy(sp.lightPreSamplingRngs, pre.data(), 8 * numPreSampledLights, hipMemcpyHostToDevice), "upload pre-sampling rng states")) { gfxh_restir_destroy(r); return 1; }
        }
        std::vector<float> d(2048);
        gfxh_spatial_neighbor_deltas(d.data());
        if (!hip_ok(hipMemcpy(deltas, d.data(), 8 * 1024, hipMemcpyHostToDevice), "upload neighbour table")) { gfxh_restir_destroy(r); return 1; }
    }
    // scene.updateASs + setupLightGeomDistributions (restir_di_main.cpp:1199, 2263-2264)
    if (gfx_accel_build(ctx, nullptr, &r->accel) || gfx_lights_build_static(ctx, nullptr)) {
        g_driverError = gfx_last_error(ctx);
        gfxh_restir_destroy(r);
        return 1;
    }
    r->camera = cfg->camera;
    r->prevCamera = cfg->camera;
    {
        const char* e = std::getenv("GFX_SERIAL_FRAMES");   // debugging aid: everything on the caller's stream
        r->pipelineFrames = !(e && e[0] == '1');
        if (!hip_ok(hipStreamCreateWithFlags(&r->gbStream, hipStreamNonBlocking), "hipStreamCreateWithFlags") ||
            !hip_ok(hipEventCreateWithFlags(&r->evPrevRead, hipEventDisableTiming), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&r->evGbuffer, hipEventDisableTiming), "hipEventCreate")) {
            gfxh_restir_destroy(r);
            return 1;
        }
    }
    *out = r;
    return 0;
}

void gfxh_restir_destroy(gfxh_restir* r) {
    if (!r) return;
    (void)hipDeviceSynchronize();
    if (r->evPrevRead) (void)hipEventDestroy(r->evPrevRead);
    if (r->evGbuffer) (void)hipEventDestroy(r->evGbuffer);
    if (r->gbStream) (void)hipStreamDestroy(r->gbStream);
    for (void* p : r->allocations) (void)hipFree(p);
    delete r;
}

int gfxh_restir_band_plan(gfxh_restir* r, gfxh_band_plan* out) {
    const gfxh_restir_config& cfg = r->cfg;
    const bool whole = cfg.rowBegin == 0 && cfg.rowEnd == 0;
    const uint32_t passes = cfg.enableSpatialReuse ? cfg.numSpatialReusePasses : 0;
    gfxh_band_plan_compute(cfg.height, whole ? 0 : cfg.rowBegin, whole ? cfg.height : cfg.rowEnd,
                           static_cast<uint32_t>(std::ceil(cfg.spatialNeighborRadius)), passes, 0, out);
    return 0;
}

int gfxh_restir_reset(gfxh_restir* r) { r->resetRequested = true; return 0; }

int gfxh_restir_set_env(gfxh_restir* r, float* texels, uint32_t w, uint32_t h, float powerCoeff, float rotation) {
    const size_t n = static_cast<size_t>(w) * h;
    std::vector<float> rowPDF(n), rowCDF(static_cast<size_t>(h) * (w + 1)), rowInt(h), topPDF(h), topCDF(h + 1);
    float topIntegral = 0;
    gfxh_env_build_importance(texels, w, h, rowPDF.data(), rowCDF.data(), rowInt.data(), topPDF.data(), topCDF.data(), &topIntegral);
    gfx_restir_static_params& sp = r->sp;
    auto up = [&](const void** dst, const void* src, size_t bytes) {
        void* p = nullptr;
        if (alloc_dev(r, &p, bytes, false)) return 1;
        if (!hip_ok(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice), "upload env")) return 1;
        *dst = p;
        return 0;
    };
    int err = 0;
    err |= up(&sp.envLightTexture, texels, 16 * n);
    err |= up(&sp.envRowPDF, rowPDF.data(), 4 * rowPDF.size());
    err |= up(&sp.envRowCDF, rowCDF.data(), 4 * rowCDF.size());
    err |= up(&sp.envRowIntegrals, rowInt.data(), 4 * rowInt.size());
    err |= up(&sp.envTopPDF, topPDF.data(), 4 * topPDF.size());
    err |= up(&sp.envTopCDF, topCDF.data(), 4 * topCDF.size());
    sp.envRowGuide = nullptr; sp.envTopGuide = nullptr;
    {
        std::vector<uint16_t> rowGuide(n), topGuide(h);
        if (gfxh_env_build_guides(rowCDF.data(), topCDF.data(), w, h, rowGuide.data(), topGuide.data())) {
            err |= up(&sp.envRowGuide, rowGuide.data(), 2 * rowGuide.size());
            err |= up(&sp.envTopGuide, topGuide.data(), 2 * topGuide.size());
        }
    }
    if (err) return 1;
    sp.envWidth = static_cast<int32_t>(w); sp.envHeight = static_cast<int32_t>(h); sp.envTopIntegral = topIntegral;
    r->envPowerCoeff = powerCoeff; r->envRotation = rotation;
    r->resetRequested = true;
    return 0;
}
int gfxh_restir_set_camera(gfxh_restir* r, const gfx_camera* cam) { r->camera = *cam; return 0; }
int gfxh_restir_rebuild_accel(gfxh_restir* r, void* stream) {
    // in place: the handle stays valid.  Ordered after everything queued on `stream`; the pipelined G-buffer pass
    // of the last frame was joined into that stream before its later passes were queued.
    if (gfx_accel_build(r->ctx, stream, &r->accel)) { g_driverError = gfx_last_error(r->ctx); return 1; }
    // the next frame's pipelined G-buffer pass must not start before the build
    if (r->evPrevRead && !hip_ok(hipEventRecord(r->evPrevRead, static_cast<hipStream_t>(stream)), "hipEventRecord")) return 1;
    r->prevReadPending = r->evPrevRead != nullptr;
    return 0;
}
void* gfxh_restir_beauty_buffer(gfxh_restir* r) { return r->sp.beautyAccumBuffer; }
uint64_t gfxh_restir_accel(gfxh_restir* r) { return r->accel; }

int gfxh_restir_get_params(gfxh_restir* r, gfx_restir_static_params* s, gfx_restir_frame_params* f,
                           uint32_t* lastReservoirIndex, uint32_t* lastSpatialNeighborBaseIndex, uint32_t* frameIndex) {
    if (s) *s = r->sp;
    if (f) *f = r->fp;
    if (lastReservoirIndex) *lastReservoirIndex = r->lastReservoirIndex;
    if (lastSpatialNeighborBaseIndex) *lastSpatialNeighborBaseIndex = r->lastSpatialNeighborBaseIndex;
    if (frameIndex) *frameIndex = r->frameIndex;
    return 0;
}

int gfxh_restir_render_frame(gfxh_restir* r, void* stream) {
    gfx_ctx* ctx = r->ctx;
    const gfxh_restir_config& cfg = r->cfg;
    const uint32_t frameIndex = r->frameIndex;
    const uint32_t bufferIndex = frameIndex % 2;                       // :1695
    gfx_restir_frame_params& fp = r->fp;
    // prevCamera = camera latched at the top of every frame (:1699), then the camera may move
    fp.prevCamera = frameIndex == 0 ? r->camera : r->prevCamera;
    fp.camera = r->camera;

    // scene.setupLightInstDistribution every frame (:2303-2309)
    if (gfx_lights_build_instances(ctx, stream, bufferIndex)) { g_driverError = gfx_last_error(ctx); return 1; }

    const bool newSequence = frameIndex == 0 || r->resetRequested;     // :2311 (no resize in a headless run)
    r->resetRequested = false;
    const bool firstAccumFrame = !cfg.enableAccumulation || newSequence;   // :2312-2313 (no animation / camera motion)
    if (firstAccumFrame) r->numAccumFrames = 0;
    else r->numAccumFrames = std::min(r->numAccumFrames + 1, 1u << cfg.log2MaxNumAccums);

    fp.travHandle = r->accel;
    fp.numAccumFrames = r->numAccumFrames;
    fp.frameIndex = frameIndex;
    fp.envLightPowerCoeff = r->envPowerCoeff;                          // pow(10, log10EnvLightPowerCoeff) (:2322)
    fp.envLightRotation = r->envRotation;
    fp.spatialNeighborRadius = cfg.spatialNeighborRadius;
    fp.radiusThresholdForSpatialVisReuse = 10.0f;
    fp.log2NumCandidateSamples = cfg.log2NumCandidateSamples;
    fp.numSpatialNeighbors = cfg.numSpatialNeighbors;
    fp.useLowDiscrepancyNeighbors = cfg.useLowDiscrepancyNeighbors;
    fp.reuseVisibility = cfg.reuseVisibility;
    fp.reuseVisibilityForTemporal = 1;
    fp.reuseVisibilityForSpatiotemporal = 0;
    fp.enableTemporalReuse = cfg.enableTemporalReuse;
    fp.enableSpatialReuse = cfg.enableSpatialReuse;
    fp.useUnbiasedEstimator = cfg.renderer == GFXH_ORIGINAL_RESTIR_UNBIASED || cfg.renderer == GFXH_REARCHITECTED_RESTIR_UNBIASED;
    fp.bufferIndex = bufferIndex;
    fp.resetFlowBuffer = newSequence;
    fp.enableJittering = cfg.enableJittering;
    fp.enableEnvLight = r->sp.envLightTexture != nullptr;
    fp.enableBumpMapping = cfg.enableBumpMapping;
    fp.useSolidAngleSampling = 0;

    uint32_t currentReservoirIndex = (r->lastReservoirIndex + 1) % 2;  // :2352
    const uint32_t W = cfg.width, H = cfg.height;
    gfxh_band_plan plan;
    gfxh_restir_band_plan(r, &plan);
#define DRV_GFX(call) do { if (call) { g_driverError = gfx_last_error(ctx); return 1; } } while (0)
    // G-buffer pass, pipelined under the previous frame when nothing forbids it: jittering advances the pixel
    // RNGs the previous frame's passes are still drawing from, and a band renderer's halo exchange has its own
    // ordering with the caller's stream.
    hipStream_t main = static_cast<hipStream_t>(stream);
    const bool wholeFrame = cfg.rowBegin == 0 && cfg.rowEnd == 0;
    const bool pipelined = r->pipelineFrames && !cfg.enableJittering && wholeFrame;
    auto gbuffer_pass = [&](bool pathTraceEntry, uint32_t rowBegin, uint32_t rowEnd) -> int {
        hipStream_t s = main;
        if (pipelined) {
            s = r->gbStream;
            if (r->prevReadPending) DRV_HIP(hipStreamWaitEvent(s, r->evPrevRead, 0));
            else { DRV_HIP(hipEventRecord(r->evPrevRead, main)); DRV_HIP(hipStreamWaitEvent(s, r->evPrevRead, 0)); }   // first frame: after whatever the caller queued
        }
        if (pathTraceEntry) DRV_GFX(gfx_pt_launch(ctx, s, GFX_PT_SETUP_GBUFFERS, W, H, cfg.maxPathLength, rowBegin, rowEnd));
        else DRV_GFX(gfx_restir_launch_rows(ctx, s, GFX_RESTIR_SETUP_GBUFFERS, W, H, rowBegin, rowEnd));
        if (pipelined) {
            DRV_HIP(hipEventRecord(r->evGbuffer, s));
            DRV_HIP(hipStreamWaitEvent(main, r->evGbuffer, 0));
        }
        return 0;
    };
    // call once the frame has queued its last pass that reads the previous frame's G-buffer
    auto prev_gbuffer_released = [&]() -> int {
        if (pipelined) { DRV_HIP(hipEventRecord(r->evPrevRead, main)); r->prevReadPending = true; }
        return 0;
    };
    DRV_GFX(gfx_restir_set_params(ctx, stream, &r->sp, &fp, currentReservoirIndex, r->lastSpatialNeighborBaseIndex));
    if (cfg.renderer == GFXH_PATH_TRACE_REGIR) {
        // regir_main.cpp:2021-2066: G-buffer, cell reservoirs (+ temporal reuse unless a new sequence), ReGIR path
        // tracing, last-access update.  Whole frame (the grid is shared state).
        DRV_GFX(gfx_regir_set_params(ctx, &r->regir));
        if (gbuffer_pass(true, 0, 0) || prev_gbuffer_released()) return 1;
        const int build = (cfg.regirEnableTemporalReuse && !newSequence) ? GFX_PT_REGIR_BUILD_CELL_RESERVOIRS_TEMPORAL : GFX_PT_REGIR_BUILD_CELL_RESERVOIRS;
        DRV_GFX(gfx_pt_launch(ctx, stream, build, W, H, cfg.maxPathLength, 0, 0));
        DRV_GFX(gfx_pt_launch(ctx, stream, GFX_PT_PATH_TRACE_REGIR, W, H, cfg.maxPathLength, 0, 0));
        DRV_GFX(gfx_pt_launch(ctx, stream, GFX_PT_REGIR_UPDATE_LAST_ACCESS, W, H, cfg.maxPathLength, 0, 0));
        r->prevCamera = r->camera;
        ++r->frameIndex;
        return 0;
    }
    if (cfg.renderer == GFXH_PATH_TRACE_BASELINE) {
        // path_tracing_main.cpp:2068-2093: G-buffer pipeline, then pathTraceBaseline.  A band needs no
        // halo: paths never read a neighbour's pixel state.
        if (gbuffer_pass(true, plan.bandBegin, plan.bandEnd) || prev_gbuffer_released()) return 1;
        DRV_GFX(gfx_pt_launch(ctx, stream, GFX_PT_PATH_TRACE_BASELINE, W, H, cfg.maxPathLength, plan.bandBegin, plan.bandEnd));
        r->prevCamera = r->camera;
        ++r->frameIndex;
        return 0;
    }
    if (gbuffer_pass(false, plan.gbufferRows[0], plan.gbufferRows[1])) return 1;                  // :2366-2367

    if (cfg.renderer == GFXH_REARCHITECTED_RESTIR_BIASED || cfg.renderer == GFXH_REARCHITECTED_RESTIR_UNBIASED) {
        // restir_di_main.cpp:2423-2487; whole-frame only (the previous frame's reservoirs, sample
        // visibility and G-buffers of halo rows are not exchanged for this renderer yet)
        const bool T = cfg.enableTemporalReuse && !newSequence, S = cfg.enableSpatialReuse && !newSequence;
        const int k = (T && S) ? 3 : T ? 1 : S ? 2 : 0;
        const int trace = GFX_RESTIR_TRACE_SHADOW_RAYS + (k == 0 ? 0 : k + (fp.useUnbiasedEstimator ? 3 : 0));
        const int shade = GFX_RESTIR_SHADE_AND_RESAMPLE + k;
        DRV_GFX(gfx_restir_launch(ctx, stream, GFX_RESTIR_LIGHT_PRESAMPLING, W, H));
        DRV_GFX(gfx_restir_launch(ctx, stream, GFX_RESTIR_PER_PIXEL_RIS, W, H));
        DRV_GFX(gfx_restir_launch(ctx, stream, trace, W, H));
        DRV_GFX(gfx_restir_launch(ctx, stream, shade, W, H));
        if (prev_gbuffer_released()) return 1;   // shadeAndResample reads the previous G-buffer and sample visibility
        ++r->lastSpatialNeighborBaseIndex;                                                     // :2486
        r->lastReservoirIndex = currentReservoirIndex;
        r->prevCamera = r->camera;
        ++r->frameIndex;
        return 0;
    }

    int entry = GFX_RESTIR_INITIAL_RIS;                                                        // :2378-2384
    if (cfg.enableTemporalReuse && !newSequence)
        entry = fp.useUnbiasedEstimator ? GFX_RESTIR_INITIAL_AND_TEMPORAL_UNBIASED : GFX_RESTIR_INITIAL_AND_TEMPORAL_BIASED;
    DRV_GFX(gfx_restir_launch_rows(ctx, stream, entry, W, H, plan.initialRows[0], plan.initialRows[1]));
    if (prev_gbuffer_released()) return 1;   // only the temporal pass reads the previous frame's G-buffer

    if (cfg.enableSpatialReuse) {                                                              // :2393-2411
        const int spatial = fp.useUnbiasedEstimator ? GFX_RESTIR_SPATIAL_UNBIASED : GFX_RESTIR_SPATIAL_BIASED;
        for (uint32_t i = 0; i < cfg.numSpatialReusePasses; ++i) {
            const uint32_t baseIndex = r->lastSpatialNeighborBaseIndex + cfg.numSpatialNeighbors * i;
            DRV_GFX(gfx_restir_set_params(ctx, stream, nullptr, nullptr, currentReservoirIndex, baseIndex));
            DRV_GFX(gfx_restir_launch_rows(ctx, stream, spatial, W, H, plan.spatialRows[i][0], plan.spatialRows[i][1]));
            currentReservoirIndex = (currentReservoirIndex + 1) % 2;
        }
        r->lastSpatialNeighborBaseIndex += cfg.numSpatialNeighbors * cfg.numSpatialReusePasses;
    }
    DRV_GFX(gfx_restir_set_params(ctx, stream, nullptr, nullptr, currentReservoirIndex, r->lastSpatialNeighborBaseIndex));
    DRV_GFX(gfx_restir_launch_rows(ctx, stream, GFX_RESTIR_SHADING, W, H, plan.shadingRows[0], plan.shadingRows[1]));   // :2418-2420
#undef DRV_GFX
    r->lastReservoirIndex = currentReservoirIndex;                                             // :2493
    r->prevCamera = r->camera;
    ++r->frameIndex;
    return 0;
}

const char* gfxh_restir_last_error(void) { return g_driverError.c_str(); }

} // extern "C"
