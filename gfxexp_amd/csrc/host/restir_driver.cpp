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

// Flags of the events that only order kernels of THIS device across streams: no system-scope fence at the record (its cache write-back
// and invalidation cost the kernels behind it; hip_runtime_api.h hipEventDisableSystemFence).  GFX_EVENT_SYSTEM_FENCE=1: the default flags.
static unsigned order_event_flags() {
    static const unsigned flags = [] { const char* e = std::getenv("GFX_EVENT_SYSTEM_FENCE"); return (e && e[0] == '1') ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence); }();
    return flags;
}
}

struct gfxh_restir {
    gfx_ctx* ctx = nullptr;
    gfxh_restir_config cfg;
    gfx_restir_static_params sp;
    gfx_restir_frame_params fp;
    std::vector<void*> allocations;
    std::vector<void*> envAllocations;          // the environment map's tables (gfxh_restir_set_env): replaced as a set
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
    // gfxh_restir_outputs_consumed: the caller's stream has passed its reads of the albedo / normal accumulators the next G-buffer pass rewrites
    hipEvent_t evConsumed = nullptr;
    bool consumedPending = false;
    // strip-exchange mode of a band renderer (gfxh_restir_set_exchange)
    gfxh_exchange_fn exchange = nullptr;
    void* exchangeUser = nullptr;
    uint32_t maxMotionRows = 0;
    bool viewMoved = false;     // the camera or an instance moved since the last frame (accumulation restarts; band seams need motion rows)
    // lanes of a band renderer (gfxexp_host.h gfxh_lane): the G-buffer strips travel on gbStream behind the pass that made them
    // (evGbStrips: they have arrived), the HDR bands on gatherStream underneath the next frame (evBandDone: the frame's last pass is
    // queued; evGather: the gather has finished)
    hipEvent_t evGbStrips = nullptr, evBandDone = nullptr, evGather = nullptr, evSeamRows = nullptr, evSeamStrips = nullptr;
    hipStream_t gatherStream = nullptr, seamStream = nullptr;
    // GFX_SEAM_FIRST=1: stripMode 2 (the seam rows of the first biased spatial pass in a launch of their own, their exchange on the seam lane).
    // Off by default: on one GPU with a transport of the same shape the split costs more than the exchange it hides (the pass is ~50 us at
    // eight bands: profiles/r06_band_host_overhead.json, r06_experiments.txt 2)
    bool seamStripsPending = false, seamFirst = false;
    int stripMode = 3;          // GFX_STRIP_MODE = 1 | 2 | 3 (gfxh_restir_frame_program): 3 = intermediate spatial passes recomputed on their halo (the default)
    bool asyncGather = false, gatherPending = false, gbStripsPending = false;
    bool stripsOnGbLane = true;   // GFX_GB_STRIPS_ON_MAIN=1 (A/B runs): the G-buffer strips on the caller's stream ahead of the candidate pass, as rounds 2-5 issued them
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
        const char* m = std::getenv("GFX_GB_STRIPS_ON_MAIN");
        r->stripsOnGbLane = !(m && m[0] == '1');
        const char* sf = std::getenv("GFX_SEAM_FIRST");
        r->seamFirst = sf && sf[0] == '1';
        const char* sm = std::getenv("GFX_STRIP_MODE");
        if (sm && sm[0] >= '1' && sm[0] <= '3') r->stripMode = sm[0] - '0';
        if (r->seamFirst) r->stripMode = 2;
        if (!hip_ok(hipStreamCreateWithFlags(&r->gbStream, hipStreamNonBlocking), "hipStreamCreateWithFlags") ||
            !hip_ok(hipEventCreateWithFlags(&r->evPrevRead, order_event_flags()), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&r->evGbuffer, order_event_flags()), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&r->evGbStrips, order_event_flags()), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&r->evBandDone, order_event_flags()), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&r->evGather, order_event_flags()), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&r->evSeamRows, order_event_flags()), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&r->evSeamStrips, order_event_flags()), "hipEventCreate") ||
            !hip_ok(hipStreamCreateWithFlags(&r->seamStream, hipStreamNonBlocking), "hipStreamCreateWithFlags") ||
            !hip_ok(hipStreamCreateWithFlags(&r->gatherStream, hipStreamNonBlocking), "hipStreamCreateWithFlags")) {
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
    if (r->evConsumed) (void)hipEventDestroy(r->evConsumed);
    if (r->evGbuffer) (void)hipEventDestroy(r->evGbuffer);
    if (r->evGbStrips) (void)hipEventDestroy(r->evGbStrips);
    if (r->evBandDone) (void)hipEventDestroy(r->evBandDone);
    if (r->evGather) (void)hipEventDestroy(r->evGather);
    if (r->gbStream) (void)hipStreamDestroy(r->gbStream);
    if (r->gatherStream) (void)hipStreamDestroy(r->gatherStream);
    if (r->seamStream) (void)hipStreamDestroy(r->seamStream);
    if (r->evSeamRows) (void)hipEventDestroy(r->evSeamRows);
    if (r->evSeamStrips) (void)hipEventDestroy(r->evSeamStrips);
    for (void* p : r->allocations) (void)hipFree(p);
    for (void* p : r->envAllocations) (void)hipFree(p);
    delete r;
}

int gfxh_strip_rows(uint32_t height, uint32_t bandBegin, uint32_t bandEnd, uint32_t rows, gfxh_exchange_desc* out) {
    const uint32_t b = bandBegin, e = bandEnd;
    out->recvAbove[0] = b > rows ? b - rows : 0u; out->recvAbove[1] = b;
    out->recvBelow[0] = e; out->recvBelow[1] = std::min(height, e + rows);
    out->sendAbove[0] = b; out->sendAbove[1] = b == 0 ? b : std::min(e, b + rows);
    out->sendBelow[0] = e >= height ? e : (e - b > rows ? e - rows : b); out->sendBelow[1] = e;
    // a strip taller than the band would have to come from a rank further away than the adjacent one
    return (rows > e - b && (b > 0 || e < height)) ? 1 : 0;
}

int gfxh_restir_set_exchange(gfxh_restir* r, gfxh_exchange_fn fn, void* user, uint32_t maxMotionRows) {
    r->exchange = fn; r->exchangeUser = user; r->maxMotionRows = maxMotionRows;
    return 0;
}

int gfxh_restir_set_async_gather(gfxh_restir* r, int enable) {
    if (!r) { g_driverError = "gfxh_restir_set_async_gather: null renderer"; return 1; }
    r->asyncGather = enable != 0;
    return 0;
}

int gfxh_restir_finish_gather(gfxh_restir* r, void* stream) {
    if (!r) { g_driverError = "gfxh_restir_finish_gather: null renderer"; return 1; }
    if (r->gatherPending && !hip_ok(hipStreamWaitEvent(static_cast<hipStream_t>(stream), r->evGather, 0), "hipStreamWaitEvent")) return 1;
    return 0;
}

int gfxh_band_rows(uint32_t height, uint32_t world, uint32_t rank, uint32_t* begin, uint32_t* end) {
    if (world == 0 || rank >= world) return 1;
    const uint32_t tiles = (height + 7) / 8, base = tiles / world, extra = tiles % world;
    uint32_t row = 0;
    for (uint32_t r = 0; r <= rank; ++r) {
        const uint32_t h = (base + (r < extra ? 1u : 0u)) * 8;
        *begin = row; row = std::min(height, row + h); *end = row;
    }
    return 0;
}

// Strip feasibility as a decision every rank takes identically: the tallest strip any frame of this configuration
// exchanges (new sequence or not) against the SMALLEST band of the partition.  gfxh_strip_rows alone looks at the
// calling rank's band: with 1080 rows over 8 ranks (7 x 136 + 128) a 130-row strip passes on ranks 0-6 and fails
// on rank 7, whose neighbours would already be inside the collective.
int gfxh_restir_check_partition(const gfxh_restir_config* cfgp, uint32_t world, uint32_t maxMotionRows) {
    if (world <= 1) return 0;
    uint32_t minBand = 0xFFFFFFFFu;
    for (uint32_t rank = 0; rank < world; ++rank) {
        uint32_t b = 0, e = 0;
        if (gfxh_band_rows(cfgp->height, world, rank, &b, &e)) return 1;
        minBand = std::min(minBand, e - b);
    }
    if (minBand == 0) { g_driverError = "gfxh_restir_check_partition: more ranks than 8-row tiles"; return 1; }
    gfxh_restir_config cfg = *cfgp;
    // any interior band will do: the program's exchange rows do not depend on where the band lies
    gfxh_band_rows(cfg.height, world, 0, &cfg.rowBegin, &cfg.rowEnd);
    const uint32_t unbiased = cfg.renderer == GFXH_ORIGINAL_RESTIR_UNBIASED || cfg.renderer == GFXH_REARCHITECTED_RESTIR_UNBIASED;
    uint32_t tallest = 0;
    for (int newSequence = 0; newSequence < 2; ++newSequence) {
        gfxh_frame_step steps[64];
        uint32_t n = 0, a = 0, c = 0;
        (void)gfxh_restir_frame_program(&cfg, 3, maxMotionRows, newSequence, 1, 0, unbiased, steps, 64, &n, &a, &c);   // (mode 3 has the tallest strips)
        for (uint32_t k = 0; k < n; ++k) if (steps[k].op == GFXH_STEP_EXCHANGE_STRIPS) tallest = std::max(tallest, steps[k].exchangeRows);
    }
    if (tallest > minBand) {
        g_driverError = "gfxh_restir_check_partition: an exchange strip of " + std::to_string(tallest) + " rows (spatial radius, motion rows) is taller than the smallest band ("
                        + std::to_string(minBand) + " rows of " + std::to_string(cfg.height) + " over " + std::to_string(world) + " ranks): fewer ranks, a smaller radius or fewer motion rows";
        return 1;
    }
    return 0;
}

static uint32_t tallest_strip(const gfxh_restir_config* cfgp, uint32_t firstBandEnd, uint32_t maxMotionRows) {
    gfxh_restir_config cfg = *cfgp;
    cfg.rowBegin = 0; cfg.rowEnd = firstBandEnd;   // any band will do: the program's exchange rows do not depend on where the band lies
    const uint32_t unbiased = cfg.renderer == GFXH_ORIGINAL_RESTIR_UNBIASED || cfg.renderer == GFXH_REARCHITECTED_RESTIR_UNBIASED;
    uint32_t tallest = 0;
    for (int newSequence = 0; newSequence < 2; ++newSequence) {
        gfxh_frame_step steps[64];
        uint32_t n = 0, a = 0, c = 0;
        (void)gfxh_restir_frame_program(&cfg, 3, maxMotionRows, newSequence, 1, 0, unbiased, steps, 64, &n, &a, &c);   // (mode 3 has the tallest strips)
        for (uint32_t k = 0; k < n; ++k) if (steps[k].op == GFXH_STEP_EXCHANGE_STRIPS) tallest = std::max(tallest, steps[k].exchangeRows);
    }
    return tallest;
}

int gfxh_restir_check_bands(const gfxh_restir_config* cfgp, uint32_t world, const uint32_t* bandBegin, uint32_t maxMotionRows) {
    if (world <= 1) return 0;
    if (!bandBegin || bandBegin[0] != 0 || bandBegin[world] != cfgp->height) { g_driverError = "gfxh_restir_check_bands: the partition must run from row 0 to the image height"; return 1; }
    uint32_t minBand = 0xFFFFFFFFu;
    for (uint32_t r = 0; r < world; ++r) {
        if (bandBegin[r + 1] <= bandBegin[r] || (r + 1 < world && bandBegin[r + 1] % 8u != 0)) {
            g_driverError = "gfxh_restir_check_bands: band boundaries must ascend on whole 8-row tiles";
            return 1;
        }
        minBand = std::min(minBand, bandBegin[r + 1] - bandBegin[r]);
    }
    const uint32_t tallest = tallest_strip(cfgp, bandBegin[1], maxMotionRows);
    if (tallest > minBand) {
        g_driverError = "gfxh_restir_check_bands: an exchange strip of " + std::to_string(tallest) + " rows is taller than the smallest band (" + std::to_string(minBand) + " rows)";
        return 1;
    }
    return 0;
}

int gfxh_balance_bands(uint32_t height, uint32_t world, const uint32_t* in, const float* ms, uint32_t minRows, uint32_t* out) {
    if (world == 0 || !in || !ms || !out || in[0] != 0 || in[world] != height) return 1;
    const uint32_t tiles = (height + 7u) / 8u;
    const uint32_t minTiles = std::max(1u, (minRows + 7u) / 8u);
    // every band but the last is whole tiles; the last one ends at `height`, which need not be a multiple of 8: it is measured in rows
    if (static_cast<uint64_t>(minTiles) * 8u * (world - 1u) + std::max(minRows, 1u) > height) return 1;
    // cost per 8-row tile: a band's time spread evenly over its tiles (double: the prefix sums decide the cuts)
    std::vector<double> cost(tiles, 0.0);
    double total = 0.0;
    for (uint32_t r = 0; r < world; ++r) {
        if (in[r + 1] <= in[r] || !(ms[r] > 0.0f) || !(ms[r] < 1e30f)) return 1;
        if (r > 0 && in[r] % 8u != 0) return 1;                     // interior boundaries sit on tile edges (gfxh_restir_check_bands)
        const uint32_t t0 = in[r] / 8u, t1 = (in[r + 1] + 7u) / 8u;
        for (uint32_t t = t0; t < t1 && t < tiles; ++t) cost[t] = static_cast<double>(ms[r]) / (t1 - t0);
        total += ms[r];
    }
    // cut k after the tile at which the running cost passes k / world of the total (nearest tile edge), each band at least
    // minTiles tall and leaving room for the bands behind it
    out[0] = 0;
    uint32_t tile = 0;
    double run = 0.0;
    for (uint32_t k = 1; k < world; ++k) {
        const double want = total * k / world;
        // bands k .. world - 2 behind the cut are at least minTiles tiles, the last one at least minRows ROWS of a possibly partial tile
        const uint32_t lo = out[k - 1] / 8u + minTiles, hi = (height - ((world - k - 1u) * minTiles * 8u + std::max(minRows, 1u))) / 8u;
        if (hi < lo) return 1;
        while (tile < tiles && run + cost[tile] <= want) run += cost[tile++];
        uint32_t cut = tile;
        if (tile < tiles && (want - run) > (run + cost[tile] - want)) cut = tile + 1;   // the nearer edge
        cut = std::min(std::max(cut, lo), hi);
        out[k] = cut * 8u;
    }
    out[world] = height;
    return 0;
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

int gfxh_env_upload(float* texels, uint32_t w, uint32_t h, gfx_restir_static_params* spOut, void** allocations, uint32_t* numAllocations) {
    const size_t n = static_cast<size_t>(w) * h;
    std::vector<float> rowPDF(n), rowCDF(static_cast<size_t>(h) * (w + 1)), rowInt(h), topPDF(h), topCDF(h + 1);
    float topIntegral = 0;
    gfxh_env_build_importance(texels, w, h, rowPDF.data(), rowCDF.data(), rowInt.data(), topPDF.data(), topCDF.data(), &topIntegral);
    gfx_restir_static_params& sp = *spOut;
    *numAllocations = 0;
    auto up = [&](const void** dst, const void* src, size_t bytes) {
        void* p = nullptr;
        if (*numAllocations >= GFXH_ENV_MAX_ALLOCATIONS) return 1;
        if (!hip_ok(hipMalloc(&p, bytes), "hipMalloc (environment map)")) return 1;
        allocations[(*numAllocations)++] = p;
        if (!hip_ok(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice), "upload env")) return 1;
        *dst = p;
        return 0;
    };
    int err = 0;
    err |= up(&sp.envLightTexture, texels, 16 * n);
    err |= up(&sp.envRowIntegrals, rowInt.data(), 4 * rowInt.size());
    err |= up(&sp.envTopPDF, topPDF.data(), 4 * topPDF.size());
    err |= up(&sp.envTopCDF, topCDF.data(), 4 * topCDF.size());
    sp.envRowPDF = nullptr; sp.envRowCDF = nullptr;
    sp.envRowGuide = nullptr; sp.envTopGuide = nullptr; sp.envRowTable = nullptr; sp.envRowSketch = nullptr;
    bool haveTable = false;
    {
        std::vector<uint16_t> rowGuide(n), topGuide(h);
        if (gfxh_env_build_guides(rowCDF.data(), topCDF.data(), w, h, rowGuide.data(), topGuide.data())) {
            err |= up(&sp.envTopGuide, topGuide.data(), 2 * topGuide.size());
            // the rows interleaved (cdf, pdf, guide, texel per record): what a light sample on the map reads, in two or three sectors
            // (GFX_ENV_ROW_TABLE=0: the separate arrays, for A/B runs)
            const char* e = std::getenv("GFX_ENV_ROW_TABLE");
            if (!(e && e[0] == '0')) {
                std::vector<uint32_t> table(8 * static_cast<size_t>(h) * GFX_ENV_ROW_STRIDE(w));
                gfxh_env_build_row_table(texels, rowPDF.data(), rowCDF.data(), rowGuide.data(), w, h, table.data());
                err |= up(&sp.envRowTable, table.data(), 4 * table.size());
                haveTable = !err;
                // ... and, with GFX_ENV_ROW_SKETCH=1, the rows' inverse-CDF sketches: a sample of a verified cell reads one line of the table and
                // no guide.  Off by default: the candidate pass of configs[4] then moves 3.45 GB through the memory-side counters instead of
                // 5.80 GB and takes 1-3 % LONGER (it runs at the L2s' sector rate and the VALU's, not HBM's: profiles/r06_experiments.txt 4)
                const char* sk = std::getenv("GFX_ENV_ROW_SKETCH");
                if (sk && sk[0] == '1') {
                    uint32_t numRecords = 0;
                    (void)gfxh_env_build_row_sketch(rowCDF.data(), w, h, nullptr, 0, &numRecords);
                    std::vector<uint32_t> sketch(static_cast<size_t>(numRecords) * GFX_ENV_SKETCH_WORDS);
                    (void)gfxh_env_build_row_sketch(rowCDF.data(), w, h, sketch.data(), numRecords, &numRecords);
                    err |= up(&sp.envRowSketch, sketch.data(), 4 * sketch.size());
                }
            }
            if (!haveTable) err |= up(&sp.envRowGuide, rowGuide.data(), 2 * rowGuide.size());
        }
    }
    // the separate row arrays are what the samplers read WITHOUT the interleaved table; with it they would be 10 bytes per texel of
    // device memory nothing reads (20 MB at 2048 x 1024, 320 MB at 8192 x 4096)
    if (!haveTable) {
        err |= up(&sp.envRowPDF, rowPDF.data(), 4 * rowPDF.size());
        err |= up(&sp.envRowCDF, rowCDF.data(), 4 * rowCDF.size());
    }
    if (err) return 1;
    sp.envWidth = static_cast<int32_t>(w); sp.envHeight = static_cast<int32_t>(h); sp.envTopIntegral = topIntegral;
    return 0;
}

int gfxh_restir_set_env(gfxh_restir* r, float* texels, uint32_t w, uint32_t h, float powerCoeff, float rotation) {
    // frames in flight may still read the tables this call replaces: wait for them, then the previous map's allocations can go
    DRV_HIP(hipDeviceSynchronize());
    for (void* p : r->envAllocations) (void)hipFree(p);
    r->envAllocations.clear();
    void* allocations[GFXH_ENV_MAX_ALLOCATIONS];
    uint32_t numAllocations = 0;
    const int err = gfxh_env_upload(texels, w, h, &r->sp, allocations, &numAllocations);
    for (uint32_t i = 0; i < numAllocations; ++i) r->envAllocations.push_back(allocations[i]);   // freed with the next map or the renderer
    if (err) return 1;
    r->envPowerCoeff = powerCoeff; r->envRotation = rotation;
    r->resetRequested = true;
    return 0;
}
int gfxh_restir_set_camera(gfxh_restir* r, const gfx_camera* cam) {
    if (std::memcmp(&r->camera, cam, sizeof(*cam)) != 0) r->viewMoved = true;
    r->camera = *cam;
    return 0;
}
int gfxh_restir_rebuild_accel(gfxh_restir* r, void* stream) {
    // in place: the handle stays valid.  Ordered after everything queued on `stream`; the pipelined G-buffer pass
    // of the last frame was joined into that stream before its later passes were queued.
    if (gfx_accel_build(r->ctx, stream, &r->accel)) { g_driverError = gfx_last_error(r->ctx); return 1; }
    // the next frame's pipelined G-buffer pass must not start before the build
    if (r->evPrevRead && !hip_ok(hipEventRecord(r->evPrevRead, static_cast<hipStream_t>(stream)), "hipEventRecord")) return 1;
    r->prevReadPending = r->evPrevRead != nullptr;
    r->viewMoved = true;   // "animate" of restir_di_main.cpp:2312-2313
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

// The frame of restir_di_main.cpp:2311-2493 (original and rearchitected ReSTIR), path_tracing_main.cpp:2068-2093 and
// regir_main.cpp:2021-2066 as a list of steps: the passes with their row ranges and the reservoir / neighbour-table
// indices in force, plus -- stripMode -- the exchange points of a band renderer (gfxexp_host.h, gfxh_restir_set_exchange).
// Pure host logic: the GPU driver executes it, and the multi-process CPU tests execute the same program with the oracle.
int gfxh_restir_frame_program(const gfxh_restir_config* cfgp, int stripMode, uint32_t maxMotionRows, int newSequence,
                              uint32_t lastReservoirIndex, uint32_t lastSpatialNeighborBaseIndex, uint32_t useUnbiasedEstimator,
                              gfxh_frame_step* steps, uint32_t capacity, uint32_t* numSteps, uint32_t* newLastReservoirIndex,
                              uint32_t* newLastSpatialNeighborBaseIndex) {
    const gfxh_restir_config& cfg = *cfgp;
    const bool whole = cfg.rowBegin == 0 && cfg.rowEnd == 0;
    const bool strips = stripMode && !whole;
    const bool seamFirst = strips && stripMode == 2;
    const uint32_t passes = cfg.enableSpatialReuse ? cfg.numSpatialReusePasses : 0;
    const uint32_t radiusRows = static_cast<uint32_t>(std::ceil(cfg.spatialNeighborRadius));
    gfxh_band_plan plan;
    gfxh_band_plan_compute(cfg.height, whole ? 0 : cfg.rowBegin, whole ? cfg.height : cfg.rowEnd, radiusRows, passes, 0, &plan);
    // stripMode 3: the spatial passes that another pass follows are RECOMPUTED on a halo that shrinks by `radius` rows per pass (the plan's
    // spatialRows), so only the first of them has an exchange in front of it -- radius x passes rows of the candidate pass's reservoirs and of
    // the pixel RNG states the halo rows' passes draw from -- and the reservoir exchanges between the spatial passes, which sit on the frame's
    // critical path around a pass of ~50 us, are gone.  A halo row's pass is the owner's computation on the owner's inputs: same bits.
    const bool recompute = strips && stripMode == 3 && passes >= 2;
    if (strips) {   // every pass on the band only (mode 3: the spatial passes but the last one also on their halo)
        plan.gbufferRows[0] = plan.initialRows[0] = plan.shadingRows[0] = plan.bandBegin;
        plan.gbufferRows[1] = plan.initialRows[1] = plan.shadingRows[1] = plan.bandEnd;
        if (!recompute) for (int i = 0; i < 8; ++i) { plan.spatialRows[i][0] = plan.bandBegin; plan.spatialRows[i][1] = plan.bandEnd; }
    }
    uint32_t n = 0;
    int tooTall = 0;
    uint32_t currentReservoirIndex = (lastReservoirIndex + 1) % 2;  // :2352
    uint32_t baseIndex = lastSpatialNeighborBaseIndex;
    auto push = [&](uint32_t op, uint32_t pass, uint32_t rb, uint32_t re, uint32_t lane = GFXH_LANE_MAIN) -> gfxh_frame_step* {
        if (n >= capacity) return nullptr;
        gfxh_frame_step& st = steps[n++];
        std::memset(&st, 0, sizeof(st));
        st.op = op; st.pass = pass; st.rowBegin = rb; st.rowEnd = re; st.lane = lane;
        st.currentReservoirIndex = currentReservoirIndex; st.spatialNeighborBaseIndex = baseIndex;
        return &st;
    };
    auto exchange = [&](uint32_t rows, uint32_t buffers, uint32_t reservoirIndex, uint32_t lane = GFXH_LANE_MAIN) {
        if (!strips || rows == 0) return false;
        gfxh_exchange_desc d;
        if (gfxh_strip_rows(cfg.height, plan.bandBegin, plan.bandEnd, rows, &d)) tooTall = 1;
        if (gfxh_frame_step* st = push(GFXH_STEP_EXCHANGE_STRIPS, 0, 0, 0, lane)) { st->exchangeRows = rows; st->buffers = buffers; st->reservoirIndex = reservoirIndex; }
        return true;
    };
    // the all-gather of the HDR bands runs on its own lane underneath the next frame; the pass that writes the beauty buffer
    // waits for the previous frame's gather first (it has read the band by then)
    auto beauty_writer = [&]() { if (strips) push(GFXH_STEP_WAIT_PREVIOUS_GATHER, 0, 0, 0); };
    auto gather = [&]() { if (strips) push(GFXH_STEP_GATHER_BANDS, 0, plan.bandBegin, plan.bandEnd, GFXH_LANE_GATHER); };
    const uint32_t motion = maxMotionRows;

    if (cfg.renderer == GFXH_PATH_TRACE_REGIR) {
        // regir_main.cpp:2021-2066: G-buffer, cell reservoirs (+ temporal reuse unless a new sequence), ReGIR path tracing,
        // last-access update.  The grid lives in world space: every rank builds all of it (same slot RNGs, same access
        // history) and traces its own rows; the per-cell access counters are summed over the ranks before they age the cells.
        const uint32_t rb = strips ? plan.bandBegin : 0, re = strips ? plan.bandEnd : 0;
        push(GFXH_STEP_PT_PASS, GFX_PT_SETUP_GBUFFERS, rb, re, GFXH_LANE_GBUFFER);
        push(GFXH_STEP_PREV_GBUFFER_RELEASED, 0, 0, 0);
        push(GFXH_STEP_PT_PASS, (cfg.regirEnableTemporalReuse && !newSequence) ? GFX_PT_REGIR_BUILD_CELL_RESERVOIRS_TEMPORAL : GFX_PT_REGIR_BUILD_CELL_RESERVOIRS, 0, 0);
        beauty_writer();
        push(GFXH_STEP_PT_PASS, GFX_PT_PATH_TRACE_REGIR, rb, re);
        if (strips) push(GFXH_STEP_ALLREDUCE_CELL_ACCESSES, 0, 0, 0);
        push(GFXH_STEP_PT_PASS, GFX_PT_REGIR_UPDATE_LAST_ACCESS, 0, 0);
        gather();
    }
    else if (cfg.renderer == GFXH_PATH_TRACE_BASELINE) {
        // path_tracing_main.cpp:2068-2093: G-buffer pipeline, then pathTraceBaseline.  Paths never read a neighbour's pixel state.
        push(GFXH_STEP_PT_PASS, GFX_PT_SETUP_GBUFFERS, plan.bandBegin, whole ? 0 : plan.bandEnd, GFXH_LANE_GBUFFER);
        push(GFXH_STEP_PREV_GBUFFER_RELEASED, 0, 0, 0);
        beauty_writer();
        push(GFXH_STEP_PT_PASS, GFX_PT_PATH_TRACE_BASELINE, plan.bandBegin, whole ? 0 : plan.bandEnd);
        gather();
    }
    else if (cfg.renderer == GFXH_REARCHITECTED_RESTIR_BIASED || cfg.renderer == GFXH_REARCHITECTED_RESTIR_UNBIASED) {
        // restir_di_main.cpp:2423-2487.  The pre-sampled lights are replicated (same RNG streams on every rank); the
        // per-pixel passes run on the band.  A band without an exchange callback renders the whole frame (its halo
        // would need the previous frame's state of other ranks).
        const uint32_t rb = strips ? plan.bandBegin : 0, re = strips ? plan.bandEnd : 0;
        push(GFXH_STEP_RESTIR_PASS, GFX_RESTIR_SETUP_GBUFFERS, rb, re, GFXH_LANE_GBUFFER);    // :2366-2367
        const bool T = cfg.enableTemporalReuse && !newSequence, S = cfg.enableSpatialReuse && !newSequence;
        const int k = (T && S) ? 3 : T ? 1 : S ? 2 : 0;
        const uint32_t trace = GFX_RESTIR_TRACE_SHADOW_RAYS + (k == 0 ? 0 : k + (useUnbiasedEstimator ? 3 : 0));
        push(GFXH_STEP_RESTIR_PASS, GFX_RESTIR_LIGHT_PRESAMPLING, 0, 0);
        push(GFXH_STEP_RESTIR_PASS, GFX_RESTIR_PER_PIXEL_RIS, rb, re);
        push(GFXH_STEP_RESTIR_PASS, trace, rb, re);
        beauty_writer();
        push(GFXH_STEP_RESTIR_PASS, GFX_RESTIR_SHADE_AND_RESAMPLE + k, rb, re);
        push(GFXH_STEP_PREV_GBUFFER_RELEASED, 0, 0, 0);   // shadeAndResample reads the previous G-buffer and sample visibility
        // everything the next frame reads as "the previous frame" around a pixel: its temporal neighbour (motion) and
        // its spatiotemporal neighbours (radius)
        const uint32_t rows = (cfg.enableSpatialReuse ? radiusRows : 0u) + (cfg.enableTemporalReuse ? motion : 0u);
        if (cfg.enableTemporalReuse || cfg.enableSpatialReuse)
            exchange(rows, GFXH_BUF_GBUFFERS | GFXH_BUF_SAMPLE_VISIBILITY | GFXH_BUF_RESERVOIRS, currentReservoirIndex);
        gather();
        ++baseIndex;                                                                           // :2486
    }
    else {
        push(GFXH_STEP_RESTIR_PASS, GFX_RESTIR_SETUP_GBUFFERS, plan.gbufferRows[0], whole ? 0 : plan.gbufferRows[1], GFXH_LANE_GBUFFER);   // :2366-2367
        // strip mode: the G-buffer rows the spatial passes (radius) and the next frame's temporal pass (motion) read.  They travel on
        // the G-buffer lane right behind the pass: the candidate + temporal pass of THIS frame reads no neighbour's current G-buffer
        // (the temporal reprojection reads the PREVIOUS frame's, whose strips arrived a frame ago), so only the first spatial pass
        // -- or, without one, the end of the frame -- waits for them
        bool gbStripsInFlight = exchange(std::max(cfg.enableSpatialReuse ? radiusRows * (recompute ? passes : 1u) : 0u, cfg.enableTemporalReuse ? motion : 0u), GFXH_BUF_GBUFFERS, 0, GFXH_LANE_GBUFFER);
        uint32_t entry = GFX_RESTIR_INITIAL_RIS;                                               // :2378-2384
        if (cfg.enableTemporalReuse && !newSequence)
            entry = useUnbiasedEstimator ? GFX_RESTIR_INITIAL_AND_TEMPORAL_UNBIASED : GFX_RESTIR_INITIAL_AND_TEMPORAL_BIASED;
        push(GFXH_STEP_RESTIR_PASS, entry, plan.initialRows[0], whole ? 0 : plan.initialRows[1]);
        push(GFXH_STEP_PREV_GBUFFER_RELEASED, 0, 0, 0);   // only the temporal pass reads the previous frame's G-buffer
        bool shadingIssued = false;
        if (cfg.enableSpatialReuse) {                                                          // :2393-2411
            const uint32_t spatial = useUnbiasedEstimator ? GFX_RESTIR_SPATIAL_UNBIASED : GFX_RESTIR_SPATIAL_BIASED;
            const uint32_t base0 = baseIndex;
            bool seamStripsInFlight = false;   // the reservoirs the pass in turn reads across the seams travel on lane SEAM (issued inside the pass before it)
            for (uint32_t i = 0; i < cfg.numSpatialReusePasses; ++i) {
                // strip mode: the reservoirs this pass resamples from, radius rows either side of the band
                if (seamStripsInFlight) { push(GFXH_STEP_WAIT_SEAM_STRIPS, 0, 0, 0); seamStripsInFlight = false; }
                else if (recompute) { if (i == 0) exchange(radiusRows * passes, GFXH_BUF_RESERVOIRS | GFXH_BUF_RNG, currentReservoirIndex); }
                else exchange(radiusRows, GFXH_BUF_RESERVOIRS, currentReservoirIndex);
                if (gbStripsInFlight) { push(GFXH_STEP_WAIT_GBUFFER_STRIPS, 0, 0, 0); gbStripsInFlight = false; }
                baseIndex = base0 + cfg.numSpatialNeighbors * i;
                // the last biased pass and the shading pass (:2418-2420) as one step where they cover the same rows (not the halo
                // scheme, whose spatial passes run over the halo too): one kernel for a band-sized launch (include/gfxexp.h)
                const bool last = i + 1 == cfg.numSpatialReusePasses;
                shadingIssued = last && !useUnbiasedEstimator && plan.spatialRows[i][0] == plan.shadingRows[0] && plan.spatialRows[i][1] == plan.shadingRows[1];
                if (shadingIssued) beauty_writer();
                // Seam rows first: a biased pass that another pass follows writes the rows its neighbours read next -- the first and the last
                // `radius` rows of the band -- in a launch of its own, their exchange goes out on lane SEAM, and the interior rows follow on
                // the frame's stream underneath it.  (A per-pixel kernel: which launch computes a pixel changes nothing.)
                const uint32_t seamAbove = plan.bandBegin > 0 ? radiusRows : 0u, seamBelow = plan.bandEnd < cfg.height ? radiusRows : 0u;
                const bool split = seamFirst && !last && !useUnbiasedEstimator && (seamAbove + seamBelow) > 0 &&
                                   plan.bandEnd - plan.bandBegin > seamAbove + seamBelow;
                if (split) {
                    const uint32_t gb = plan.bandBegin + seamAbove, ge = plan.bandEnd - seamBelow;
                    if (gfxh_frame_step* st = push(GFXH_STEP_RESTIR_PASS, spatial, plan.bandBegin, plan.bandEnd)) { st->gapBegin = gb; st->gapEnd = ge; }
                    exchange(radiusRows, GFXH_BUF_RESERVOIRS, (currentReservoirIndex + 1) % 2, GFXH_LANE_SEAM);
                    seamStripsInFlight = true;
                    push(GFXH_STEP_RESTIR_PASS, spatial, gb, ge);
                }
                else
                push(GFXH_STEP_RESTIR_PASS, shadingIssued ? static_cast<uint32_t>(GFX_RESTIR_SPATIAL_BIASED_AND_SHADING) : spatial,
                     plan.spatialRows[i][0], whole ? 0 : plan.spatialRows[i][1]);
                currentReservoirIndex = (currentReservoirIndex + 1) % 2;
            }
            baseIndex = base0 + cfg.numSpatialNeighbors * cfg.numSpatialReusePasses;
        }
        if (!shadingIssued) {
            beauty_writer();
            push(GFXH_STEP_RESTIR_PASS, GFX_RESTIR_SHADING, plan.shadingRows[0], whole ? 0 : plan.shadingRows[1]);   // :2418-2420
        }
        // strip mode: the final reservoirs the next frame's temporal pass reads across the seams
        if (cfg.enableTemporalReuse) exchange(motion, GFXH_BUF_RESERVOIRS, currentReservoirIndex);
        // no spatial pass waited for the G-buffer strips: the next frame's temporal pass is their first reader
        if (gbStripsInFlight) push(GFXH_STEP_WAIT_GBUFFER_STRIPS, 0, 0, 0);
        gather();
    }
    *numSteps = n;
    *newLastReservoirIndex = (cfg.renderer == GFXH_PATH_TRACE_REGIR || cfg.renderer == GFXH_PATH_TRACE_BASELINE) ? lastReservoirIndex : currentReservoirIndex;   // :2493
    *newLastSpatialNeighborBaseIndex = baseIndex;
    return (n >= capacity || tooTall) ? 1 : 0;
}

// What an exchange step moves, as the descriptor the callback receives: built from the launch parameters alone, so the
// CPU tests fill it from the oracle's host buffers with the same code the GPU driver uses.
int gfxh_frame_step_exchange_desc(const gfxh_restir_config* cfg, const gfxh_frame_step* st, uint32_t stepIndex, const gfx_restir_static_params* sp,
                                  const gfx_regir_params* regir, uint32_t bufferIndex, gfxh_exchange_desc* out) {
    gfxh_exchange_desc& d = *out;
    std::memset(&d, 0, sizeof(d));
    const uint32_t W = cfg->width, H = cfg->height;
    const size_t numPixelsAll = static_cast<size_t>(W) * H;
    d.stage = stepIndex; d.width = W; d.height = H; d.lane = st->lane;
    d.bandBegin = cfg->rowBegin; d.bandEnd = cfg->rowEnd;
    auto add = [&](void* base, uint32_t bytesPerPixel, uint32_t planes) {
        gfxh_exchange_buffer& b = d.buffers[d.numBuffers++];
        b.base = base; b.bytesPerPixel = bytesPerPixel; b.numPlanes = planes; b.planeStride = static_cast<uint64_t>(bytesPerPixel) * numPixelsAll;
    };
    switch (st->op) {
    case GFXH_STEP_EXCHANGE_STRIPS:
        d.kind = GFXH_EXCHANGE_STRIPS;
        if (gfxh_strip_rows(H, cfg->rowBegin, cfg->rowEnd, st->exchangeRows, &d)) return 1;
        // GBuffer1 (motion vectors) is only ever read at a pass's own pixel
        if (st->buffers & GFXH_BUF_GBUFFERS) { add(sp->gbuffer0[bufferIndex], 16, 1); add(sp->gbuffer2[bufferIndex], 16, 1); add(sp->gbuffer3[bufferIndex], 16, 1); }
        if (st->buffers & GFXH_BUF_SAMPLE_VISIBILITY) add(sp->sampleVisibilityBuffer[bufferIndex], 4, 1);
        if (st->buffers & GFXH_BUF_RESERVOIRS) { add(sp->reservoirBuffer[st->reservoirIndex], 16, 3); add(sp->reservoirInfoBuffer[st->reservoirIndex], 8, 1); }
        if (st->buffers & GFXH_BUF_RNG) add(sp->rngBuffer, 8, 1);
        return 0;
    case GFXH_STEP_ALLREDUCE_CELL_ACCESSES:
        if (!regir) return 1;
        d.kind = GFXH_EXCHANGE_ALLREDUCE_SUM_U32;
        d.counters = regir->perCellNumAccesses;
        d.numCounters = static_cast<uint64_t>(regir->gridDimension[0]) * regir->gridDimension[1] * regir->gridDimension[2];
        return 0;
    case GFXH_STEP_GATHER_BANDS:
        d.kind = GFXH_EXCHANGE_GATHER_BANDS;
        add(sp->beautyAccumBuffer, 16, 1);
        return 0;
    default: return 1;
    }
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
    const bool viewMoved = r->viewMoved;
    const bool firstAccumFrame = !cfg.enableAccumulation || newSequence || viewMoved;   // :2312-2313 (animate || cameraIsActuallyMoving)
    const uint32_t numAccumFrames = firstAccumFrame ? 0u : std::min(r->numAccumFrames + 1, 1u << cfg.log2MaxNumAccums);

    // ---- everything that can refuse the frame is decided before any renderer state changes, so a caller that fixes the
    // configuration and retries gets the frame it would have got the first time.  The decisions below depend on the
    // configuration and on calls every rank of a band split makes alike (camera, rebuilds), never on this rank's own band
    // -- except the strip-height test of the program, which gfxh_restir_check_partition takes for all ranks at install time.
    const bool wholeFrame = cfg.rowBegin == 0 && cfg.rowEnd == 0;
    const bool strips = r->exchange != nullptr && !wholeFrame;
    const bool useUnbiased = cfg.renderer == GFXH_ORIGINAL_RESTIR_UNBIASED || cfg.renderer == GFXH_REARCHITECTED_RESTIR_UNBIASED;
    if (!wholeFrame && !strips) {
        // halo recompute: the halo comes from the neighbours' bands, which tilesplit / gfxh_rccl cut within 8 rows of this one
        gfxh_band_plan plan;
        gfxh_restir_band_plan(r, &plan);
        if (plan.haloRows + 8 > cfg.rowEnd - cfg.rowBegin && plan.haloRows > 0 && !(cfg.rowBegin == 0 && cfg.rowEnd >= cfg.height)) {
            g_driverError = "gfxh_restir_render_frame: the halo (radius x spatial passes) is taller than a neighbour's band; "
                            "install a strip exchange (gfxh_restir_set_exchange) or use fewer ranks";
            return 1;
        }
    }
    // A band renderer reads the previous frame across its seams in the temporal pass.  A strip exchange moves `maxMotionRows` rows of
    // that state for it; with none -- or in the halo-recompute scheme, whose one refresh per frame covers radius x passes rows of the
    // *final* state and nothing of what a moved camera or instance makes the temporal pass reproject into beyond them -- a frame that
    // follows a move would silently read stale rows there: refuse it.  Not a concern of a frame that reads no previous frame (new
    // sequence, temporal reuse off).
    if (!wholeFrame && viewMoved && (!strips || r->maxMotionRows == 0) && cfg.enableTemporalReuse && !newSequence) {
        g_driverError = strips ? "gfxh_restir_render_frame: the camera or an instance moved, but this band renderer exchanges no motion rows "
                                 "(gfxh_restir_set_exchange with maxMotionRows > 0)"
                               : "gfxh_restir_render_frame: the camera or an instance moved, but a halo-recompute band renderer refreshes no motion rows "
                                 "(install a strip exchange with maxMotionRows > 0, or call gfxh_restir_reset to start a new sequence)";
        return 1;
    }
    gfxh_frame_step steps[64];
    uint32_t numSteps = 0, newLastRes = 0, newLastBase = 0;
    if (gfxh_restir_frame_program(&cfg, strips ? (r->stripMode == 2 && !r->pipelineFrames ? 1 : r->stripMode) : 0, r->maxMotionRows, newSequence ? 1 : 0, r->lastReservoirIndex, r->lastSpatialNeighborBaseIndex,
                                  useUnbiased, steps, 64, &numSteps, &newLastRes, &newLastBase)) {
        g_driverError = "gfxh_restir_render_frame: the exchange strip is taller than the band (fewer ranks or a smaller radius; "
                        "gfxh_restir_check_partition decides this for all ranks at once)";
        return 1;
    }
    r->resetRequested = false;
    r->viewMoved = false;
    r->numAccumFrames = numAccumFrames;

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

    // the frame as a program (gfxh_restir_frame_program, built above): passes with their row ranges and index bookkeeping,
    // and -- for a band renderer with an exchange callback -- the points where rows owned by other ranks have to arrive
    const uint32_t W = cfg.width, H = cfg.height;
#define DRV_GFX(call) do { if (call) { g_driverError = gfx_last_error(ctx); return 1; } } while (0)
    // G-buffer pass, pipelined under the previous frame when nothing forbids it: jittering advances the pixel
    // RNGs the previous frame's passes are still drawing from.  A strip-exchange band renderer pipelines too: the pass
    // writes only this band's rows of the OTHER G-buffer half, the exchange that follows it is issued on the caller's
    // stream after that stream has waited for the pass, and the strips a neighbour sends into this frame's half never
    // touch the half being written.  (A band's launches are too small to fill the GPU -- 261 k rays on 262 k traversal
    // lanes at 8 bands -- so running the next G-buffer pass underneath the reuse passes is worth more there than on the
    // whole frame: profiles/r03_band_compute_bound.json.)  The halo-recompute scheme stays serial.
    hipStream_t main = static_cast<hipStream_t>(stream);
    const bool pipelined = r->pipelineFrames && !cfg.enableJittering && (wholeFrame || strips);
    const bool asyncGather = strips && r->asyncGather && r->pipelineFrames;
    if (cfg.renderer == GFXH_PATH_TRACE_REGIR) DRV_GFX(gfx_regir_set_params(ctx, &r->regir));
    bool gbWaited = false;      // main has been made to wait for this frame's G-buffer pass
    auto lane_stream = [&](uint32_t lane) -> hipStream_t {
        return lane == GFXH_LANE_GBUFFER ? (pipelined ? r->gbStream : main) : lane == GFXH_LANE_GATHER ? (asyncGather ? r->gatherStream : main)
             : lane == GFXH_LANE_SEAM ? (pipelined ? r->seamStream : main) : main;
    };
    auto exchange_stream = [&](const gfxh_frame_step& st) -> hipStream_t {
        if (st.op == GFXH_STEP_EXCHANGE_STRIPS && st.lane == GFXH_LANE_GBUFFER && !r->stripsOnGbLane) return main;
        return lane_stream(st.lane);
    };
    for (uint32_t k = 0; k < numSteps; ++k) {
        const gfxh_frame_step& st = steps[k];
        switch (st.op) {
        case GFXH_STEP_RESTIR_PASS:
        case GFXH_STEP_PT_PASS: {
            hipStream_t s = lane_stream(st.lane);
            const bool gb = st.lane == GFXH_LANE_GBUFFER;
            if (gb && pipelined) {
                if (r->prevReadPending) DRV_HIP(hipStreamWaitEvent(s, r->evPrevRead, 0));
                else { DRV_HIP(hipEventRecord(r->evPrevRead, main)); DRV_HIP(hipStreamWaitEvent(s, r->evPrevRead, 0)); }   // first frame: after whatever the caller queued
                if (r->consumedPending) { DRV_HIP(hipStreamWaitEvent(s, r->evConsumed, 0)); r->consumedPending = false; }   // gfxh_restir_outputs_consumed
            }
            if (!gb && pipelined && !gbWaited) { DRV_HIP(hipStreamWaitEvent(main, r->evGbuffer, 0)); gbWaited = true; }
            DRV_GFX(gfx_restir_set_params(ctx, s, &r->sp, &fp, st.currentReservoirIndex, st.spatialNeighborBaseIndex));
            if (st.op == GFXH_STEP_PT_PASS) DRV_GFX(gfx_pt_launch(ctx, s, static_cast<int>(st.pass), W, H, cfg.maxPathLength, st.rowBegin, st.rowEnd));
            else if (st.gapEnd > st.gapBegin) DRV_GFX(gfx_restir_launch_rows_gap(ctx, s, static_cast<int>(st.pass), W, H, st.rowBegin, st.rowEnd, st.gapBegin, st.gapEnd));
            else DRV_GFX(gfx_restir_launch_rows(ctx, s, static_cast<int>(st.pass), W, H, st.rowBegin, st.rowEnd));
            if (gb && pipelined) DRV_HIP(hipEventRecord(r->evGbuffer, s));
            break;
        }
        case GFXH_STEP_PREV_GBUFFER_RELEASED:   // the frame has queued its last pass that reads the previous frame's G-buffer
            if (pipelined) { DRV_HIP(hipEventRecord(r->evPrevRead, main)); r->prevReadPending = true; }
            break;
        case GFXH_STEP_WAIT_GBUFFER_STRIPS:
            if (pipelined && r->gbStripsPending) { DRV_HIP(hipStreamWaitEvent(main, r->evGbStrips, 0)); r->gbStripsPending = false; }
            break;
        case GFXH_STEP_WAIT_SEAM_STRIPS:
            if (r->seamStripsPending) { DRV_HIP(hipStreamWaitEvent(main, r->evSeamStrips, 0)); r->seamStripsPending = false; }
            break;
        case GFXH_STEP_WAIT_PREVIOUS_GATHER:
            if (r->gatherPending) { DRV_HIP(hipStreamWaitEvent(main, r->evGather, 0)); r->gatherPending = false; }
            break;
        case GFXH_STEP_EXCHANGE_STRIPS:
        case GFXH_STEP_ALLREDUCE_CELL_ACCESSES:
        case GFXH_STEP_GATHER_BANDS: {
            gfxh_exchange_desc d;
            gfxh_frame_step_exchange_desc(&cfg, &st, k, &r->sp, cfg.renderer == GFXH_PATH_TRACE_REGIR ? &r->regir : nullptr, bufferIndex, &d);
            hipStream_t s = exchange_stream(st);
            if (s == main) d.lane = GFXH_LANE_MAIN;            // the lane the exchange actually runs on (its communicator)
            if (s == main && pipelined && !gbWaited) { DRV_HIP(hipStreamWaitEvent(main, r->evGbuffer, 0)); gbWaited = true; }
            if (st.op == GFXH_STEP_GATHER_BANDS && s != main) {       // behind the frame's last pass
                DRV_HIP(hipEventRecord(r->evBandDone, main));
                DRV_HIP(hipStreamWaitEvent(s, r->evBandDone, 0));
            }
            const bool seam = st.op == GFXH_STEP_EXCHANGE_STRIPS && st.lane == GFXH_LANE_SEAM && s != main;
            if (seam) {                                                // behind the seam rows the frame's stream has just queued
                DRV_HIP(hipEventRecord(r->evSeamRows, main));
                DRV_HIP(hipStreamWaitEvent(s, r->evSeamRows, 0));
            }
            if (r->exchange(r->exchangeUser, s, &d)) { g_driverError = "gfxh_restir_render_frame: the exchange callback failed"; return 1; }
            if (st.op == GFXH_STEP_GATHER_BANDS && s != main) { DRV_HIP(hipEventRecord(r->evGather, s)); r->gatherPending = true; }
            if (seam) { DRV_HIP(hipEventRecord(r->evSeamStrips, s)); r->seamStripsPending = true; }
            else if (st.op == GFXH_STEP_EXCHANGE_STRIPS && s != main) { DRV_HIP(hipEventRecord(r->evGbStrips, s)); r->gbStripsPending = true; }
            break;
        }
        default: break;
        }
    }
    if (pipelined && !gbWaited) DRV_HIP(hipStreamWaitEvent(main, r->evGbuffer, 0));   // a program without a pass on the main lane
#undef DRV_GFX
    r->lastReservoirIndex = newLastRes;                                                        // :2493
    r->lastSpatialNeighborBaseIndex = newLastBase;
    r->prevCamera = r->camera;
    ++r->frameIndex;
    return 0;
}

int gfxh_restir_outputs_consumed(gfxh_restir* r, void* stream) {
    if (!r) { g_driverError = "gfxh_restir_outputs_consumed: null renderer"; return 1; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!r->evConsumed && !hip_ok(hipEventCreateWithFlags(&r->evConsumed, order_event_flags()), "hipEventCreate")) return 1;
    if (!hip_ok(hipEventRecord(r->evConsumed, s), "hipEventRecord")) return 1;
    r->consumedPending = true;
    return 0;
}

const char* gfxh_restir_last_error(void) { return g_driverError.c_str(); }

} // extern "C"
