// nrc_driver.cpp -- headless neural-radiance-caching frame driver (include/gfxexp_host.h).
//
// Re-creates the part of neural_radiance_caching/neural_radiance_caching_main.cpp that surrounds the
// hot path: NRC buffer allocation and the LCG shuffler table (:1145-1200), per-frame offsets from
// mt19937(72139121) (:1602, :2276-2277), and the frame sequencing (:2225-2370).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../../../include/gfxexp_host.h"

namespace {
thread_local std::string g_nrcError;
bool nrc_hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    g_nrcError = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}
#define NRC_HIP(call) do { if (!nrc_hip_ok((call), #call)) return 1; } while (0)
#define NRC_GFX(call) do { if (call) { g_nrcError = gfx_last_error(r->ctx); return 1; } } while (0)
constexpr uint32_t kNumTrainingDataPerFrame = 1u << 16;   // neural_radiance_caching_shared.h:8
constexpr uint32_t kTrainBufferSize = 2u << 16;            // :9
}


// Flags of the events that only order kernels of THIS device across streams: no system-scope fence at the record (its cache write-back
// and invalidation cost the kernels behind it; hip_runtime_api.h hipEventDisableSystemFence).  GFX_EVENT_SYSTEM_FENCE=1: the default flags.
static unsigned nrc_order_event_flags() {
    static const unsigned flags = [] { const char* e = std::getenv("GFX_EVENT_SYSTEM_FENCE"); return (e && e[0] == '1') ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence); }();
    return flags;
}

struct gfxh_nrc {
    gfx_ctx* ctx = nullptr;
    gfxh_nrc_config cfg;
    gfx_restir_static_params sp;
    gfx_restir_frame_params fp;
    gfx_nrc_params np;
    gfx_regir_params regir;      // neeSampler == 1
    // neeSampler == 2: the ReSTIR DI passes ahead of the tracer -- reservoir ping-pong and neighbour-table index as restir_di_main.cpp:1686, 2402-2411
    uint32_t lastReservoirIndex = 1, lastSpatialNeighborBaseIndex = 0;
    std::vector<void*> allocations;
    std::vector<void*> envAllocations;        // the environment map's tables (gfxh_nrc_set_env): replaced as a set
    uint64_t accel = 0, network = 0;
    uint32_t frameIndex = 0, numAccumFrames = 0;
    bool viewMoved = false;      // an instance moved since the last frame: accumulation restarts
    float envPowerCoeff = 1.0f, envRotation = 0.0f;       // gfxh_nrc_set_env
    std::mt19937 perFrameRng{ 72139121 };                  // main:1602
    gfx_camera prevCamera;
    uint32_t lastNumTrainingData = 0, lastTileSize[2] = { 8, 8 }, lastNumInferenceQueries = 0;
    // whole-frame renderers never wait for the GPU inside a frame: the inference batch size is formed on the device
    // (GFX_PT_NRC_COUNT_QUERIES + gfx_nrc_infer_indirect) and the figures gfxh_nrc_stats reports are copied into pinned
    // host words asynchronously (evStats marks their arrival)
    uint32_t* hostStats = nullptr;            // pinned: [0] numTrainingData, [1..2] tileSize, [3] numInferenceQueries
    void* dQueryCount = nullptr;
    hipEvent_t evStats = nullptr;
    bool statsPending = false;
    // The four training steps of frame N only feed the inference of frame N + 1, so they run on their own
    // stream underneath the G-buffer / path-tracing kernels of frame N + 1 (a training step is ~300 single-wave
    // blocks: it leaves most of the chip idle on its own).  evData: shuffled training data ready;
    // evTrained: weights of the last step packed.
    hipStream_t trainStream = nullptr;
    hipEvent_t evData = nullptr, evTrained = nullptr;
    // Frame pipelining (as gfxh_restir): the G-buffer pass of a frame depends on nothing the previous frame computes -- it writes the
    // other half of the double-buffered G-buffers from its own ray queue / hit / ticket scratch inside the context -- so it runs on
    // gbStream, where it overlaps the previous frame's inference, accumulation and propagation that are still queued on the caller's
    // stream.  evGb: G-buffer written; evGbFree: the caller's stream has passed the point up to which it read these buffers.
    hipStream_t gbStream = nullptr;
    hipEvent_t evGb = nullptr, evGbFree = nullptr;
    bool pipelineFrames = true, gbFreePending = false;
    hipEvent_t evConsumed = nullptr;          // gfxh_nrc_outputs_consumed: the caller's reads of the albedo / normal accumulators are behind this event
    bool consumedPending = false;
    bool trainPending = false, overlapTraining = true;
    // band split: every rank trains its own copy of the network on the gathered batch (training is reproducible bit for bit, nrc.hip
    // k_nrc_grid_scatter, so the copies stay identical); GFX_NRC_TRAIN_ON_RANK0=1: rank 0 trains and broadcasts its inference images (rounds 3-4)
    bool trainOnRank0 = false;
    // ... and every kChecksumPeriod-th frame the ranks compare a checksum of the images they infer with (one 8-byte all-reduce): a copy
    // that has drifted -- an order-dependent gradient sum, a rank that missed a batch -- fails the frame instead of drawing seams
    void* dChecksum = nullptr;                // device uint32[2]: {checksum, 1}
    // band renderer (gfxh_nrc_set_exchange)
    gfxh_exchange_fn exchange = nullptr; void* exchangeUser = nullptr; int rank = 0;
    uint32_t gatherCounts[2] = { 0, 0 };
};

static int nrc_alloc(gfxh_nrc* r, void** p, size_t bytes) {
    NRC_HIP(hipMalloc(p, bytes));
    r->allocations.push_back(*p);
    NRC_HIP(hipMemset(*p, 0, bytes));
    return 0;
}

extern "C" {

void gfxh_nrc_default_config(gfxh_nrc_config* cfg, uint32_t width, uint32_t height) {
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->width = width; cfg->height = height;
    cfg->positionEncoding = GFX_NRC_HASH_GRID; cfg->numHiddenLayers = 2; cfg->learningRate = 1e-2f;
    cfg->maxPathLength = 5; cfg->radianceScale = 1.0f; cfg->train = 1; cfg->enableAccumulation = 0;
    cfg->neeSampler = 0;
    cfg->regirGridDimension[0] = 32; cfg->regirGridDimension[1] = 8; cfg->regirGridDimension[2] = 32;   // regir_main.cpp:1112
    cfg->regirLog2CandidatesPerLightSlot = 3; cfg->regirLog2CandidatesPerCell = 2;                       // :1733-1734
    cfg->regirEnableTemporalReuse = 1; cfg->regirEnableCellRandomization = 1;                             // :1735-1736
    cfg->camera.aspect = static_cast<float>(width) / height;
    cfg->camera.fovY = 50 * 3.14159265358979323846f / 180;
    const float ident[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    std::memcpy(cfg->camera.orientation, ident, sizeof(ident));
}

void gfxh_nrc_destroy(gfxh_nrc* r) {
    if (!r) return;
    (void)hipDeviceSynchronize();
    if (r->evStats) (void)hipEventDestroy(r->evStats);
    if (r->hostStats) (void)hipHostFree(r->hostStats);
    if (r->evData) (void)hipEventDestroy(r->evData);
    if (r->evTrained) (void)hipEventDestroy(r->evTrained);
    if (r->trainStream) (void)hipStreamDestroy(r->trainStream);
    if (r->evGb) (void)hipEventDestroy(r->evGb);
    if (r->evGbFree) (void)hipEventDestroy(r->evGbFree);
    if (r->evConsumed) (void)hipEventDestroy(r->evConsumed);
    if (r->gbStream) (void)hipStreamDestroy(r->gbStream);
    if (r->network) (void)gfx_nrc_destroy(r->ctx, r->network);
    for (void* p : r->allocations) (void)hipFree(p);
    for (void* p : r->envAllocations) (void)hipFree(p);
    delete r;
}

int gfxh_nrc_create(gfx_ctx* ctx, const gfxh_nrc_config* cfg, gfxh_nrc** out) {
    *out = nullptr;
    gfxh_nrc* r = new gfxh_nrc();
    r->ctx = ctx; r->cfg = *cfg;
    const size_t n = static_cast<size_t>(cfg->width) * cfg->height;
    gfx_restir_static_params& sp = r->sp; gfx_nrc_params& np = r->np;
    std::memset(&sp, 0, sizeof(sp)); std::memset(&r->fp, 0, sizeof(r->fp)); std::memset(&np, 0, sizeof(np));
    sp.imageSizeX = static_cast<int32_t>(cfg->width); sp.imageSizeY = static_cast<int32_t>(cfg->height);
    int err = 0;
    err |= nrc_alloc(r, &sp.rngBuffer, 8 * n);
    for (int i = 0; i < 2; ++i) {
        err |= nrc_alloc(r, &sp.gbuffer0[i], sizeof(gfx_gbuffer0) * n);
        err |= nrc_alloc(r, &sp.gbuffer1[i], sizeof(gfx_gbuffer1) * n);
        err |= nrc_alloc(r, &sp.gbuffer2[i], sizeof(gfx_gbuffer2) * n);
        err |= nrc_alloc(r, &sp.gbuffer3[i], sizeof(gfx_gbuffer3) * n);
    }
    err |= nrc_alloc(r, &sp.beautyAccumBuffer, 16 * n);
    err |= nrc_alloc(r, &sp.albedoAccumBuffer, 16 * n);
    err |= nrc_alloc(r, &sp.normalAccumBuffer, 16 * n);
    // main:1150-1185
    np.maxNumTrainingSuffixes = static_cast<uint32_t>(n / 16);
    const size_t cap = (n + np.maxNumTrainingSuffixes + 255) / 256 * 256;
    for (int i = 0; i < 2; ++i) {
        err |= nrc_alloc(r, &np.numTrainingData[i], 4);
        err |= nrc_alloc(r, &np.tileSize[i], 8);
        err |= nrc_alloc(r, &np.targetMinMax[i], 24);
        err |= nrc_alloc(r, &np.targetAvg[i], 12);
        err |= nrc_alloc(r, &np.trainRadianceQueryBuffer[i], 56ull * kTrainBufferSize);
        err |= nrc_alloc(r, &np.trainTargetBuffer[i], 12ull * kTrainBufferSize);
    }
    err |= nrc_alloc(r, &np.offsetToSelectUnbiasedTile, 4);
    err |= nrc_alloc(r, &np.offsetToSelectTrainingPath, 4);
    err |= nrc_alloc(r, &np.inferenceRadianceQueryBuffer, 56 * cap);
    err |= nrc_alloc(r, &np.inferenceTerminalInfoBuffer, 16 * n);
    err |= nrc_alloc(r, &np.inferredRadianceBuffer, 12 * cap);
    err |= nrc_alloc(r, &np.perFrameContributionBuffer, 12 * n);
    err |= nrc_alloc(r, &np.trainVertexInfoBuffer, 16ull * kTrainBufferSize);
    err |= nrc_alloc(r, &np.trainSuffixTerminalInfoBuffer, 4ull * np.maxNumTrainingSuffixes);
    err |= nrc_alloc(r, &np.dataShufflerBuffer, 4ull * kNumTrainingDataPerFrame);
    std::memset(&r->regir, 0, sizeof(r->regir));
    if (cfg->neeSampler == 1) {   // the light-slot grid of regir_main.cpp:1071-1097 over the scene box
        gfx_regir_params& g = r->regir;
        const uint32_t* d = cfg->regirGridDimension;
        const size_t numCells = static_cast<size_t>(d[0]) * d[1] * d[2], numSlots = numCells * 512;
        if (numCells == 0) { g_nrcError = "gfxh_nrc_create: empty ReGIR grid"; gfxh_nrc_destroy(r); return 1; }
        if (!(cfg->rowBegin == 0 && cfg->rowEnd == 0)) { g_nrcError = "gfxh_nrc_create: the ReGIR NEE sampler is not wired to band renderers"; gfxh_nrc_destroy(r); return 1; }
        for (int i = 0; i < 2; ++i) {
            err |= nrc_alloc(r, &g.reservoirs[i], 48 * numSlots);
            err |= nrc_alloc(r, &g.reservoirInfos[i], 8 * numSlots);
            err |= nrc_alloc(r, &g.numActiveCells[i], 4);
        }
        err |= nrc_alloc(r, &g.lightSlotRngs, 8 * numSlots);
        err |= nrc_alloc(r, &g.perCellNumAccesses, 4 * numCells);
        err |= nrc_alloc(r, &g.lastAccessFrameIndices, 4 * numCells);
        if (!err) {
            std::vector<uint64_t> states(numSlots);
            gfxh_seed_rng_states(states.data(), numSlots, 591842031321323413ull);
            if (!nrc_hip_ok(hipMemcpy(g.lightSlotRngs, states.data(), 8 * numSlots, hipMemcpyHostToDevice), "upload light-slot rng states") ||
                !nrc_hip_ok(hipMemset(g.lastAccessFrameIndices, 0xFF, 4 * numCells), "fill lastAccessFrameIndices")) err = 1;
        }
        for (int k = 0; k < 3; ++k) {
            g.gridOrigin[k] = cfg->sceneAabbMin[k];
            g.gridCellSize[k] = (cfg->sceneAabbMax[k] - cfg->sceneAabbMin[k]) / static_cast<float>(d[k]);
            g.gridDimension[k] = d[k];
        }
        g.log2NumCandidatesPerLightSlot = cfg->regirLog2CandidatesPerLightSlot;
        g.log2NumCandidatesPerCell = cfg->regirLog2CandidatesPerCell;
        g.enableCellRandomization = cfg->regirEnableCellRandomization;
    }
    else if (cfg->neeSampler == 2) {   // the per-pixel reservoirs and the neighbour table of restir_di_main.cpp:1233-1325
        if (!(cfg->rowBegin == 0 && cfg->rowEnd == 0)) { g_nrcError = "gfxh_nrc_create: the ReSTIR NEE sampler is not wired to band renderers"; gfxh_nrc_destroy(r); return 1; }
        for (int i = 0; i < 2; ++i) {
            err |= nrc_alloc(r, &sp.reservoirBuffer[i], 48 * n);
            err |= nrc_alloc(r, &sp.reservoirInfoBuffer[i], 8 * n);
        }
        void* deltas = nullptr;
        err |= nrc_alloc(r, &deltas, 8 * 1024);
        if (!err) {
            std::vector<float> host(2 * 1024);
            gfxh_spatial_neighbor_deltas(host.data());
            if (!nrc_hip_ok(hipMemcpy(deltas, host.data(), 8 * 1024, hipMemcpyHostToDevice), "upload the neighbour table")) err = 1;
            sp.spatialNeighborDeltas = deltas;
        }
    }
    else if (cfg->neeSampler != 0) { g_nrcError = "gfxh_nrc_create: unknown neeSampler"; gfxh_nrc_destroy(r); return 1; }
    if (err) { gfxh_nrc_destroy(r); return 1; }
    {
        std::vector<uint64_t> states(n);
        gfxh_seed_rng_states(states.data(), n, 591842031321323413ull);
        const uint32_t tile[2] = { 8, 8 };
        std::vector<uint32_t> shufflers(kNumTrainingDataPerFrame);
        uint32_t lcg = 471313181u;                                             // main:1188
        for (uint32_t i = 0; i < kNumTrainingDataPerFrame; ++i) { lcg = (lcg * 1103515245u + 12345u) % (1u << 31); shufflers[i] = lcg; }
        bool ok = nrc_hip_ok(hipMemcpy(sp.rngBuffer, states.data(), 8 * n, hipMemcpyHostToDevice), "upload rng states");
        for (int i = 0; i < 2 && ok; ++i) ok = nrc_hip_ok(hipMemcpy(np.tileSize[i], tile, 8, hipMemcpyHostToDevice), "upload tile size");
        ok = ok && nrc_hip_ok(hipMemcpy(np.dataShufflerBuffer, shufflers.data(), 4ull * kNumTrainingDataPerFrame, hipMemcpyHostToDevice), "upload shufflers");
        if (!ok) { gfxh_nrc_destroy(r); return 1; }
    }
    std::memcpy(np.sceneAabbMin, cfg->sceneAabbMin, 12); std::memcpy(np.sceneAabbMax, cfg->sceneAabbMax, 12);
    if (gfx_accel_build(ctx, nullptr, &r->accel) || gfx_lights_build_static(ctx, nullptr) ||
        gfx_nrc_create(ctx, cfg->positionEncoding, cfg->numHiddenLayers, cfg->learningRate, &r->network)) {
        g_nrcError = gfx_last_error(ctx);
        gfxh_nrc_destroy(r);
        return 1;
    }
    r->prevCamera = cfg->camera;
    {
        const char* e = std::getenv("GFX_NRC_SERIAL_TRAINING");   // debugging aid: train on the caller's stream
        r->overlapTraining = !(e && e[0] == '1');
        const char* e0 = std::getenv("GFX_NRC_TRAIN_ON_RANK0");
        r->trainOnRank0 = e0 && e0[0] == '1';
        // non-blocking: the caller's stream may be the legacy default stream, which would serialise a blocking one
        const char* sf = std::getenv("GFX_SERIAL_FRAMES");   // debugging aid: everything on the caller's stream
        r->pipelineFrames = !(sf && sf[0] == '1');
        if (!nrc_hip_ok(hipStreamCreateWithFlags(&r->gbStream, hipStreamNonBlocking), "hipStreamCreateWithFlags") ||
            !nrc_hip_ok(hipEventCreateWithFlags(&r->evGb, nrc_order_event_flags()), "hipEventCreate") ||
            !nrc_hip_ok(hipEventCreateWithFlags(&r->evGbFree, nrc_order_event_flags()), "hipEventCreate") ||
            !nrc_hip_ok(hipStreamCreateWithFlags(&r->trainStream, hipStreamNonBlocking), "hipStreamCreateWithFlags") ||
            !nrc_hip_ok(hipEventCreateWithFlags(&r->evData, nrc_order_event_flags()), "hipEventCreate") ||
            !nrc_hip_ok(hipEventCreateWithFlags(&r->evTrained, nrc_order_event_flags()), "hipEventCreate") ||
            !nrc_hip_ok(hipEventCreateWithFlags(&r->evStats, hipEventDisableTiming), "hipEventCreate") ||
            !nrc_hip_ok(hipHostMalloc(reinterpret_cast<void**>(&r->hostStats), 64, hipHostMallocDefault), "hipHostMalloc")) {
            gfxh_nrc_destroy(r);
            return 1;
        }
        std::memset(r->hostStats, 0, 64);
        if (gfx_nrc_query_count_ptr(ctx, &r->dQueryCount)) { g_nrcError = gfx_last_error(ctx); gfxh_nrc_destroy(r); return 1; }
    }
    *out = r;
    return 0;
}

int gfxh_nrc_render_frame(gfxh_nrc* r, void* stream, float* lossOut) {
    gfx_ctx* ctx = r->ctx;
    const gfxh_nrc_config& cfg = r->cfg;
    const uint32_t W = cfg.width, H = cfg.height;
    const uint32_t frameIndex = r->frameIndex, bufferIndex = frameIndex % 2;
    gfx_restir_frame_params& fp = r->fp;
    NRC_GFX(gfx_lights_build_instances(ctx, stream, bufferIndex));
    const bool newSequence = frameIndex == 0;                                       // main:2228
    const bool viewMoved = r->viewMoved;                                // animate || cameraIsActuallyMoving (neural_radiance_caching_main.cpp firstAccumFrame)
    r->viewMoved = false;
    if (!cfg.enableAccumulation || newSequence || viewMoved) r->numAccumFrames = 0;
    else r->numAccumFrames = std::min(r->numAccumFrames + 1, 1u << 16);
    fp.travHandle = r->accel; fp.numAccumFrames = r->numAccumFrames; fp.frameIndex = frameIndex;
    fp.prevCamera = frameIndex == 0 ? cfg.camera : r->prevCamera;
    fp.camera = cfg.camera;
    fp.envLightPowerCoeff = r->envPowerCoeff; fp.envLightRotation = r->envRotation;
    fp.bufferIndex = bufferIndex; fp.resetFlowBuffer = newSequence; fp.enableJittering = 0; fp.enableEnvLight = r->sp.envLightTexture != nullptr; fp.enableBumpMapping = cfg.enableBumpMapping; fp.useSolidAngleSampling = 0;
    r->np.radianceScale = cfg.radianceScale;
    r->np.preprocessOffsetToSelectUnbiasedTile = static_cast<uint32_t>(r->perFrameRng());   // main:2276-2277
    r->np.preprocessOffsetToSelectTrainingPath = static_cast<uint32_t>(r->perFrameRng());
    r->np.isNewSequence = newSequence;
    NRC_GFX(gfx_restir_set_params(ctx, stream, &r->sp, &fp, 0, 0));
    NRC_GFX(gfx_nrc_set_render_params(ctx, &r->np));
    const bool band = !(cfg.rowBegin == 0 && cfg.rowEnd == 0);
    if (band && !r->exchange) { g_nrcError = "gfxh_nrc_render_frame: a band renderer needs gfxh_nrc_set_exchange"; return 1; }
    if (band && (cfg.rowEnd > H || cfg.rowBegin >= cfg.rowEnd)) { g_nrcError = "gfxh_nrc_render_frame: row band outside the image"; return 1; }
    const uint32_t rb = band ? cfg.rowBegin : 0, re = band ? cfg.rowEnd : 0;
    const bool regirNee = cfg.neeSampler == 1, restirNee = cfg.neeSampler == 2;
    if (restirNee) {   // the reference's ReSTIR DI defaults (restir_di_main.cpp:1944-1967): 32 candidates, temporal + 2 x 5 biased spatial reuse, radius 20
        fp.spatialNeighborRadius = 20.0f; fp.radiusThresholdForSpatialVisReuse = 10.0f;
        fp.log2NumCandidateSamples = 5; fp.numSpatialNeighbors = 5; fp.useLowDiscrepancyNeighbors = 1;
        fp.reuseVisibility = 1; fp.reuseVisibilityForTemporal = 1; fp.reuseVisibilityForSpatiotemporal = 0;
        fp.enableTemporalReuse = 1; fp.enableSpatialReuse = 1; fp.useUnbiasedEstimator = 0;
        NRC_GFX(gfx_restir_set_params(ctx, stream, &r->sp, &fp, 0, 0));
    }
    // The G-buffer pass: pipelined under the previous frame's tail unless something it depends on was queued on the caller's stream
    // since (an instance moved: the BVH update) or it is not a whole-frame renderer.  It overwrites the G-buffers of two frames ago,
    // which the caller's stream read last in that frame's path-tracing pass (evGbFree).
    const bool pipelined = r->pipelineFrames && !band && !viewMoved && !newSequence;
    if (pipelined) {
        if (r->gbFreePending) NRC_HIP(hipStreamWaitEvent(r->gbStream, r->evGbFree, 0));
        if (r->consumedPending) { NRC_HIP(hipStreamWaitEvent(r->gbStream, r->evConsumed, 0)); r->consumedPending = false; }
        NRC_GFX(gfx_pt_launch(ctx, r->gbStream, GFX_PT_SETUP_GBUFFERS, W, H, cfg.maxPathLength, rb, re));
        NRC_HIP(hipEventRecord(r->evGb, r->gbStream));
        NRC_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), r->evGb, 0));
    }
    else {
        // (a pass still running on gbStream belongs to an earlier frame and was joined by that frame)
        NRC_GFX(gfx_pt_launch(ctx, stream, GFX_PT_SETUP_GBUFFERS, W, H, cfg.maxPathLength, rb, re));
    }
    if (restirNee) {   // restir_di_main.cpp:2365-2421 without the shading pass: its direct term is formed at the tracer's first vertex
        uint32_t cur = (r->lastReservoirIndex + 1) % 2, base = r->lastSpatialNeighborBaseIndex;
        NRC_GFX(gfx_restir_set_params(ctx, stream, &r->sp, &fp, cur, base));
        NRC_GFX(gfx_restir_launch(ctx, stream, newSequence ? GFX_RESTIR_INITIAL_RIS : GFX_RESTIR_INITIAL_AND_TEMPORAL_BIASED, W, H));
        for (uint32_t i = 0; i < 2; ++i) {
            NRC_GFX(gfx_restir_set_params(ctx, stream, &r->sp, &fp, cur, base + fp.numSpatialNeighbors * i));
            NRC_GFX(gfx_restir_launch(ctx, stream, GFX_RESTIR_SPATIAL_BIASED, W, H));
            cur = (cur + 1) % 2;
        }
        base += fp.numSpatialNeighbors * 2;
        r->lastReservoirIndex = cur; r->lastSpatialNeighborBaseIndex = base;
        NRC_GFX(gfx_restir_set_params(ctx, stream, &r->sp, &fp, cur, base));      // the tracer reads reservoirs[cur]
    }
    if (regirNee) {   // regir_main.cpp:2031-2066 around the tracer: build (with temporal reuse past the first frame), trace, age
        NRC_GFX(gfx_regir_set_params(ctx, &r->regir));
        NRC_GFX(gfx_pt_launch(ctx, stream, (cfg.regirEnableTemporalReuse && !newSequence) ? GFX_PT_REGIR_BUILD_CELL_RESERVOIRS_TEMPORAL : GFX_PT_REGIR_BUILD_CELL_RESERVOIRS,
                              W, H, cfg.maxPathLength, 0, 0));
    }
    NRC_GFX(gfx_pt_launch(ctx, stream, GFX_PT_NRC_PREPROCESS, W, H, cfg.maxPathLength, 0, 0));
    NRC_GFX(gfx_pt_launch(ctx, stream, regirNee ? GFX_PT_PATH_TRACE_NRC_REGIR : restirNee ? GFX_PT_PATH_TRACE_NRC_RESTIR : GFX_PT_PATH_TRACE_NRC, W, H, cfg.maxPathLength, rb, re));
    if (regirNee) NRC_GFX(gfx_pt_launch(ctx, stream, GFX_PT_REGIR_UPDATE_LAST_ACCESS, W, H, cfg.maxPathLength, 0, 0));
    if (r->pipelineFrames && !band) {   // nothing later in the frame reads the G-buffers
        NRC_HIP(hipEventRecord(r->evGbFree, static_cast<hipStream_t>(stream)));
        r->gbFreePending = true;
    }
    // main:2293-2303: the inference batch size needs the tile size of this frame.  The reference synchronises the stream and
    // reads it back; a band renderer does the same here (the record gather needs the counts on the host anyway).  The whole-
    // frame renderer forms the batch size on the device instead and never waits for the GPU inside a frame.
    uint32_t tilesX = 0, tilesY = 0, numInferenceQueries = 0;
    if (band) {
        NRC_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
        NRC_HIP(hipMemcpy(&r->lastNumTrainingData, r->np.numTrainingData[bufferIndex], 4, hipMemcpyDeviceToHost));
        NRC_HIP(hipMemcpy(r->lastTileSize, r->np.tileSize[bufferIndex], 8, hipMemcpyDeviceToHost));
        tilesX = (W + r->lastTileSize[0] - 1) / r->lastTileSize[0]; tilesY = (H + r->lastTileSize[1] - 1) / r->lastTileSize[1];
        numInferenceQueries = (W * H + tilesX * tilesY + 127) / 128 * 128;
        r->lastNumInferenceQueries = numInferenceQueries;
    }
    else {
        NRC_GFX(gfx_pt_launch(ctx, stream, GFX_PT_NRC_COUNT_QUERIES, W, H, cfg.maxPathLength, 0, 0));
        if (r->statsPending) { NRC_HIP(hipEventSynchronize(r->evStats)); r->statsPending = false; }   // the previous frame's copies (long done)
        hipStream_t s = static_cast<hipStream_t>(stream);
        NRC_HIP(hipMemcpyAsync(r->hostStats + 0, r->np.numTrainingData[bufferIndex], 4, hipMemcpyDeviceToHost, s));
        NRC_HIP(hipMemcpyAsync(r->hostStats + 1, r->np.tileSize[bufferIndex], 8, hipMemcpyDeviceToHost, s));
        NRC_HIP(hipMemcpyAsync(r->hostStats + 3, r->dQueryCount, 4, hipMemcpyDeviceToHost, s));
        NRC_HIP(hipEventRecord(r->evStats, s));
        r->statsPending = true;
    }
    if (r->trainPending) {   // the weights this frame infers with come from the previous frame's training
        NRC_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), r->evTrained, 0));
        r->trainPending = false;
    }
    auto exchange = [&](gfxh_exchange_desc& d) -> int {
        d.width = W; d.height = H; d.bandBegin = cfg.rowBegin; d.bandEnd = cfg.rowEnd;
        if (r->exchange(r->exchangeUser, stream, &d)) { g_nrcError = "gfxh_nrc_render_frame: the exchange callback failed"; return 1; }
        return 0;
    };
    if (!band) {
        // sized for the smallest tile (4 x 4: W * H / 16 = maxNumTrainingSuffixes tiles); the kernel reads the real count
        const uint32_t maxQueries = static_cast<uint32_t>((static_cast<size_t>(W) * H + r->np.maxNumTrainingSuffixes + 255) / 256 * 256);
        NRC_GFX(gfx_nrc_infer_indirect(ctx, stream, r->network, r->np.inferenceRadianceQueryBuffer, r->dQueryCount, maxQueries, r->np.inferredRadianceBuffer));
    }
    else {
        // the band's pixels, then the suffix queries of the training tiles (all of them: the ones whose training pixel lies
        // in another band are never read).  Rounding a batch up to 128 touches queries this rank does not use.
        char* q = static_cast<char*>(r->np.inferenceRadianceQueryBuffer); char* y = static_cast<char*>(r->np.inferredRadianceBuffer);
        const size_t first = static_cast<size_t>(cfg.rowBegin) * W;
        const uint32_t bandQueries = ((cfg.rowEnd - cfg.rowBegin) * W + 127) / 128 * 128;
        NRC_GFX(gfx_nrc_infer(ctx, stream, r->network, q + 56 * first, bandQueries, y + 12 * first));
        const uint32_t tileQueries = (tilesX * tilesY + 127) / 128 * 128;
        NRC_GFX(gfx_nrc_infer(ctx, stream, r->network, q + 56ull * W * H, tileQueries, y + 12ull * W * H));
    }
    NRC_GFX(gfx_pt_launch(ctx, stream, GFX_PT_NRC_ACCUMULATE, W, H, cfg.maxPathLength, rb, re));
    if (cfg.train) {
        NRC_GFX(gfx_pt_launch(ctx, stream, GFX_PT_NRC_PROPAGATE, W, H, cfg.maxPathLength, 0, 0));
        if (band) {   // every rank ends up with every band's records (rank order) and the global count
            gfxh_exchange_desc d; std::memset(&d, 0, sizeof(d));
            d.kind = GFXH_EXCHANGE_GATHER_RECORDS;
            d.numBuffers = 2;
            d.buffers[0].base = r->np.trainRadianceQueryBuffer[0]; d.buffers[0].bytesPerPixel = 56; d.buffers[0].numPlanes = 1;
            d.buffers[1].base = r->np.trainTargetBuffer[0]; d.buffers[1].bytesPerPixel = 12; d.buffers[1].numPlanes = 1;
            r->gatherCounts[0] = std::min(r->lastNumTrainingData, kTrainBufferSize); r->gatherCounts[1] = 0;
            d.counters = r->gatherCounts; d.numCounters = kTrainBufferSize;
            if (exchange(d)) return 1;
            r->lastNumTrainingData = r->gatherCounts[0];
            NRC_HIP(hipMemcpyAsync(r->np.numTrainingData[bufferIndex], &r->gatherCounts[0], 4, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
        }
        NRC_GFX(gfx_pt_launch(ctx, stream, GFX_PT_NRC_SHUFFLE, W, H, cfg.maxPathLength, 0, 0));
        constexpr uint32_t batchSize = kNumTrainingDataPerFrame / 4;                 // main:2350
        void* ts = stream;
        const bool rank0Trains = band && r->trainOnRank0;
        const bool overlap = r->overlapTraining && !rank0Trains;     // (the broadcast of rank 0's images has to follow the training on the caller's stream)
        if (overlap) {
            NRC_HIP(hipEventRecord(r->evData, static_cast<hipStream_t>(stream)));
            NRC_HIP(hipStreamWaitEvent(r->trainStream, r->evData, 0));
            ts = r->trainStream;
        }
        if (!rank0Trains || r->rank == 0) {
            for (uint32_t step = 0; step < 4; ++step) {
                const char* q = static_cast<const char*>(r->np.trainRadianceQueryBuffer[1]) + 56ull * step * batchSize;
                const char* t = static_cast<const char*>(r->np.trainTargetBuffer[1]) + 12ull * step * batchSize;
                NRC_GFX(gfx_nrc_train(ctx, ts, r->network, q, t, batchSize, (lossOut && step == 3) ? lossOut : nullptr));
            }
        }
        else if (lossOut) *lossOut = 0.0f;
        constexpr uint32_t kChecksumPeriod = 16;
        const bool compare = band && !rank0Trains && (r->frameIndex % kChecksumPeriod) == kChecksumPeriod - 1;
        uint32_t ownChecksum = 0;
        if (compare) {
            if (!r->dChecksum && nrc_alloc(r, &r->dChecksum, 8)) return 1;
            const uint32_t init[2] = { 0u, 1u };
            NRC_HIP(hipMemcpyAsync(r->dChecksum, init, 8, hipMemcpyHostToDevice, static_cast<hipStream_t>(ts)));
            NRC_GFX(gfx_nrc_params_checksum(ctx, ts, r->network, r->dChecksum));
            NRC_HIP(hipMemcpyAsync(&ownChecksum, r->dChecksum, 4, hipMemcpyDeviceToHost, static_cast<hipStream_t>(ts)));
        }
        if (overlap) {
            NRC_HIP(hipEventRecord(r->evTrained, r->trainStream));
            r->trainPending = true;
        }
        if (compare) {
            if (overlap) { NRC_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), r->evTrained, 0)); r->trainPending = false; }
            gfxh_exchange_desc d; std::memset(&d, 0, sizeof(d));
            d.kind = GFXH_EXCHANGE_ALLREDUCE_SUM_U32; d.counters = r->dChecksum; d.numCounters = 2;
            if (exchange(d)) return 1;
            uint32_t sum[2] = { 0u, 0u };
            NRC_HIP(hipMemcpyAsync(sum, r->dChecksum, 8, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
            NRC_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
            if (sum[0] != ownChecksum * sum[1]) {     // (mod 2^32: equal copies sum to world x own)
                g_nrcError = "gfxh_nrc_render_frame: the ranks' copies of the network have diverged (checksum of the inference images, frame " +
                             std::to_string(r->frameIndex) + "); GFX_NRC_TRAIN_ON_RANK0=1 trains on rank 0 and broadcasts";
                return 1;
            }
        }
        if (rank0Trains) {   // rank 0's freshly packed inference images -> everyone
            gfxh_exchange_desc d; std::memset(&d, 0, sizeof(d));
            d.kind = GFXH_EXCHANGE_BROADCAST;
            for (int which = 0; which < 2; ++which) {
                void* p = nullptr; uint64_t bytes = 0;
                NRC_GFX(gfx_nrc_inference_image_async(ctx, ts, r->network, which, &p, &bytes));   // packed on the training stream, behind the four steps
                if (!p || !bytes) continue;
                gfxh_exchange_buffer& b = d.buffers[d.numBuffers++];
                b.base = p; b.bytesPerPixel = 1; b.numPlanes = 1; b.planeStride = bytes;
            }
            if (exchange(d)) return 1;
        }
    }
    else if (band) {   // the tile size of the next frame adapts to the global record count
        gfxh_exchange_desc d; std::memset(&d, 0, sizeof(d));
        d.kind = GFXH_EXCHANGE_ALLREDUCE_SUM_U32; d.counters = r->np.numTrainingData[bufferIndex]; d.numCounters = 1;
        if (exchange(d)) return 1;
    }
    if (band) {
        gfxh_exchange_desc d; std::memset(&d, 0, sizeof(d));
        d.kind = GFXH_EXCHANGE_GATHER_BANDS; d.numBuffers = 1;
        d.buffers[0].base = r->sp.beautyAccumBuffer; d.buffers[0].bytesPerPixel = 16; d.buffers[0].numPlanes = 1;
        d.buffers[0].planeStride = 16ull * W * H;
        if (exchange(d)) return 1;
    }
    r->prevCamera = cfg.camera;
    ++r->frameIndex;
    return 0;
}

int gfxh_nrc_set_exchange(gfxh_nrc* r, gfxh_exchange_fn fn, void* user, int rank) {
    r->exchange = fn; r->exchangeUser = user; r->rank = rank;
    // Every rank training its own copy needs a training step that is reproducible bit for bit: the default hash-grid gradient (summed in
    // LDS tables in a defined order) is; GFX_NRC_GRID_GRAD = f32 | f16atomic sums with global atomics in arrival order, so the copies
    // would drift apart: those modes train on rank 0 and broadcast.
    if (const char* e = std::getenv("GFX_NRC_GRID_GRAD"))
        if (std::strcmp(e, "f32") == 0 || std::strcmp(e, "f16atomic") == 0) r->trainOnRank0 = true;
    return 0;
}

int gfxh_nrc_rebuild_accel(gfxh_nrc* r, void* stream) {
    r->viewMoved = true;
    if (gfx_accel_build(r->ctx, stream, &r->accel)) { g_nrcError = gfx_last_error(r->ctx); return 1; }
    return 0;
}
void* gfxh_nrc_beauty_buffer(gfxh_nrc* r) { return r->sp.beautyAccumBuffer; }
int gfxh_nrc_set_env(gfxh_nrc* r, float* texels, uint32_t w, uint32_t h, float powerCoeff, float rotation) {
    // frames in flight may still read the tables this call replaces: wait for them, then the previous map's allocations can go
    NRC_HIP(hipDeviceSynchronize());
    for (void* p : r->envAllocations) (void)hipFree(p);
    r->envAllocations.clear();
    void* allocations[GFXH_ENV_MAX_ALLOCATIONS];
    uint32_t numAllocations = 0;
    const int err = gfxh_env_upload(texels, w, h, &r->sp, allocations, &numAllocations);
    for (uint32_t i = 0; i < numAllocations; ++i) r->envAllocations.push_back(allocations[i]);
    if (err) { g_nrcError = std::string("gfxh_env_upload: ") + gfxh_restir_last_error(); return 1; }
    r->envPowerCoeff = powerCoeff; r->envRotation = rotation;
    r->viewMoved = true;
    return 0;
}

uint64_t gfxh_nrc_network(gfxh_nrc* r) {
    if (r->trainStream) (void)hipStreamSynchronize(r->trainStream);   // whoever asks for the network sees it trained
    return r->network;
}
int gfxh_nrc_stats(gfxh_nrc* r, uint32_t* numTrainingData, uint32_t tileSize[2], uint32_t* numInferenceQueries) {
    if (r->statsPending) {      // whole-frame renderer: the last frame's figures arrive through pinned memory
        if (hipEventSynchronize(r->evStats) != hipSuccess) return 1;
        r->statsPending = false;
    }
    if (r->hostStats && !(r->exchange && !(r->cfg.rowBegin == 0 && r->cfg.rowEnd == 0)) && r->frameIndex > 0) {
        r->lastNumTrainingData = r->hostStats[0]; r->lastTileSize[0] = r->hostStats[1]; r->lastTileSize[1] = r->hostStats[2];
        r->lastNumInferenceQueries = r->hostStats[3];
    }
    if (numTrainingData) *numTrainingData = r->lastNumTrainingData;
    if (tileSize) { tileSize[0] = r->lastTileSize[0]; tileSize[1] = r->lastTileSize[1]; }
    if (numInferenceQueries) *numInferenceQueries = r->lastNumInferenceQueries;
    return 0;
}
int gfxh_nrc_outputs_consumed(gfxh_nrc* r, void* stream) {
    if (!r) { g_nrcError = "gfxh_nrc_outputs_consumed: null renderer"; return 1; }
    if (!r->evConsumed) NRC_HIP(hipEventCreateWithFlags(&r->evConsumed, nrc_order_event_flags()));
    NRC_HIP(hipEventRecord(r->evConsumed, static_cast<hipStream_t>(stream)));
    r->consumedPending = true;
    return 0;
}
const char* gfxh_nrc_last_error(void) { return g_nrcError.c_str(); }

} // extern "C"
