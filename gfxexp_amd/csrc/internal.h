// internal.h -- host-side state of the library behind include/gfxexp.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "device_types.h"

namespace gfx {

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };
#define GFX_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) \
    throw ::gfx::HipError(std::string(#call) + ": " + hipGetErrorString(e__)); } while (0)

// Growable device allocation owned by the library (BVH memory, ray queues, scratch).
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void reserve(size_t n) {
        if (n <= bytes) return;
        if (p) GFX_HIP(hipFree(p));
        p = nullptr; bytes = 0;
        GFX_HIP(hipMalloc(&p, n));
        bytes = n;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

struct HostTexture {
    uint32_t width = 0, height = 0, format = 0;
    std::vector<uint8_t> texels;   // tightly packed rows in `format`
};
struct HostGeom {
    std::vector<DevVertex> vertices;
    std::vector<uint32_t> triangles;   // 3 per triangle
    uint32_t materialSlot = 0;
};
struct HostInstance {
    uint32_t group = 0;
    float transform[12];
    float prevTransform[12];
    // set by gfx_instance_set_transform (InstanceController::update): curToPrevTransform = prev * invert(cur),
    // optional caller-supplied normal matrix (row-major 3x3)
    bool animated = false, hasNormalMatrix = false;
    // lives in the animated subtree of the BVH (gfx_instance_set_dynamic, or implied by the first
    // gfx_instance_set_transform): later transform updates rebuild only that subtree
    bool dynamic = false;
    float curToPrev[12];
    float normalMatrix[9];
};
void instance_cur_to_prev(const float prev[12], const float cur[12], float out[12]);   // scene.cpp

struct Accel {
    // nodes and triangle records are both 64-byte items and live in ONE allocation (nodes first, triangle
    // records from item `triItemOffset`), so the traversal kernel fetches any item as base + (index << 6)
    DevBuf nodes, links, triIds;
    uint32_t triItemOffset = 0;
    uint32_t numNodes = 0, numTris = 0, numInputTris = 0, maxDepth = 0;
    // split tree (static + animated subtree under a two-child root, lbvh.hip): world boxes of the two subtree
    // roots (2 x 8 floats, device), triangle count and node range of the static part
    bool split = false;
    DevBuf rootBoxes;
    uint32_t numStaticTris = 0, staticNodeEnd = 0, staticDepth = 0;   // staticDepth: levels of the static subtree (lbvh_update_dynamic re-derives maxDepth)
    Bvh8Tri* trisPtr() const { return reinterpret_cast<Bvh8Tri*>(nodes.as<Bvh8Node>() + triItemOffset); }
    DevAccel dev() const {
        DevAccel a; a.nodes = nodes.as<Bvh8Node>(); a.links = links.as<Bvh8Link>(); a.tris = trisPtr(); a.numNodes = numNodes; a.numTris = numTris;
        a.triItemOffset = triItemOffset;
        return a;
    }
};

struct KernelTiming { double ms = 0; uint32_t calls = 0; };

struct RestirParams {
    gfx_restir_static_params s;
    gfx_restir_frame_params f;
    uint32_t currentReservoirIndex = 0;
    uint32_t spatialNeighborBaseIndex = 0;
    bool valid = false;
};

struct NrcNet;

// Scheduling knobs of a context (none changes a result).  Defaults are the measured best on MI355X; gfx_ctx_create
// overrides them from the environment (GFX_PIXEL_MAP, GFX_SUPER_X, GFX_SUPER_Y, GFX_TRACE_BLOCKS_PER_CU, GFX_TRACE_REFILL,
// GFX_TRACE_BATCH) and gfx_tunable_set changes them per context at run time (profiles/, tools/).
struct Tunables {
    int pixelMap = 2;                // restir_common.hip.h PixelGrid::mode: 0 scan lines, 1 8x8 tiles, 2 tiles + XCD supertiles
    int superShiftX = 2, superShiftY = 2;   // supertile = 2^2 x 2^2 blocks of 16 x 16 pixels = 64 x 64 pixels (profiles/r03_pixel_map_supers.jsonl)
    int traceBlocksPerCU = 4;        // persistent traversal grid: blocks of 256 per CU (LDS: 4 x 40 KiB)
    int traceRefill = 8;             // refill a wave when at least this many lanes are idle
    int traceBatch = 64;             // rays bought per device atomic (32 and 128 are slower)
    int temporalHints = 1;           // primary rays test the triangle their pixel hit one frame ago first (trace.hip)
    int candidateSplit = 0;          // k_initial_candidates: lanes per pixel (1, 2, 4); 0 = by launch size (restir.hip)
    int blockOrder = 1;              // k_initial_fused: blocks start by decreasing cost of one frame ago (restir.hip k_order_blocks); 0 = index order
    int fusePasses = 0;              // ReSTIR ray passes as one kernel each (restir.hip k_*_fused): 0 = small launches only, 1 never, 2 always
    int nrcStagedInfer = 0;          // k_nrc_infer_staged (hash-grid levels through LDS): 0 = large batches only, 1 never, 2 always, 3 always in the software-pipelined form k_nrc_infer_piped (nrc.hip)
    int ptDiag = 0;                  // one-kernel path tracers count wave iterations and the lanes that held a ray in them (gfx_pt_diag_read)
    int ptRegen = 0;                 // baseline path tracer, one-kernel form: blocks of 256 per CU of the regenerating launch (pathtrace.hip k_pt_regen); 0 = k_pt_fused
                                     // (the default: regeneration fills the lanes -- 0.30 -> 0.6 of them hold a ray -- and still takes 17 % longer on the
                                     // 512 x 512 bunny frame: profiles/r06_experiments.txt 3)
    int ptRegenMin = 16;             // ... and the idle lanes a wave waits for before it refills (1 = at once)
    int ptOverlap = 1;               // path tracers: the NEE (any-hit) trace + its apply kernel of a bounce run on a second stream underneath
                                     // the extension (closest-hit) trace of the same bounce (pathtrace.hip)
};

struct Context {
    int device = 0;
    int numCUs = 0;                  // of `device` (gfx_ctx_create)
    Tunables tune;
    uint32_t nrcInferPipedLds = 0;            // dynamic LDS bytes k_nrc_infer_piped has been enabled for on this device
    bool nrcInferStagedConfigured = false;   // k_nrc_infer_staged has been given its 128 KiB of dynamic LDS on this device
    size_t nrcTrainLdsConfigured = 0; // dynamic LDS bytes k_nrc_train has been enabled for on this device
    std::string lastError;
    // scene (host mirror)
    std::vector<gfx_material> materials;
    std::vector<HostTexture> textures;       // indexed by slot; slot 0 stays empty ("no texture")
    std::vector<HostGeom> geoms;
    std::vector<std::vector<uint32_t>> groups;
    std::vector<HostInstance> insts;
    bool sceneDirty = true;
    // scene (device)
    DevBuf dMaterials, dGeomInsts, dInsts, dVertices, dTriangles, dSlotPool, dFlatGeoms, dLightW, dLightP, dLightCDF, dLightRefs, dEmitterRecs, dEmitterRecExtras, dLightNormalMatrices, dInstMatrixIndex, dTextures, dTexelPool, dSrgbLut, dEmitterTexRefs;
    bool anyEmittanceTexture = false;
    std::vector<LightGeomRef> hLightRefs;
    uint32_t numEmitterRecs = 0;
    std::vector<DevGeomInst> hGeomInsts;
    std::vector<DevInstance> hInsts;
    std::vector<DevFlatGeom> hFlatGeoms;
    uint32_t totalTriangles = 0;
    // the flattened geometry list split by HostInstance::dynamic: [0] static, [1] animated (lbvh.hip builds one
    // subtree over each when both exist); instances whose transform changed since the last upload
    std::vector<SubsetGeom> hSubset[2];
    DevBuf dSubset[2];
    uint32_t subsetTris[2] = { 0, 0 };
    bool transformsDirty = false;
    std::vector<uint32_t> movedInsts;
    bool emitterRecsDirty = false;
    bool instDistValid = false;      // the instance-level distribution on the device matches the current transforms
    uint32_t lightPoolSize = 0;
    uint32_t lightInstDistOffset = 0;
    DevBuf dLightInstIntegral;       // float[4]; [0] = integral of the instance-level distribution, [1..2] guide header
    DevBuf dLightInstGuide;          // uint16[lightInstGuideCells]
    uint32_t lightInstGuideCells = 0;
    bool lightsStaticBuilt = false;
    // emitter interval table (emitter_spans.h): spans[numEmitterRecs], guide[spanGuideCells], header uint32[4],
    // instance interval starts uint32[numInsts + 1] (build scratch)
    DevBuf dSpans, dSpanGuide, dSpanHeader, dSpanInstBegin;
    uint32_t spanGuideCells = 0;
    // ticket areas of k_trace per counter buffer ([0] smallCounters, [1] any other): which of the two areas the next launch draws from
    // (the launch before it zeroed that one), whether both have been zeroed once, and the stream of the last launch (a launch on another
    // stream waits for the event recorded behind that launch and re-zeroes: the hand-over between launches is stream order)
    struct TicketState { bool zeroed = false; uint32_t next = 0; void* buffer = nullptr; hipStream_t stream = nullptr; hipEvent_t lastLaunch = nullptr; } ticketState[3];
    uint32_t numLightMatrices = 0;   // distinct normal matrices of the emitter instances (scene.cpp light_matrices_upload)
    // pinned staging of light_matrices_upload (two buffers, an event each): the per-frame re-upload after an animated emitter moved is
    // truly asynchronous -- a host wait there would stall the frame loop and with it the overlap of consecutive frames
    struct PinnedStage { void* p = nullptr; size_t bytes = 0; hipEvent_t done = nullptr; } lightStage[2];
    uint32_t lightStageNext = 0;
    DevScene devScene() const;
    // accels
    std::vector<Accel*> accels;
    uint32_t maxLeafTris = 4;
    // ray scratch
    DevBuf rayOrg, rayDir, rayOut, rayHits, spill, pixelRaySlot, shadeScratch, spatialScratch, smallCounters;
    // The G-buffer pass has its own ray queue / hit / stack-spill / ticket scratch, so a driver may run the next
    // frame's G-buffer pass on a second stream underneath the tail of the current frame (restir_driver.cpp).
    DevBuf gbRayOrg, gbRayDir, gbRayHits, gbSpill, gbCounters;
    // ... and the path tracers a third (stack spill + ticket areas) for the NEE trace that runs on `auxStream` underneath the extension
    // trace of the same bounce (pathtrace.hip); auxFork / auxJoin order the two streams
    DevBuf auxSpill, auxCounters;
    // fused ReSTIR kernels (restir.hip): step counts per block of the last launch and the block order made from them, for one launch shape
    // ([0] k_initial_fused, [1] k_shading_fused, [2] k_gbuffer_fused, [3] k_initial_candidates, [4] k_pt_fused).  k_order_blocks runs on auxStream behind the launch that wrote the costs (`counted`);
    // the next launch of that kind waits for `ordered`.
    struct BlockOrder { DevBuf cost, order; uint64_t key = 0; uint32_t blocks = 0; bool valid = false; hipEvent_t counted = nullptr, ordered = nullptr; } blockOrders[5];
    hipStream_t auxStream = nullptr;
    hipEvent_t auxFork = nullptr, auxJoin = nullptr;
    // path tracer scratch (pathtrace.hip)
    DevBuf ptPending, ptExtOrg, ptExtDir, ptExtOwner, ptState;
    DevBuf rearchSlots;
    DevBuf nrcState, neeTrainIdx;
    DevBuf nrcQueryCount;            // u32: inference batch size of the NRC frame (GFX_PT_NRC_COUNT_QUERIES)
    // build scratch
    DevBuf bTris, bBoxes, bKeys, bKeysAlt, bVals, bValsAlt, bSortTemp, bNodesLR, bParents, bFlags, bNodeBoxes, bRanges, bQueueA, bQueueB, bCounters, bCosts, bDec, bFlatIdx;
    // restir
    RestirParams restir;
    gfx_regir_params regir;
    std::vector<NrcNet*> nrcNets;
    gfx_nrc_params nrcRender;
    bool nrcRenderValid = false;
    bool regirValid = false;
    // instrumentation
    bool timingEnabled = false;
    std::map<std::string, KernelTiming> timings;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pendingEvents;
    bool countersEnabled = false;
    DevBuf dTraceCounters;     // u64[8]: any-hit launches {nodes, triangles, rays, spills}, closest-hit launches {same}
    bool countersSplit = true; // false while gfx_trace counts into a caller-supplied u64[4]
    DevBuf dTraceDiag;         // u64[8] scheduling diagnostics of counting launches
    DevBuf ptDiag;             // u64[8] of the one-kernel path tracers ("pt_diag")
    ~Context();
};

// Time one kernel launch with HIP events on `stream` when timing is enabled.
struct ScopedKernelTimer {
    Context& ctx; hipStream_t stream; const char* name; hipEvent_t a = nullptr, b = nullptr;
    ScopedKernelTimer(Context& c, hipStream_t s, const char* n) : ctx(c), stream(s), name(n) {
        if (!ctx.timingEnabled) return;
        GFX_HIP(hipEventCreate(&a)); GFX_HIP(hipEventCreate(&b));
        GFX_HIP(hipEventRecord(a, stream));
    }
    ~ScopedKernelTimer() {
        if (!a) return;
        (void)hipEventRecord(b, stream);
        ctx.pendingEvents.push_back({ name, { a, b } });
    }
};

// ---- scene.cpp
void scene_upload(Context& ctx, hipStream_t stream);
void transforms_upload(Context& ctx, hipStream_t stream);   // moved instances only (scene.cpp)
// ---- lbvh.hip
void lbvh_build(Context& ctx, hipStream_t stream, Accel& out);
bool lbvh_update_dynamic(Context& ctx, hipStream_t stream, Accel& out);   // false: a full lbvh_build is needed
// ---- trace.hip
// The small per-context counter buffers (Context::smallCounters / gbCounters): bytes [0, 1024) hold the ray-queue counts of the passes
// (restir.hip, pathtrace.hip), bytes [1024, 1024 + 2 x 4096) the two ticket-counter areas of k_trace (trace.hip: a launch draws from one and zeroes the other for the next launch).  Always reserved at full size,
// so that no later reserve() moves it under a pointer a pass already holds.
constexpr size_t kSmallCountersBytes = 1024 + 2 * 4096;
constexpr size_t kSmallCountersTicketOffset = 1024;
struct TraceLaunch {
    DevAccel accel;
    const float4* rayOrgTmin; const float4* rayDirTmax;
    uint32_t numRays; const uint32_t* numRaysPtr;   // device-side count overrides numRays when non-null
    void* out;
    int mode;
    DevBuf* spill = nullptr;        // stack-spill area / ticket word; null = the context's shared ones
    DevBuf* counters = nullptr;
    uint32_t* perRayItems = nullptr; // counting launches: items fetched per ray
    uint32_t* zeroWords[2] = { nullptr, nullptr };   // device words the launch sets to zero (queue heads the NEXT pass appends to:
                                    // saves the path tracers two memsets per bounce); must not be this launch's own count
    bool hintFromOut = false;       // closest-hit: out[] still holds the previous launch's results for the same rays (primary rays of
                                    // the previous frame): each ray tests that triangle first (trace.hip)
};
void trace_launch(Context& ctx, hipStream_t stream, const TraceLaunch& t);
// ---- restir.hip: cost-ordered block start (Context::blockOrders)
void block_order_begin(Context& ctx, hipStream_t stream, int which, uint32_t blocks, uint64_t key, uint32_t minBlocks, const uint32_t*& order, uint32_t*& cost);
void block_order_end(Context& ctx, hipStream_t stream, int which, uint32_t blocks, uint32_t* cost);
// ---- diag.hip
void stream_copy(Context& ctx, hipStream_t stream, void* dDst, const void* dSrc, size_t bytes);
// ---- textures.hip
void texture_sample(Context& ctx, hipStream_t stream, uint32_t texSlot, const void* dUv, uint32_t n, void* dOut, int gather);
// ---- lights.hip
void light_matrices_upload(Context& ctx, hipStream_t stream);
void lights_build_static(Context& ctx, hipStream_t stream);
void lights_build_instances(Context& ctx, hipStream_t stream, uint32_t bufferIndex);
// ---- restir.hip
void restir_launch(Context& ctx, hipStream_t stream, int pass, uint32_t width, uint32_t height, uint32_t rowBegin, uint32_t rowEnd, uint32_t gapBegin = 0, uint32_t gapEnd = 0);
void restir_copy_to_linear(Context& ctx, hipStream_t stream, void* color, void* albedo, void* normal, void* motion);
void restir_visualize(Context& ctx, hipStream_t stream, const void* linearBuffer, int bufferType, float mvOffset, float mvScale, uint32_t width, uint32_t height, void* out);
// ---- nrc.hip
struct NrcNet;
NrcNet* nrc_create(Context& ctx, int posEnc, uint32_t numHiddenLayers, float learningRate);
void nrc_destroy(NrcNet* net);
uint32_t nrc_num_params(const NrcNet* net);
void nrc_set_params(Context& ctx, hipStream_t stream, NrcNet* net, const float* hostParams, uint32_t count);
void nrc_get_params(NrcNet* net, int which, float* hostOut, uint32_t count);
void nrc_inference_image(Context& ctx, NrcNet* net, int which, void** dPtr, uint64_t* bytes, hipStream_t stream = nullptr, bool onStream = false);
void nrc_params_checksum(Context& ctx, hipStream_t stream, NrcNet* net, uint32_t* dOut);
void nrc_infer(Context& ctx, hipStream_t stream, NrcNet* net, const float* dInputs, uint32_t numData, float* dPredictions, const uint32_t* dNumData = nullptr);
void nrc_train(Context& ctx, hipStream_t stream, NrcNet* net, const float* dInputs, const float* dTargets, uint32_t numData, float* lossOnCPU);
// ---- pathtrace.hip
void pathtrace_launch(Context& ctx, hipStream_t stream, int pass, uint32_t width, uint32_t height,
                      uint32_t maxPathLength, uint32_t rowBegin, uint32_t rowEnd);

} // namespace gfx
