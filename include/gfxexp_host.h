/*
 * gfxexp_host.h -- host layer above the C ABI of gfxexp.h: scene construction and the headless
 * ReSTIR DI frame driver.  It mirrors what the reference's host program does around the hot path
 * (restir_di/restir_di_main.cpp) without the window / ImGui / OptiX parts:
 *   scene building      createTriangleMeshes / createRectangleLight / createInstance
 *                       (common/common_host.cpp:2178-2429, 2431-2476, 2582-2656)
 *   material constants  immediate 1x1 textures, 8-bit + sRGB decode (common_host.cpp:1045-1073,
 *                       1602-1659; basic_types.h:5396-5402)
 *   frame loop          buffer allocation + seeding (restir_di_main.cpp:1210-1325), Halton disk table
 *                       (:1487-1542), per-frame sequencing and index bookkeeping (:2311-2493)
 * Everything here is plain C ABI as well so tests and bench.py drive it through ctypes.
 */
#ifndef GFXEXP_HOST_H
#define GFXEXP_HOST_H

#include "gfxexp.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gfxh_scene gfxh_scene;

gfxh_scene* gfxh_scene_create(void);
void gfxh_scene_destroy(gfxh_scene* s);
const char* gfxh_last_error(void);

/* createDiffuseAndSpecularMaterial with immediate values (common_host.cpp:1560-1700): diffuse and
 * specular go through 8-bit quantisation + sRGB decode, smoothness through 8-bit quantisation,
 * emittance is kept as float; hasEmittance = any(emittance != 0). Returns the material slot. */
uint32_t gfxh_scene_add_material_traditional(gfxh_scene* s, const float diffuse[3], const float specular[3],
                                             float smoothness, const float emittance[3]);
/* Raw material (values as sampled). */
uint32_t gfxh_scene_add_material(gfxh_scene* s, const gfx_material* m);

/* Textures.  Slots are 1-based (0 = "no texture" in gfx_material).  gfxh_scene_load_texture is loadTexture
 * (common_host.cpp:1163-1244) for the formats this build decodes itself -- binary PPM / PGM, PFM, uncompressed BMP and
 * TGA; DDS / PNG / JPEG assets are converted offline -- cached per path: 8-bit images are stored as `format8`
 * (GFX_TEX_RGBA8_SRGB for colour maps, GFX_TEX_RGBA8_UNORM for normal maps, GFX_TEX_R8_UNORM / RG8 take the first
 * channels), float images as GFX_TEX_RGBA32F.  Both return the slot, 0 on failure. */
uint32_t gfxh_scene_add_texture(gfxh_scene* s, uint32_t width, uint32_t height, uint32_t format, const void* texels);
uint32_t gfxh_scene_load_texture(gfxh_scene* s, const char* path, uint32_t format8);
uint32_t gfxh_scene_num_textures(gfxh_scene* s);
int gfxh_scene_get_texture(gfxh_scene* s, uint32_t slot, uint32_t* width, uint32_t* height, uint32_t* format, const void** texels);

/* Geometry / groups / instances (return slot indices). */
uint32_t gfxh_scene_add_geom(gfxh_scene* s, const gfx_vertex* v, uint32_t nv, const uint32_t* tris, uint32_t nt, uint32_t matSlot);
uint32_t gfxh_scene_add_group(gfxh_scene* s, const uint32_t* geomSlots, uint32_t n);
uint32_t gfxh_scene_add_instance(gfxh_scene* s, uint32_t group, const float xfm[12]);

/* OBJ + MTL reader ("-obj <path> <scale> trad"): one geometry instance per material, one group.
 * Returns the group index or 0xFFFFFFFF on failure.  The scale belongs to the instance transform
 * (the reference passes it as the mesh pre-transform, restir_di_main.cpp:1119-1124). */
uint32_t gfxh_scene_load_obj(gfxh_scene* s, const char* path);
/* The same with the material convention of "-obj <path> <scale> trad|simple_pbr" (restir_di_main.cpp:701-726,
 * MaterialConvention common_host.h:594-597): simple_pbr reads Kd / map_Kd as base colour and Ks / map_Ks as
 * (occlusion, roughness, metallic) for the SimplePBR BRDF. */
enum gfxh_material_convention { GFXH_MATCONV_TRADITIONAL = 0, GFXH_MATCONV_SIMPLE_PBR = 1 };
uint32_t gfxh_scene_load_obj_conv(gfxh_scene* s, const char* path, int materialConvention);
/* createRectangleLight (common_host.cpp:2431-2476): XZ rectangle facing -Y. Returns the group. */
uint32_t gfxh_scene_add_rectangle(gfxh_scene* s, float width, float depth, const float emittance[3]);
/* "-rect-emitter-tex <path>" (restir_di_main.cpp:693-699): the rectangle's emittance read from an image. */
uint32_t gfxh_scene_add_rectangle_textured(gfxh_scene* s, float width, float depth, const float emittance[3], const char* emitterTexturePath);

/* Procedural "street" stand-in for Bistro Exterior (the asset is not redistributable / absent):
 * tessellated ground, facade blocks with window grids, instanced props, and many small emitters. */
typedef struct gfxh_street_params {
    uint32_t seed;
    uint32_t groundTess;        /* ground is groundTess x groundTess quads */
    uint32_t numBuildings;
    uint32_t facadeTess;        /* window grid resolution per facade */
    uint32_t numProps;          /* instanced icospheres / crates */
    uint32_t propSubdiv;        /* icosphere subdivision level (0..5) */
    uint32_t numLamps;          /* small box emitters on poles */
    uint32_t numSigns;          /* emissive quads on facades */
    float extent;               /* half size of the street block in metres */
    float lampEmittance;
    float signEmittance;
    uint32_t textured;          /* 1: ground / facades / crates get albedo + smoothness + normal maps, signs a float emittance map */
    /* depth complexity (0 = none; the geometry above does not depend on these): */
    uint32_t numTrees;          /* instanced trees: trunk + leavesPerTree randomly oriented 12-30 cm leaf cards in the crown */
    uint32_t leavesPerTree;
    uint32_t numWires;          /* sagging 3-cm cables across the street (8 segments each) */
    uint32_t numRailings;       /* 3-m railing segments (2 rails + 24 bars of 1.6 cm) along the kerbs */
} gfxh_street_params;
int gfxh_scene_make_street(gfxh_scene* s, const gfxh_street_params* p);

/* Enumeration (to feed the same arrays to another consumer, e.g. the test oracle). */
int gfxh_scene_counts(gfxh_scene* s, uint32_t counts[5]); /* materials, geoms, groups, insts, triangles(total over instances) */
int gfxh_scene_get_material(gfxh_scene* s, uint32_t i, gfx_material* out);
int gfxh_scene_get_geom(gfxh_scene* s, uint32_t i, const gfx_vertex** v, uint32_t* nv, const uint32_t** tris, uint32_t* nt, uint32_t* matSlot);
int gfxh_scene_get_group(gfxh_scene* s, uint32_t i, const uint32_t** geomSlots, uint32_t* n);
int gfxh_scene_get_instance(gfxh_scene* s, uint32_t i, uint32_t* group, float xfm[12]);
/* World-space bounds of all instances: {minx,miny,minz,maxx,maxy,maxz}. */
int gfxh_scene_bounds(gfxh_scene* s, float bounds[6]);

/* Push the scene through gfx_material_set / gfx_geom_create / gfx_group_create / gfx_instance_create. */
int gfxh_scene_upload(gfxh_scene* s, gfx_ctx* ctx);

/* 3x4 row-major transform from scale, yaw/pitch/roll (degrees, the reference's CLI convention) and
 * translation: T * R * S. */
void gfxh_make_transform(float scale, float rollDeg, float pitchDeg, float yawDeg, const float pos[3], float out[12]);
/* Camera orientation from roll/pitch/yaw in degrees (restir_di_main.cpp:2667-2676 qFromEulerAngles), row-major 3x3. */
void gfxh_make_orientation(float rollDeg, float pitchDeg, float yawDeg, float out[9]);

/* restir_di_main.cpp:1316-1321 / :1217: PCG32 states from std::mt19937_64(seed). */
void gfxh_seed_rng_states(uint64_t* states, uint64_t count, uint64_t seed);
/* restir_di_main.cpp:1487-1542: 1024 Halton(2,3) samples mapped to the unit disk (float2 x 1024). */
void gfxh_spatial_neighbor_deltas(float* out2x1024);

/* Environment light (loadEnvironmentalTexture, common/common_host.cpp:2658-2711 and the
 * RegularConstantContinuousDistribution2D build, :204-357): importance = luminance * sin(theta) per
 * texel of a lat-long float4 image; per-row piecewise-constant PDFs/CDFs (Kahan sums) and the
 * marginal over rows.  texels are clamped to [0, 65504] in place like the reference.  Output sizes:
 * rowPDF h*w, rowCDF h*(w+1), rowIntegrals h, topPDF h, topCDF h+1. */
int gfxh_env_build_importance(float* texels4, uint32_t w, uint32_t h, float* rowPDF, float* rowCDF,
                              float* rowIntegrals, float* topPDF, float* topCDF, float* topIntegral);
/* Guide tables for the CDFs gfxh_env_build_importance produced (gfx_restir_static_params::envRowGuide / envTopGuide):
 * the samplers' log2(n)-step searches become a table lookup plus a one- or two-entry bracket search with identical
 * results.  Returns 1 when the tables are usable (every CDF monotone, w and h <= 65536), 0 otherwise (pass NULL). */
int gfxh_env_build_guides(const float* rowCDF, const float* topCDF, uint32_t w, uint32_t h, uint16_t* rowGuide, uint16_t* topGuide);
/* gfx_restir_static_params::envRowTable: the rows of the map interleaved -- h rows of GFX_ENV_ROW_STRIDE(w) records of 32 bytes
 * {cdf, pdf, guide, r, g, b, next record's cdf, 0} (records w + 1 .. stride - 1 of a row are zero) from the (clamped) texels, the
 * conditional PDFs / CDFs and a usable row guide (gfxh_env_build_guides returned 1).  outRecords: 32 x h x GFX_ENV_ROW_STRIDE(w) bytes.
 * Same samples as with the separate arrays; a third of the memory traffic per sample. */
void gfxh_env_build_row_table(const float* texels4, const float* rowPDF, const float* rowCDF, const uint16_t* rowGuide, uint32_t w, uint32_t h, void* outRecords);
/* gfx_restir_static_params::envRowSketch (described there).  Writes *numRecords = h + the child records the map needs; the records
 * themselves (GFX_ENV_SKETCH_WORDS 32-bit words each) go to outSketch when it is non-NULL and capacityRecords >= *numRecords (call once
 * with NULL to size the buffer).  Returns the number of first-level cells (of 32 h) whose prediction the builder verified -- for every
 * column that a cell's range of u reaches, at both ends of the range, with the arithmetic the device uses -- to lie within one column
 * of the bisection's result. */
uint32_t gfxh_env_build_row_sketch(const float* rowCDF, uint32_t w, uint32_t h, void* outSketch, uint32_t capacityRecords, uint32_t* numRecords);
/* The device side of "-env-texture" (restir_di_main.cpp:1188-1197, common_host.cpp:204-357) in one call, for the renderers below and for
 * callers that fill gfx_restir_static_params themselves: the importance tables, guides and row table of a lat-long float4 map (the three
 * functions above) are built, map and tables uploaded, and the env* fields of `sp` set.  Synchronous.  The device allocations it made
 * (at most GFXH_ENV_MAX_ALLOCATIONS, also on failure) are returned for the caller to hipFree once no launch reads `sp` any more. */
#define GFXH_ENV_MAX_ALLOCATIONS 10
int gfxh_env_upload(float* texels4, uint32_t w, uint32_t h, gfx_restir_static_params* sp, void** allocations, uint32_t* numAllocations);
/* Synthetic lat-long sky (gradient + sun disc) used as the stand-in environment map. */
void gfxh_env_make_sky(uint32_t w, uint32_t h, float sunElevationDeg, float sunAzimuthDeg, float sunRadiance, float* texels4);

/* ---------------------------------------------------------------- headless ReSTIR DI renderer */
typedef struct gfxh_restir gfxh_restir;

enum gfxh_renderer {
    GFXH_ORIGINAL_RESTIR_BIASED = 0,   /* restir_di_main.cpp:1958-1977 Renderer enum */
    GFXH_ORIGINAL_RESTIR_UNBIASED = 1,
    GFXH_REARCHITECTED_RESTIR_BIASED = 2,   /* frame loop :2423-2487, configs (5, 1, 1) :1966-1969 */
    GFXH_REARCHITECTED_RESTIR_UNBIASED = 3,
    GFXH_PATH_TRACE_BASELINE = 4,      /* path_tracing/path_tracing_main.cpp:2068-2093 frame loop */
    GFXH_PATH_TRACE_REGIR = 5          /* regir/regir_main.cpp:2021-2066 frame loop (grid 32 x 8 x 32 over the scene box) */
};
typedef struct gfxh_restir_config {
    uint32_t width, height;
    int renderer;
    uint32_t log2NumCandidateSamples;   /* 5 */
    uint32_t enableTemporalReuse;       /* 1 */
    uint32_t enableSpatialReuse;        /* 1 */
    uint32_t numSpatialReusePasses;     /* 2 biased / 1 unbiased */
    uint32_t numSpatialNeighbors;       /* 5 biased / 3 unbiased */
    float spatialNeighborRadius;        /* 20 */
    uint32_t useLowDiscrepancyNeighbors;/* 1 */
    uint32_t reuseVisibility;           /* 1 */
    uint32_t enableAccumulation;        /* 0 */
    uint32_t log2MaxNumAccums;          /* 16 */
    gfx_camera camera;                  /* fovY = 50 deg in the reference (:1613) */
    /* rows [rowBegin, rowEnd) owned by this process when a frame is split across GPUs; 0,0 = all. */
    uint32_t rowBegin, rowEnd;
    uint32_t maxPathLength;             /* GFXH_PATH_TRACE_BASELINE only; 5 (path_tracing_main.cpp:1519) */
    uint32_t enableJittering;           /* 0 (path_tracing_main.cpp:1515, restir_di_main.cpp) */
    /* GFXH_PATH_TRACE_REGIR only (regir_main.cpp:1112, 1733-1736) */
    float regirAabbMin[3], regirAabbMax[3];   /* scene.initialSceneAabb */
    uint32_t regirGridDimension[3];           /* 32 x 8 x 32 */
    uint32_t regirLog2CandidatesPerLightSlot; /* 3 */
    uint32_t regirLog2CandidatesPerCell;      /* 2 */
    uint32_t regirEnableTemporalReuse;        /* 1 */
    uint32_t regirEnableCellRandomization;    /* 1 */
    uint32_t enableBumpMapping;               /* 0 (restir_di_main.cpp:1986); normal maps through applyBumpMapping */
} gfxh_restir_config;

void gfxh_restir_default_config(gfxh_restir_config* cfg, uint32_t width, uint32_t height, int renderer);

/* Row-band plan of one rank for one frame.  A band [bandBegin, bandEnd) owns its rows;
 * every reuse pass reads neighbours up to `radiusRows` away, so earlier passes run on a halo that
 * shrinks by radiusRows per spatial pass:
 *   G-buffer + initial/temporal RIS : band +- (radiusRows * numSpatialPasses + maxMotionRows)
 *   spatial pass i                  : band +- radiusRows * (numSpatialPasses - 1 - i)
 *   shading                         : band
 * Within a frame halo pixels reproduce bit for bit what their owner computes (per-pixel RNG, same
 * inputs).  Across frames the owner's FINAL reservoir / ReservoirInfo / RNG state of the halo rows
 * must be refreshed once per frame: recv* are the row ranges to receive from the rank above
 * (rank - 1) / below (rank + 1), send* the own rows those ranks need.  All ranges are [begin, end). */
typedef struct gfxh_band_plan {
    uint32_t bandBegin, bandEnd, haloRows;
    uint32_t gbufferRows[2], initialRows[2], spatialRows[8][2], shadingRows[2];
    uint32_t recvAbove[2], sendAbove[2], recvBelow[2], sendBelow[2];
} gfxh_band_plan;
void gfxh_band_plan_compute(uint32_t height, uint32_t bandBegin, uint32_t bandEnd, uint32_t radiusRows,
                            uint32_t numSpatialPasses, uint32_t maxMotionRows, gfxh_band_plan* out);
/* The plan the renderer uses (band from cfg.rowBegin/rowEnd; whole frame when both are 0). */
int gfxh_restir_band_plan(gfxh_restir* r, gfxh_band_plan* out);
/* ---- band renderers without redundant work: strip exchange between the passes (second alternative).
 * With an exchange callback installed a band renderer (cfg.rowBegin / rowEnd) runs EVERY pass on its own rows only
 * and, at the points of a frame where a pass is about to read rows another rank owns, hands the callback a
 * description of what has to move: device buffers (base pointer, bytes per pixel, planes), the row ranges to send to
 * / receive from the rank above (rank - 1) and below (rank + 1), or a counter array to sum over all ranks, or the
 * band of the HDR buffer to all-gather.  The callback performs the transfer ordered with `stream` (RCCL:
 * ncclSend / ncclRecv / ncclAllReduce / ncclAllGather on that stream -- gfxh_rccl_exchange below; bench.py and the
 * gloo tests: torch.distributed).  Per-pixel RNG streams are advanced by their owner only, so nothing else is shared and
 * the result is bit-identical to the single-GPU frame.  Exchange points per frame:
 *   original ReSTIR     G-buffers (16+16+16 B/pixel) after the G-buffer pass, rows max(radius x passes, motion); reservoirs +
 *                       infos + RNG states (64 B/pixel) ONCE behind the candidate pass, rows radius x passes -- the spatial passes
 *                       another pass follows are recomputed on a halo that shrinks by radius rows per pass (stripMode 3, the
 *                       default; stripMode 1: reservoirs + infos, 56 B/pixel, before every spatial pass, rows radius); final
 *                       reservoirs + infos after the last reuse pass, rows motion (the next frame's temporal pass)
 *   rearchitected       everything the next frame reads from "the previous frame" (G-buffers, sample visibility,
 *                       reservoirs, infos: 108 B/pixel) once at the end of the frame, rows radius + motion
 *   ReGIR path tracer   all-reduce(sum) of perCellNumAccesses (one u32 per cell) before the last-access update
 *   every renderer      all-gather of the float4 HDR bands at the end of the frame
 * maxMotionRows bounds |motion vector y| over the run (0 = static camera and scene: a camera or instance that moves
 * then makes gfxh_restir_render_frame fail instead of silently dropping temporal reuse along the seams). */
enum gfxh_exchange_kind {
    GFXH_EXCHANGE_STRIPS = 0, GFXH_EXCHANGE_ALLREDUCE_SUM_U32 = 1, GFXH_EXCHANGE_GATHER_BANDS = 2,
    /* NRC band renderers (gfxh_nrc_set_exchange): */
    GFXH_EXCHANGE_GATHER_RECORDS = 3,   /* buffers[k] = record arrays (bytesPerPixel = bytes per record); `counters` is a HOST uint32_t[2]:
                                         * in [0] = this rank's record count, out [0] = the total; afterwards every rank holds all
                                         * records, rank 0's first, then rank 1's, ... (numCounters = capacity in records) */
    GFXH_EXCHANGE_BROADCAST = 4         /* buffers[k]: planeStride bytes at base, from rank 0 to everyone */
};
typedef struct gfxh_exchange_buffer {
    void* base;               /* full-frame device buffer, row-major */
    uint32_t bytesPerPixel;   /* per plane */
    uint32_t numPlanes;       /* reservoirs: 3 planes of 16 B */
    uint64_t planeStride;     /* bytes between planes */
} gfxh_exchange_buffer;
/* The streams of a renderer ("lanes").  A band renderer issues every exchange on the lane whose work it belongs to, so that an
 * exchange never waits behind kernels it does not depend on and nothing waits for an exchange it does not read:
 *   MAIN     the caller's stream: the reuse passes and the reservoir strips between them
 *   GBUFFER  the renderer's G-buffer stream: the G-buffer pass of frame N + 1 runs underneath frame N's reuse passes, and so does
 *            the exchange of its strips (the candidate pass reads no neighbour's G-buffer; only the spatial passes wait for them)
 *   GATHER   the renderer's gather stream: the all-gather of the HDR bands runs underneath the NEXT frame (nothing of that frame
 *            reads other ranks' pixels; its shading pass waits for the gather to have read the band it overwrites)
 *   SEAM     the renderer's seam stream: a spatial pass that is followed by another one runs its seam rows (the rows the neighbours'
 *            next pass reads) FIRST, their exchange travels on this lane while the caller's stream runs the interior rows, and only
 *            the next pass waits for it (stripMode 2 of gfxh_restir_frame_program; the biased estimator's passes)
 * A transport that keeps one communicator per lane (gfxh_rccl_create_lanes; a process group per lane in tilesplit.StripExchange)
 * lets the three run concurrently; with one communicator they still run, in issue order. */
enum gfxh_lane { GFXH_LANE_MAIN = 0, GFXH_LANE_GBUFFER = 1, GFXH_LANE_GATHER = 2, GFXH_LANE_SEAM = 3, GFXH_NUM_LANES = 4 };
typedef struct gfxh_exchange_desc {
    uint32_t kind;                       /* enum gfxh_exchange_kind */
    uint32_t stage;                      /* ordinal of the exchange point inside the frame (diagnostics) */
    uint32_t lane;                       /* enum gfxh_lane: `stream` of the callback is that lane's stream */
    uint32_t reserved;
    uint32_t width, height;
    uint32_t bandBegin, bandEnd;
    uint32_t sendAbove[2], recvAbove[2], sendBelow[2], recvBelow[2];   /* STRIPS: row ranges [begin, end) */
    uint32_t numBuffers;
    gfxh_exchange_buffer buffers[8];     /* STRIPS: all of them; GATHER_BANDS: buffers[0] = the float4 HDR buffer */
    void* counters; uint64_t numCounters; /* ALLREDUCE_SUM_U32 */
} gfxh_exchange_desc;
typedef int (*gfxh_exchange_fn)(void* user, void* stream, const gfxh_exchange_desc* desc);
int gfxh_restir_set_exchange(gfxh_restir* r, gfxh_exchange_fn fn, void* user, uint32_t maxMotionRows);
/* The all-gather of the HDR bands on the renderer's gather stream, underneath the next frame (lane GATHER above).  Off (the default):
 * the gather is issued on the caller's stream and the frame buffer holds every rank's rows of THIS frame when that stream has passed
 * gfxh_restir_render_frame's work.  On: other ranks' rows arrive while the next frame renders; a reader of the whole frame first calls
 * gfxh_restir_finish_gather(r, stream), which makes `stream` wait for the outstanding gather (a bench loop calls it once, after the
 * last frame). */
int gfxh_restir_set_async_gather(gfxh_restir* r, int enable);
int gfxh_restir_finish_gather(gfxh_restir* r, void* stream);
/* Row ranges of a strip exchange of `rows` rows for the band [bandBegin, bandEnd) of a frame of `height` rows, into
 * the send* / recv* members of `out` (nothing is sent above row 0 / below the last row).  Returns 1 when `rows`
 * exceeds the band height of this rank: the strip would have to come from a rank further away. */
int gfxh_strip_rows(uint32_t height, uint32_t bandBegin, uint32_t bandEnd, uint32_t rows, gfxh_exchange_desc* out);
/* The row band of `rank` when a frame of `height` rows is split over `world` ranks: whole 8-row tiles, the remainder
 * spread from rank 0 (1080 rows / 8 ranks = 7 x 136 + 128).  The one partition every part of this library uses
 * (gfxh_rccl_create, tilesplit.band_rows, bench.py). */
int gfxh_band_rows(uint32_t height, uint32_t world, uint32_t rank, uint32_t* bandBegin, uint32_t* bandEnd);
/* Whether `cfg` can be split over `world` ranks with strip exchanges of up to `maxMotionRows` motion rows: the tallest strip
 * of any frame against the SMALLEST band of the partition.  A pure function of its arguments, so every rank reaches the
 * same verdict -- call it where the exchange is installed (tilesplit.StripExchange, bench.py and restir_di_headless do);
 * the per-frame test inside gfxh_restir_render_frame only sees the calling rank's band.  1 + gfxh_restir_last_error()
 * when it does not fit. */
int gfxh_restir_check_partition(const gfxh_restir_config* cfg, uint32_t world, uint32_t maxMotionRows);
/* The same test for a partition given explicitly: band r = rows [bandBegin[r], bandBegin[r + 1]), world + 1 entries from 0 to
 * cfg->height, every boundary a multiple of 8 (except the last).  For cost-balanced bands (gfxh_balance_bands). */
int gfxh_restir_check_bands(const gfxh_restir_config* cfg, uint32_t world, const uint32_t* bandBegin, uint32_t maxMotionRows);
/* Cost-balanced bands.  Equal rows are not equal work (sky rows are cheap, street-level rows are not): given the partition a
 * frame was rendered with and the time every rank took for its band (all-gathered by the caller, so that every rank passes the
 * same numbers), writes the partition that would have equalised the times, assuming the cost is uniform inside each old band:
 * boundaries on whole 8-row tiles, no band shorter than minRows rows (the tallest exchange strip; at least one tile).  A pure
 * function of its arguments.  bandBeginIn / bandBeginOut: world + 1 entries.  Returns 1 on invalid arguments. */
int gfxh_balance_bands(uint32_t height, uint32_t world, const uint32_t* bandBeginIn, const float* bandMilliseconds, uint32_t minRows,
                       uint32_t* bandBeginOut);
/* An exchange callback over RCCL for C++ host programs (librccl is loaded with dlopen on first use; one process per
 * GPU).  Create with the ncclUniqueId bytes rank 0 obtained from gfxh_rccl_unique_id and distributed its own way.  It exchanges
 * the DEFAULT partition (gfxh_band_rows) unless gfxh_rccl_set_bands names another: gfxh_rccl_create fails when the default leaves a
 * rank without a band (more ranks than 8-row tiles), and a renderer whose band is not its rank's band of the partition in force is
 * refused at the first gather. */
typedef struct gfxh_rccl gfxh_rccl;
int gfxh_rccl_unique_id(void* id128);
int gfxh_rccl_create(const void* id128, int rank, int world, uint32_t height, gfxh_rccl** out);
/* One communicator per lane (enum gfxh_lane): `ids` = numLanes x 128 bytes, each from its own gfxh_rccl_unique_id call on rank 0
 * (1 <= numLanes <= GFXH_NUM_LANES; a lane beyond numLanes shares communicator 0).  With all of them, the G-buffer strips, the seam strips, the reservoir
 * strips and the band gather never queue behind each other. */
int gfxh_rccl_create_lanes(const void* ids, uint32_t numLanes, int rank, int world, uint32_t height, gfxh_rccl** out);
/* Another partition than gfxh_band_rows' (the cost-balanced bands of gfxh_balance_bands): bandBegin = world + 1 ascending rows from 0
 * to the image height, the same on every rank.  The renderers' cfg.rowBegin / rowEnd must be this partition's. */
int gfxh_rccl_set_bands(gfxh_rccl* c, const uint32_t* bandBegin);
void gfxh_rccl_destroy(gfxh_rccl* c);
int gfxh_rccl_exchange(void* user /* gfxh_rccl* */, void* stream, const gfxh_exchange_desc* desc);
const char* gfxh_rccl_last_error(void);

/* One frame as a list of steps (host logic only: the driver executes it on the GPU, the multi-process CPU tests execute
 * the same list with the oracle): passes with their row ranges and the index bookkeeping of restir_di_main.cpp:2311-2493,
 * and the exchange points of strip mode. */
enum gfxh_step_op {
    GFXH_STEP_RESTIR_PASS = 0,            /* gfx_restir_launch_rows(pass, rowBegin, rowEnd) after gfx_restir_set_params(current indices) */
    GFXH_STEP_PT_PASS = 1,                /* gfx_pt_launch(pass, rowBegin, rowEnd) */
    GFXH_STEP_EXCHANGE_STRIPS = 2,        /* exchangeRows rows of `buffers` (reservoirs: those of reservoirIndex) */
    GFXH_STEP_ALLREDUCE_CELL_ACCESSES = 3,
    GFXH_STEP_GATHER_BANDS = 4,
    GFXH_STEP_PREV_GBUFFER_RELEASED = 5,  /* the frame no longer reads the previous frame's G-buffer (frame pipelining) */
    GFXH_STEP_WAIT_GBUFFER_STRIPS = 6,    /* the next pass reads the neighbours' G-buffer rows: MAIN waits for the exchange issued on GBUFFER */
    GFXH_STEP_WAIT_PREVIOUS_GATHER = 7,   /* the next pass overwrites the HDR band the previous frame's gather (lane GATHER) sends */
    GFXH_STEP_WAIT_SEAM_STRIPS = 8        /* the next pass reads the strips the exchange on lane SEAM brings: MAIN waits for it */
};
enum gfxh_exchange_buffers { GFXH_BUF_GBUFFERS = 1 /* GBuffer 0, 2, 3 of the frame */, GFXH_BUF_RESERVOIRS = 2 /* + ReservoirInfo */, GFXH_BUF_SAMPLE_VISIBILITY = 4,
                             GFXH_BUF_RNG = 8 /* the pixels' PCG32 states (stripMode 3: the halo rows' passes draw from them) */ };
typedef struct gfxh_frame_step {
    uint32_t op, pass;
    uint32_t rowBegin, rowEnd;                                 /* 0, 0 = all rows */
    uint32_t currentReservoirIndex, spatialNeighborBaseIndex;  /* launch parameters in force for a pass */
    uint32_t exchangeRows, buffers, reservoirIndex;            /* GFXH_STEP_EXCHANGE_STRIPS */
    uint32_t lane;                                             /* enum gfxh_lane the step is issued on (a program run in list order, as the CPU tests do, is one valid schedule) */
    uint32_t gapBegin, gapEnd;                                 /* a pass over rows [rowBegin, gapBegin) + [gapEnd, rowEnd) (gfx_restir_launch_rows_gap); 0, 0 = no gap */
} gfxh_frame_step;
/* stripMode: 0 = whole frame / halo recompute; 1 = strip exchange, every pass over the band in one launch, reservoirs exchanged before every
 * spatial pass; 2 = 1 + seam rows first (lane SEAM) for the spatial passes that another pass follows (GFX_SEAM_FIRST=1; measured slower);
 * 3 = the spatial passes that another pass follows are recomputed on a halo that shrinks by `radius` rows per pass: ONE reservoir exchange per
 * frame (radius x passes rows of reservoirs, infos and pixel RNG states behind the candidate pass; G-buffer strips as tall) instead of one per
 * spatial pass -- what gfxh_restir_render_frame runs (GFX_STRIP_MODE=1|2|3); the same frames bit for bit in every mode. */
int gfxh_restir_frame_program(const gfxh_restir_config* cfg, int stripMode, uint32_t maxMotionRows, int newSequence,
                              uint32_t lastReservoirIndex, uint32_t lastSpatialNeighborBaseIndex, uint32_t useUnbiasedEstimator,
                              gfxh_frame_step* steps, uint32_t capacity, uint32_t* numSteps, uint32_t* newLastReservoirIndex,
                              uint32_t* newLastSpatialNeighborBaseIndex);
/* The descriptor an exchange step hands to the callback, from the launch parameters the buffers live in (device pointers in
 * the driver, host pointers when the CPU tests run the program with the oracle).  `regir` may be null unless the step is
 * GFXH_STEP_ALLREDUCE_CELL_ACCESSES.  Returns 1 for a step that is not an exchange or a strip taller than the band. */
int gfxh_frame_step_exchange_desc(const gfxh_restir_config* cfg, const gfxh_frame_step* step, uint32_t stepIndex,
                                  const gfx_restir_static_params* sp, const gfx_regir_params* regir, uint32_t bufferIndex,
                                  gfxh_exchange_desc* out);

/* Allocates every per-pixel buffer (hipMalloc), seeds the RNG buffer, uploads the Halton table,
 * builds BVH and light distributions for the uploaded scene. */
int gfxh_restir_create(gfx_ctx* ctx, const gfxh_restir_config* cfg, gfxh_restir** out);
void gfxh_restir_destroy(gfxh_restir* r);
/* One frame: light-instance distribution, G-buffer, initial(+temporal) RIS, spatial passes, shading.
 * Frame pipelining: the G-buffer pass of this frame may run on a private stream underneath the passes of the
 * PREVIOUS frame that are still queued on `stream`.  It rewrites the albedo / normal accumulation buffers (single-buffered running
 * means) and this frame's G-buffer half: work the caller queued on its own stream after the previous gfxh_restir_render_frame returned
 * -- a denoiser, a read-back of those buffers -- is NOT ordered before that pass unless the caller says where it ends:
 * gfxh_restir_outputs_consumed(r, stream) records that point (an event on `stream`) and the next frame's G-buffer pass waits for it.
 * The beauty buffer is only written by passes on `stream` and needs no such call.  GFX_SERIAL_FRAMES=1 switches the pipelining off. */
int gfxh_restir_render_frame(gfxh_restir* r, void* stream);
int gfxh_restir_outputs_consumed(gfxh_restir* r, void* stream);
/* Restart the sequence (newSequence, restir_di_main.cpp:2311). */
int gfxh_restir_reset(gfxh_restir* r);
/* "-env-texture": upload a lat-long float4 environment map (host pointer) + its importance map and
 * enable environment lighting with the given power coefficient and rotation (restir_di_main.cpp:1188-1197). */
int gfxh_restir_set_env(gfxh_restir* r, float* texels4, uint32_t w, uint32_t h, float powerCoeff, float rotation);
int gfxh_restir_set_camera(gfxh_restir* r, const gfx_camera* cam);
/* Scene::updateASs of an animated frame (restir_di_main.cpp:2258-2264): call after gfx_instance_set_transform on
 * the renderer's context and before the next gfxh_restir_render_frame; rebuilds the renderer's BVH in place
 * (same handle) on `stream`. */
int gfxh_restir_rebuild_accel(gfxh_restir* r, void* stream);
/* Device pointer of the float4 beauty accumulation buffer (W*H). */
void* gfxh_restir_beauty_buffer(gfxh_restir* r);
/* Copies of the static parameters (device pointers) and of the last frame parameters. */
int gfxh_restir_get_params(gfxh_restir* r, gfx_restir_static_params* s, gfx_restir_frame_params* f,
                           uint32_t* lastReservoirIndex, uint32_t* lastSpatialNeighborBaseIndex, uint32_t* frameIndex);
uint64_t gfxh_restir_accel(gfxh_restir* r);
/* Message of the last gfxh_restir_* call that returned non-zero on this thread. */
const char* gfxh_restir_last_error(void);

/* ---------------------------------------------------------------- headless NRC renderer -------- */
/* The frame loop of neural_radiance_caching_main.cpp:2225-2370 over the C ABI: G-buffer, preprocessNRC,
 * pathTraceNRC, infer (W*H + #tiles queries rounded up to 128), accumulate, and -- when training --
 * propagate, shuffle, 4 training steps of 16 384 records.  Like the reference it reads the tile size
 * back from the device once per frame to size the inference batch (:2293-2303). */
typedef struct gfxh_nrc gfxh_nrc;
typedef struct gfxh_nrc_config {
    uint32_t width, height;
    int positionEncoding;        /* gfx_nrc_position_encoding; HashGrid (main:458) */
    uint32_t numHiddenLayers;    /* 2 (main:459) */
    float learningRate;          /* 1e-2 (main:460) */
    uint32_t maxPathLength;      /* 5; 0 = unlimited bounces (main:1860-1861, 2246) */
    float radianceScale;         /* pow(10, log10RadianceScale) (main:2240) */
    uint32_t train;              /* 1 */
    uint32_t enableAccumulation; /* 0 */
    gfx_camera camera;
    float sceneAabbMin[3], sceneAabbMax[3];   /* scene.initialSceneAabb (main:1139) */
    uint32_t rowBegin, rowEnd;   /* rows [rowBegin, rowEnd) of a band renderer (needs gfxh_nrc_set_exchange); 0, 0 = the whole frame */
    /* next-event estimation of the NRC tracer: 0 = the emitter distributions (the reference), 1 = the ReGIR grid
     * (GFX_PT_PATH_TRACE_NRC_REGIR: an extension, the reference lists the combination as open, README.md:80-81).  The grid
     * spans sceneAabb; its parameters default to regir_main.cpp:1112, 1733-1736), 2 = the pixel's ReSTIR DI reservoir at the first
     * path vertex (GFX_PT_PATH_TRACE_NRC_RESTIR: the other half of the same open item; the frame runs the original ReSTIR DI passes
     * -- 32 candidates, temporal + 2 x 5 biased spatial reuse, radius 20, restir_di_main.cpp:1944-1967, 2365-2421 -- ahead of the
     * tracer).  1 and 2: whole-frame renderers only. */
    uint32_t neeSampler;
    uint32_t regirGridDimension[3];
    uint32_t regirLog2CandidatesPerLightSlot, regirLog2CandidatesPerCell;
    uint32_t regirEnableTemporalReuse, regirEnableCellRandomization;
    uint32_t enableBumpMapping;  /* 0 */
} gfxh_nrc_config;
void gfxh_nrc_default_config(gfxh_nrc_config* cfg, uint32_t width, uint32_t height);
int gfxh_nrc_create(gfx_ctx* ctx, const gfxh_nrc_config* cfg, gfxh_nrc** out);
void gfxh_nrc_destroy(gfxh_nrc* r);
/* lossOut (optional): the loss of the fourth training step (main:2363).  The G-buffer pass is pipelined as in gfxh_restir_render_frame:
 * gfxh_nrc_outputs_consumed(r, stream) marks the point on `stream` behind which the caller no longer reads the albedo / normal
 * accumulation buffers of the previous frame. */
int gfxh_nrc_render_frame(gfxh_nrc* r, void* stream, float* lossOut);
int gfxh_nrc_outputs_consumed(gfxh_nrc* r, void* stream);
/* Row-band split of the NRC frame over the GPUs of a node (no reference counterpart; one process per GPU, `rank` of them).
 * Every rank path-traces, infers and accumulates its own rows; the training records of all bands are gathered in rank order
 * (GFXH_EXCHANGE_GATHER_RECORDS: 68 B per record, <= 2^17 records), every rank shuffles the same batch and EVERY rank runs the four
 * training steps on its own copy of the network, on its training stream underneath the next frame like the whole-frame renderer: a
 * training step is reproducible bit for bit (nrc.hip k_nrc_grid_scatter: the hash-grid gradient is summed in a defined order), so the
 * copies stay identical and nothing but records and HDR bands crosses the links.  (GFX_NRC_TRAIN_ON_RANK0=1 keeps the scheme of rounds
 * 3-4: rank 0 trains on the caller's stream and its inference images -- gfx_nrc_inference_image_async: 20 KB + 2 MB -- are broadcast,
 * GFXH_EXCHANGE_BROADCAST.)  The tile size adapts to the global record count; the HDR bands are gathered like gfxh_restir's. */
int gfxh_nrc_set_exchange(gfxh_nrc* r, gfxh_exchange_fn fn, void* user, int rank);
/* Scene::updateASs of an animated frame: rebuild the renderer's BVH in place after gfx_instance_set_transform. */
int gfxh_nrc_rebuild_accel(gfxh_nrc* r, void* stream);
/* "-env-texture" of the NRC sample (neural_radiance_caching_main.cpp: the same option and loader as restir_di_main.cpp:1188-1197): as
 * gfxh_restir_set_env.  Paths that leave the scene pick the map up, next-event estimation samples it with probability 0.25; the
 * accumulation restarts. */
int gfxh_nrc_set_env(gfxh_nrc* r, float* texels4, uint32_t w, uint32_t h, float powerCoeff, float rotation);
void* gfxh_nrc_beauty_buffer(gfxh_nrc* r);
uint64_t gfxh_nrc_network(gfxh_nrc* r);
int gfxh_nrc_stats(gfxh_nrc* r, uint32_t* numTrainingData, uint32_t tileSize[2], uint32_t* numInferenceQueries);
const char* gfxh_nrc_last_error(void);

/* ---- output chain (common/common_host.cpp:2725-2922 saveImage / saveImageHDR) -------------------------------------
 * rgba: host copy of a float4 accumulation buffer (gfx_read_device of the beauty buffer), row-major, top row first.
 * SDR: optional tone map on the luminance (1 - exp(-brightnessScale * Y), chroma kept), optional sRGB gamma, 8 bits
 * per channel as min(uint(v * 255), 255); written as 24-bit .bmp or binary .ppm by extension.  HDR: the fp32 values
 * times brightnessScale as a .pfm (the reference writes the same numbers as an OpenEXR file through tinyexr). */
typedef struct gfxh_sdr_config {
    float alphaForOverride;              /* kept for layout parity with SDRImageSaverConfig (common_host.h:1520-1532); unused */
    float brightnessScale;
    uint32_t applyToneMap, apply_sRGB_gammaCorrection, flipY;
} gfxh_sdr_config;
int gfxh_save_image_sdr(const char* path, uint32_t width, uint32_t height, const float* rgba, const gfxh_sdr_config* cfg);
int gfxh_save_image_hdr(const char* path, uint32_t width, uint32_t height, float brightnessScale, const float* rgba, int flipY);
/* The 8-bit pixels gfxh_save_image_sdr would write (R | G << 8 | B << 16 | A << 24, common_host.cpp:2886-2890). */
void gfxh_tonemap_sdr(uint32_t width, uint32_t height, const float* rgba, const gfxh_sdr_config* cfg, uint32_t* out);

/* ---- the ABI as the compiler sees it -------------------------------------------------------------------------------
 * Layout of every struct of gfxexp.h / gfxexp_host.h in the library as built: a binding that mirrors the structs by hand
 * (ctypes, cgo, JNI) asserts itself against these instead of against a second hand-written copy.  gfxh_abi_layout: fieldName NULL
 * -> sizeof(struct) in *size (offset 0); otherwise offsetof / sizeof of the field.  Returns 1 for an unknown struct or field.
 * gfxh_abi_entry enumerates the table (a struct's own entry has fieldName NULL and precedes its fields, in declaration order). */
int gfxh_abi_layout(const char* structName, const char* fieldName, uint64_t* offset, uint64_t* size);
uint32_t gfxh_abi_num_entries(void);
int gfxh_abi_entry(uint32_t index, const char** structName, const char** fieldName, uint64_t* offset, uint64_t* size);

#ifdef __cplusplus
}
#endif
#endif /* GFXEXP_HOST_H */
