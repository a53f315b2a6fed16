/*
 * gfxexp.h -- C ABI of the MI355X-native path-tracing inner loop.
 *
 * This is the drop-in boundary.  Every entry point names the reference
 * interface (file:line under the GfxExp tree) whose job it takes over.  Signatures carry plain
 * pointers and sizes only; device memory is caller-owned unless stated otherwise
 * (reference ownership model: restir_di/restir_di_main.cpp:1233-1325).
 *
 * All functions return 0 on success, non-zero on failure; gfx_last_error() gives the text
 * (the reference throws std::runtime_error from CUDADRV_CHECK, utils/cuda_util.cpp:58-69).
 * Everything is asynchronous with respect to `stream` (a hipStream_t passed as void*), and a
 * gfx_ctx is not thread-safe (same as the reference: one host thread, restir_di_main.cpp:1705).
 * A context also owns ONE set of library streams and events behind the launches (the side stream the path tracers' NEE traces and the
 * block-order sorts run on, with its fork / join events, and the scratch sets of the traversal): the launches of
 * one context (gfx_pt_launch, gfx_restir_launch*) are to be issued from one host thread, and renderers that run concurrently on
 * different streams take a context each (as the band renderers of tests/ and tools/ do).  With gfx_counters_enable the
 * per-launch diagnostics are those of the calling stream's order only when "pt_overlap" is 0.
 */
#ifndef GFXEXP_H
#define GFXEXP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFX_INVALID_SLOT 0xFFFFFFFFu

typedef struct gfx_ctx gfx_ctx;

/* ---------------------------------------------------------------- POD layouts in HBM -------- */

/* common/common_shared.h:1109-1114 (shared::Vertex, 44 B). */
typedef struct gfx_vertex {
    float position[3];
    float normal[3];
    float texCoord0Dir[3];
    float texCoord[2];
} gfx_vertex;

/* Materials.  Every value of shared::MaterialData (common/common_shared.h:1144-1177) is a texture read through
 * tex2DLod at mip level 0 (common/common_device.cuh:143-147, 778-826).  Here a value is either a texture slot
 * (gfx_texture_set; slots are 1-based, 0 = none) or, with slot 0, the constant a / b / smoothness / emittance: what
 * the reference's 1x1 "immediate" textures return (common/common_host.cpp:1045-1073, 1602-1659), i.e. the *sampled*
 * (linear) value. */
enum gfx_bsdf_type {
    GFX_BSDF_LAMBERT = 0,              /* common/common_device.cuh:335-374 */
    GFX_BSDF_DIFFUSE_AND_SPECULAR = 1, /* common/common_device.cuh:443-765 */
    GFX_BSDF_SIMPLE_PBR = 2            /* common/common_device.cuh:767-776 */
};
enum gfx_bump_type {                   /* BumpMapTextureType -> readModifiedNormal*, common/common_device.cuh:205-240 */
    GFX_BUMP_NORMAL_MAP = 0,           /* RGB normal map: n = 2 t - 1 */
    GFX_BUMP_NORMAL_MAP_2CH = 1,       /* two channels, z = sqrt(1 - x^2 - y^2) */
    GFX_BUMP_HEIGHT_MAP = 2,           /* finite differences of the four texels under the bilinear footprint */
    GFX_BUMP_LEFT_HANDED = 0x100       /* flag: flip y (TexDimInfo::isLeftHanded) */
};
typedef struct gfx_material {
    uint32_t bsdfType;
    float a[3];          /* lambert reflectance | diffuse | baseColor */
    float b[3];          /* - | specular F0 | (occlusion, roughness, metallic) */
    float smoothness;    /* diffuse+specular only */
    float emittance[3];
    uint32_t hasEmittance; /* "mat.emittance != 0" in the reference */
    uint32_t texA;          /* texture slot for a (reflectance | diffuse | baseColor_opacity), 0 = the constant */
    uint32_t texB;          /* texture slot for b (specular | occlusion_roughness_metallic), 0 = the constant */
    uint32_t texSmoothness; /* one-channel texture for smoothness, 0 = the constant */
    uint32_t texNormal;     /* normal / height map read when gfx_restir_frame_params.enableBumpMapping, 0 = flat (0.5, 0.5, 1) */
    uint32_t texEmittance;  /* 0 = the constant */
    uint32_t bumpMapType;   /* enum gfx_bump_type, optionally | GFX_BUMP_LEFT_HANDED */
    uint32_t pad[2];
} gfx_material;

/* Texel formats and the sampler each one is read through (createTextureObject, common/common_host.cpp:1462-1481):
 * all samplers are bilinear, wrap = repeat in both directions, mip level 0. */
enum gfx_tex_format {
    GFX_TEX_RGBA8_SRGB = 0,   /* 8-bit RGBA, sampler_sRGB: R, G, B decoded sRGB -> linear per texel, A / 255 */
    GFX_TEX_RGBA8_UNORM = 1,  /* 8-bit RGBA, sampler_normFloat: c / 255 */
    GFX_TEX_R8_UNORM = 2,     /* one 8-bit channel (smoothness, height maps) */
    GFX_TEX_RG8_UNORM = 3,    /* two 8-bit channels (two-channel normal maps, BC5 decoded offline) */
    GFX_TEX_RGBA32F = 4       /* float RGBA, sampler_float (HDR emittance) */
};

/* restir_di/restir_di_shared.h:182-204 -- same element structs, row-major linear arrays. */
typedef struct gfx_gbuffer0 { uint32_t instSlot, geomInstSlot, primIndex; uint16_t qbcB, qbcC; } gfx_gbuffer0;
typedef struct gfx_gbuffer1 { float motionVector[2]; } gfx_gbuffer1;
typedef struct gfx_gbuffer2 { float positionInWorld[3]; uint32_t qGeometricNormal; } gfx_gbuffer2;
typedef struct gfx_gbuffer3 { uint32_t qShadingNormal, qShadingTangent, qTexCoord, matSlot; } gfx_gbuffer3;

/* restir_di/restir_di_shared.h:89-96,106-139.  Reservoir<LightSample> is 48 B; in HBM it is
 * stored as THREE planes of 16 B per pixel (plane k starts at k * W*H*16 bytes) so a wavefront's
 * loads are contiguous:
 *   plane0 = (emittance.r, emittance.g, emittance.b, position.x)
 *   plane1 = (position.y, position.z, normal.x, normal.y)
 *   plane2 = (normal.z, atInfinity as u32 bits, sumWeights, streamLength as u32 bits) */
typedef struct gfx_reservoir_info { float recPDFEstimate, targetDensity; } gfx_reservoir_info; /* :141-144 */

/* restir_di/restir_di_shared.h:45-60.  orientation is row-major: o[r*3+c]. */
typedef struct gfx_camera {
    float aspect;
    float fovY;
    float position[3];
    float orientation[9];
} gfx_camera;

/* closest-hit record written by gfx_trace (common/common_shared.h:1065-1078 HitObject, compacted). */
typedef struct gfx_hit {
    float dist;          /* +inf / tmax on miss */
    float bcB, bcC;
    uint32_t triIndex;   /* index into the BVH's triangle records, GFX_INVALID_SLOT on miss */
} gfx_hit;
typedef struct gfx_tri_ids { uint32_t instSlot, geomInstSlot, primIndex; } gfx_tri_ids;

/* ---------------------------------------------------------------- context ------------------- */

/* restir_di/restir_di_main.cpp:128-136 (cuInit, cuCtxCreate, optixu::Context::create). */
int gfx_ctx_create(int device, gfx_ctx** out);
void gfx_ctx_destroy(gfx_ctx* ctx);
const char* gfx_last_error(gfx_ctx* ctx);
/* Library build identification: "gfxexp_amd <version> gfx950". */
const char* gfx_version(void);

/* ---------------------------------------------------------------- scene --------------------- */

/* common/common_host.cpp:1454-1815 (create*Material): constants and / or texture slots. */
int gfx_material_set(gfx_ctx* ctx, uint32_t matSlot, const gfx_material* mat);
/* Texture upload (loadTexture + cudau::Array::write + createTextureObject, common/common_host.cpp:1163-1244,
 * 1462-1481).  texSlot >= 1; texels are tightly packed rows, row 0 first, in `format`.
 * The CUDA texture unit is replaced by a software tex2DLod with a written contract ("Textures"):
 *   x = (u - floor(u)) * W - 0.5, i = floor(x), alpha = round_to_8_fraction_bits(x - i)   (likewise y, beta)
 *   T = ((1-alpha)(1-beta)) T[i,j] + (alpha (1-beta)) T[i+1,j] + ((1-alpha) beta) T[i,j+1] + (alpha beta) T[i+1,j+1]
 * with indices wrapped modulo W / H, fp32 arithmetic in exactly this order, 8-bit texels decoded per texel
 * before filtering (c / 255, or the exact sRGB formula rounded to fp32). */
int gfx_texture_set(gfx_ctx* ctx, uint32_t texSlot, uint32_t width, uint32_t height, uint32_t format, const void* texels);
/* Inspection: tex2DLod<float4> (gather = 0) or tex2Dgather<float4> of component 0 (gather = 1) of one texture at n
 * coordinates; dUv = device float2[n], dOut = device float4[n].  Runs the functions the shading kernels call. */
int gfx_texture_sample(gfx_ctx* ctx, void* stream, uint32_t texSlot, const void* dUv, uint32_t n, void* dOut, int gather);

/* common/common_host.cpp:1817-1905 createGeometryInstance: host vertex/triangle arrays in,
 * geomInstSlot out.  `vertexStride` >= sizeof(gfx_vertex). */
int gfx_geom_create(gfx_ctx* ctx, const void* vertices, uint32_t vertexStride, uint32_t numVertices,
                    const uint32_t* triangles, uint32_t numTriangles, uint32_t matSlot,
                    uint32_t* geomInstSlot);
/* common/common_host.cpp:2051-2078 createGeometryGroup. */
int gfx_group_create(gfx_ctx* ctx, const uint32_t* geomInstSlots, uint32_t n, uint32_t* group);
/* common/common_host.cpp:2582-2656 createInstance: xfm is 3x4 row-major (the float[12] handed to
 * optixInst.setTransform).  instSlot doubles as the OptiX instance id. */
int gfx_instance_create(gfx_ctx* ctx, uint32_t group, const float xfm[12], uint32_t* instSlot);
/* common/common_host.h:837-855 InstanceController::update, the InstanceData part: xfm becomes the instance's
 * object-to-world matrix and curToPrevTransform = (previous matrix) * invert(xfm), so the next G-buffer pass
 * writes motion vectors for the instance (optix_gbuffer_kernels.cu:132).  Like the reference's controllers, call
 * it every frame for an animated instance (a frame without a call keeps the last curToPrevTransform), then
 * gfx_accel_build again (Scene::updateASs, restir_di_main.cpp:2263-2264); the emitter distributions and
 * records are rebuilt by the next gfx_lights_build_instances.  The normal matrix is recomputed as
 * transpose(invert(upper-left 3x3)) like at creation; ..._and_normal_matrix takes the controller's own
 * matRot / curScale (row-major 3x3, common_host.h:843) for a bit-exact shim. */
int gfx_instance_set_transform(gfx_ctx* ctx, uint32_t instSlot, const float xfm[12]);
int gfx_instance_set_transform_and_normal_matrix(gfx_ctx* ctx, uint32_t instSlot, const float xfm[12], const float normalMatrix[9]);
/* Declares an instance animated before the first gfx_accel_build.  Animated instances live in their own BVH
 * subtree under a two-child root; gfx_instance_set_transform on them followed by gfx_accel_build (same handle)
 * rebuilds only that subtree -- a few small kernels for a light or two -- instead of the whole tree.  An instance
 * that was not declared becomes animated at its first gfx_instance_set_transform (one full rebuild). */
int gfx_instance_set_dynamic(gfx_ctx* ctx, uint32_t instSlot, int dynamic);

/* common/common_host.h:1027-1100 Scene::updateASs -> OptixTraversableHandle.  Builds the HIP
 * LBVH -> BVH8 over all instances (world space).  The 64-bit handle fits the reference's
 * perFramePlp.travHandle field (restir_di_main.cpp:2264). */
int gfx_accel_build(gfx_ctx* ctx, void* stream, uint64_t* handle);
/* Leaf size limit of the collapse step, 1..4 (default 4; a node's leaf triangles fit one 32-bit mask); GeometryBVHBuildConfig::maxNumPrimsPerLeaf,
 * common/bvh_builder.h:38-44. */
int gfx_accel_set_max_leaf(gfx_ctx* ctx, uint32_t maxLeafTris);
/* Build statistics of the last gfx_accel_build: {numTriangles, numNodes, numTriRecords, maxDepth}. */
int gfx_accel_stats(gfx_ctx* ctx, uint64_t handle, uint32_t stats[4]);
/* Device pointer to the gfx_tri_ids table of the BVH (indexed by gfx_hit.triIndex). */
int gfx_accel_tri_ids(gfx_ctx* ctx, uint64_t handle, const void** dTriIds, uint32_t* count);

/* common/common_host.h:1102-1266 setupLightGeomDistributions (once) and
 * :1268-1359 setupLightInstDistribution (per frame; restir_di_main.cpp:2303-2309). */
int gfx_lights_build_static(gfx_ctx* ctx, void* stream);
int gfx_lights_build_instances(gfx_ctx* ctx, void* stream, uint32_t bufferIndex);
/* Read back the three-level distribution for inspection/tests: level 0 = instances,
 * 1 = geomInsts of instance `index`, 2 = emitter triangles of geomInst `index`.
 * weights/cdf may be NULL; returns the element count in *n and the integral in *integral. */
int gfx_lights_read(gfx_ctx* ctx, uint32_t level, uint32_t index, float* weights, float* cdf,
                    uint32_t capacity, uint32_t* n, float* integral);

/* State of the emitter interval table the last gfx_lights_build_instances produced (the three searches of
 * sampleLight, restir_di_shared.h:366-415, flattened into one lookup): info = { usable (the build verified it
 * against the three searches; 0 = kernels run the searches themselves), records verified, records, guide cells,
 * distinct normal matrices among the emitter instances, interior guide cells (a lookup that lands in one is done after
 * one load), 0, 0 }. */
int gfx_lights_table_info(gfx_ctx* ctx, uint32_t info[8]);

/* ---------------------------------------------------------------- ray queries ---------------- */

/* Replaces optixTrace (utils/optix_util.h:557-603) for wavefront ray queues and the scalar
 * bvh::traverse (common/bvh_builder.cpp:1272-1649).
 *   rayOrgTmin[i] = (org.xyz, tmin), rayDirTmax[i] = (dir.xyz, tmax); intervals are exclusive.
 *   mode CLOSEST: out = gfx_hit[numRays];  mode ANY: out = uint32_t[numRays] (1 = occluded).
 * counters (optional, device u64[4]): node fetches, triangle fetches, rays, stack spills. */
enum gfx_trace_mode { GFX_TRACE_CLOSEST = 0, GFX_TRACE_ANY = 1 };
int gfx_trace(gfx_ctx* ctx, void* stream, uint64_t accel, int mode,
              const void* dRayOrgTmin, const void* dRayDirTmax, uint32_t numRays,
              void* dOut, void* dCounters);
/* The same through the counting instantiation, additionally writing how many 64-byte items (BVH8 nodes + triangle
 * records) each ray fetched to dPerRayItems (device u32[numRays]): the work distribution a scene asks of the traversal
 * (tools/bvh_quality.py reports its histogram).  dCounters must be given. */
int gfx_trace_counted(gfx_ctx* ctx, void* stream, uint64_t accel, int mode,
                      const void* dRayOrgTmin, const void* dRayDirTmax, uint32_t numRays,
                      void* dOut, void* dCounters, void* dPerRayItems);

/* ---------------------------------------------------------------- ReSTIR DI ------------------ */

/* restir_di/restir_di_shared.h:208-239 StaticPipelineLaunchParameters with surfaces/textures
 * replaced by plain device pointers (row-major, pixel p = y*W + x). */
typedef struct gfx_restir_static_params {
    int32_t imageSizeX, imageSizeY;
    void* rngBuffer;                 /* uint64_t[W*H]  (PCG32RNG state) */
    void* gbuffer0[2];
    void* gbuffer1[2];
    void* gbuffer2[2];
    void* gbuffer3[2];
    void* reservoirBuffer[2];        /* 3 planes x W*H x 16 B */
    void* reservoirInfoBuffer[2];    /* gfx_reservoir_info[W*H] */
    void* sampleVisibilityBuffer[2]; /* uint32_t[W*H] (rearchitected only) */
    const void* spatialNeighborDeltas; /* float2[1024] */
    void* beautyAccumBuffer;         /* float4[W*H] */
    void* albedoAccumBuffer;
    void* normalAccumBuffer;
    int32_t numTilesX, numTilesY;    /* rearchitected only */
    void* lightPreSamplingRngs;      /* uint64_t[131072] */
    void* preSampledLights;          /* 48 B x 131072 */
    /* environment light (restir_di_shared.h:221-222); NULL = none */
    const void* envLightTexture;     /* float4[envW*envH], lat-long */
    int32_t envWidth, envHeight;
    const void* envRowPDF;           /* float[envH*envW]      conditional PDFs per row */
    const void* envRowCDF;           /* float[envH*(envW+1)] */
    const void* envRowIntegrals;     /* float[envH] */
    const void* envTopPDF;           /* float[envH] */
    const void* envTopCDF;           /* float[envH+1] */
    float envTopIntegral;
    /* Optional guide tables over the environment CDFs (gfxh_env_build_guides; NULL = plain binary search, same
     * results): uint16_t[envH*envW] / uint16_t[envH], entry k = largest index whose CDF value lies in cell <= k
     * of envW (envH) equal cells of [0,1). */
    const void* envRowGuide;
    const void* envTopGuide;
    /* Optional: the rows of the map with everything a light sample on it reads in one place (gfxh_env_build_row_table; NULL = the
     * separate arrays above, same results): envH x (envW + 1) records of 32 bytes {float cdf, pdf; uint32 guide; float r, g, b; 8 B
     * unused} -- record (row, i) = envRowCDF[row * (envW + 1) + i], envRowPDF[row * envW + i], envRowGuide[row * envW + i] and texel
     * (i, row) (zero where i = envW has none); the seventh word is the NEXT record's cdf.  Rows are GFX_ENV_ROW_STRIDE(envW) records apart
     * (a multiple of four: groups of four records are 128-byte lines).  A sample then touches two or three 64-byte sectors instead of six
     * or seven. */
    const void* envRowTable;
    /* Optional, with envRowTable (gfxh_env_build_row_sketch; NULL = the guide inside the records): records of GFX_ENV_SKETCH_WORDS words --
     * 33 floats = an inverse CDF (in columns) at 33 equidistant u, a 32-bit mask, the index of the record's first child record, one unused
     * word.  Record `row` (0 .. envH - 1) covers u in [0, 1) of that row: mask bit k set = for every u of cell k = [k/32, (k+1)/32) the linear
     * interpolation of knots k and k + 1 lands within one column of the column the bisection finds (verified by the builder for every
     * column of the cell, with the device's arithmetic); the sign bit of knot k repeats the verdict (set = failed), so a sample of a
     * verified cell reads the two knots and nothing else.  A cell that fails has a CHILD record (envH + first child + its rank among the
     * record's failing cells) that covers the cell's range of u the same way at 1/32 of the step; a sub-cell that fails again keeps the
     * guide.  A sample of a verified (sub-)cell reads ONE 128-byte line of the row table (the group of four records around the
     * prediction; a neighbouring line in the few cases the column sits across its edge) instead of the guide's line plus the column's;
     * the records (envH + a few hundred) stay in L2.  Same column, same sample. */
    const void* envRowSketch;
} gfx_restir_static_params;
#define GFX_ENV_ROW_STRIDE(envW) ((((uint32_t)(envW)) + 1u + 3u) & ~3u)
#define GFX_ENV_SKETCH_CELLS 32u
#define GFX_ENV_SKETCH_WORDS 36u

/* restir_di/restir_di_shared.h:241-281 PerFramePipelineLaunchParameters. */
typedef struct gfx_restir_frame_params {
    uint64_t travHandle;
    uint32_t numAccumFrames;
    uint32_t frameIndex;
    gfx_camera camera;
    gfx_camera prevCamera;
    float envLightPowerCoeff;
    float envLightRotation;
    float spatialNeighborRadius;
    float radiusThresholdForSpatialVisReuse;
    uint32_t log2NumCandidateSamples;
    uint32_t numSpatialNeighbors;
    uint32_t useLowDiscrepancyNeighbors;
    uint32_t reuseVisibility;
    uint32_t reuseVisibilityForTemporal;
    uint32_t reuseVisibilityForSpatiotemporal;
    uint32_t enableTemporalReuse;
    uint32_t enableSpatialReuse;
    uint32_t useUnbiasedEstimator;
    uint32_t bufferIndex;
    uint32_t resetFlowBuffer;
    uint32_t enableJittering;
    uint32_t enableEnvLight;
    uint32_t enableBumpMapping;
    /* path tracers: sampleLight<true> / computeSurfacePoint<.., true> -- emitter triangles sampled uniformly in the
     * solid angle they subtend from the shading point (restir_di_shared.h:417-483, path_tracing_shared.h:319-385,
     * 550-568; a compile-time `useSolidAngleSampling = false` in the reference, a run-time switch here) */
    uint32_t useSolidAngleSampling;
} gfx_restir_frame_params;

/* restir_di/restir_di_main.cpp:2350-2359: the three cuMemcpyHtoDAsync of plp/perFramePlp.
 * currentReservoirIndex is 1 bit, spatialNeighborBaseIndex wraps to 10 bits
 * (restir_di_shared.h:283-288). */
int gfx_restir_set_params(gfx_ctx* ctx, void* stream,
                          const gfx_restir_static_params* s, const gfx_restir_frame_params* f,
                          uint32_t currentReservoirIndex, uint32_t spatialNeighborBaseIndex);

/* restir_di/restir_di_main.cpp:72-95 entry-point enums; pipeline.launch(stream, plp, W, H, 1)
 * at :2366-2420. */
enum gfx_restir_pass {
    GFX_RESTIR_SETUP_GBUFFERS = 0,                 /* optix_gbuffer_kernels.cu:5-243 */
    GFX_RESTIR_INITIAL_RIS = 1,                    /* optix_restir_di_kernels.cu:289-291 */
    GFX_RESTIR_INITIAL_AND_TEMPORAL_BIASED = 2,    /* :293-295 */
    GFX_RESTIR_INITIAL_AND_TEMPORAL_UNBIASED = 3,  /* :297-299 */
    GFX_RESTIR_SPATIAL_BIASED = 4,                 /* :549-551 */
    GFX_RESTIR_SPATIAL_UNBIASED = 5,               /* :553-555 */
    GFX_RESTIR_SHADING = 6,                        /* :559-637 */
    /* Rearchitected ReSTIR (restir_di_main.cpp:2423-2487).  preSampledLights = 131072 x 48 B
     * ([emittance.rgb, position.x] [position.yz, normal.xy] [normal.z, atInfinity, areaPDensity, 0]),
     * lightPreSamplingRngs = 131072 x u64 seeded from mt19937_64(894213312210) (:1216-1219),
     * sampleVisibilityBuffer = the SampleVisibility bit union of restir_di_shared.h:146-164. */
    GFX_RESTIR_LIGHT_PRESAMPLING = 7,              /* per_pixel_ris.cu:6-40 */
    GFX_RESTIR_PER_PIXEL_RIS = 8,                  /* per_pixel_ris.cu:44-128 (8x8 tiles) */
    /* optix_restir_di_rearch_kernels.cu:227-253, in the order of RearchitectedReSTIREntryPoint
     * (restir_di_main.cpp:83-95) */
    GFX_RESTIR_TRACE_SHADOW_RAYS = 9,
    GFX_RESTIR_TRACE_SHADOW_RAYS_TEMPORAL_BIASED = 10,
    GFX_RESTIR_TRACE_SHADOW_RAYS_SPATIAL_BIASED = 11,
    GFX_RESTIR_TRACE_SHADOW_RAYS_SPATIOTEMPORAL_BIASED = 12,
    GFX_RESTIR_TRACE_SHADOW_RAYS_TEMPORAL_UNBIASED = 13,
    GFX_RESTIR_TRACE_SHADOW_RAYS_SPATIAL_UNBIASED = 14,
    GFX_RESTIR_TRACE_SHADOW_RAYS_SPATIOTEMPORAL_UNBIASED = 15,
    GFX_RESTIR_SHADE_AND_RESAMPLE = 16,            /* :649-663 */
    GFX_RESTIR_SHADE_AND_RESAMPLE_TEMPORAL = 17,
    GFX_RESTIR_SHADE_AND_RESAMPLE_SPATIAL = 18,
    GFX_RESTIR_SHADE_AND_RESAMPLE_SPATIOTEMPORAL = 19,
    /* No reference entry point of its own: GFX_RESTIR_SPATIAL_BIASED followed by GFX_RESTIR_SHADING of the reservoirs that pass wrote
     * (currentReservoirIndex is the spatial pass's; the shading reads the other buffer), as restir_di_main.cpp:2393-2420 issues them for
     * the last spatial pass of a frame.  Both on the same rows.  One kernel where the launch is small (a row band), two launches otherwise;
     * same results as the two calls. */
    GFX_RESTIR_SPATIAL_BIASED_AND_SHADING = 20,
    GFX_RESTIR_NUM_PASSES
};
int gfx_restir_launch(gfx_ctx* ctx, void* stream, int pass, uint32_t width, uint32_t height);
/* The same pass restricted to image rows [rowBegin, rowEnd): the unit of the multi-GPU row-band split
 * (no reference counterpart -- restir_di_main.cpp is single GPU, :130-133).  Buffers keep their
 * full-frame size and indexing.  rowBegin == rowEnd == 0 -> every row. */
int gfx_restir_launch_rows(gfx_ctx* ctx, void* stream, int pass, uint32_t width, uint32_t height,
                           uint32_t rowBegin, uint32_t rowEnd);
/* Rows [rowBegin, gapBegin) and [gapEnd, rowEnd) in ONE launch: the seam rows of a band (what the neighbours' next pass reads), which
 * a band renderer runs ahead of the interior [gapBegin, gapEnd) so that their exchange overlaps the interior (gfxexp_host.h, lane
 * SEAM).  GFX_RESTIR_SPATIAL_BIASED only (a per-pixel kernel: which launch computes a pixel changes nothing). */
int gfx_restir_launch_rows_gap(gfx_ctx* ctx, void* stream, int pass, uint32_t width, uint32_t height,
                               uint32_t rowBegin, uint32_t rowEnd, uint32_t gapBegin, uint32_t gapEnd);

/* Output chain (restir_di/gpu_kernels/copy_buffers.cu:6-80; host calls restir_di_main.cpp:2497-2571).
 * gfx_restir_copy_to_linear = copyToLinearBuffers: beauty / albedo / normal accumulation buffers (normal normalised
 * unless zero) and GBuffer1's motion vectors of the current gfx_restir_set_params into caller-owned linear device
 * arrays (float4 x 3, float2).  gfx_visualize = visualizeToOutputBuffer: one linear buffer -> float4 display values. */
enum gfx_buffer_to_display {                       /* BufferToDisplay, restir_di_shared.h:292-298 */
    GFX_DISPLAY_NOISY_BEAUTY = 0, GFX_DISPLAY_ALBEDO = 1, GFX_DISPLAY_NORMAL = 2, GFX_DISPLAY_FLOW = 3, GFX_DISPLAY_DENOISED_BEAUTY = 4
};
int gfx_restir_copy_to_linear(gfx_ctx* ctx, void* stream, void* dLinearColor, void* dLinearAlbedo, void* dLinearNormal, void* dLinearMotionVector);
int gfx_visualize(gfx_ctx* ctx, void* stream, const void* dLinearBuffer, int bufferTypeToDisplay, float motionVectorOffset, float motionVectorScale,
                  uint32_t width, uint32_t height, void* dOutputFloat4);

/* ---------------------------------------------------------------- path tracing ---------------- */

/* The baseline path tracer (path_tracing/path_tracing_main.cpp:2068-2093: G-buffer pipeline, then
 * the pathTraceBaseline pipeline; kernels path_tracing/gpu_kernels/optix_pathtracing_kernels.cu:74-341:
 * NEE + BSDF sampling with power-heuristic MIS, Russian roulette, implicit light hits).
 * Parameters are the ones set by gfx_restir_set_params: path_tracing_shared.h:137-173 is the subset
 * {imageSize, rngBuffer, GBuffer0/1, beauty/albedo/normal accumulation, env light} of the static block
 * and {travHandle, numAccumFrames, camera, prevCamera, envLightPowerCoeff/Rotation, bufferIndex,
 * resetFlowBuffer, enableJittering, enableEnvLight} of the per-frame block; maxPathLength
 * (path_tracing_shared.h:165, 4-bit field, UI range 2..15, default 5 at path_tracing_main.cpp:1519)
 * is passed here.  Rows [rowBegin, rowEnd) as in gfx_restir_launch_rows; rowEnd == 0 -> whole frame.  Of the NRC passes
 * GFX_PT_PATH_TRACE_NRC and GFX_PT_NRC_ACCUMULATE are per pixel and honour the rows; the others run over tiles / records. */
enum gfx_pt_pass {
    GFX_PT_SETUP_GBUFFERS = 0,        /* path_tracing/gpu_kernels/optix_gbuffer_kernels.cu */
    GFX_PT_PATH_TRACE_BASELINE = 1,   /* optix_pathtracing_kernels.cu:298-341 (pathTraceBaseline RG/CH/MS) */
    /* ReGIR (regir/regir_main.cpp:2031-2066); need gfx_regir_set_params */
    GFX_PT_REGIR_BUILD_CELL_RESERVOIRS = 2,           /* regir/gpu_kernels/build_cell_reservoirs.cu:221-223 */
    GFX_PT_REGIR_BUILD_CELL_RESERVOIRS_TEMPORAL = 3,  /* :225-227 */
    GFX_PT_PATH_TRACE_REGIR = 4,                      /* regir/gpu_kernels/optix_pathtracing_kernels.cu:425-433 */
    GFX_PT_REGIR_UPDATE_LAST_ACCESS = 5,              /* build_cell_reservoirs.cu:229-243 */
    /* Neural radiance caching (neural_radiance_caching_main.cpp:2270-2370); need gfx_nrc_set_render_params.
     * The network calls between them are gfx_nrc_infer / gfx_nrc_train. */
    GFX_PT_NRC_PREPROCESS = 6,                        /* nrc_setup_kernels.cu:6-49 */
    GFX_PT_PATH_TRACE_NRC = 7,                        /* neural_radiance_caching/gpu_kernels/optix_pathtracing_kernels.cu:693-703 */
    GFX_PT_NRC_ACCUMULATE = 8,                        /* nrc_setup_kernels.cu:51-93 */
    GFX_PT_NRC_PROPAGATE = 9,                         /* :95-137 */
    GFX_PT_NRC_SHUFFLE = 10,                          /* :139-216 */
    GFX_PT_NRC_VISUALIZE_PREDICTION = 11,             /* optix_pathtracing_kernels.cu:705-778 */
    /* numInferenceQueries = (W * H + #tiles of tileSize[bufferIndex]) rounded up to 128 -- what the reference computes on the
     * host after a stream synchronisation and a device read (neural_radiance_caching_main.cpp:2293-2303) -- written to the
     * context's device word (gfx_nrc_query_count_ptr) for gfx_nrc_infer_indirect: the frame needs no host round trip. */
    GFX_PT_NRC_COUNT_QUERIES = 12,
    /* GFX_PT_PATH_TRACE_NRC with many-light next-event estimation: every path vertex draws its light sample from the ReGIR
     * grid cell under it (sampleFromCell, regir/gpu_kernels/optix_pathtracing_kernels.cu:18-82) instead of the emitter
     * distributions (neural_radiance_caching/gpu_kernels/optix_pathtracing_kernels.cu:38-63).  The combination is an open
     * item of the reference (README.md:80-81 "Combine with many-light sampling techniques like ReSTIR/ReGIR"), so its
     * definition is this build's: the ReGIR estimate has no evaluable density, hence no MIS -- NEE carries all direct light
     * at a path vertex and emitters found by BSDF sampling (path length >= 2, incl. the environment) contribute nothing;
     * Russian roulette, cache termination and the training records are those of the NRC tracer.  Needs gfx_regir_set_params
     * (grid built by GFX_PT_REGIR_BUILD_CELL_RESERVOIRS* before, GFX_PT_REGIR_UPDATE_LAST_ACCESS after) AND
     * gfx_nrc_set_render_params. */
    GFX_PT_PATH_TRACE_NRC_REGIR = 13,
    /* GFX_PT_PATH_TRACE_NRC whose FIRST path vertex takes its next-event estimation from the pixel's ReSTIR DI reservoir -- the other
     * half of the reference's open item (README.md:80-81 "... like ReSTIR/ReGIR"; NEE site neural_radiance_caching/gpu_kernels/
     * optix_pathtracing_kernels.cu:38-63), again a composition of two things the reference has: the frame first runs the original
     * ReSTIR DI passes (GFX_RESTIR_INITIAL_* and the spatial passes, restir_di_main.cpp:2365-2421, up to but excluding SHADING) on the
     * same G-buffers and per-pixel RNGs; this pass then adds, at the first vertex, recPDFEstimate x performDirectLighting of the
     * final reservoir sample at the G-buffer's shading point with its shadow ray (the direct term of the shading pass,
     * optix_restir_di_kernels.cu:574-606) in place of the tracer's own light sample, and draws no random numbers there.  The
     * reservoir estimates all direct light of that vertex and has no density, so what the first extension ray finds emitting (path
     * length 2: surface or environment) contributes nothing; deeper vertices keep the tracer's NEE + MIS.  The reservoirs are the
     * ones of `currentReservoirIndex` (gfx_restir_set_params).  Whole-frame renderers only. */
    GFX_PT_PATH_TRACE_NRC_RESTIR = 14
};

/* The ReGIR members of regir/regir_shared.h:200-263 (grid of cells x 512 light slots).  Light-slot
 * reservoirs use the three-plane layout of the pixel reservoirs: plane k of buffer b at
 * reservoirs[b] + 16 * (k * numLightSlots + slot), numLightSlots = cells * 512.
 * lightSlotRngs are seeded row-major from mt19937_64(591842031321323413) (regir_main.cpp:1086-1092),
 * lastAccessFrameIndices start at 0xFFFFFFFF (regir_main.cpp:1096, 1112: fill(-1)).
 * The build passes read frame.frameIndex and frame.bufferIndex. */
typedef struct gfx_regir_params {
    void* reservoirs[2];
    void* reservoirInfos[2];         /* gfx_reservoir_info[numLightSlots] */
    void* lightSlotRngs;             /* uint64_t[numLightSlots] */
    void* perCellNumAccesses;        /* uint32_t[numCells] */
    void* lastAccessFrameIndices;    /* uint32_t[numCells] */
    void* numActiveCells[2];         /* uint32_t each (statistics) */
    float gridOrigin[3];
    float gridCellSize[3];
    uint32_t gridDimension[3];       /* 32 x 8 x 32 in the reference (regir_main.cpp:1112) */
    uint32_t log2NumCandidatesPerLightSlot;   /* 3 (regir_main.cpp:1733) */
    uint32_t log2NumCandidatesPerCell;        /* 2 (:1734) */
    uint32_t enableCellRandomization;         /* 1 (:1736) */
} gfx_regir_params;
int gfx_regir_set_params(gfx_ctx* ctx, const gfx_regir_params* p);
int gfx_pt_launch(gfx_ctx* ctx, void* stream, int pass, uint32_t width, uint32_t height,
                  uint32_t maxPathLength, uint32_t rowBegin, uint32_t rowEnd);

/* NRC render-side state: the NRC members of neural_radiance_caching_shared.h:229-265 (+ radianceScale of
 * the per-frame block, :277, and the three arguments of preprocessNRC, nrc_setup_kernels.cu:6-9).
 *   RadianceQuery              14 floats = one column of the network input (:118-137)
 *   TerminalInfo               float3 alpha + u32 {hasQuery:1, pathLength:8, isTrainingPixel:1, isUnbiasedTile:1}
 *   TrainingVertexInfo         float3 localThroughput + u32 {prevVertexDataIndex:23, pathLength:8}
 *   TrainingSuffixTerminalInfo u32 {prevVertexDataIndex:23, hasQuery:1, pathLength:8}
 *   dataShuffler               u32 LCG state; entry i = state after i + 1 steps from 471313181
 *                              (neural_radiance_caching_main.cpp:1186-1194)
 * numPixels = W*H; maxNumTrainingSuffixes = W*H/16 (:1150); train buffers hold 131072 records (:9). */
typedef struct gfx_nrc_params {
    float sceneAabbMin[3], sceneAabbMax[3];
    uint32_t maxNumTrainingSuffixes;
    void* numTrainingData[2];            /* uint32_t */
    void* tileSize[2];                   /* uint32_t[2], initialised to 8 x 8 */
    void* targetMinMax[2];               /* int32_t[6] ordered-int min rgb, max rgb */
    void* targetAvg[2];                  /* float[3] */
    void* offsetToSelectUnbiasedTile;    /* uint32_t */
    void* offsetToSelectTrainingPath;    /* uint32_t */
    void* inferenceRadianceQueryBuffer;  /* 56 B x (numPixels + maxNumTrainingSuffixes, rounded up to 256) */
    void* inferenceTerminalInfoBuffer;   /* 16 B x numPixels */
    void* inferredRadianceBuffer;        /* float3 x (numPixels + maxNumTrainingSuffixes, rounded up to 256) */
    void* perFrameContributionBuffer;    /* float3 x numPixels */
    void* trainRadianceQueryBuffer[2];   /* 56 B x 131072 */
    void* trainTargetBuffer[2];          /* float3 x 131072 */
    void* trainVertexInfoBuffer;         /* 16 B x 131072 */
    void* trainSuffixTerminalInfoBuffer; /* uint32_t x maxNumTrainingSuffixes */
    void* dataShufflerBuffer;            /* uint32_t x 65536 */
    float radianceScale;
    uint32_t preprocessOffsetToSelectUnbiasedTile;   /* perFrameRng() draws, :2276-2277 (mt19937(72139121)) */
    uint32_t preprocessOffsetToSelectTrainingPath;
    uint32_t isNewSequence;
} gfx_nrc_params;
int gfx_nrc_set_render_params(gfx_ctx* ctx, const gfx_nrc_params* p);

/* ---------------------------------------------------------------- neural radiance cache -------- */

/* NeuralRadianceCache (neural_radiance_caching/network_interface.h:14-28): the tiny-cuda-nn network of
 * network_interface.cu:48-132 -- Composite{HashGrid | TriangleWave (3 dims), OneBlob 4 bins (5 dims),
 * Identity (6 dims)} -> fully fused MLP, 64 neurons, numHiddenLayers (2 or 5), ReLU; loss
 * RelativeL2Luminance; optimizer EMA(0.99) o Adam(lr, 0.9, 0.99, l2_reg 1e-6).  Here: bf16 MFMA,
 * fp32 accumulation and fp32 master parameters.
 *   gfx_nrc_create    = initialize(posEnc, numHiddenLayers, learningRate)      network_interface.cu:48-132
 *   gfx_nrc_destroy   = finalize()                                            :134-139
 *   gfx_nrc_infer     = infer(stream, inputData, numData, predictionData)     :141-147
 *   gfx_nrc_train     = train(stream, inputData, targetData, numData, lossOnCPU)   :149-157
 * inputData: device fp32 [14, numData] column-major (RadianceQuery, neural_radiance_caching_shared.h:118-127),
 * predictionData / targetData: device fp32 [3, numData]; numData must be a multiple of 128 (:143, :151).
 * Parameter blob (fp32): W0 [64][64], W1.. [64][64] x (numHiddenLayers - 1), Wout [16][64] (rows >= 3
 * unused), then the hash grid (level tables back to back, 2 features per entry); W[out][in] with `in`
 * in the canonical feature order [position | one-blob 4 x 5 | identity 6 | ones].
 * get_params which: 0 training weights, 1 EMA (inference) weights, 2 Adam first moment, 3 second moment. */
enum gfx_nrc_position_encoding { GFX_NRC_TRIANGLE_WAVE = 0, GFX_NRC_HASH_GRID = 1 };   /* network_interface.h:5-8 */
int gfx_nrc_create(gfx_ctx* ctx, int positionEncoding, uint32_t numHiddenLayers, float learningRate, uint64_t* outHandle);
int gfx_nrc_destroy(gfx_ctx* ctx, uint64_t handle);
int gfx_nrc_infer(gfx_ctx* ctx, void* stream, uint64_t handle, const void* dInputData, uint32_t numData, void* dPredictionData);
int gfx_nrc_train(gfx_ctx* ctx, void* stream, uint64_t handle, const void* dInputData, const void* dTargetData,
                  uint32_t numData, float* lossOnCPU);
/* gfx_nrc_infer with the batch size read on the device: *dNumData queries (a multiple of 128, <= maxNumData) are inferred;
 * the launch is sized for maxNumData.  dNumData = gfx_nrc_query_count_ptr after GFX_PT_NRC_COUNT_QUERIES in the NRC frame. */
int gfx_nrc_infer_indirect(gfx_ctx* ctx, void* stream, uint64_t handle, const void* dInputData, const void* dNumData, uint32_t maxNumData,
                           void* dPredictionData);
int gfx_nrc_query_count_ptr(gfx_ctx* ctx, void** dNumData);
int gfx_nrc_num_params(gfx_ctx* ctx, uint64_t handle, uint32_t* outCount);
int gfx_nrc_set_params(gfx_ctx* ctx, uint64_t handle, const float* hostParams, uint32_t count);
int gfx_nrc_get_params(gfx_ctx* ctx, uint64_t handle, int which, float* hostOut, uint32_t count);
/* The device images gfx_nrc_infer reads: which = 0 the packed bf16 MLP fragments, 1 the packed bf16 hash grid (null / 0 for
 * the triangle-wave encoding).  They are brought up to date with the trained (EMA) weights when somebody asks -- this call, its _async
 * form, gfx_nrc_infer -- not after every training step (a frame trains four steps and infers once); whoever packs them first waits, on its
 * stream, for an event gfx_nrc_train records behind its optimizer, so no ordering between the caller's streams is assumed and no stream
 * handle of an earlier call is kept.  gfx_nrc_inference_image has no stream to order with: it packs on a library-owned stream and
 * returns when the images are complete (a host wait for the last training step if the images were stale).  The pointers stay valid for the
 * network's lifetime, the CONTENTS are rewritten by the next refresh after a training step -- a caller that caches the pointers must
 * order its reads before that (e.g. call one of these functions again after training).
 * gfx_nrc_inference_image_async: the same, packed on `stream`; the caller uses the images in `stream` order, no host wait.
 * A process that does not train (a band renderer other than rank 0, gfxh_nrc_set_exchange) receives these bytes from the one that does. */
int gfx_nrc_inference_image(gfx_ctx* ctx, uint64_t handle, int which, void** dPtr, uint64_t* bytes);
int gfx_nrc_inference_image_async(gfx_ctx* ctx, void* stream, uint64_t handle, int which, void** dPtr, uint64_t* bytes);
/* A 32-bit checksum of both inference images (position-weighted word sum, packed on `stream` first if stale) ADDED to the device word
 * *dOutU32: processes that each train a copy of the network on the same batches (the band-split NRC renderer) compare it to notice a
 * copy that has drifted (gfxexp_host.h gfxh_nrc_set_exchange). */
int gfx_nrc_params_checksum(gfx_ctx* ctx, void* stream, uint64_t handle, void* dOutU32);

/* Blocking device-to-host copy of library- or caller-owned device memory (TypedBuffer::read,
 * utils/cuda_util.h; used for pick info at restir_di_main.cpp:2010). */
int gfx_read_device(gfx_ctx* ctx, const void* dSrc, void* hostDst, size_t bytes);

/* Per-kernel HIP-event timing of the launches issued since the last reset
 * (cudau::Timer, utils/cuda_util.h:441-485).  names/ms arrays sized by capacity. */
int gfx_timing_enable(gfx_ctx* ctx, int enable);
int gfx_timing_collect(gfx_ctx* ctx, char names[][48], float* totalMs, uint32_t* calls,
                       uint32_t capacity, uint32_t* n);
/* Scheduling knobs of a context; none changes a result (no reference counterpart: OptiX schedules in the driver).
 *   "pixel_map" 0|1|2           which pixels share a wave / block / XCD in the per-pixel kernels: scan lines, 8 x 8 tiles
 *                               (the tiling of restir_di/gpu_kernels/per_pixel_ris.cu:44-61), tiles + XCD-aware supertiles (default)
 *   "super_x", "super_y"        log2 of the supertile size in 16 x 16-pixel blocks (mode 2; default 2, 2)
 *   "trace_blocks_per_cu", "trace_refill", "trace_batch"   persistent traversal grid, lane-refill threshold, rays per ticket
 *   "temporal_hints" 0|1        the primary ray of a pixel first tests the triangle it hit one frame ago (default 1; GFX_TEMPORAL_HINTS)
 *   "pt_overlap" 0|1            path tracers (gfx_pt_launch): the NEE any-hit trace of a bounce and the kernel that applies it run on a
 *                               library-owned second stream underneath the extension closest-hit trace of the same bounce -- the two
 *                               read and write disjoint buffers; the bounce kernel waits for both (default 1; GFX_PT_OVERLAP)
 *   "fuse_passes" 0|1|2         ray passes as ONE kernel each -- the thread that makes a ray traces it and consumes the result
 *                               (csrc/trace_local.hip.h) -- instead of producer kernel, k_trace launch, consumer kernel.  0 (default): the
 *                               G-buffer pass (GFX_RESTIR_SETUP_GBUFFERS, GFX_PT_SETUP_GBUFFERS) at every size; GFX_RESTIR_INITIAL_*, _SHADING
 *                               and _SPATIAL_BIASED_AND_SHADING for launches of up to about half a full-HD frame (a row band of a multi-GPU
 *                               frame); GFX_PT_PATH_TRACE_BASELINE / _REGIR (the whole path of a pixel in one kernel) for launches of about
 *                               one round of waves (512 x 512).  1 never, 2 always (GFX_FUSE_PASSES).  Counting launches
 *                               (gfx_counters_enable) always take the k_trace form.
 *   "block_order" 0|1           fused GFX_RESTIR_INITIAL_* / _SHADING kernels whose launch is several rounds of blocks start their blocks by
 *                               decreasing cost (traversal steps of the block's longest ray) of the same launch one frame ago instead of in
 *                               index order: the launch ends when its last tracing wave does (default 1; GFX_BLOCK_ORDER)
 *   "candidate_split" 0|1|2|4   lanes per pixel in the candidate loop of the initial-RIS passes (GFX_RESTIR_INITIAL_*): the lanes take the
 *                               pixel's candidates round robin and the reservoir is formed as the sequential loop forms it; 0 (default)
 *                               = by launch size: 4 when the launch fills the GPU's wave slots at most ~1.5 times (a row band of an
 *                               8-way split frame), else 1 (GFX_CANDIDATE_SPLIT)
 *   "nrc_staged_infer" 0|1|2    gfx_nrc_infer with the hash-grid encoding: 0 (default) a batch that gives every CU at least one pass of 3 072 queries is
 *                               encoded level by level out of LDS copies of the level tables (one persistent block per CU; same predictions bit for
 *                               bit), smaller batches gather from the tables in place; 1 never, 2 always (GFX_NRC_STAGED_INFER)
 * The same knobs are read once from the environment by gfx_ctx_create (GFX_PIXEL_MAP, GFX_SUPER_X, GFX_SUPER_Y,
 * GFX_TRACE_BLOCKS_PER_CU, GFX_TRACE_REFILL, GFX_TRACE_BATCH). */
int gfx_tunable_set(gfx_ctx* ctx, const char* name, int value);
/* Ray-traversal counters accumulated by the renderer passes: {node fetches, triangle fetches, rays (queue entries with an empty
 * interval are not counted), stack spills}
 * of the any-hit launches in [0..3] and of the closest-hit launches in [4..7]. */
int gfx_counters_enable(gfx_ctx* ctx, int enable);
int gfx_counters_read(gfx_ctx* ctx, uint64_t counters[8], int reset);
/* Scheduling diagnostics of the counting trace launches: [0] wave iterations, [1] lanes that held an
 * item summed over iterations, [2] / [3] the same after the ray queue ran dry (drain phase), [4] clock cycles summed over
 * the waves, of which [5] in the ray refill (ticket, ray loads, setup), [6] waiting for the item fetch, [7] processing
 * items (the rest: item selection, loop overhead). */
int gfx_trace_diag_read(gfx_ctx* ctx, uint64_t diag[8], int reset);
/* The one-kernel path tracers (k_pt_fused, k_pt_regen) with the tunable "pt_diag" set: [0] bounce iterations summed over the waves, [1] lanes
 * that held a ray in them, [2] traversal steps, [3] waves, [4] refills (k_pt_regen). */
int gfx_pt_diag_read(gfx_ctx* ctx, uint64_t diag[8], int reset);
/* Measurement utility ("measure peak with a streaming-copy microbenchmark on the box, don't quote the datasheet"): copies
 * `bytes` (a multiple of 16; both pointers 16-byte aligned, device memory) with 16-byte non-temporal loads and stores per lane, on
 * `stream`; dDst == NULL makes it a read-only pass over dSrc.  bench.py times both with HIP events for roofline.peak_measured /
 * peak_measured_read_only; no renderer calls it. */
int gfx_stream_copy(gfx_ctx* ctx, void* dDst, const void* dSrc, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GFXEXP_H */
