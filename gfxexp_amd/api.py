"""ctypes bindings of libgfxexp.so (include/gfxexp.h + include/gfxexp_host.h).

The library is the product: hand-written HIP kernels for gfx950 behind a C ABI.  There is no CPU
fallback -- if the shared object is missing or a call fails, these bindings raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GFX_LIB") or os.path.join(_HERE, "libgfxexp.so")   # GFX_LIB: an experiment build (build.build_variant)

GFX_INVALID_SLOT = 0xFFFFFFFF
TRACE_CLOSEST, TRACE_ANY = 0, 1
(PASS_SETUP_GBUFFERS, PASS_INITIAL_RIS, PASS_INITIAL_TEMPORAL_BIASED, PASS_INITIAL_TEMPORAL_UNBIASED,
 PASS_SPATIAL_BIASED, PASS_SPATIAL_UNBIASED, PASS_SHADING) = range(7)
PASS_SPATIAL_BIASED_AND_SHADING = 20
# rearchitected ReSTIR: trace-shadow-rays / shade-and-resample entry points in the order of
# RearchitectedReSTIREntryPoint (restir_di_main.cpp:83-95)
PASS_LIGHT_PRESAMPLING, PASS_PER_PIXEL_RIS, PASS_TRACE_SHADOW_RAYS = 7, 8, 9
PASS_SHADE_AND_RESAMPLE = 16


def rearch_passes(temporal, spatial, unbiased, new_sequence):
    """(traceShadowRays pass, shadeAndResample pass) as restir_di_main.cpp:2446-2484 selects them."""
    if new_sequence or not (temporal or spatial):
        return PASS_TRACE_SHADOW_RAYS, PASS_SHADE_AND_RESAMPLE
    k = (1 if temporal and not spatial else 2 if spatial and not temporal else 3)
    return PASS_TRACE_SHADOW_RAYS + k + (3 if unbiased else 0), PASS_SHADE_AND_RESAMPLE + k
RENDERER_BIASED, RENDERER_UNBIASED, RENDERER_REARCH_BIASED, RENDERER_REARCH_UNBIASED, RENDERER_PATH_TRACE, RENDERER_PATH_TRACE_REGIR = 0, 1, 2, 3, 4, 5
(PT_SETUP_GBUFFERS, PT_PATH_TRACE_BASELINE, PT_REGIR_BUILD_CELLS, PT_REGIR_BUILD_CELLS_TEMPORAL,
 PT_PATH_TRACE_REGIR, PT_REGIR_UPDATE_LAST_ACCESS, PT_NRC_PREPROCESS, PT_PATH_TRACE_NRC, PT_NRC_ACCUMULATE,
 PT_NRC_PROPAGATE, PT_NRC_SHUFFLE, PT_NRC_VISUALIZE_PREDICTION, PT_NRC_COUNT_QUERIES, PT_PATH_TRACE_NRC_REGIR, PT_PATH_TRACE_NRC_RESTIR) = range(15)


class GfxError(RuntimeError):
    pass


class GfxMaterial(C.Structure):
    _fields_ = [("bsdfType", C.c_uint32), ("a", C.c_float * 3), ("b", C.c_float * 3),
                ("smoothness", C.c_float), ("emittance", C.c_float * 3), ("hasEmittance", C.c_uint32),
                ("texA", C.c_uint32), ("texB", C.c_uint32), ("texSmoothness", C.c_uint32), ("texNormal", C.c_uint32),
                ("texEmittance", C.c_uint32), ("bumpMapType", C.c_uint32), ("pad", C.c_uint32 * 2)]


class GfxCamera(C.Structure):
    _fields_ = [("aspect", C.c_float), ("fovY", C.c_float), ("position", C.c_float * 3),
                ("orientation", C.c_float * 9)]


class GfxRestirStaticParams(C.Structure):
    _fields_ = [
        ("imageSizeX", C.c_int32), ("imageSizeY", C.c_int32),
        ("rngBuffer", C.c_void_p),
        ("gbuffer0", C.c_void_p * 2), ("gbuffer1", C.c_void_p * 2),
        ("gbuffer2", C.c_void_p * 2), ("gbuffer3", C.c_void_p * 2),
        ("reservoirBuffer", C.c_void_p * 2), ("reservoirInfoBuffer", C.c_void_p * 2),
        ("sampleVisibilityBuffer", C.c_void_p * 2),
        ("spatialNeighborDeltas", C.c_void_p),
        ("beautyAccumBuffer", C.c_void_p), ("albedoAccumBuffer", C.c_void_p), ("normalAccumBuffer", C.c_void_p),
        ("numTilesX", C.c_int32), ("numTilesY", C.c_int32),
        ("lightPreSamplingRngs", C.c_void_p), ("preSampledLights", C.c_void_p),
        ("envLightTexture", C.c_void_p), ("envWidth", C.c_int32), ("envHeight", C.c_int32),
        ("envRowPDF", C.c_void_p), ("envRowCDF", C.c_void_p), ("envRowIntegrals", C.c_void_p),
        ("envTopPDF", C.c_void_p), ("envTopCDF", C.c_void_p), ("envTopIntegral", C.c_float),
        ("envRowGuide", C.c_void_p), ("envTopGuide", C.c_void_p), ("envRowTable", C.c_void_p), ("envRowSketch", C.c_void_p),
    ]


class GfxRestirFrameParams(C.Structure):
    _fields_ = [
        ("travHandle", C.c_uint64), ("numAccumFrames", C.c_uint32), ("frameIndex", C.c_uint32),
        ("camera", GfxCamera), ("prevCamera", GfxCamera),
        ("envLightPowerCoeff", C.c_float), ("envLightRotation", C.c_float),
        ("spatialNeighborRadius", C.c_float), ("radiusThresholdForSpatialVisReuse", C.c_float),
        ("log2NumCandidateSamples", C.c_uint32), ("numSpatialNeighbors", C.c_uint32),
        ("useLowDiscrepancyNeighbors", C.c_uint32), ("reuseVisibility", C.c_uint32),
        ("reuseVisibilityForTemporal", C.c_uint32), ("reuseVisibilityForSpatiotemporal", C.c_uint32),
        ("enableTemporalReuse", C.c_uint32), ("enableSpatialReuse", C.c_uint32),
        ("useUnbiasedEstimator", C.c_uint32), ("bufferIndex", C.c_uint32),
        ("resetFlowBuffer", C.c_uint32), ("enableJittering", C.c_uint32),
        ("enableEnvLight", C.c_uint32), ("enableBumpMapping", C.c_uint32), ("useSolidAngleSampling", C.c_uint32),
    ]


class GfxRegirParams(C.Structure):
    _fields_ = [("reservoirs", C.c_void_p * 2), ("reservoirInfos", C.c_void_p * 2), ("lightSlotRngs", C.c_void_p),
                ("perCellNumAccesses", C.c_void_p), ("lastAccessFrameIndices", C.c_void_p),
                ("numActiveCells", C.c_void_p * 2), ("gridOrigin", C.c_float * 3), ("gridCellSize", C.c_float * 3),
                ("gridDimension", C.c_uint32 * 3), ("log2NumCandidatesPerLightSlot", C.c_uint32),
                ("log2NumCandidatesPerCell", C.c_uint32), ("enableCellRandomization", C.c_uint32)]


class GfxNrcParams(C.Structure):
    _fields_ = [("sceneAabbMin", C.c_float * 3), ("sceneAabbMax", C.c_float * 3), ("maxNumTrainingSuffixes", C.c_uint32),
                ("numTrainingData", C.c_void_p * 2), ("tileSize", C.c_void_p * 2), ("targetMinMax", C.c_void_p * 2),
                ("targetAvg", C.c_void_p * 2), ("offsetToSelectUnbiasedTile", C.c_void_p),
                ("offsetToSelectTrainingPath", C.c_void_p), ("inferenceRadianceQueryBuffer", C.c_void_p),
                ("inferenceTerminalInfoBuffer", C.c_void_p), ("inferredRadianceBuffer", C.c_void_p),
                ("perFrameContributionBuffer", C.c_void_p), ("trainRadianceQueryBuffer", C.c_void_p * 2),
                ("trainTargetBuffer", C.c_void_p * 2), ("trainVertexInfoBuffer", C.c_void_p),
                ("trainSuffixTerminalInfoBuffer", C.c_void_p), ("dataShufflerBuffer", C.c_void_p),
                ("radianceScale", C.c_float), ("preprocessOffsetToSelectUnbiasedTile", C.c_uint32),
                ("preprocessOffsetToSelectTrainingPath", C.c_uint32), ("isNewSequence", C.c_uint32)]


TEX_RGBA8_SRGB, TEX_RGBA8_UNORM, TEX_R8_UNORM, TEX_RG8_UNORM, TEX_RGBA32F = 0, 1, 2, 3, 4
BUMP_NORMAL_MAP, BUMP_NORMAL_MAP_2CH, BUMP_HEIGHT_MAP, BUMP_LEFT_HANDED = 0, 1, 2, 0x100


class GfxhStreetParams(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("groundTess", C.c_uint32), ("numBuildings", C.c_uint32),
                ("facadeTess", C.c_uint32), ("numProps", C.c_uint32), ("propSubdiv", C.c_uint32),
                ("numLamps", C.c_uint32), ("numSigns", C.c_uint32), ("extent", C.c_float),
                ("lampEmittance", C.c_float), ("signEmittance", C.c_float), ("textured", C.c_uint32),
                ("numTrees", C.c_uint32), ("leavesPerTree", C.c_uint32), ("numWires", C.c_uint32), ("numRailings", C.c_uint32)]


class GfxhRestirConfig(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("renderer", C.c_int),
                ("log2NumCandidateSamples", C.c_uint32), ("enableTemporalReuse", C.c_uint32),
                ("enableSpatialReuse", C.c_uint32), ("numSpatialReusePasses", C.c_uint32),
                ("numSpatialNeighbors", C.c_uint32), ("spatialNeighborRadius", C.c_float),
                ("useLowDiscrepancyNeighbors", C.c_uint32), ("reuseVisibility", C.c_uint32),
                ("enableAccumulation", C.c_uint32), ("log2MaxNumAccums", C.c_uint32),
                ("camera", GfxCamera), ("rowBegin", C.c_uint32), ("rowEnd", C.c_uint32),
                ("maxPathLength", C.c_uint32), ("enableJittering", C.c_uint32),
                ("regirAabbMin", C.c_float * 3), ("regirAabbMax", C.c_float * 3), ("regirGridDimension", C.c_uint32 * 3),
                ("regirLog2CandidatesPerLightSlot", C.c_uint32), ("regirLog2CandidatesPerCell", C.c_uint32),
                ("regirEnableTemporalReuse", C.c_uint32), ("regirEnableCellRandomization", C.c_uint32),
                ("enableBumpMapping", C.c_uint32)]


class GfxhFrameStep(C.Structure):
    _fields_ = [("op", C.c_uint32), ("pass_", C.c_uint32), ("rowBegin", C.c_uint32), ("rowEnd", C.c_uint32),
                ("currentReservoirIndex", C.c_uint32), ("spatialNeighborBaseIndex", C.c_uint32),
                ("exchangeRows", C.c_uint32), ("buffers", C.c_uint32), ("reservoirIndex", C.c_uint32), ("lane", C.c_uint32),
                ("gapBegin", C.c_uint32), ("gapEnd", C.c_uint32)]


class GfxhExchangeBuffer(C.Structure):
    _fields_ = [("base", C.c_void_p), ("bytesPerPixel", C.c_uint32), ("numPlanes", C.c_uint32), ("planeStride", C.c_uint64)]


class GfxhExchangeDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("stage", C.c_uint32), ("lane", C.c_uint32), ("reserved", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("bandBegin", C.c_uint32), ("bandEnd", C.c_uint32),
                ("sendAbove", C.c_uint32 * 2), ("recvAbove", C.c_uint32 * 2), ("sendBelow", C.c_uint32 * 2), ("recvBelow", C.c_uint32 * 2),
                ("numBuffers", C.c_uint32), ("buffers", GfxhExchangeBuffer * 8),
                ("counters", C.c_void_p), ("numCounters", C.c_uint64)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(GfxhExchangeDesc))
EXCHANGE_STRIPS, EXCHANGE_ALLREDUCE_SUM_U32, EXCHANGE_GATHER_BANDS, EXCHANGE_GATHER_RECORDS, EXCHANGE_BROADCAST = 0, 1, 2, 3, 4
(STEP_RESTIR_PASS, STEP_PT_PASS, STEP_EXCHANGE_STRIPS, STEP_ALLREDUCE_CELL_ACCESSES, STEP_GATHER_BANDS, STEP_PREV_GBUFFER_RELEASED,
 STEP_WAIT_GBUFFER_STRIPS, STEP_WAIT_PREVIOUS_GATHER, STEP_WAIT_SEAM_STRIPS) = range(9)
LANE_MAIN, LANE_GBUFFER, LANE_GATHER, LANE_SEAM, NUM_LANES = 0, 1, 2, 3, 4
BUF_GBUFFERS, BUF_RESERVOIRS, BUF_SAMPLE_VISIBILITY, BUF_RNG = 1, 2, 4, 8


def frame_program(cfg, strip_mode, max_motion_rows, new_sequence, last_res, last_base, unbiased):
    """gfxh_restir_frame_program -> (steps, new_last_res, new_last_base)."""
    steps = (GfxhFrameStep * 64)()
    n, nr, nb = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = lib().gfxh_restir_frame_program(C.byref(cfg), C.c_int(int(strip_mode)), C.c_uint32(max_motion_rows), C.c_int(int(new_sequence)),
                                         C.c_uint32(last_res), C.c_uint32(last_base), C.c_uint32(int(unbiased)), steps, C.c_uint32(64),
                                         C.byref(n), C.byref(nr), C.byref(nb))
    if rc:
        raise GfxError("gfxh_restir_frame_program: the exchange strip is taller than the band")
    return [steps[i] for i in range(n.value)], nr.value, nb.value


def exchange_desc(cfg, step, step_index, static_params, regir_params, buffer_index):
    """gfxh_frame_step_exchange_desc: the descriptor of an exchange step over the buffers of `static_params` (any structure
    with the gfx_restir_static_params layout, device or host pointers)."""
    d = GfxhExchangeDesc()
    rc = lib().gfxh_frame_step_exchange_desc(C.byref(cfg), C.byref(step), C.c_uint32(step_index), C.byref(static_params),
                                             C.byref(regir_params) if regir_params is not None else None, C.c_uint32(buffer_index), C.byref(d))
    if rc:
        raise GfxError("gfxh_frame_step_exchange_desc: not an exchange step, or the strip is taller than the band")
    return d


def abi_layout():
    """gfxh_abi_entry: {struct: {"size": sizeof, "fields": [(name, offset, size), ...]}} as the library's compiler laid the
    structs of include/gfxexp.h and include/gfxexp_host.h out."""
    L = lib()
    out = {}
    for i in range(L.gfxh_abi_num_entries()):
        sn, fn = C.c_char_p(), C.c_char_p()
        off, size = C.c_uint64(), C.c_uint64()
        L.gfxh_abi_entry(C.c_uint32(i), C.byref(sn), C.byref(fn), C.byref(off), C.byref(size))
        e = out.setdefault(sn.value.decode(), {"size": None, "fields": []})
        if fn.value is None:
            e["size"] = size.value
        else:
            e["fields"].append((fn.value.decode(), off.value, size.value))
    return out


# which ctypes class mirrors which C struct (tests/test_abi_and_host.py checks every one against abi_layout())
def abi_mirrors():
    return {"gfx_material": GfxMaterial, "gfx_camera": GfxCamera, "gfx_restir_static_params": GfxRestirStaticParams,
            "gfx_restir_frame_params": GfxRestirFrameParams, "gfx_regir_params": GfxRegirParams, "gfx_nrc_params": GfxNrcParams,
            "gfxh_street_params": GfxhStreetParams, "gfxh_restir_config": GfxhRestirConfig, "gfxh_frame_step": GfxhFrameStep,
            "gfxh_exchange_buffer": GfxhExchangeBuffer, "gfxh_exchange_desc": GfxhExchangeDesc, "gfxh_band_plan": GfxhBandPlan,
            "gfxh_nrc_config": GfxhNrcConfig, "gfxh_sdr_config": GfxhSdrConfig}


class RcclExchange:
    """gfxh_rccl: the C++ exchange callback over RCCL (csrc/host/rccl_exchange.cpp), one communicator per lane.  `ids` = the
    bytes of `lanes` ncclUniqueIds from unique_ids() on rank 0, distributed by the caller."""

    @staticmethod
    def unique_ids(lanes=NUM_LANES):
        L = lib()
        L.gfxh_rccl_last_error.restype = C.c_char_p
        raw = (C.c_uint8 * (128 * lanes))()
        for k in range(lanes):
            if L.gfxh_rccl_unique_id(C.byref(raw, 128 * k)):
                raise GfxError("gfxh_rccl_unique_id: " + L.gfxh_rccl_last_error().decode(errors="replace"))
        return bytes(raw)

    def __init__(self, ids, rank, world, height, bands=None):
        self.L = lib()
        self.L.gfxh_rccl_last_error.restype = C.c_char_p
        self.rank, self.world = rank, world
        lanes = len(ids) // 128
        raw = (C.c_uint8 * len(ids)).from_buffer_copy(ids)
        h = C.c_void_p()
        if self.L.gfxh_rccl_create_lanes(raw, C.c_uint32(lanes), C.c_int(rank), C.c_int(world), C.c_uint32(height), C.byref(h)):
            raise GfxError("gfxh_rccl_create_lanes: " + self.L.gfxh_rccl_last_error().decode(errors="replace"))
        self.h = h
        if bands is not None:
            self.set_bands(bands)

    def set_bands(self, bands):
        begins = (C.c_uint32 * (len(bands) + 1))(*([b for b, _ in bands] + [bands[-1][1]]))
        if self.L.gfxh_rccl_set_bands(self.h, begins):
            raise GfxError("gfxh_rccl_set_bands: " + self.L.gfxh_rccl_last_error().decode(errors="replace"))

    def install(self, renderer, max_motion_rows=0):
        renderer.set_exchange_native(self.L.gfxh_rccl_exchange, self.h, max_motion_rows)

    def close(self):
        if self.h:
            self.L.gfxh_rccl_destroy(self.h)
            self.h = None


def band_rows(height, world, rank):
    """gfxh_band_rows: the row band of `rank` (whole 8-row tiles, remainder spread from rank 0)."""
    b, e = C.c_uint32(), C.c_uint32()
    if lib().gfxh_band_rows(C.c_uint32(height), C.c_uint32(world), C.c_uint32(rank), C.byref(b), C.byref(e)):
        raise GfxError("gfxh_band_rows: rank outside the world")
    return b.value, e.value


def check_partition(cfg, world, max_motion_rows=0):
    """gfxh_restir_check_partition: raises when a strip of this configuration is taller than the smallest band of `world`
    ranks -- the same verdict on every rank, before anyone enters a collective."""
    if lib().gfxh_restir_check_partition(C.byref(cfg), C.c_uint32(world), C.c_uint32(max_motion_rows)):
        raise GfxError(lib().gfxh_restir_last_error().decode(errors="replace"))


def check_bands(cfg, bands, max_motion_rows=0):
    """gfxh_restir_check_bands: the strip-feasibility verdict for an explicit partition [(begin, end)] (cost-balanced bands)."""
    begins = (C.c_uint32 * (len(bands) + 1))(*([b for b, _ in bands] + [bands[-1][1]]))
    if lib().gfxh_restir_check_bands(C.byref(cfg), C.c_uint32(len(bands)), begins, C.c_uint32(max_motion_rows)):
        raise GfxError(lib().gfxh_restir_last_error().decode(errors="replace"))


def balance_bands(height, bands, band_ms, min_rows=8):
    """gfxh_balance_bands: the partition that would have equalised the band times `band_ms` measured with `bands`
    ([(begin, end)] per rank; every rank must pass the same numbers).  Returns the new [(begin, end)]."""
    world = len(bands)
    begins = (C.c_uint32 * (world + 1))(*([b for b, _ in bands] + [bands[-1][1]]))
    ms = (C.c_float * world)(*[float(t) for t in band_ms])
    out = (C.c_uint32 * (world + 1))()
    if lib().gfxh_balance_bands(C.c_uint32(height), C.c_uint32(world), begins, ms, C.c_uint32(min_rows), out):
        raise GfxError("gfxh_balance_bands: invalid partition or times")
    return [(int(out[r]), int(out[r + 1])) for r in range(world)]


def strip_rows(height, band_begin, band_end, rows):
    d = GfxhExchangeDesc()
    rc = lib().gfxh_strip_rows(C.c_uint32(height), C.c_uint32(band_begin), C.c_uint32(band_end), C.c_uint32(rows), C.byref(d))
    return d, rc


class GfxhBandPlan(C.Structure):
    _fields_ = [("bandBegin", C.c_uint32), ("bandEnd", C.c_uint32), ("haloRows", C.c_uint32),
                ("gbufferRows", C.c_uint32 * 2), ("initialRows", C.c_uint32 * 2), ("spatialRows", (C.c_uint32 * 2) * 8),
                ("shadingRows", C.c_uint32 * 2), ("recvAbove", C.c_uint32 * 2), ("sendAbove", C.c_uint32 * 2),
                ("recvBelow", C.c_uint32 * 2), ("sendBelow", C.c_uint32 * 2)]


HIT_DTYPE = np.dtype([("dist", "<f4"), ("bcB", "<f4"), ("bcC", "<f4"), ("triIndex", "<u4")])
TRI_IDS_DTYPE = np.dtype([("instSlot", "<u4"), ("geomInstSlot", "<u4"), ("primIndex", "<u4")])
VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("texCoord0Dir", "<f4", 3),
                         ("texCoord", "<f4", 2)])
GBUFFER0_DTYPE = np.dtype([("instSlot", "<u4"), ("geomInstSlot", "<u4"), ("primIndex", "<u4"),
                           ("qbcB", "<u2"), ("qbcC", "<u2")])
GBUFFER2_DTYPE = np.dtype([("positionInWorld", "<f4", 3), ("qGeometricNormal", "<u4")])
GBUFFER3_DTYPE = np.dtype([("qShadingNormal", "<u4"), ("qShadingTangent", "<u4"), ("qTexCoord", "<u4"),
                           ("matSlot", "<u4")])

# every symbol include/gfxexp.h and include/gfxexp_host.h declare
C_ABI_SYMBOLS = [
    "gfx_ctx_create", "gfx_ctx_destroy", "gfx_last_error", "gfx_version", "gfx_material_set", "gfx_texture_set", "gfx_texture_sample", "gfx_geom_create",
    "gfx_group_create", "gfx_instance_create", "gfx_instance_set_transform", "gfx_instance_set_transform_and_normal_matrix", "gfx_instance_set_dynamic",
    "gfx_accel_build",
    "gfx_accel_set_max_leaf", "gfx_accel_stats", "gfx_accel_tri_ids", "gfx_lights_build_static",
    "gfx_lights_build_instances", "gfx_lights_read", "gfx_lights_table_info", "gfx_trace", "gfx_trace_counted", "gfx_restir_set_params", "gfx_restir_copy_to_linear", "gfx_visualize", "gfx_restir_launch",
    "gfx_restir_launch_rows", "gfx_restir_launch_rows_gap", "gfx_pt_launch", "gfx_regir_set_params",
    "gfx_nrc_create", "gfx_nrc_destroy", "gfx_nrc_infer", "gfx_nrc_infer_indirect", "gfx_nrc_query_count_ptr", "gfx_nrc_train", "gfx_nrc_num_params", "gfx_nrc_set_params",
    "gfx_nrc_get_params", "gfx_nrc_inference_image", "gfx_nrc_inference_image_async", "gfx_nrc_params_checksum", "gfx_nrc_set_render_params",
    "gfx_read_device", "gfx_timing_enable", "gfx_timing_collect", "gfx_counters_enable", "gfx_counters_read", "gfx_trace_diag_read", "gfx_pt_diag_read",
    "gfx_tunable_set", "gfx_stream_copy",
]
HOST_ABI_SYMBOLS = [
    "gfxh_scene_create", "gfxh_scene_destroy", "gfxh_last_error", "gfxh_scene_add_material_traditional",
    "gfxh_scene_add_material", "gfxh_scene_add_texture", "gfxh_scene_load_texture", "gfxh_scene_num_textures", "gfxh_scene_get_texture", "gfxh_scene_add_geom", "gfxh_scene_add_group", "gfxh_scene_add_instance",
    "gfxh_scene_load_obj", "gfxh_scene_load_obj_conv", "gfxh_scene_add_rectangle_textured", "gfxh_scene_add_rectangle", "gfxh_scene_make_street", "gfxh_scene_counts",
    "gfxh_scene_get_material", "gfxh_scene_get_geom", "gfxh_scene_get_group", "gfxh_scene_get_instance",
    "gfxh_scene_bounds", "gfxh_scene_upload", "gfxh_make_transform", "gfxh_make_orientation",
    "gfxh_seed_rng_states", "gfxh_spatial_neighbor_deltas", "gfxh_restir_default_config", "gfxh_band_plan_compute",
    "gfxh_restir_band_plan", "gfxh_restir_set_exchange", "gfxh_strip_rows", "gfxh_band_rows", "gfxh_restir_check_partition", "gfxh_restir_check_bands", "gfxh_balance_bands", "gfxh_restir_frame_program", "gfxh_frame_step_exchange_desc",
    "gfxh_rccl_unique_id", "gfxh_rccl_create", "gfxh_rccl_create_lanes", "gfxh_rccl_set_bands", "gfxh_rccl_destroy", "gfxh_rccl_exchange", "gfxh_rccl_last_error", "gfxh_restir_create",
    "gfxh_restir_set_async_gather", "gfxh_restir_finish_gather", "gfxh_abi_layout", "gfxh_abi_num_entries", "gfxh_abi_entry",
    "gfxh_env_build_importance", "gfxh_env_build_guides", "gfxh_env_build_row_table", "gfxh_env_build_row_sketch", "gfxh_env_upload", "gfxh_env_make_sky", "gfxh_restir_set_env",
    "gfxh_restir_destroy", "gfxh_restir_render_frame", "gfxh_restir_outputs_consumed", "gfxh_restir_reset", "gfxh_restir_set_camera", "gfxh_restir_rebuild_accel",
    "gfxh_restir_beauty_buffer", "gfxh_restir_get_params", "gfxh_restir_accel",
    "gfxh_nrc_default_config", "gfxh_nrc_create", "gfxh_nrc_destroy", "gfxh_nrc_render_frame", "gfxh_nrc_outputs_consumed", "gfxh_nrc_set_exchange", "gfxh_nrc_rebuild_accel", "gfxh_nrc_set_env", "gfxh_nrc_beauty_buffer",
    "gfxh_nrc_network", "gfxh_nrc_stats", "gfxh_save_image_sdr", "gfxh_save_image_hdr", "gfxh_tonemap_sdr",
]

_lib = None


def lib():
    """Load libgfxexp.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GfxError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 /
        # libhsa-runtime64.so.1.  Importing torch first makes the loader resolve this library's
        # NEEDED libamdhip64.so.7 to that already-loaded copy (same SONAME); the other order loads a
        # second runtime and torch then reports "No HIP GPUs are available".
        try:
            import torch  # noqa: F401  (device memory / streams / torch.distributed plumbing)
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.gfx_last_error.restype = C.c_char_p
        L.gfx_version.restype = C.c_char_p
        L.gfxh_last_error.restype = C.c_char_p
        L.gfxh_restir_last_error.restype = C.c_char_p
        L.gfxh_scene_create.restype = C.c_void_p
        L.gfxh_restir_beauty_buffer.restype = C.c_void_p
        L.gfxh_restir_accel.restype = C.c_uint64
        for name in ("gfxh_scene_add_material_traditional", "gfxh_scene_add_material", "gfxh_scene_add_geom",
                     "gfxh_scene_add_group", "gfxh_scene_add_instance", "gfxh_scene_load_obj", "gfxh_scene_load_obj_conv",
                     "gfxh_scene_add_rectangle", "gfxh_scene_add_rectangle_textured"):
            getattr(L, name).restype = C.c_uint32
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


# ---------------------------------------------------------------- host layer: scenes
class HostScene:
    """gfxh_scene: materials / geometry / groups / instances on the host."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.gfxh_scene_create())

    def close(self):
        if self.h:
            self.L.gfxh_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_material_traditional(self, diffuse, specular, smoothness, emittance=(0, 0, 0)):
        return self.L.gfxh_scene_add_material_traditional(self.h, _f3(diffuse), _f3(specular), C.c_float(smoothness),
                                                          _f3(emittance))

    def add_material(self, mat):
        return self.L.gfxh_scene_add_material(self.h, C.byref(mat))

    def add_texture(self, texels, fmt):
        """texels: (H, W, 4) uint8 for the RGBA8 formats, (H, W) uint8 for R8, (H, W, 2) uint8 for RG8, (H, W, 4) float32 for
        RGBA32F.  Returns the 1-based texture slot."""
        t = np.ascontiguousarray(texels)
        h, w = t.shape[0], t.shape[1]
        slot = self.L.gfxh_scene_add_texture(self.h, C.c_uint32(w), C.c_uint32(h), C.c_uint32(fmt), _p(t))
        if slot == 0:
            raise GfxError(self.L.gfxh_last_error().decode(errors="replace"))
        return slot

    def load_texture(self, path, fmt8=0):
        slot = self.L.gfxh_scene_load_texture(self.h, path.encode(), C.c_uint32(fmt8))
        if slot == 0:
            raise GfxError(self.L.gfxh_last_error().decode(errors="replace"))
        return slot

    def textures(self):
        """[(slot, width, height, format, texel bytes)] of every texture."""
        out = []
        for slot in range(1, self.L.gfxh_scene_num_textures(self.h) + 1):
            w, h, f, ptr = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_void_p()
            self.L.gfxh_scene_get_texture(self.h, C.c_uint32(slot), C.byref(w), C.byref(h), C.byref(f), C.byref(ptr))
            bpp = {TEX_RGBA8_SRGB: 4, TEX_RGBA8_UNORM: 4, TEX_R8_UNORM: 1, TEX_RG8_UNORM: 2, TEX_RGBA32F: 16}[f.value]
            data = np.frombuffer((C.c_char * (bpp * w.value * h.value)).from_address(ptr.value), dtype=np.uint8).copy()
            out.append((slot, w.value, h.value, f.value, data))
        return out

    def add_geom(self, vertices, triangles, mat_slot):
        v = np.ascontiguousarray(vertices)
        assert v.dtype == VERTEX_DTYPE
        t = np.ascontiguousarray(triangles, np.uint32).reshape(-1, 3)
        return self.L.gfxh_scene_add_geom(self.h, _p(v), C.c_uint32(len(v)), _p(t), C.c_uint32(len(t)), C.c_uint32(mat_slot))

    def add_group(self, geoms):
        g = np.ascontiguousarray(geoms, np.uint32)
        return self.L.gfxh_scene_add_group(self.h, _p(g), C.c_uint32(len(g)))

    def add_instance(self, group, xfm12):
        x = np.ascontiguousarray(xfm12, np.float32).reshape(12)
        return self.L.gfxh_scene_add_instance(self.h, C.c_uint32(group), _p(x))

    def load_obj(self, path, simple_pbr=False):
        if simple_pbr:
            g = self.L.gfxh_scene_load_obj_conv(self.h, path.encode(), C.c_int(1))
            if g == 0xFFFFFFFF:
                raise GfxError("gfxh_scene_load_obj: " + self.L.gfxh_last_error().decode(errors="replace"))
            return g
        return self._load_obj_trad(path)

    def _load_obj_trad(self, path):
        g = self.L.gfxh_scene_load_obj(self.h, path.encode())
        if g == GFX_INVALID_SLOT:
            raise GfxError(self.L.gfxh_last_error().decode(errors="replace"))
        return g

    def add_rectangle(self, width, depth, emittance):
        return self.L.gfxh_scene_add_rectangle(self.h, C.c_float(width), C.c_float(depth), _f3(emittance))

    def make_street(self, params):
        if self.L.gfxh_scene_make_street(self.h, C.byref(params)):
            raise GfxError(self.L.gfxh_last_error().decode(errors="replace"))

    def counts(self):
        c = (C.c_uint32 * 5)()
        self.L.gfxh_scene_counts(self.h, c)
        return dict(materials=c[0], geoms=c[1], groups=c[2], insts=c[3], triangles=c[4])

    def bounds(self):
        b = (C.c_float * 6)()
        self.L.gfxh_scene_bounds(self.h, b)
        return np.array(list(b), np.float32)

    def materials(self):
        out = []
        for i in range(self.counts()["materials"]):
            m = GfxMaterial()
            self.L.gfxh_scene_get_material(self.h, C.c_uint32(i), C.byref(m))
            out.append(m)
        return out

    def geoms(self):
        out = []
        for i in range(self.counts()["geoms"]):
            vp, tp = C.c_void_p(), C.c_void_p()
            nv, nt, mat = C.c_uint32(), C.c_uint32(), C.c_uint32()
            self.L.gfxh_scene_get_geom(self.h, C.c_uint32(i), C.byref(vp), C.byref(nv), C.byref(tp), C.byref(nt), C.byref(mat))
            v = np.frombuffer((C.c_char * (44 * nv.value)).from_address(vp.value), dtype=VERTEX_DTYPE).copy()
            t = np.frombuffer((C.c_char * (12 * nt.value)).from_address(tp.value), dtype=np.uint32).reshape(-1, 3).copy()
            out.append((v, t, mat.value))
        return out

    def groups(self):
        out = []
        for i in range(self.counts()["groups"]):
            gp, n = C.c_void_p(), C.c_uint32()
            self.L.gfxh_scene_get_group(self.h, C.c_uint32(i), C.byref(gp), C.byref(n))
            out.append(np.frombuffer((C.c_char * (4 * n.value)).from_address(gp.value), dtype=np.uint32).copy())
        return out

    def instances(self):
        out = []
        for i in range(self.counts()["insts"]):
            g = C.c_uint32()
            x = (C.c_float * 12)()
            self.L.gfxh_scene_get_instance(self.h, C.c_uint32(i), C.byref(g), x)
            out.append((g.value, np.array(list(x), np.float32)))
        return out

    def upload(self, ctx):
        if self.L.gfxh_scene_upload(self.h, ctx.h):
            raise GfxError(self.L.gfxh_last_error().decode(errors="replace"))


def make_transform(scale=1.0, roll=0.0, pitch=0.0, yaw=0.0, pos=(0, 0, 0)):
    out = (C.c_float * 12)()
    lib().gfxh_make_transform(C.c_float(scale), C.c_float(roll), C.c_float(pitch), C.c_float(yaw), _f3(pos), out)
    return np.array(list(out), np.float32)


def make_camera(width, height, pos, roll=0.0, pitch=0.0, yaw=0.0, fov_y_deg=50.0):
    cam = GfxCamera()
    cam.aspect = float(width) / float(height)
    cam.fovY = np.float32(fov_y_deg * np.pi / 180)
    cam.position = _f3(pos)
    ori = (C.c_float * 9)()
    lib().gfxh_make_orientation(C.c_float(roll), C.c_float(pitch), C.c_float(yaw), ori)
    cam.orientation = ori
    return cam


def band_plan(height, band_begin, band_end, radius_rows, num_spatial_passes, max_motion_rows=0):
    plan = GfxhBandPlan()
    lib().gfxh_band_plan_compute(C.c_uint32(height), C.c_uint32(band_begin), C.c_uint32(band_end), C.c_uint32(radius_rows),
                                 C.c_uint32(num_spatial_passes), C.c_uint32(max_motion_rows), C.byref(plan))
    return plan


class GfxhSdrConfig(C.Structure):
    _fields_ = [("alphaForOverride", C.c_float), ("brightnessScale", C.c_float), ("applyToneMap", C.c_uint32),
                ("apply_sRGB_gammaCorrection", C.c_uint32), ("flipY", C.c_uint32)]


def sdr_config(brightness=1.0, tone_map=True, gamma=True, flip_y=False):
    return GfxhSdrConfig(-1.0, brightness, int(tone_map), int(gamma), int(flip_y))


def tonemap_sdr(rgba, width, height, cfg):
    """8-bit pixels (R | G << 8 | B << 16 | A << 24) of a float4 image: saveImage's tone map + sRGB gamma."""
    src = np.ascontiguousarray(rgba, np.float32).reshape(-1)
    out = np.zeros(width * height, np.uint32)
    lib().gfxh_tonemap_sdr(C.c_uint32(width), C.c_uint32(height), _p(src), C.byref(cfg), _p(out))
    return out.reshape(height, width)


def save_image_sdr(path, rgba, width, height, cfg):
    src = np.ascontiguousarray(rgba, np.float32).reshape(-1)
    if lib().gfxh_save_image_sdr(path.encode(), C.c_uint32(width), C.c_uint32(height), _p(src), C.byref(cfg)):
        raise GfxError(lib().gfxh_last_error().decode(errors="replace"))


def save_image_hdr(path, rgba, width, height, brightness=1.0, flip_y=False):
    src = np.ascontiguousarray(rgba, np.float32).reshape(-1)
    if lib().gfxh_save_image_hdr(path.encode(), C.c_uint32(width), C.c_uint32(height), C.c_float(brightness), _p(src), C.c_int(int(flip_y))):
        raise GfxError(lib().gfxh_last_error().decode(errors="replace"))


def env_make_sky(w, h, sun_elevation=35.0, sun_azimuth=40.0, sun_radiance=400.0):
    t = np.zeros((h * w, 4), np.float32)
    lib().gfxh_env_make_sky(C.c_uint32(w), C.c_uint32(h), C.c_float(sun_elevation), C.c_float(sun_azimuth), C.c_float(sun_radiance), _p(t))
    return t


def env_build_importance(texels, w, h):
    out = dict(rowPDF=np.zeros(h * w, np.float32), rowCDF=np.zeros(h * (w + 1), np.float32), rowIntegrals=np.zeros(h, np.float32),
               topPDF=np.zeros(h, np.float32), topCDF=np.zeros(h + 1, np.float32))
    integ = C.c_float()
    lib().gfxh_env_build_importance(_p(texels), C.c_uint32(w), C.c_uint32(h), _p(out["rowPDF"]), _p(out["rowCDF"]),
                                    _p(out["rowIntegrals"]), _p(out["topPDF"]), _p(out["topCDF"]), C.byref(integ))
    out["topIntegral"] = integ.value
    out["rowGuide"], out["topGuide"] = np.zeros(h * w, np.uint16), np.zeros(h, np.uint16)
    out["guidesUsable"] = bool(lib().gfxh_env_build_guides(_p(out["rowCDF"]), _p(out["topCDF"]), C.c_uint32(w), C.c_uint32(h),
                                                           _p(out["rowGuide"]), _p(out["topGuide"])))
    if out["guidesUsable"]:      # the interleaved rows (gfx_restir_static_params::envRowTable) and their sketches (envRowSketch)
        stride = (w + 1 + 3) & ~3
        out["rowTable"] = np.zeros(8 * h * stride, np.uint32)
        lib().gfxh_env_build_row_table(_p(texels), _p(out["rowPDF"]), _p(out["rowCDF"]), _p(out["rowGuide"]), C.c_uint32(w), C.c_uint32(h), _p(out["rowTable"]))
        lib().gfxh_env_build_row_sketch.restype = C.c_uint32
        nrec = C.c_uint32()
        lib().gfxh_env_build_row_sketch(_p(out["rowCDF"]), C.c_uint32(w), C.c_uint32(h), None, C.c_uint32(0), C.byref(nrec))
        out["rowSketch"] = np.zeros(36 * nrec.value, np.uint32)       # h row records + the child records of the cells that failed
        out["sketchCells"] = int(lib().gfxh_env_build_row_sketch(_p(out["rowCDF"]), C.c_uint32(w), C.c_uint32(h), _p(out["rowSketch"]), nrec, C.byref(nrec)))
        out["sketchRecords"] = nrec.value
    return out


def seed_rng_states(count, seed):
    out = np.zeros(count, np.uint64)
    lib().gfxh_seed_rng_states(_p(out), C.c_uint64(count), C.c_uint64(seed))
    return out


def spatial_neighbor_deltas():
    out = np.zeros((1024, 2), np.float32)
    lib().gfxh_spatial_neighbor_deltas(_p(out))
    return out


# ---------------------------------------------------------------- device context
class Context:
    """gfx_ctx: scene tables, BVH and kernels on one GPU."""

    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        if self.L.gfx_ctx_create(C.c_int(device), C.byref(h)):
            raise GfxError("gfx_ctx_create: " + self.L.gfx_last_error(None).decode(errors="replace"))
        self.h = h

    def close(self):
        if self.h:
            self.L.gfx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise GfxError(self.L.gfx_last_error(self.h).decode(errors="replace"))

    def instance_set_transform(self, inst_slot, xfm12, normal_matrix9=None):
        """InstanceController::update for one instance (previous matrix kept for the motion vectors).  Rebuild the
        acceleration structure (accel_build with the old handle) before the next trace / frame."""
        x = np.ascontiguousarray(xfm12, np.float32).reshape(12)
        if normal_matrix9 is None:
            self._check(self.L.gfx_instance_set_transform(self.h, C.c_uint32(inst_slot), _p(x)))
        else:
            nm = np.ascontiguousarray(normal_matrix9, np.float32).reshape(9)
            self._check(self.L.gfx_instance_set_transform_and_normal_matrix(self.h, C.c_uint32(inst_slot), _p(x), _p(nm)))

    def instance_set_dynamic(self, inst_slot, dynamic=True):
        self._check(self.L.gfx_instance_set_dynamic(self.h, C.c_uint32(inst_slot), C.c_int(int(dynamic))))

    def accel_build(self, stream=0, handle=0):
        hd = C.c_uint64(handle)
        self._check(self.L.gfx_accel_build(self.h, C.c_void_p(stream), C.byref(hd)))
        return hd.value

    def accel_set_max_leaf(self, n):
        self._check(self.L.gfx_accel_set_max_leaf(self.h, C.c_uint32(n)))

    def accel_stats(self, handle):
        s = (C.c_uint32 * 4)()
        self._check(self.L.gfx_accel_stats(self.h, C.c_uint64(handle), s))
        return dict(triangles=s[0], nodes=s[1], triRecords=s[2], maxDepth=s[3])

    def accel_tri_ids_ptr(self, handle):
        p, n = C.c_void_p(), C.c_uint32()
        self._check(self.L.gfx_accel_tri_ids(self.h, C.c_uint64(handle), C.byref(p), C.byref(n)))
        return p.value, n.value

    def texture_set(self, slot, texels, fmt):
        t = np.ascontiguousarray(texels)
        self._check(self.L.gfx_texture_set(self.h, C.c_uint32(slot), C.c_uint32(t.shape[1]), C.c_uint32(t.shape[0]), C.c_uint32(fmt), _p(t)))

    def texture_sample(self, slot, d_uv, n, d_out, gather=False, stream=0):
        self._check(self.L.gfx_texture_sample(self.h, C.c_void_p(stream), C.c_uint32(slot), C.c_void_p(d_uv), C.c_uint32(n), C.c_void_p(d_out),
                                              C.c_int(1 if gather else 0)))

    def lights_build_static(self, stream=0):
        self._check(self.L.gfx_lights_build_static(self.h, C.c_void_p(stream)))

    def lights_build_instances(self, stream=0, buffer_index=0):
        self._check(self.L.gfx_lights_build_instances(self.h, C.c_void_p(stream), C.c_uint32(buffer_index)))

    def lights_read(self, level, index=0):
        n, integ = C.c_uint32(), C.c_float()
        self._check(self.L.gfx_lights_read(self.h, C.c_uint32(level), C.c_uint32(index), None, None, C.c_uint32(0),
                                           C.byref(n), C.byref(integ)))
        w = np.zeros(n.value, np.float32)
        c = np.zeros(n.value, np.float32)
        self._check(self.L.gfx_lights_read(self.h, C.c_uint32(level), C.c_uint32(index), _p(w), _p(c), C.c_uint32(n.value),
                                           C.byref(n), C.byref(integ)))
        return w, c, integ.value

    def lights_table_info(self):
        """{usable, verified, records, cells, matrices, interior_cells} of the emitter interval table (emitter_spans.h)."""
        info = (C.c_uint32 * 8)()
        self._check(self.L.gfx_lights_table_info(self.h, info))
        return {"usable": int(info[0]), "verified": int(info[1]), "records": int(info[2]), "cells": int(info[3]),
                "matrices": int(info[4]), "interior_cells": int(info[5])}

    def trace(self, accel, mode, d_ray_org, d_ray_dir, num_rays, d_out, d_counters=0, stream=0, d_per_ray_items=0):
        if d_per_ray_items:
            self._check(self.L.gfx_trace_counted(self.h, C.c_void_p(stream), C.c_uint64(accel), C.c_int(mode), C.c_void_p(d_ray_org),
                                                 C.c_void_p(d_ray_dir), C.c_uint32(num_rays), C.c_void_p(d_out), C.c_void_p(d_counters),
                                                 C.c_void_p(d_per_ray_items)))
            return
        self._check(self.L.gfx_trace(self.h, C.c_void_p(stream), C.c_uint64(accel), C.c_int(mode), C.c_void_p(d_ray_org),
                                     C.c_void_p(d_ray_dir), C.c_uint32(num_rays), C.c_void_p(d_out), C.c_void_p(d_counters)))

    def restir_set_params(self, static_params, frame_params, cur_res_index, base_index, stream=0):
        self._check(self.L.gfx_restir_set_params(self.h, C.c_void_p(stream),
                                                 C.byref(static_params) if static_params is not None else None,
                                                 C.byref(frame_params) if frame_params is not None else None,
                                                 C.c_uint32(cur_res_index), C.c_uint32(base_index)))

    def restir_copy_to_linear(self, d_color, d_albedo, d_normal, d_motion, stream=0):
        self._check(self.L.gfx_restir_copy_to_linear(self.h, C.c_void_p(stream), C.c_void_p(d_color), C.c_void_p(d_albedo), C.c_void_p(d_normal), C.c_void_p(d_motion)))

    def visualize(self, d_linear, buffer_type, width, height, d_out, mv_offset=0.5, mv_scale=0.02, stream=0):
        self._check(self.L.gfx_visualize(self.h, C.c_void_p(stream), C.c_void_p(d_linear), C.c_int(buffer_type), C.c_float(mv_offset), C.c_float(mv_scale),
                                         C.c_uint32(width), C.c_uint32(height), C.c_void_p(d_out)))

    def restir_launch_rows(self, pass_id, width, height, row_begin, row_end, stream=0):
        self._check(self.L.gfx_restir_launch_rows(self.h, C.c_void_p(stream), C.c_int(pass_id), C.c_uint32(width), C.c_uint32(height),
                                                  C.c_uint32(row_begin), C.c_uint32(row_end)))

    def restir_launch_rows_gap(self, pass_id, width, height, row_begin, row_end, gap_begin, gap_end, stream=0):
        self._check(self.L.gfx_restir_launch_rows_gap(self.h, C.c_void_p(stream), C.c_int(pass_id), C.c_uint32(width), C.c_uint32(height),
                                                      C.c_uint32(row_begin), C.c_uint32(row_end), C.c_uint32(gap_begin), C.c_uint32(gap_end)))

    def restir_launch(self, pass_id, width, height, stream=0):
        self._check(self.L.gfx_restir_launch(self.h, C.c_void_p(stream), C.c_int(pass_id), C.c_uint32(width), C.c_uint32(height)))

    def nrc_inference_image(self, net, which):
        """(device pointer, bytes) of the packed inference image gfx_nrc_infer reads: 0 = MLP fragments, 1 = hash grid."""
        ptr, n = C.c_void_p(), C.c_uint64()
        self._check(self.L.gfx_nrc_inference_image(self.h, C.c_uint64(net), C.c_int(which), C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def nrc_set_render_params(self, params):
        self._check(self.L.gfx_nrc_set_render_params(self.h, C.byref(params)))

    def regir_set_params(self, params):
        self._check(self.L.gfx_regir_set_params(self.h, C.byref(params)))

    def pt_launch(self, pass_id, width, height, max_path_length, row_begin=0, row_end=0, stream=0):
        self._check(self.L.gfx_pt_launch(self.h, C.c_void_p(stream), C.c_int(pass_id), C.c_uint32(width), C.c_uint32(height),
                                         C.c_uint32(max_path_length), C.c_uint32(row_begin), C.c_uint32(row_end)))

    def read_device(self, dptr, nbytes):
        out = np.zeros(nbytes, np.uint8)
        self._check(self.L.gfx_read_device(self.h, C.c_void_p(dptr), _p(out), C.c_size_t(nbytes)))
        return out

    def timing_enable(self, on=True):
        self._check(self.L.gfx_timing_enable(self.h, C.c_int(1 if on else 0)))

    def timing_collect(self):
        cap = 64
        names = ((C.c_char * 48) * cap)()
        ms = (C.c_float * cap)()
        calls = (C.c_uint32 * cap)()
        n = C.c_uint32()
        self._check(self.L.gfx_timing_collect(self.h, names, ms, calls, C.c_uint32(cap), C.byref(n)))
        return {names[i].value.decode(): (ms[i], calls[i]) for i in range(min(n.value, cap))}

    def tunable_set(self, name, value):
        """Scheduling knob of this context ("pixel_map", "super_x", "super_y", "trace_blocks_per_cu", "trace_refill",
        "trace_batch", "temporal_hints", "pt_overlap", "candidate_split", "fuse_passes", "block_order"); changes no result."""
        self._check(self.L.gfx_tunable_set(self.h, name.encode(), C.c_int(int(value))))

    def stream_copy(self, d_dst, d_src, nbytes, stream=0):
        """Measurement utility: device-to-device copy with 16-byte accesses per lane (bench.py times it for roofline.peak_measured)."""
        self._check(self.L.gfx_stream_copy(self.h, C.c_void_p(d_dst), C.c_void_p(d_src), C.c_size_t(nbytes), C.c_void_p(stream)))

    def counters_enable(self, on=True):
        self._check(self.L.gfx_counters_enable(self.h, C.c_int(1 if on else 0)))

    def trace_diag_read(self, reset=True):
        c = (C.c_uint64 * 8)()
        self._check(self.L.gfx_trace_diag_read(self.h, c, C.c_int(1 if reset else 0)))
        return dict(iterations=c[0], itemLanes=c[1], drainIterations=c[2], drainItemLanes=c[3],
                    waveCycles=c[4], refillCycles=c[5], fetchCycles=c[6], processCycles=c[7])

    def pt_diag_read(self, reset=True):
        """gfx_pt_diag_read (tunable "pt_diag"): wave iterations, lanes with a ray, traversal steps, waves, refills of the one-kernel path tracers."""
        c = (C.c_uint64 * 8)()
        self._check(self.L.gfx_pt_diag_read(self.h, c, C.c_int(1 if reset else 0)))
        return dict(iterations=c[0], lanes=c[1], steps=c[2], waves=c[3], refills=c[4])

    def counters_read(self, reset=True):
        c = (C.c_uint64 * 8)()
        self._check(self.L.gfx_counters_read(self.h, c, C.c_int(1 if reset else 0)))
        names = ("nodeFetches", "triFetches", "rays", "spills")
        out = {n: c[i] + c[4 + i] for i, n in enumerate(names)}
        out["any"] = {n: c[i] for i, n in enumerate(names)}
        out["closest"] = {n: c[4 + i] for i, n in enumerate(names)}
        return out


NRC_TRIANGLE_WAVE, NRC_HASH_GRID = 0, 1


class NeuralRadianceCache:
    """gfx_nrc_*: the network behind NeuralRadianceCache (network_interface.h:14-28).  Inputs and
    outputs are device pointers to column-major fp32 [14, N] / [3, N] (torch tensors of shape [N, 14] /
    [N, 3], contiguous)."""

    def __init__(self, ctx, position_encoding=NRC_HASH_GRID, num_hidden_layers=2, learning_rate=1e-2):
        self.ctx, self.L = ctx, lib()
        h = C.c_uint64()
        ctx._check(self.L.gfx_nrc_create(ctx.h, C.c_int(position_encoding), C.c_uint32(num_hidden_layers),
                                         C.c_float(learning_rate), C.byref(h)))
        self.h = h.value

    def close(self):
        if self.h:
            self.L.gfx_nrc_destroy(self.ctx.h, C.c_uint64(self.h))
            self.h = 0

    def num_params(self):
        n = C.c_uint32()
        self.ctx._check(self.L.gfx_nrc_num_params(self.ctx.h, C.c_uint64(self.h), C.byref(n)))
        return n.value

    def set_params(self, params):
        p = np.ascontiguousarray(params, np.float32)
        self.ctx._check(self.L.gfx_nrc_set_params(self.ctx.h, C.c_uint64(self.h), _p(p), C.c_uint32(p.size)))

    def get_params(self, which=0):
        out = np.zeros(self.num_params(), np.float32)
        self.ctx._check(self.L.gfx_nrc_get_params(self.ctx.h, C.c_uint64(self.h), C.c_int(which), _p(out), C.c_uint32(out.size)))
        return out

    def infer(self, d_inputs, num_data, d_predictions, stream=0):
        self.ctx._check(self.L.gfx_nrc_infer(self.ctx.h, C.c_void_p(stream), C.c_uint64(self.h), C.c_void_p(d_inputs),
                                             C.c_uint32(num_data), C.c_void_p(d_predictions)))

    def train(self, d_inputs, d_targets, num_data, want_loss=False, stream=0):
        loss = C.c_float(0.0)
        self.ctx._check(self.L.gfx_nrc_train(self.ctx.h, C.c_void_p(stream), C.c_uint64(self.h), C.c_void_p(d_inputs),
                                             C.c_void_p(d_targets), C.c_uint32(num_data), C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None


class GfxhNrcConfig(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("positionEncoding", C.c_int), ("numHiddenLayers", C.c_uint32),
                ("learningRate", C.c_float), ("maxPathLength", C.c_uint32), ("radianceScale", C.c_float), ("train", C.c_uint32),
                ("enableAccumulation", C.c_uint32), ("camera", GfxCamera), ("sceneAabbMin", C.c_float * 3),
                ("sceneAabbMax", C.c_float * 3), ("rowBegin", C.c_uint32), ("rowEnd", C.c_uint32),
                ("neeSampler", C.c_uint32), ("regirGridDimension", C.c_uint32 * 3),
                ("regirLog2CandidatesPerLightSlot", C.c_uint32), ("regirLog2CandidatesPerCell", C.c_uint32),
                ("regirEnableTemporalReuse", C.c_uint32), ("regirEnableCellRandomization", C.c_uint32), ("enableBumpMapping", C.c_uint32)]


NRC_NEE_LIGHTS, NRC_NEE_REGIR, NRC_NEE_RESTIR = 0, 1, 2


class NrcRenderer:
    """gfxh_nrc: the headless frame loop of neural_radiance_caching_main.cpp over the C ABI."""

    def __init__(self, ctx, cfg):
        self.L, self.ctx, self.cfg = lib(), ctx, cfg
        self.L.gfxh_nrc_last_error.restype = C.c_char_p
        self.L.gfxh_nrc_beauty_buffer.restype = C.c_void_p
        self.L.gfxh_nrc_network.restype = C.c_uint64
        h = C.c_void_p()
        if self.L.gfxh_nrc_create(ctx.h, C.byref(cfg), C.byref(h)):
            raise GfxError("gfxh_nrc_create: " + self.L.gfxh_nrc_last_error().decode(errors="replace"))
        self.h = h

    @staticmethod
    def default_config(width, height, bounds):
        cfg = GfxhNrcConfig()
        lib().gfxh_nrc_default_config(C.byref(cfg), C.c_uint32(width), C.c_uint32(height))
        for k in range(3):
            cfg.sceneAabbMin[k] = float(bounds[k]); cfg.sceneAabbMax[k] = float(bounds[3 + k])
        return cfg

    def close(self):
        if self.h:
            self.L.gfxh_nrc_destroy(self.h)
            self.h = None

    def set_exchange(self, fn, rank):
        """Band renderer (cfg.rowBegin / rowEnd): fn(stream, desc: GfxhExchangeDesc) as for RestirRenderer.set_exchange."""
        def thunk(user, stream, desc):
            try:
                fn(stream, desc.contents)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        self._exchange_cb = EXCHANGE_FN(thunk)
        self.L.gfxh_nrc_set_exchange(self.h, self._exchange_cb, None, C.c_int(rank))

    def network(self):
        return self.L.gfxh_nrc_network(self.h)

    def rebuild_accel(self, stream=0):
        if self.L.gfxh_nrc_rebuild_accel(self.h, C.c_void_p(stream)):
            raise GfxError("gfxh_nrc_rebuild_accel: " + self.L.gfxh_nrc_last_error().decode(errors="replace"))

    def set_env(self, texels, w, h, power_coeff=1.0, rotation=0.0):
        t = np.ascontiguousarray(texels, np.float32)
        if self.L.gfxh_nrc_set_env(self.h, _p(t), C.c_uint32(w), C.c_uint32(h), C.c_float(power_coeff), C.c_float(rotation)):
            raise GfxError("gfxh_nrc_set_env: " + self.L.gfxh_nrc_last_error().decode(errors="replace"))

    def render_frame(self, stream=0, want_loss=False):
        loss = C.c_float(0.0)
        if self.L.gfxh_nrc_render_frame(self.h, C.c_void_p(stream), C.byref(loss) if want_loss else None):
            raise GfxError("gfxh_nrc_render_frame: " + self.L.gfxh_nrc_last_error().decode(errors="replace"))
        return loss.value if want_loss else None

    def outputs_consumed(self, stream=0):
        """gfxh_nrc_outputs_consumed: as RestirRenderer.outputs_consumed."""
        if self.L.gfxh_nrc_outputs_consumed(self.h, C.c_void_p(stream)):
            raise GfxError("gfxh_nrc_outputs_consumed: " + self.L.gfxh_nrc_last_error().decode(errors="replace"))

    def beauty_ptr(self):
        return self.L.gfxh_nrc_beauty_buffer(self.h)

    def stats(self):
        n, q = C.c_uint32(), C.c_uint32()
        t = (C.c_uint32 * 2)()
        self.L.gfxh_nrc_stats(self.h, C.byref(n), t, C.byref(q))
        return dict(numTrainingData=n.value, tileSize=(t[0], t[1]), numInferenceQueries=q.value)


class RestirRenderer:
    """gfxh_restir: the headless frame loop of restir_di_main.cpp over the C ABI."""

    def __init__(self, ctx, cfg):
        self.L = lib()
        self.ctx = ctx
        self.cfg = cfg
        h = C.c_void_p()
        if self.L.gfxh_restir_create(ctx.h, C.byref(cfg), C.byref(h)):
            raise GfxError("gfxh_restir_create: " + self.L.gfxh_restir_last_error().decode(errors="replace"))
        self.h = h

    @staticmethod
    def default_config(width, height, renderer=RENDERER_BIASED):
        cfg = GfxhRestirConfig()
        lib().gfxh_restir_default_config(C.byref(cfg), C.c_uint32(width), C.c_uint32(height), C.c_int(renderer))
        return cfg

    def close(self):
        if self.h:
            self.L.gfxh_restir_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def band_plan(self):
        plan = GfxhBandPlan()
        self.L.gfxh_restir_band_plan(self.h, C.byref(plan))
        return plan

    def set_exchange(self, fn, max_motion_rows=0):
        """Install the strip-exchange callback of a band renderer: fn(stream, desc: GfxhExchangeDesc) -> None / raises.
        A callback that knows its world size (tilesplit.StripExchange) gets the partition checked here, on every rank
        alike (gfxh_restir_check_partition)."""
        world = getattr(fn, "world", None)
        bands = getattr(fn, "custom_bands", None)
        if bands:
            check_bands(self.cfg, bands, max_motion_rows)
        elif world:
            check_partition(self.cfg, int(world), max_motion_rows)

        def thunk(user, stream, desc):
            try:
                fn(stream, desc.contents)
                return 0
            except Exception:       # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._exchange_cb = EXCHANGE_FN(thunk)     # keep the trampoline alive
        self.L.gfxh_restir_set_exchange(self.h, self._exchange_cb, None, C.c_uint32(max_motion_rows))

    def set_exchange_native(self, fn_ptr, user, max_motion_rows=0):
        """Install a native exchange callback (gfxh_rccl_exchange with its gfxh_rccl*): no Python between the passes."""
        self._exchange_cb = None
        self.L.gfxh_restir_set_exchange(self.h, C.cast(fn_ptr, C.c_void_p), user, C.c_uint32(max_motion_rows))

    def set_async_gather(self, enable=True):
        """The band gather on the renderer's gather stream underneath the next frame; finish_gather() before reading other ranks' rows."""
        if self.L.gfxh_restir_set_async_gather(self.h, C.c_int(int(enable))):
            raise GfxError("gfxh_restir_set_async_gather: " + self.L.gfxh_restir_last_error().decode(errors="replace"))

    def finish_gather(self, stream=0):
        if self.L.gfxh_restir_finish_gather(self.h, C.c_void_p(stream)):
            raise GfxError("gfxh_restir_finish_gather: " + self.L.gfxh_restir_last_error().decode(errors="replace"))

    def render_frame(self, stream=0):
        if self.L.gfxh_restir_render_frame(self.h, C.c_void_p(stream)):
            raise GfxError("gfxh_restir_render_frame: " + self.L.gfxh_restir_last_error().decode(errors="replace"))

    def outputs_consumed(self, stream=0):
        """gfxh_restir_outputs_consumed: `stream` has passed its reads of the albedo / normal accumulators of the last frame (a
        denoiser, a read-back); the next frame's pipelined G-buffer pass, which rewrites them, waits for this point."""
        if self.L.gfxh_restir_outputs_consumed(self.h, C.c_void_p(stream)):
            raise GfxError("gfxh_restir_outputs_consumed: " + self.L.gfxh_restir_last_error().decode(errors="replace"))

    def reset(self):
        self.L.gfxh_restir_reset(self.h)

    def set_env(self, texels, w, h, power_coeff=1.0, rotation=0.0):
        t = np.ascontiguousarray(texels, np.float32)
        if self.L.gfxh_restir_set_env(self.h, _p(t), C.c_uint32(w), C.c_uint32(h), C.c_float(power_coeff), C.c_float(rotation)):
            raise GfxError("gfxh_restir_set_env: " + self.L.gfxh_restir_last_error().decode(errors="replace"))

    def set_camera(self, cam):
        self.L.gfxh_restir_set_camera(self.h, C.byref(cam))

    def rebuild_accel(self, stream=0):
        """After Context.instance_set_transform: rebuild this renderer's BVH in place (Scene::updateASs)."""
        if self.L.gfxh_restir_rebuild_accel(self.h, C.c_void_p(stream)):
            raise GfxError("gfxh_restir_rebuild_accel: " + self.L.gfxh_restir_last_error().decode(errors="replace"))

    def beauty_ptr(self):
        return self.L.gfxh_restir_beauty_buffer(self.h)

    def accel(self):
        return self.L.gfxh_restir_accel(self.h)

    def params(self):
        s, f = GfxRestirStaticParams(), GfxRestirFrameParams()
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self.L.gfxh_restir_get_params(self.h, C.byref(s), C.byref(f), C.byref(a), C.byref(b), C.byref(c))
        return s, f, a.value, b.value, c.value
