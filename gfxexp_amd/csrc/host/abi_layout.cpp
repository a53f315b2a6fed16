// abi_layout.cpp -- the layout of every struct of include/gfxexp.h and include/gfxexp_host.h as THIS compiler lays it out
// (gfxh_abi_layout): language bindings that mirror the structs by hand (gfxexp_amd/api.py's ctypes classes, a cgo / JNI stub)
// check themselves against it instead of against another hand-written mirror.  A field renamed or removed in a header fails
// to compile here; a field added to a header changes sizeof and fails tests/test_abi_and_host.py by struct name until it is
// listed here and in the binding.
#include <cstddef>
#include <cstring>
#include "../../../include/gfxexp_host.h"

namespace {
struct Field { const char* strct; const char* field; uint64_t offset, size; };
#define S(T) { #T, nullptr, 0, sizeof(T) }
#define F(T, f) { #T, #f, offsetof(T, f), sizeof(((T*)nullptr)->f) }
const Field kFields[] = {
    S(gfx_vertex), F(gfx_vertex, position), F(gfx_vertex, normal), F(gfx_vertex, texCoord0Dir), F(gfx_vertex, texCoord),
    S(gfx_material), F(gfx_material, bsdfType), F(gfx_material, a), F(gfx_material, b), F(gfx_material, smoothness), F(gfx_material, emittance),
    F(gfx_material, hasEmittance), F(gfx_material, texA), F(gfx_material, texB), F(gfx_material, texSmoothness), F(gfx_material, texNormal),
    F(gfx_material, texEmittance), F(gfx_material, bumpMapType), F(gfx_material, pad),
    S(gfx_gbuffer0), F(gfx_gbuffer0, instSlot), F(gfx_gbuffer0, geomInstSlot), F(gfx_gbuffer0, primIndex), F(gfx_gbuffer0, qbcB), F(gfx_gbuffer0, qbcC),
    S(gfx_gbuffer1), F(gfx_gbuffer1, motionVector),
    S(gfx_gbuffer2), F(gfx_gbuffer2, positionInWorld), F(gfx_gbuffer2, qGeometricNormal),
    S(gfx_gbuffer3), F(gfx_gbuffer3, qShadingNormal), F(gfx_gbuffer3, qShadingTangent), F(gfx_gbuffer3, qTexCoord), F(gfx_gbuffer3, matSlot),
    S(gfx_reservoir_info), F(gfx_reservoir_info, recPDFEstimate), F(gfx_reservoir_info, targetDensity),
    S(gfx_camera), F(gfx_camera, aspect), F(gfx_camera, fovY), F(gfx_camera, position), F(gfx_camera, orientation),
    S(gfx_hit), F(gfx_hit, dist), F(gfx_hit, bcB), F(gfx_hit, bcC), F(gfx_hit, triIndex),
    S(gfx_tri_ids), F(gfx_tri_ids, instSlot), F(gfx_tri_ids, geomInstSlot), F(gfx_tri_ids, primIndex),
    S(gfx_restir_static_params), F(gfx_restir_static_params, imageSizeX), F(gfx_restir_static_params, imageSizeY), F(gfx_restir_static_params, rngBuffer),
    F(gfx_restir_static_params, gbuffer0), F(gfx_restir_static_params, gbuffer1), F(gfx_restir_static_params, gbuffer2), F(gfx_restir_static_params, gbuffer3),
    F(gfx_restir_static_params, reservoirBuffer), F(gfx_restir_static_params, reservoirInfoBuffer), F(gfx_restir_static_params, sampleVisibilityBuffer),
    F(gfx_restir_static_params, spatialNeighborDeltas), F(gfx_restir_static_params, beautyAccumBuffer), F(gfx_restir_static_params, albedoAccumBuffer),
    F(gfx_restir_static_params, normalAccumBuffer), F(gfx_restir_static_params, numTilesX), F(gfx_restir_static_params, numTilesY),
    F(gfx_restir_static_params, lightPreSamplingRngs), F(gfx_restir_static_params, preSampledLights), F(gfx_restir_static_params, envLightTexture),
    F(gfx_restir_static_params, envWidth), F(gfx_restir_static_params, envHeight), F(gfx_restir_static_params, envRowPDF), F(gfx_restir_static_params, envRowCDF),
    F(gfx_restir_static_params, envRowIntegrals), F(gfx_restir_static_params, envTopPDF), F(gfx_restir_static_params, envTopCDF),
    F(gfx_restir_static_params, envTopIntegral), F(gfx_restir_static_params, envRowGuide), F(gfx_restir_static_params, envTopGuide), F(gfx_restir_static_params, envRowTable), F(gfx_restir_static_params, envRowSketch),
    S(gfx_restir_frame_params), F(gfx_restir_frame_params, travHandle), F(gfx_restir_frame_params, numAccumFrames), F(gfx_restir_frame_params, frameIndex),
    F(gfx_restir_frame_params, camera), F(gfx_restir_frame_params, prevCamera), F(gfx_restir_frame_params, envLightPowerCoeff), F(gfx_restir_frame_params, envLightRotation),
    F(gfx_restir_frame_params, spatialNeighborRadius), F(gfx_restir_frame_params, radiusThresholdForSpatialVisReuse), F(gfx_restir_frame_params, log2NumCandidateSamples),
    F(gfx_restir_frame_params, numSpatialNeighbors), F(gfx_restir_frame_params, useLowDiscrepancyNeighbors), F(gfx_restir_frame_params, reuseVisibility),
    F(gfx_restir_frame_params, reuseVisibilityForTemporal), F(gfx_restir_frame_params, reuseVisibilityForSpatiotemporal), F(gfx_restir_frame_params, enableTemporalReuse),
    F(gfx_restir_frame_params, enableSpatialReuse), F(gfx_restir_frame_params, useUnbiasedEstimator), F(gfx_restir_frame_params, bufferIndex),
    F(gfx_restir_frame_params, resetFlowBuffer), F(gfx_restir_frame_params, enableJittering), F(gfx_restir_frame_params, enableEnvLight),
    F(gfx_restir_frame_params, enableBumpMapping), F(gfx_restir_frame_params, useSolidAngleSampling),
    S(gfx_regir_params), F(gfx_regir_params, reservoirs), F(gfx_regir_params, reservoirInfos), F(gfx_regir_params, lightSlotRngs), F(gfx_regir_params, perCellNumAccesses),
    F(gfx_regir_params, lastAccessFrameIndices), F(gfx_regir_params, numActiveCells), F(gfx_regir_params, gridOrigin), F(gfx_regir_params, gridCellSize),
    F(gfx_regir_params, gridDimension), F(gfx_regir_params, log2NumCandidatesPerLightSlot), F(gfx_regir_params, log2NumCandidatesPerCell), F(gfx_regir_params, enableCellRandomization),
    S(gfx_nrc_params), F(gfx_nrc_params, sceneAabbMin), F(gfx_nrc_params, sceneAabbMax), F(gfx_nrc_params, maxNumTrainingSuffixes), F(gfx_nrc_params, numTrainingData),
    F(gfx_nrc_params, tileSize), F(gfx_nrc_params, targetMinMax), F(gfx_nrc_params, targetAvg), F(gfx_nrc_params, offsetToSelectUnbiasedTile),
    F(gfx_nrc_params, offsetToSelectTrainingPath), F(gfx_nrc_params, inferenceRadianceQueryBuffer), F(gfx_nrc_params, inferenceTerminalInfoBuffer),
    F(gfx_nrc_params, inferredRadianceBuffer), F(gfx_nrc_params, perFrameContributionBuffer), F(gfx_nrc_params, trainRadianceQueryBuffer), F(gfx_nrc_params, trainTargetBuffer),
    F(gfx_nrc_params, trainVertexInfoBuffer), F(gfx_nrc_params, trainSuffixTerminalInfoBuffer), F(gfx_nrc_params, dataShufflerBuffer), F(gfx_nrc_params, radianceScale),
    F(gfx_nrc_params, preprocessOffsetToSelectUnbiasedTile), F(gfx_nrc_params, preprocessOffsetToSelectTrainingPath), F(gfx_nrc_params, isNewSequence),
    S(gfxh_street_params), F(gfxh_street_params, seed), F(gfxh_street_params, groundTess), F(gfxh_street_params, numBuildings), F(gfxh_street_params, facadeTess),
    F(gfxh_street_params, numProps), F(gfxh_street_params, propSubdiv), F(gfxh_street_params, numLamps), F(gfxh_street_params, numSigns), F(gfxh_street_params, extent),
    F(gfxh_street_params, lampEmittance), F(gfxh_street_params, signEmittance), F(gfxh_street_params, textured), F(gfxh_street_params, numTrees),
    F(gfxh_street_params, leavesPerTree), F(gfxh_street_params, numWires), F(gfxh_street_params, numRailings),
    S(gfxh_restir_config), F(gfxh_restir_config, width), F(gfxh_restir_config, height), F(gfxh_restir_config, renderer), F(gfxh_restir_config, log2NumCandidateSamples),
    F(gfxh_restir_config, enableTemporalReuse), F(gfxh_restir_config, enableSpatialReuse), F(gfxh_restir_config, numSpatialReusePasses), F(gfxh_restir_config, numSpatialNeighbors),
    F(gfxh_restir_config, spatialNeighborRadius), F(gfxh_restir_config, useLowDiscrepancyNeighbors), F(gfxh_restir_config, reuseVisibility), F(gfxh_restir_config, enableAccumulation),
    F(gfxh_restir_config, log2MaxNumAccums), F(gfxh_restir_config, camera), F(gfxh_restir_config, rowBegin), F(gfxh_restir_config, rowEnd), F(gfxh_restir_config, maxPathLength),
    F(gfxh_restir_config, enableJittering), F(gfxh_restir_config, regirAabbMin), F(gfxh_restir_config, regirAabbMax), F(gfxh_restir_config, regirGridDimension),
    F(gfxh_restir_config, regirLog2CandidatesPerLightSlot), F(gfxh_restir_config, regirLog2CandidatesPerCell), F(gfxh_restir_config, regirEnableTemporalReuse),
    F(gfxh_restir_config, regirEnableCellRandomization), F(gfxh_restir_config, enableBumpMapping),
    S(gfxh_band_plan), F(gfxh_band_plan, bandBegin), F(gfxh_band_plan, bandEnd), F(gfxh_band_plan, haloRows), F(gfxh_band_plan, gbufferRows), F(gfxh_band_plan, initialRows),
    F(gfxh_band_plan, spatialRows), F(gfxh_band_plan, shadingRows), F(gfxh_band_plan, recvAbove), F(gfxh_band_plan, sendAbove), F(gfxh_band_plan, recvBelow), F(gfxh_band_plan, sendBelow),
    S(gfxh_exchange_buffer), F(gfxh_exchange_buffer, base), F(gfxh_exchange_buffer, bytesPerPixel), F(gfxh_exchange_buffer, numPlanes), F(gfxh_exchange_buffer, planeStride),
    S(gfxh_exchange_desc), F(gfxh_exchange_desc, kind), F(gfxh_exchange_desc, stage), F(gfxh_exchange_desc, lane), F(gfxh_exchange_desc, reserved), F(gfxh_exchange_desc, width),
    F(gfxh_exchange_desc, height), F(gfxh_exchange_desc, bandBegin), F(gfxh_exchange_desc, bandEnd), F(gfxh_exchange_desc, sendAbove), F(gfxh_exchange_desc, recvAbove),
    F(gfxh_exchange_desc, sendBelow), F(gfxh_exchange_desc, recvBelow), F(gfxh_exchange_desc, numBuffers), F(gfxh_exchange_desc, buffers), F(gfxh_exchange_desc, counters),
    F(gfxh_exchange_desc, numCounters),
    S(gfxh_frame_step), F(gfxh_frame_step, op), F(gfxh_frame_step, pass), F(gfxh_frame_step, rowBegin), F(gfxh_frame_step, rowEnd), F(gfxh_frame_step, currentReservoirIndex),
    F(gfxh_frame_step, spatialNeighborBaseIndex), F(gfxh_frame_step, exchangeRows), F(gfxh_frame_step, buffers), F(gfxh_frame_step, reservoirIndex), F(gfxh_frame_step, lane), F(gfxh_frame_step, gapBegin), F(gfxh_frame_step, gapEnd),
    S(gfxh_nrc_config), F(gfxh_nrc_config, width), F(gfxh_nrc_config, height), F(gfxh_nrc_config, positionEncoding), F(gfxh_nrc_config, numHiddenLayers), F(gfxh_nrc_config, learningRate),
    F(gfxh_nrc_config, maxPathLength), F(gfxh_nrc_config, radianceScale), F(gfxh_nrc_config, train), F(gfxh_nrc_config, enableAccumulation), F(gfxh_nrc_config, camera),
    F(gfxh_nrc_config, sceneAabbMin), F(gfxh_nrc_config, sceneAabbMax), F(gfxh_nrc_config, rowBegin), F(gfxh_nrc_config, rowEnd), F(gfxh_nrc_config, neeSampler),
    F(gfxh_nrc_config, regirGridDimension), F(gfxh_nrc_config, regirLog2CandidatesPerLightSlot), F(gfxh_nrc_config, regirLog2CandidatesPerCell),
    F(gfxh_nrc_config, regirEnableTemporalReuse), F(gfxh_nrc_config, regirEnableCellRandomization), F(gfxh_nrc_config, enableBumpMapping),
    S(gfxh_sdr_config), F(gfxh_sdr_config, alphaForOverride), F(gfxh_sdr_config, brightnessScale), F(gfxh_sdr_config, applyToneMap),
    F(gfxh_sdr_config, apply_sRGB_gammaCorrection), F(gfxh_sdr_config, flipY),
};
#undef S
#undef F
}

extern "C" {

uint32_t gfxh_abi_num_entries(void) { return static_cast<uint32_t>(sizeof(kFields) / sizeof(kFields[0])); }

int gfxh_abi_entry(uint32_t index, const char** structName, const char** fieldName, uint64_t* offset, uint64_t* size) {
    if (index >= gfxh_abi_num_entries()) return 1;
    const Field& f = kFields[index];
    *structName = f.strct; *fieldName = f.field; *offset = f.offset; *size = f.size;
    return 0;
}

int gfxh_abi_layout(const char* structName, const char* fieldName, uint64_t* offset, uint64_t* size) {
    for (const Field& f : kFields) {
        if (std::strcmp(f.strct, structName) != 0) continue;
        if ((fieldName == nullptr) != (f.field == nullptr)) continue;
        if (fieldName && std::strcmp(f.field, fieldName) != 0) continue;
        if (offset) *offset = f.offset;
        if (size) *size = f.size;
        return 0;
    }
    return 1;
}

} // extern "C"
