// tests/shim/shim_main.cpp -- a miniature of the reference's host program written against the shim
// (hip_backend.h, network_interface.h): scene creation through the C ABI calls of INTEGRATION.md section 1, the
// frame loop of restir_di_main.cpp:2311-2493 with its index bookkeeping, and the NeuralRadianceCache class.
// Built by tests/shim/build.py with hipcc against include/ and libgfxexp.so; run by tests/test_gpu_shim.py.
// Prints "shim ok ..." on success; any C-ABI failure surfaces as std::runtime_error like in the reference
// (main catches once and returns -1, restir_di_main.cpp:2679-2682).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "gfxexp_host.h"
#include "hip_backend.h"
#include "network_interface.h"

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) throw std::runtime_error(std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

template <typename T>
static T* device_alloc(size_t count, bool zero = true) {
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, sizeof(T) * count));
    if (zero) HIP_CHECK(hipMemset(p, 0, sizeof(T) * count));
    return static_cast<T*>(p);
}

static uint32_t add_quad(uint32_t matSlot, const float p[4][3], const float n[3]) {
    gfx_vertex v[4];
    const float uv[4][2] = { { 0, 0 }, { 1, 0 }, { 1, 1 }, { 0, 1 } };
    for (int i = 0; i < 4; ++i) {
        std::memcpy(v[i].position, p[i], 12); std::memcpy(v[i].normal, n, 12);
        v[i].texCoord0Dir[0] = 1; v[i].texCoord0Dir[1] = 0; v[i].texCoord0Dir[2] = 0;
        v[i].texCoord[0] = uv[i][0]; v[i].texCoord[1] = uv[i][1];
    }
    const uint32_t tris[6] = { 0, 1, 2, 0, 2, 3 };
    uint32_t slot;
    GFX_CHECK(gfx_geom_create(g_gfx, v, sizeof(gfx_vertex), 4, tris, 2, matSlot, &slot));
    return slot;
}

int main() try {
    hipbackend::GPUEnvironment gpuEnv;
    gpuEnv.initialize();
    hipStream_t stream;
    HIP_CHECK(hipStreamCreate(&stream));

    // ---- scene (createDiffuseAndSpecularMaterial / createGeometryInstance / createGeometryGroup / createInstance)
    gfx_material floorMat; std::memset(&floorMat, 0, sizeof(floorMat));
    floorMat.bsdfType = GFX_BSDF_DIFFUSE_AND_SPECULAR;
    floorMat.a[0] = floorMat.a[1] = floorMat.a[2] = 0.4f; floorMat.b[0] = floorMat.b[1] = floorMat.b[2] = 0.04f; floorMat.smoothness = 0.3f;
    gfx_material lightMat = floorMat;
    lightMat.a[0] = lightMat.a[1] = lightMat.a[2] = 0.01f;
    lightMat.emittance[0] = 30; lightMat.emittance[1] = 28; lightMat.emittance[2] = 24; lightMat.hasEmittance = 1;
    GFX_CHECK(gfx_material_set(g_gfx, 0, &floorMat));
    GFX_CHECK(gfx_material_set(g_gfx, 1, &lightMat));
    const float floorP[4][3] = { { -5, 0, 5 }, { 5, 0, 5 }, { 5, 0, -5 }, { -5, 0, -5 } }, up[3] = { 0, 1, 0 };
    const float lightP[4][3] = { { -1, 3, -1 }, { 1, 3, -1 }, { 1, 3, 1 }, { -1, 3, 1 } }, down[3] = { 0, -1, 0 };
    const uint32_t gFloor = add_quad(0, floorP, up), gLight = add_quad(1, lightP, down);
    uint32_t grpFloor, grpLight, inst;
    GFX_CHECK(gfx_group_create(g_gfx, &gFloor, 1, &grpFloor));
    GFX_CHECK(gfx_group_create(g_gfx, &gLight, 1, &grpLight));
    const float ident[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    GFX_CHECK(gfx_instance_create(g_gfx, grpFloor, ident, &inst));
    GFX_CHECK(gfx_instance_create(g_gfx, grpLight, ident, &inst));
    const uint64_t travHandle = hipbackend::updateASs(stream);
    GFX_CHECK(gfx_lights_build_static(g_gfx, stream));

    // ---- per-pixel buffers owned by the host program (restir_di_main.cpp:1233-1325)
    const uint32_t W = 64, H = 48, N = W * H;
    gfx_restir_static_params sp; std::memset(&sp, 0, sizeof(sp));
    sp.imageSizeX = W; sp.imageSizeY = H;
    std::vector<uint64_t> rngStates(N);
    gfxh_seed_rng_states(rngStates.data(), N, 591842031321323413ull);
    uint64_t* dRng = device_alloc<uint64_t>(N, false);
    HIP_CHECK(hipMemcpy(dRng, rngStates.data(), 8ull * N, hipMemcpyHostToDevice));
    sp.rngBuffer = dRng;
    for (int k = 0; k < 2; ++k) {
        sp.gbuffer0[k] = device_alloc<gfx_gbuffer0>(N); sp.gbuffer1[k] = device_alloc<gfx_gbuffer1>(N);
        sp.gbuffer2[k] = device_alloc<gfx_gbuffer2>(N); sp.gbuffer3[k] = device_alloc<gfx_gbuffer3>(N);
        sp.reservoirBuffer[k] = device_alloc<float>(12ull * N); sp.reservoirInfoBuffer[k] = device_alloc<gfx_reservoir_info>(N);
        sp.sampleVisibilityBuffer[k] = device_alloc<uint32_t>(N);
    }
    std::vector<float> deltas(2 * 1024);
    gfxh_spatial_neighbor_deltas(deltas.data());
    float* dDeltas = device_alloc<float>(2 * 1024, false);
    HIP_CHECK(hipMemcpy(dDeltas, deltas.data(), 4ull * deltas.size(), hipMemcpyHostToDevice));
    sp.spatialNeighborDeltas = dDeltas;
    sp.beautyAccumBuffer = device_alloc<float>(4ull * N); sp.albedoAccumBuffer = device_alloc<float>(4ull * N); sp.normalAccumBuffer = device_alloc<float>(4ull * N);

    gfx_restir_frame_params fp; std::memset(&fp, 0, sizeof(fp));
    fp.travHandle = travHandle;
    fp.camera.aspect = static_cast<float>(W) / H; fp.camera.fovY = 50.0f * 3.14159265f / 180.0f;
    fp.camera.position[0] = 0; fp.camera.position[1] = 2.5f; fp.camera.position[2] = 7;
    float ori[9]; gfxh_make_orientation(0.0f, 15.0f, 180.0f, ori);
    std::memcpy(fp.camera.orientation, ori, sizeof(ori));
    fp.envLightPowerCoeff = 1; fp.spatialNeighborRadius = 20; fp.radiusThresholdForSpatialVisReuse = 10;
    fp.log2NumCandidateSamples = 5; fp.numSpatialNeighbors = 5; fp.useLowDiscrepancyNeighbors = 1;
    fp.reuseVisibility = 1; fp.reuseVisibilityForTemporal = 1; fp.enableTemporalReuse = 1; fp.enableSpatialReuse = 1;

    // ---- the frame loop (restir_di_main.cpp:2311-2493), original biased renderer
    hipbackend::Pipeline restir;
    uint32_t lastReservoirIndex = 1, lastSpatialNeighborBaseIndex = 0;
    for (uint32_t frameIndex = 0; frameIndex < 3; ++frameIndex) {
        const bool newSequence = frameIndex == 0;
        fp.prevCamera = fp.camera;
        fp.frameIndex = frameIndex; fp.bufferIndex = frameIndex % 2; fp.resetFlowBuffer = newSequence; fp.numAccumFrames = 0;
        GFX_CHECK(gfx_lights_build_instances(g_gfx, stream, fp.bufferIndex));
        uint32_t currentReservoirIndex = (lastReservoirIndex + 1) % 2;
        uint32_t baseIndex = lastSpatialNeighborBaseIndex;
        restir.setEntryPoint(GFX_RESTIR_SETUP_GBUFFERS);
        restir.launch(stream, sp, fp, currentReservoirIndex, baseIndex, W, H);
        restir.setEntryPoint(newSequence ? GFX_RESTIR_INITIAL_RIS : GFX_RESTIR_INITIAL_AND_TEMPORAL_BIASED);
        restir.launch(stream, sp, fp, currentReservoirIndex, baseIndex, W, H);
        for (int pass = 0; pass < 2; ++pass) {
            restir.setEntryPoint(GFX_RESTIR_SPATIAL_BIASED);
            restir.launch(stream, sp, fp, currentReservoirIndex, baseIndex, W, H);
            baseIndex += fp.numSpatialNeighbors;
            currentReservoirIndex = (currentReservoirIndex + 1) % 2;
        }
        lastSpatialNeighborBaseIndex = baseIndex;
        restir.setEntryPoint(GFX_RESTIR_SHADING);
        restir.launch(stream, sp, fp, currentReservoirIndex, baseIndex, W, H);
        lastReservoirIndex = currentReservoirIndex;
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<float> beauty(4ull * N);
    HIP_CHECK(hipMemcpy(beauty.data(), sp.beautyAccumBuffer, 16ull * N, hipMemcpyDeviceToHost));
    double mean = 0; bool finite = true;
    for (uint32_t i = 0; i < N; ++i) for (int c = 0; c < 3; ++c) { mean += beauty[4 * i + c]; finite = finite && std::isfinite(beauty[4 * i + c]); }
    mean /= 3.0 * N;
    if (!finite || !(mean > 1e-3)) { std::printf("shim FAILED: beauty mean %g finite %d\n", mean, finite ? 1 : 0); return 1; }

    // ---- errors surface as exceptions (utils/cuda_util.cpp:58-69)
    bool threw = false;
    try { restir.setEntryPoint(12345); restir.launch(stream, sp, fp, 0, 0, W, H); }
    catch (const std::runtime_error&) { threw = true; }
    if (!threw) { std::printf("shim FAILED: an invalid entry point did not throw\n"); return 1; }

    // ---- NeuralRadianceCache (network_interface.h:14-28)
    NeuralRadianceCache nrc;
    nrc.initialize(PositionEncoding::HashGrid, 2, 1e-2f);
    const uint32_t numData = 256;
    std::vector<float> in(14ull * numData), tgt(3ull * numData);
    uint32_t lcg = 12345u;
    for (float& v : in) { lcg = lcg * 1664525u + 1013904223u; v = (lcg >> 8) * (1.0f / 16777216.0f); }
    for (float& v : tgt) { lcg = lcg * 1664525u + 1013904223u; v = (lcg >> 8) * (1.0f / 16777216.0f); }
    float* dIn = device_alloc<float>(in.size(), false); float* dTgt = device_alloc<float>(tgt.size(), false); float* dOut = device_alloc<float>(3ull * numData);
    HIP_CHECK(hipMemcpy(dIn, in.data(), 4ull * in.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dTgt, tgt.data(), 4ull * tgt.size(), hipMemcpyHostToDevice));
    float loss0 = 0, loss1 = 0;
    nrc.train(stream, dIn, dTgt, numData, &loss0);
    for (int k = 0; k < 30; ++k) nrc.train(stream, dIn, dTgt, numData);
    nrc.train(stream, dIn, dTgt, numData, &loss1);
    nrc.infer(stream, dIn, numData, dOut);
    HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<float> out(3ull * numData);
    HIP_CHECK(hipMemcpy(out.data(), dOut, 4ull * out.size(), hipMemcpyDeviceToHost));
    bool ok = std::isfinite(loss0) && std::isfinite(loss1) && loss1 < loss0;
    for (float v : out) ok = ok && std::isfinite(v);
    nrc.finalize();
    if (!ok) { std::printf("shim FAILED: NRC loss %g -> %g\n", loss0, loss1); return 1; }

    gpuEnv.finalize();
    std::printf("shim ok beauty_mean %.6f nrc_loss %.5f -> %.5f\n", mean, loss0, loss1);
    return 0;
}
catch (const std::exception& ex) {
    std::printf("shim FAILED: %s\n", ex.what());
    return -1;
}
