// tests/shim/hip_backend.h -- the file INTEGRATION.md section 2 tells a maintainer of the reference to add next to
// restir_di_main.cpp, as a real translation unit: the reference's class / call shapes forwarding to the C ABI.
// Compiled against include/gfxexp.h only (no torch, no package code) and driven by tests/shim/shim_main.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdexcept>
#include <string>
#include "gfxexp.h"

inline gfx_ctx* g_gfx = nullptr;
// CUDADRV_CHECK -> throw std::runtime_error (utils/cuda_util.cpp:58-69)
#define GFX_CHECK(call) do { if (call) throw std::runtime_error(std::string(#call) + ": " + gfx_last_error(g_gfx)); } while (0)

namespace hipbackend {

struct GPUEnvironment {                         // replaces struct GPUEnvironment (restir_di_main.cpp:98-450)
    void initialize() { if (gfx_ctx_create(0, &g_gfx)) throw std::runtime_error(gfx_last_error(nullptr)); }
    void finalize() { gfx_ctx_destroy(g_gfx); g_gfx = nullptr; }
};

// Pipeline::setRayGenerationProgram(entryPoints[e]) + launch (restir_di_main.cpp:117-119, 2366-2420)
struct Pipeline {
    int entryPoint = GFX_RESTIR_SETUP_GBUFFERS;
    void setEntryPoint(int e) { entryPoint = e; }
    void launch(hipStream_t stream, const gfx_restir_static_params& s, const gfx_restir_frame_params& f,
                uint32_t currentReservoirIndex, uint32_t spatialNeighborBaseIndex, uint32_t w, uint32_t h) {
        GFX_CHECK(gfx_restir_set_params(g_gfx, stream, &s, &f, currentReservoirIndex, spatialNeighborBaseIndex));
        GFX_CHECK(gfx_restir_launch(g_gfx, stream, entryPoint, w, h));
    }
};

// Scene::updateASs (common_host.h:1027-1100): returns the traversable handle that goes into perFramePlp.travHandle
inline uint64_t updateASs(hipStream_t stream, uint64_t handle = 0) {
    GFX_CHECK(gfx_accel_build(g_gfx, stream, &handle));
    return handle;
}

} // namespace hipbackend
