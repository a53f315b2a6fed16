// tests/shim/network_interface_hip.cpp -- replaces neural_radiance_caching/network_interface.cu (:48-157): the pimpl
// class forwards to gfx_nrc_* (INTEGRATION.md section 5).  No tiny-cuda-nn.
#include "network_interface.h"
#include "hip_backend.h"

class NeuralRadianceCache::Priv { public: uint64_t net = 0; };
NeuralRadianceCache::NeuralRadianceCache() { m = new Priv(); }
NeuralRadianceCache::~NeuralRadianceCache() { delete m; }
void NeuralRadianceCache::initialize(PositionEncoding posEnc, uint32_t numHiddenLayers, float learningRate) {
    GFX_CHECK(gfx_nrc_create(g_gfx, posEnc == PositionEncoding::HashGrid ? GFX_NRC_HASH_GRID : GFX_NRC_TRIANGLE_WAVE,
                             numHiddenLayers, learningRate, &m->net));
}
void NeuralRadianceCache::finalize() { GFX_CHECK(gfx_nrc_destroy(g_gfx, m->net)); m->net = 0; }
void NeuralRadianceCache::infer(CUstream stream, float* inputData, uint32_t numData, float* predictionData) {
    GFX_CHECK(gfx_nrc_infer(g_gfx, stream, m->net, inputData, numData, predictionData));
}
void NeuralRadianceCache::train(CUstream stream, float* inputData, float* targetData, uint32_t numData, float* lossOnCPU) {
    GFX_CHECK(gfx_nrc_train(g_gfx, stream, m->net, inputData, targetData, numData, lossOnCPU));
}
