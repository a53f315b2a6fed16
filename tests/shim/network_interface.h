// tests/shim/network_interface.h -- a class with the member signatures of the reference's NeuralRadianceCache
// (neural_radiance_caching/network_interface.h:14-28), declared here so that the replacement translation unit
// network_interface_hip.cpp is checked by a compiler against exactly that interface.  CUstream becomes hipStream_t.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

using CUstream = hipStream_t;

enum class PositionEncoding {
    TriangleWave,
    HashGrid,
};

class NeuralRadianceCache {
    class Priv;
    Priv* m = nullptr;

public:
    NeuralRadianceCache();
    ~NeuralRadianceCache();

    void initialize(PositionEncoding posEnc, uint32_t numHiddenLayers, float learningRate);
    void finalize();

    void infer(CUstream stream, float* inputData, uint32_t numData, float* predictionData);
    void train(CUstream stream, float* inputData, float* targetData, uint32_t numData, float* lossOnCPU = nullptr);
};
