// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h header).
//
// orc_texture.h: tex2DLod / tex2Dgather as the reference uses them (common/common_device.cuh:143-147, 205-240),
// restated from the WRITTEN sampler contract of include/gfxexp.h (gfx_texture_set) -- bilinear, repeat wrap,
// mip level 0, 8 fraction bits in the filter weights, texels decoded before filtering -- independently of
// gfxexp_amd/csrc/texture.hip.h.  Parity unpinned against CUDA's texture unit (a hardware filter has no bit-level
// specification); pinned against a numpy statement of the same contract in tests/test_oracle_textures.py.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>
#include "orc_math.h"

namespace orc {

enum TexFormat { TexRGBA8_sRGB = 0, TexRGBA8_UNorm = 1, TexR8_UNorm = 2, TexRG8_UNorm = 3, TexRGBA32F = 4 };

struct Texel4 { float x, y, z, w; };

// sampler_sRGB's decode of an 8-bit channel: the formula of common/basic_types.h:5396-5402 in fp32
static inline float srgbDegamma8(uint32_t c) {
    const float v = static_cast<float>(c) / 255.0f;
    if (v <= 0.04045f) return v / 12.92f;
    return std::pow((v + 0.055f) / 1.055f, 2.4f);
}

struct Texture {
    uint32_t width = 0, height = 0, format = 0;
    std::vector<uint8_t> texels;
    mutable float lut[256];
    mutable bool lutReady = false;
    bool present() const { return width != 0; }

    Texel4 texel(int64_t i, int64_t j) const {
        const int64_t W = width, H = height;
        i %= W; if (i < 0) i += W;
        j %= H; if (j < 0) j += H;
        const size_t idx = static_cast<size_t>(j) * width + static_cast<size_t>(i);
        if (format == TexRGBA32F) {
            const float* p = reinterpret_cast<const float*>(texels.data()) + 4 * idx;
            return { p[0], p[1], p[2], p[3] };
        }
        if (format == TexR8_UNorm) return { static_cast<float>(texels[idx]) / 255.0f, 0.0f, 0.0f, 1.0f };
        if (format == TexRG8_UNorm) return { static_cast<float>(texels[2 * idx]) / 255.0f, static_cast<float>(texels[2 * idx + 1]) / 255.0f, 0.0f, 1.0f };
        const uint8_t* p = texels.data() + 4 * idx;
        if (format == TexRGBA8_sRGB) {
            if (!lutReady) { for (uint32_t c = 0; c < 256; ++c) lut[c] = srgbDegamma8(c); lutReady = true; }
            return { lut[p[0]], lut[p[1]], lut[p[2]], static_cast<float>(p[3]) / 255.0f };
        }
        return { static_cast<float>(p[0]) / 255.0f, static_cast<float>(p[1]) / 255.0f, static_cast<float>(p[2]) / 255.0f, static_cast<float>(p[3]) / 255.0f };
    }

    struct Footprint { int64_t i, j; float alpha, beta; };
    Footprint footprint(float u, float v) const {
        const float x = (u - std::floor(u)) * static_cast<float>(width) - 0.5f;
        const float y = (v - std::floor(v)) * static_cast<float>(height) - 0.5f;
        const float fx = std::floor(x), fy = std::floor(y);
        Footprint f;
        f.alpha = std::floor((x - fx) * 256.0f + 0.5f) / 256.0f;
        f.beta = std::floor((y - fy) * 256.0f + 0.5f) / 256.0f;
        f.i = static_cast<int64_t>(f2i(fx));
        f.j = static_cast<int64_t>(f2i(fy));
        return f;
    }

    // tex2DLod<float4>(tex, u, v, 0)
    Texel4 sample(float u, float v) const {
        const Footprint f = footprint(u, v);
        const Texel4 t00 = texel(f.i, f.j), t10 = texel(f.i + 1, f.j), t01 = texel(f.i, f.j + 1), t11 = texel(f.i + 1, f.j + 1);
        const float w00 = (1 - f.alpha) * (1 - f.beta), w10 = f.alpha * (1 - f.beta), w01 = (1 - f.alpha) * f.beta, w11 = f.alpha * f.beta;
        Texel4 r;
        r.x = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
        r.y = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
        r.z = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
        r.w = w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w;
        return r;
    }
    // tex2Dgather<float4>(tex, u, v, 0): component 0 of the footprint texels, CUDA's order
    Texel4 gatherR(float u, float v) const {
        const Footprint f = footprint(u, v);
        return { texel(f.i, f.j + 1).x, texel(f.i + 1, f.j + 1).x, texel(f.i + 1, f.j).x, texel(f.i, f.j).x };
    }
};

} // namespace orc
