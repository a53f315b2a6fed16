// texture.hip.h -- software tex2DLod: the CUDA texture unit of the reference, as written arithmetic.
//
// The reference reads every material value through tex2DLod at mip level 0 with bilinear, repeat-wrapped
// samplers (common/common_device.cuh:143-147; sampler setup common/common_host.cpp:1462-1481; texture upload
// :1163-1244).  A hardware texture unit is not bit-specified (9-bit filter weights, an sRGB decode table), so this
// build fixes ONE definition -- the contract of include/gfxexp.h gfx_texture_set -- and the CPU oracle restates it
// independently (oracle/orc_texture.h):
//     x = (u - floor(u)) * W - 0.5          i = floor(x)      alpha = floor((x - i) * 256 + 0.5) / 256
//     y = (v - floor(v)) * H - 0.5          j = floor(y)      beta  = floor((y - j) * 256 + 0.5) / 256
//     T = ((1 - alpha) (1 - beta)) T[i, j] + (alpha (1 - beta)) T[i+1, j] + ((1 - alpha) beta) T[i, j+1] + (alpha beta) T[i+1, j+1]
// indices wrapped modulo W / H; fp32, this operation order, no contraction.  The 8 fraction bits of the weights are
// what the CUDA programming guide documents for linear filtering; the rounding of that quantisation is this build's
// choice.  8-bit texels are decoded per texel BEFORE filtering: c / 255, or (sRGB formats) the table
// srgbLut[c] = degamma(c / 255) computed once on the host (basic_types.h:5396-5402 states the formula).
#pragma once
#include "device_types.h"
#include "gm_math.hip.h"

namespace gfx {

struct TexelQuad { uint32_t i0, i1, j0, j1; float w00, w10, w01, w11; };

// Footprint and weights of the bilinear filter at (u, v).
GFX_DEV TexelQuad tex_footprint(uint32_t W, uint32_t H, float u, float v) {
    const float x = (u - floorf(u)) * static_cast<float>(W) - 0.5f;
    const float y = (v - floorf(v)) * static_cast<float>(H) - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float alpha = floorf((x - fx) * 256.0f + 0.5f) / 256.0f;
    const float beta = floorf((y - fy) * 256.0f + 0.5f) / 256.0f;
    // x lies in [-0.5, W - 0.5] for every finite u (a NaN converts to 0), so i is in [-1, W - 1] and i + 1 in [0, W]:
    // the modulo of the contract is one conditional add / subtract
    const int32_t ix = f2i_sat(fx), iy = f2i_sat(fy);
    const int32_t w = static_cast<int32_t>(W), h = static_cast<int32_t>(H);
    auto wrap = [](int32_t k, int32_t n) { const int32_t r = k < 0 ? k + n : (k >= n ? k - n : k); return static_cast<uint32_t>(r < 0 ? 0 : (r >= n ? n - 1 : r)); };
    TexelQuad q;
    q.i0 = wrap(ix, w); q.i1 = wrap(ix + 1, w);
    q.j0 = wrap(iy, h); q.j1 = wrap(iy + 1, h);
    q.w00 = (1 - alpha) * (1 - beta);
    q.w10 = alpha * (1 - beta);
    q.w01 = (1 - alpha) * beta;
    q.w11 = alpha * beta;
    return q;
}

// One texel as four floats (missing channels: 0, 0, 1 like a CUDA array read of fewer channels).
GFX_DEV float4 tex_texel(const DevScene& sc, const DevTexture& t, uint32_t i, uint32_t j) {
    const size_t idx = static_cast<size_t>(j) * t.width + i;
    const uint32_t* pool = sc.texelPool + t.offset;
    if (t.format == GFX_TEX_RGBA32F) return reinterpret_cast<const float4*>(pool)[idx];   // the common case of emittance maps first
    switch (t.format) {
    case GFX_TEX_RGBA8_SRGB: {
        const uint32_t c = pool[idx];
        return make_float4(sc.srgbLut[c & 0xFFu], sc.srgbLut[(c >> 8) & 0xFFu], sc.srgbLut[(c >> 16) & 0xFFu], static_cast<float>(c >> 24) / 255.0f);
    }
    case GFX_TEX_RGBA8_UNORM: {
        const uint32_t c = pool[idx];
        return make_float4(static_cast<float>(c & 0xFFu) / 255.0f, static_cast<float>((c >> 8) & 0xFFu) / 255.0f,
                           static_cast<float>((c >> 16) & 0xFFu) / 255.0f, static_cast<float>(c >> 24) / 255.0f);
    }
    case GFX_TEX_R8_UNORM: {
        const uint32_t c = (pool[idx >> 2] >> ((idx & 3u) * 8u)) & 0xFFu;
        return make_float4(static_cast<float>(c) / 255.0f, 0.0f, 0.0f, 1.0f);
    }
    case GFX_TEX_RG8_UNORM: {
        const uint32_t c = (pool[idx >> 1] >> ((idx & 1u) * 16u)) & 0xFFFFu;
        return make_float4(static_cast<float>(c & 0xFFu) / 255.0f, static_cast<float>(c >> 8) / 255.0f, 0.0f, 1.0f);
    }
    default: {   // GFX_TEX_RGBA32F
        return reinterpret_cast<const float4*>(pool)[idx];
    }
    }
}

// tex2DLod<float4>(tex, u, v, 0)
GFX_DEV float4 tex2d_desc(const DevScene& sc, const DevTexture& t, float u, float v);
GFX_DEV float4 tex2d(const DevScene& sc, uint32_t texSlot, float u, float v) { return tex2d_desc(sc, sc.textures[texSlot], u, v); }
GFX_DEV float4 tex2d_desc(const DevScene& sc, const DevTexture& t, float u, float v) {
    const TexelQuad q = tex_footprint(t.width, t.height, u, v);
    const float4 t00 = tex_texel(sc, t, q.i0, q.j0), t10 = tex_texel(sc, t, q.i1, q.j0);
    const float4 t01 = tex_texel(sc, t, q.i0, q.j1), t11 = tex_texel(sc, t, q.i1, q.j1);
    float4 r;
    r.x = q.w00 * t00.x + q.w10 * t10.x + q.w01 * t01.x + q.w11 * t11.x;
    r.y = q.w00 * t00.y + q.w10 * t10.y + q.w01 * t01.y + q.w11 * t11.y;
    r.z = q.w00 * t00.z + q.w10 * t10.z + q.w01 * t01.z + q.w11 * t11.z;
    r.w = q.w00 * t00.w + q.w10 * t10.w + q.w01 * t01.w + q.w11 * t11.w;
    return r;
}

// tex2Dgather<float4>(tex, u, v, 0): component 0 of the four texels of the bilinear footprint, in CUDA's order
// (x: (i, j+1), y: (i+1, j+1), z: (i+1, j), w: (i, j)).
GFX_DEV float4 tex2d_gather_r(const DevScene& sc, uint32_t texSlot, float u, float v) {
    const DevTexture t = sc.textures[texSlot];
    const TexelQuad q = tex_footprint(t.width, t.height, u, v);
    return make_float4(tex_texel(sc, t, q.i0, q.j1).x, tex_texel(sc, t, q.i1, q.j1).x, tex_texel(sc, t, q.i1, q.j0).x, tex_texel(sc, t, q.i0, q.j0).x);
}

} // namespace gfx
