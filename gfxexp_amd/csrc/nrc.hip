// nrc.hip -- the neural radiance cache network: input encoding + fully fused 64-wide MLP on the
// MFMA units (v_mfma_f32_32x32x16_bf16), training step (forward, relative-L2-luminance loss,
// backward, Adam + EMA) and the C ABI behind NeuralRadianceCache
// (neural_radiance_caching/network_interface.h:14-28, network_interface.cu:48-157).
//
// tiny-cuda-nn is not vendored in the reference (ext/tiny-cuda-nn is an empty submodule), so the
// arithmetic follows the published algorithms as restated in oracle/nrc_net.py (parity unpinned).
//
// Formulation.  Every layer is computed transposed, H_out^T [64 x batch] = W [64 x 64] . H_in^T [64 x batch]:
// the weight tile is the MFMA A operand (M = out feature, K = in feature), the activations are the
// B operand (K = in feature, N = batch column).  With the 32x32x16 shape a wave owns 64 batch
// columns (two N tiles); lane (n = lane & 31, h = lane >> 5) holds, for batch column n of each tile,
// the 16 accumulator rows (reg & 3) + 8 * (reg >> 2) + 4 * h of each 32-row M tile.  The B operand
// of the NEXT layer wants 8 consecutive K slots per lane -- instead of shuffling, the K order of
// every weight matrix is permuted once when it is packed: K slot (s, h, i) carries feature
//     f(s, h, i) = 16 s + 8 (i >> 2) + 4 h + (i & 3),
// which is exactly accumulator register 8 (s & 1) + i of M tile s >> 1.  Activations therefore stay
// in registers from the encoding to the output with no LDS round trip and no cross-lane traffic;
// LDS holds the packed bf16 weight fragments (one ds_read_b128 per fragment per lane).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "internal.h"
#include "gm_math.hip.h"

#define GFX_HOSTDEV __host__ __device__

namespace gfx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kNrcIn = 14, kNrcOut = 3, kNrcOutPad = 16;
constexpr int kHashLevels = 16, kLog2Hashmap = 15, kBaseRes = 16;
constexpr int kTriFreqs = 12;
constexpr int kMaxHidden = 5;
constexpr int kFragElems = 64 * 8;                 // bf16 per (mt, s) fragment: 64 lanes x 8
constexpr int kMatFwdElems = 2 * 4 * kFragElems;   // 64 x 64 matrix, forward image (8 KiB)
constexpr int kOutFwdElems = 1 * 4 * kFragElems;   // 16(32) x 64 output matrix, forward image
constexpr int kOutBwdElems = 2 * 2 * kFragElems;   // 64 x 16(32) transposed output matrix
constexpr float kLossScale = 128.0f;
constexpr int kTStride = 72;                       // bf16 per row of the [feature][batch] LDS images (144 B)

struct NrcLevel { float scale; uint32_t res; uint32_t entries; uint32_t offset; };

struct NrcDev {
    int posEnc;                 // 1 = hash grid, 0 = triangle wave
    int numHidden;              // hidden layers (2 or 5): W0 + (numHidden - 1) hidden matrices + Wout
    uint32_t gridOff;           // fp32 element offset of the grid inside the parameter blob
    uint32_t total;             // parameters
    NrcLevel levels[kHashLevels];
};

// fp32 -> bf16, round to nearest even: gfx950's v_cvt_pk_bf16_f32 converts two values per instruction (the integer form
// (u + 0x7FFF + lsb) >> 16 costs four per value; same bits for every finite input, NaNs stay NaNs).
typedef float NrcF2 __attribute__((ext_vector_type(2)));
typedef __bf16 NrcBf2 __attribute__((ext_vector_type(2)));
GFX_DEV uint32_t pack_bf16x2(float lo, float hi) {
    const NrcF2 v = { lo, hi };
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, NrcBf2));
}
GFX_DEV uint32_t to_bf16_bits(float x) { return pack_bf16x2(x, 0.0f) & 0xFFFFu; }
GFX_DEV float from_bf16_bits(uint32_t b) { return bits2f(b << 16); }
GFX_DEV float bf16_lo(uint32_t pair) { return bits2f(pair << 16); }
GFX_DEV float bf16_hi(uint32_t pair) { return bits2f(pair & 0xFFFF0000u); }
// The level's two features of one query: trilinear blend of the 8 corner entries (bf16 pairs), fused multiply-adds in corner order.
GFX_DEV void blend_corners(const uint32_t e[8], const float w[8], float& a0, float& a1) {
    a0 = w[0] * bf16_lo(e[0]); a1 = w[0] * bf16_hi(e[0]);
#pragma unroll
    for (int c = 1; c < 8; ++c) {
        a0 = __builtin_fmaf(w[c], bf16_lo(e[c]), a0);
        a1 = __builtin_fmaf(w[c], bf16_hi(e[c]), a1);
    }
}
GFX_HOSTDEV inline int nrc_feature_of_slot(int s, int h, int i) { return 16 * s + 8 * (i >> 2) + 4 * h + (i & 3); }

// ---------------------------------------------------------------- weight packing
// fp32 parameters -> bf16 MFMA fragments.  Forward image of W [out][in]: fragment (mt, s), lane (m, h),
// element i = W[32 mt + m][f(s, h, i)]; transposed image (backward): fragment (mt, s) = W[f(s, h, i)][32 mt + m].
__global__ void k_nrc_pack(NrcDev d, const float* __restrict__ params, uint16_t* __restrict__ fwd, uint16_t* __restrict__ bwd,
                           uint32_t* __restrict__ gridOut) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t numMats = d.numHidden;                       // 64 x 64 matrices
    const uint32_t fwdTotal = numMats * kMatFwdElems + kOutFwdElems;
    if (t < fwdTotal) {
        const bool isOut = t >= numMats * kMatFwdElems;
        const uint32_t mat = isOut ? numMats : t / kMatFwdElems;
        const uint32_t r = isOut ? t - numMats * kMatFwdElems : t % kMatFwdElems;
        const uint32_t frag = r / kFragElems, lane = (r % kFragElems) / 8, i = r % 8;
        const uint32_t mt = isOut ? 0 : frag / 4, s = frag % 4;
        const uint32_t m = lane & 31, h = lane >> 5;
        const uint32_t row = 32 * mt + m, col = nrc_feature_of_slot(s, h, i);
        float v = 0.0f;
        if (!isOut || row < static_cast<uint32_t>(kNrcOutPad)) v = params[mat * 4096 + row * 64 + col];
        fwd[t] = static_cast<uint16_t>(to_bf16_bits(v));
    }
    if (bwd) {
        const uint32_t bwdTotal = numMats * kMatFwdElems + kOutBwdElems;
        if (t < bwdTotal) {
            const bool isOut = t >= numMats * kMatFwdElems;
            const uint32_t mat = isOut ? numMats : t / kMatFwdElems;
            const uint32_t r = isOut ? t - numMats * kMatFwdElems : t % kMatFwdElems;
            const uint32_t frag = r / kFragElems, lane = (r % kFragElems) / 8, i = r % 8;
            const uint32_t mt = isOut ? frag / 2 : frag / 4, s = isOut ? frag % 2 : frag % 4;
            const uint32_t m = lane & 31, h = lane >> 5;
            const uint32_t inF = 32 * mt + m, outF = nrc_feature_of_slot(s, h, i);
            float v = 0.0f;
            if (!isOut || outF < static_cast<uint32_t>(kNrcOutPad)) v = params[mat * 4096 + outF * 64 + inF];
            bwd[t] = static_cast<uint16_t>(to_bf16_bits(v));
        }
    }
    if (d.posEnc == 1) {
        const uint32_t entries = (d.total - d.gridOff) / 2;
        for (uint32_t e = t; e < entries; e += gridDim.x * blockDim.x) {
            const float a = params[d.gridOff + 2 * e], b = params[d.gridOff + 2 * e + 1];
            gridOut[e] = pack_bf16x2(a, b);
        }
    }
}

// ---------------------------------------------------------------- encoding
GFX_DEV float quartic_cdf(float x, float invRadius) {
    const float u = x * invRadius;
    const float u2 = u * u;
    const float u4 = u2 * u2;
    const float v = (15.0f / 16.0f) * u * (1 - (2.0f / 3.0f) * u2 + (1.0f / 5.0f) * u4) + 0.5f;
    return fmin2(fmax2(v, 0.0f), 1.0f);
}
// One-blob encoding, 4 bins, quartic kernel of radius 1/4, wrapped: cdf(b) = Q(b/4 - x) + Q(b/4 - x - 1) + Q(b/4 - x + 1), feature b = cdf(b + 1) - cdf(b).
// For x in [0, 1] -- every encoded input is: normalised direction angles, roughness -- ten of the fifteen kernel integrals sit on the
// clamp: Q(b/4 - x - 1) is 0 unless b = 4, where it is Q(-x); Q(b/4 - x + 1) is 1 unless b = 0, where it is Q(1 - x).  A lane in range takes
// the five-integral form, a lane out of range (NaN included) the fifteen-integral form -- a branch no lane of a wave normally enters; a
// query's features do not depend on the queries it shares a wave with.  The two forms agree to 2.4e-7 (the polynomial reaches the clamp to
// within an ulp of 1/2, not exactly), four orders of magnitude under the bf16 rounding of the features.
GFX_DEV void oneblob4(float x, float out[4]) {
    float q[5], below[5] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f }, above[5] = { 0.0f, 1.0f, 1.0f, 1.0f, 1.0f };
#pragma unroll
    for (int b = 0; b <= 4; ++b) q[b] = quartic_cdf(b * 0.25f - x, 4.0f);
    below[4] = q[0]; above[0] = q[4];
    if (!(x >= 0.0f && x <= 1.0f)) {
#pragma unroll
        for (int b = 0; b <= 4; ++b) {
            const float left = b * 0.25f;
            below[b] = quartic_cdf(left - x - 1.0f, 4.0f); above[b] = quartic_cdf(left - x + 1.0f, 4.0f);
        }
    }
    float cdf[5];
#pragma unroll
    for (int b = 0; b <= 4; ++b) cdf[b] = q[b] + below[b] + above[b];
#pragma unroll
    for (int b = 0; b < 4; ++b) out[b] = cdf[b + 1] - cdf[b];
}
// Table indices (level offset included) and trilinear weights of the 8 corners of one level.
// Hashed levels: x ^ y * 2654435761 ^ z * 805459861 (uint32) modulo the table size; dense levels:
// x + y res + z res^2 modulo the table size.  Table sizes are powers of two for the configuration
// the reference uses (4096, 32768), where the modulo is a mask.
// DENSE / POW2: 0 or 1 when the caller has decided (both are properties of the level, the same for every query), -1 = decided here.
template <int DENSE, int POW2>
GFX_DEV void grid_corners_t(const NrcLevel& lv, float px, float py, float pz, uint32_t idx[8], float w[8]) {
    const float x = px * lv.scale + 0.5f, y = py * lv.scale + 0.5f, z = pz * lv.scale + 0.5f;
    const float bx = floorf(x), by = floorf(y), bz = floorf(z);
    const float fx = x - bx, fy = y - by, fz = z - bz;
    const uint32_t ix = static_cast<uint32_t>(static_cast<int32_t>(bx));
    const uint32_t iy = static_cast<uint32_t>(static_cast<int32_t>(by));
    const uint32_t iz = static_cast<uint32_t>(static_cast<int32_t>(bz));
    const bool dense = DENSE >= 0 ? DENSE != 0 : static_cast<unsigned long long>(lv.res) * lv.res * lv.res <= lv.entries;
    const bool pow2 = POW2 >= 0 ? POW2 != 0 : (lv.entries & (lv.entries - 1)) == 0;
    uint32_t tx[2], ty[2], tz[2];
    if (dense) {
        tx[0] = ix; tx[1] = ix + 1;
        ty[0] = iy * lv.res; ty[1] = (iy + 1) * lv.res;
        tz[0] = iz * lv.res * lv.res; tz[1] = (iz + 1) * lv.res * lv.res;
    }
    else {
        tx[0] = ix; tx[1] = ix + 1;
        ty[0] = iy * 2654435761u; ty[1] = (iy + 1) * 2654435761u;
        tz[0] = iz * 805459861u; tz[1] = (iz + 1) * 805459861u;
    }
    const float wx[2] = { 1 - fx, fx }, wy[2] = { 1 - fy, fy }, wz[2] = { 1 - fz, fz };
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ox = c & 1, oy = (c >> 1) & 1, oz = (c >> 2) & 1;
        const uint32_t raw = dense ? tx[ox] + ty[oy] + tz[oz] : tx[ox] ^ ty[oy] ^ tz[oz];
        idx[c] = lv.offset + (pow2 ? raw & (lv.entries - 1) : raw % lv.entries);
        w[c] = wx[ox] * wy[oy] * wz[oz];
    }
}
GFX_DEV void grid_corners(const NrcLevel& lv, float px, float py, float pz, uint32_t idx[8], float w[8]) { grid_corners_t<-1, -1>(lv, px, py, pz, idx, w); }

// The 8 table entries of one level: eight 4-byte gathers.  (Fetching the x-neighbour pair of a corner with one 16-byte block
// load was measured slower -- the pair already shares its 64-byte sector in ~90 % of the cases and the gathers are bound by
// L2 -> L1 sector fills, profiles/r02_nrc.txt.)
#ifndef GFX_NRC_INFER_WAVES
#define GFX_NRC_INFER_WAVES 2       // waves per SIMD k_nrc_infer is compiled for: 3 and 4 measured slower (same file)
#endif
GFX_DEV void gather_corners(const uint32_t* __restrict__ grid, const uint32_t idx[8], uint32_t e[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) e[c] = grid[idx[c]];
}

// Canonical features f0 .. f0 + 3 of one batch column that do not come out of the hash grid: triangle wave (posEnc 0), one-blob,
// identity, padding ones.
GFX_DEV void encode_plain_group(int posFeatures /* 32: hash grid, 36: triangle wave */, const float x[kNrcIn], int f0, float v[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = f0 + r;
        float val = 1.0f;                                       // padding
        if (f < posFeatures) {                                  // triangle wave: feature 12 dim + freq
            const int dim = f / kTriFreqs, freq = f % kTriFreqs;
            const float xs = ldexpf(x[dim], freq - 1);
            val = fabsf(xs - floorf(xs) - 0.5f) * 4 - 1;
        }
        else if (f < posFeatures + 20) {
            const int o = f - posFeatures;
            float ob[4];
            oneblob4(x[3 + (o >> 2)], ob);
            val = ob[o & 3];
        }
        else if (f < posFeatures + 26) val = x[8 + (f - posFeatures - 20)];
        v[r] = val;
    }
}

// The 32 canonical features a lane (half h) supplies for one batch column, in K-slot order
// out[s * 8 + i] = feature f(s, h, i).  Canonical order: [position 32|36] [one-blob 20] [identity 6] [ones].
GFX_DEV void encode_half(const NrcDev& d, const uint32_t* __restrict__ grid, const float x[kNrcIn], int h, float out[32]) {
    // groups of 4 consecutive canonical features: group g = 4 g .. 4 g + 3; this half owns groups with (g & 1) == h
#pragma unroll
    for (int q = 0; q < 8; ++q) {                 // q-th owned group -> slots [4 q, 4 q + 4): s = q >> 1, i = 4 (q & 1) + r
        const int g = 2 * q + h;
        float v[4];
        const int f0 = 4 * g;
        if (d.posEnc == 1 && f0 < 32) {           // two hash-grid levels: 2 g, 2 g + 1
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const NrcLevel lv = d.levels[2 * g + k];
                uint32_t idx[8]; float w[8]; uint32_t e[8];
                grid_corners(lv, x[0], x[1], x[2], idx, w);
                gather_corners(grid, idx, e);
                blend_corners(e, w, v[2 * k], v[2 * k + 1]);
            }
        }
        else encode_plain_group(d.posEnc == 1 ? 32 : 3 * kTriFreqs, x, f0, v);
#pragma unroll
        for (int r = 0; r < 4; ++r) out[4 * q + r] = v[r];
    }
}

GFX_DEV uint4 pack8(const float v[8]) {
    uint4 r;
    r.x = pack_bf16x2(v[0], v[1]);
    r.y = pack_bf16x2(v[2], v[3]);
    r.z = pack_bf16x2(v[4], v[5]);
    r.w = pack_bf16x2(v[6], v[7]);
    return r;
}
GFX_DEV void unpack8(uint4 p, float v[8]) {
    v[0] = from_bf16_bits(p.x & 0xFFFFu); v[1] = from_bf16_bits(p.x >> 16);
    v[2] = from_bf16_bits(p.y & 0xFFFFu); v[3] = from_bf16_bits(p.y >> 16);
    v[4] = from_bf16_bits(p.z & 0xFFFFu); v[5] = from_bf16_bits(p.z >> 16);
    v[6] = from_bf16_bits(p.w & 0xFFFFu); v[7] = from_bf16_bits(p.w >> 16);
}

// B operand (4 K steps) of one batch column from 32 slot-ordered fp32 values
GFX_DEV void to_operand(const float v[32], uint4 b[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) b[s] = pack8(v + 8 * s);
}
// accumulators of a 64-row layer (two M tiles) -> ReLU -> next B operand
GFX_DEV void relu_to_operand(const f32x16 acc[2], uint4 b[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmax2(acc[s >> 1][8 * (s & 1) + i], 0.0f);
        b[s] = pack8(v);
    }
}

// one 64 x 64 layer for one N tile: acc[mt] = sum_s A(mt, s) . B(s); `frags` = the matrix' fragment image
template <typename FragPtr>
GFX_DEV void layer64(FragPtr frags, int lane, const uint4 b[4], f32x16 acc[2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint4 a = frags[(mt * 4 + s) * 64 + lane];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b[s]), c, 0, 0, 0);
        }
        acc[mt] = c;
    }
}

// ---------------------------------------------------------------- inference
// inputs [14, N] column-major fp32, predictions [3, N] column-major fp32 (network_interface.cu:141-147)
constexpr int kInferBlock = 256;
__global__ __launch_bounds__(kInferBlock) __attribute__((amdgpu_waves_per_eu(GFX_NRC_INFER_WAVES, GFX_NRC_INFER_WAVES))) void k_nrc_infer(NrcDev d, const uint16_t* __restrict__ fwd, const uint32_t* __restrict__ grid,
                                                           const float* __restrict__ inputs, uint32_t numDataArg, const uint32_t* __restrict__ numDataPtr,
                                                           float* __restrict__ predictions) {
    extern __shared__ __attribute__((aligned(16))) uint4 ldsW[];
    const uint32_t numData = numDataPtr ? min(*numDataPtr, numDataArg) : numDataArg;   // device-side batch size (gfx_nrc_infer_indirect)
    const uint32_t fwdElems = d.numHidden * kMatFwdElems + kOutFwdElems;
    for (uint32_t i = threadIdx.x; i < fwdElems / 8; i += kInferBlock) ldsW[i] = reinterpret_cast<const uint4*>(fwd)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    const uint32_t wave = blockIdx.x * (kInferBlock / 64) + (threadIdx.x >> 6);
    const uint32_t numWaves = gridDim.x * (kInferBlock / 64);
    const uint32_t numTiles = (numData + 63) / 64;
    // the tile's 64 x 14 inputs are one contiguous 3 584-byte run: 14 coalesced loads (4 cache lines each) into a
    // wave-private LDS image, read back per column, instead of 14 strided scalars per lane and half (28 lines per load)
    float* ldsX = reinterpret_cast<float*>(ldsW + fwdElems / 8) + (threadIdx.x >> 6) * (64 * kNrcIn);
    for (uint32_t tile = wave; tile < numTiles; tile += numWaves) {
        {
            const size_t base = static_cast<size_t>(tile) * 64 * kNrcIn;
            const size_t limit = static_cast<size_t>(numData) * kNrcIn;
            float v[kNrcIn];
#pragma unroll
            for (int j = 0; j < kNrcIn; ++j) { const size_t e = base + 64 * j + lane; v[j] = e < limit ? inputs[e] : 0.0f; }
            __builtin_amdgcn_wave_barrier();          // the previous tile's reads are done (same wave, program order)
#pragma unroll
            for (int j = 0; j < kNrcIn; ++j) ldsX[64 * j + lane] = v[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        uint4 b[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float x[kNrcIn];
#pragma unroll
            for (int k = 0; k < kNrcIn; ++k) x[k] = ldsX[(32 * nt + n) * kNrcIn + k];
            float enc[32];
            encode_half(d, grid, x, h, enc);
            to_operand(enc, b[nt]);
        }
        for (int layer = 0; layer < d.numHidden; ++layer) {
            const uint4* frags = ldsW + layer * (kMatFwdElems / 8);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x16 acc[2];
                layer64(frags, lane, b[nt], acc);
                relu_to_operand(acc, b[nt]);
            }
        }
        const uint4* fragsOut = ldsW + d.numHidden * (kMatFwdElems / 8);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x16 c;
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fragsOut[s * 64 + lane]),
                                                            __builtin_bit_cast(bf16x8, b[nt][s]), c, 0, 0, 0);
            const uint32_t col = tile * 64 + 32 * nt + n;
            if (h == 0 && col < numData) {       // rows 0..2 live in registers 0..2 of the h == 0 lanes
                float* o = predictions + static_cast<size_t>(col) * kNrcOut;
                o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
            }
        }
    }
}

// ---------------------------------------------------------------- inference, hash-grid levels staged through LDS
// k_nrc_infer above gathers 128 four-byte table entries per query through the vector-memory path: 270 M L1 fills of a 64-byte sector
// each per full-HD batch, 0.89 TCP requests per CU and clock -- that, not the matrix pipe (MFMA busy 3 %), is what it runs at
// (profiles/r04_nrc_pmc.txt).  A level's table is at most 2^15 entries x (2 x bf16) = 128 KiB and a CU has 160 KiB of LDS, so a large
// batch is encoded level-synchronously instead: one persistent 12-wave block per CU; per pass it owns 12 x kStagedTiles tiles of 64
// queries; for each of the 16 levels the block copies the level's table into LDS with global->LDS DMA (coalesced, 2 MB of L2 reads per
// pass and CU) and every lane computes the level's two features of ONE query of each of its wave's tiles -- eight ds_read_b32 per
// corner set, no L1 fill, no 64-byte sector per 4-byte corner -- packs them to the bf16 pair the MFMA operand wants and hands the
// pair to the lane that owns that (query, feature group) in the operand layout (lane n of half h: queries n and 32 + n of the tile, the
// levels with bit 1 == h), which is its partner lane ^ 32 or itself.  The hash half of the operands (16 registers per tile) waits in
// registers; when the 16 levels are done the LDS holds the weight fragments and a staging area instead, and the wave runs the rest of
// k_nrc_infer's body per tile: one-blob / identity features, the layers, the output.  Per query the same operations in the same order as
// k_nrc_infer: the outputs are bit-equal (tests/test_gpu_nrc_net.py).  Small batches (the training tiles' suffix queries) keep k_nrc_infer:
// a pass costs 2 MB of table copies whatever it encodes.
// Waves per SIMD against queries per pass (the operand registers of a pass are the budget; a pass pays 16 table copies of ~2 us whatever it
// encodes): 8 waves x 8 tiles 0.340 ms per 2.1 M queries, 12 x 5 0.327 (42 spilled registers), 12 x 4 0.277, 16 x 3 0.272 (50 spilled),
// 16 x 4 0.358 (90 spilled), 4 x 16 0.522 (profiles/r05_experiments.txt 4): both phases wait on latencies that two waves per SIMD do not cover.
#ifndef GFX_NRC_STAGED_BLOCK
#define GFX_NRC_STAGED_BLOCK 768    // 12 waves, 3 per SIMD (170 registers each)
#define GFX_NRC_STAGED_TILES 4      // tiles per wave and pass: 12 x 4 x 64 = 3 072 queries per pass and CU
#endif
constexpr int kStagedBlock = GFX_NRC_STAGED_BLOCK;
constexpr int kStagedTiles = GFX_NRC_STAGED_TILES;
constexpr uint32_t kStagedTableBytes = 4u << kLog2Hashmap;
GFX_DEV uint32_t staged_word(const uint4& v, int w) { return w == 0 ? v.x : w == 1 ? v.y : w == 2 ? v.z : v.w; }
GFX_DEV void staged_set_word(uint4& v, int w, uint32_t x) { if (w == 0) v.x = x; else if (w == 1) v.y = x; else if (w == 2) v.z = x; else v.w = x; }
// Byte offsets into the LDS copy of a level's table + trilinear weights of the 8 corners: grid_corners_t's indices (without the level
// offset) times four, the weights in its order.  Power-of-two tables only need the low log2(entries) + 2 bits of the scaled index, so the
// products are multiplies by the constant times four (v_mul_u32_u24; v_mul_lo_u32 issues at the same 4.2 cycles on gfx950, valu_rate.hip),
// nothing is shifted per corner, and the mask merges into the last xor (v_bitop3).
template <int DENSE, int POW2>
GFX_DEV void staged_corners(const NrcLevel& lv, float px, float py, float pz, uint32_t off[8], float w[8]) {
    if (POW2 != 1) {
        grid_corners_t<DENSE, POW2>(lv, px, py, pz, off, w);
#pragma unroll
        for (int c = 0; c < 8; ++c) off[c] <<= 2;
        return;
    }
    const float x = px * lv.scale + 0.5f, y = py * lv.scale + 0.5f, z = pz * lv.scale + 0.5f;
    const float bx = floorf(x), by = floorf(y), bz = floorf(z);
    const float fx = x - bx, fy = y - by, fz = z - bz;
    const uint32_t ix = static_cast<uint32_t>(static_cast<int32_t>(bx));
    const uint32_t iy = static_cast<uint32_t>(static_cast<int32_t>(by));
    const uint32_t iz = static_cast<uint32_t>(static_cast<int32_t>(bz));
    const uint32_t mask = (lv.entries - 1u) << 2;
    const uint32_t sy = (DENSE == 1 ? lv.res << 2 : 2654435761u << 2) & 0xFFFFFFu;
    const uint32_t sz = (DENSE == 1 ? (lv.res * lv.res) << 2 : 805459861u << 2) & 0xFFFFFFu;
    uint32_t tx[2], ty[2], tz[2];
    tx[0] = ix << 2; tx[1] = tx[0] + 4u;
    ty[0] = __umul24(iy, sy); ty[1] = ty[0] + sy;
    tz[0] = __umul24(iz, sz); tz[1] = tz[0] + sz;
    const float wx[2] = { 1 - fx, fx }, wy[2] = { 1 - fy, fy }, wz[2] = { 1 - fz, fz };
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ox = c & 1, oy = (c >> 1) & 1, oz = (c >> 2) & 1;
        off[c] = (DENSE == 1 ? tx[ox] + ty[oy] + tz[oz] : tx[ox] ^ ty[oy] ^ tz[oz]) & mask;
        w[c] = wx[ox] * wy[oy] * wz[oz];
    }
}
// One level for the wave's kStagedTiles tiles: the level's two features of query `lane` of every tile out of the LDS copy of the table,
// as the bf16 pair of the operand layout, for the lane that owns it (half hL: queries n and 32 + n of the tile).
template <int DENSE, int POW2>
GFX_DEV void staged_level(const NrcLevel& lv, const uint32_t* ldsTable, const float (&px)[kStagedTiles], const float (&py)[kStagedTiles], const float (&pz)[kStagedTiles],
                          int h, int hL, bool lastSlotUsed, uint32_t (&w0)[kStagedTiles], uint32_t (&w1)[kStagedTiles]) {
#pragma unroll
    for (int t = 0; t < kStagedTiles; ++t) {
        if (t == kStagedTiles - 1 && !lastSlotUsed) continue;             // wave-uniform
        uint32_t off[8]; float w[8]; uint32_t e[8];
        staged_corners<DENSE, POW2>(lv, px[t], py[t], pz[t], off, w);
#pragma unroll
        for (int c = 0; c < 8; ++c) e[c] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ldsTable) + off[c]);
        float a0, a1;
        blend_corners(e, w, a0, a1);
        // query `lane` of the tile holds the pair; the lane that owns it in the operand layout is lane n of the tile's half: queries n
        // (computed by lane n) and 32 + n (computed by lane 32 + n).  v_permlane32_swap: first = {lanes 0 .. 31 their own, lanes 32 .. 63 that
        // of lane - 32} = query n for (n, h); second = {lanes 0 .. 31 that of lane + 32, lanes 32 .. 63 their own} = query 32 + n
        const uint32_t mine = pack_bf16x2(a0, a1);
        const auto both = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
        if (h == hL) { w0[t] = both[0]; w1[t] = both[1]; }
    }
}
__global__ __launch_bounds__(kStagedBlock) __attribute__((amdgpu_waves_per_eu(kStagedBlock / 256, kStagedBlock / 256)))
void k_nrc_infer_staged(NrcDev d, const uint16_t* __restrict__ fwd, const uint32_t* __restrict__ grid, const float* __restrict__ inputs,
                        uint32_t numDataArg, const uint32_t* __restrict__ numDataPtr, float* __restrict__ predictions) {
    extern __shared__ __attribute__((aligned(16))) uint4 ldsAll[];         // kStagedTableBytes: a level's table, then weights + staging
    const uint32_t numData = numDataPtr ? min(*numDataPtr, numDataArg) : numDataArg;
    const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t numTiles = (numData + 63) / 64;
    // Passes of equal size: the batch is cut into gridDim.x x rounds passes of `passTiles` <= 12 x kStagedTiles tiles (a batch of 2.1 M
    // queries: 3 rounds of 43 tiles on every CU instead of 3 rounds of 48 on 181 CUs and 2 on the rest); inside a pass tile k belongs to
    // wave k mod 12, slot k / 12, so only a wave's last slot can be empty.
    constexpr uint32_t kWaves = kStagedBlock / 64, kTilesPerPass = kWaves * kStagedTiles;
    const uint32_t rounds = (numTiles + gridDim.x * kTilesPerPass - 1) / (gridDim.x * kTilesPerPass);
    const uint32_t numPasses = gridDim.x * rounds;
    const uint32_t passTiles = (numTiles + numPasses - 1) / numPasses;
    const uint32_t fwdElems = d.numHidden * kMatFwdElems + kOutFwdElems;
    const uint32_t* ldsTable = reinterpret_cast<const uint32_t*>(ldsAll);
    GFX_CYC_BEGIN          // profiling builds (GFX_LANE_PROFILE, tools/nrc_infer_profile.py): where a wave's cycles go
    for (uint32_t pass = blockIdx.x; pass < numPasses && pass * passTiles < numTiles; pass += gridDim.x) {
        GFX_CYC(0);        // positions of the pass's queries
        const uint32_t tile0 = pass * passTiles + wave, tileEnd = min((pass + 1) * passTiles, numTiles);   // slot t: tile0 + t * kWaves
        const bool lastSlotUsed = tile0 + (kStagedTiles - 1) * kWaves < tileEnd;                              // wave-uniform
        // the position of query `lane` of each tile (inputs are [14] per query: three strided loads)
        float px[kStagedTiles], py[kStagedTiles], pz[kStagedTiles];
#pragma unroll
        for (int t = 0; t < kStagedTiles; ++t) {
            const uint32_t tile = tile0 + t * kWaves;
            const size_t col = static_cast<size_t>(tile) * 64 + lane;
            const bool ok = tile < tileEnd && col < numData;
            px[t] = ok ? inputs[col * kNrcIn] : 0.0f; py[t] = ok ? inputs[col * kNrcIn + 1] : 0.0f; pz[t] = ok ? inputs[col * kNrcIn + 2] : 0.0f;
        }
        uint4 hb[kStagedTiles][2][2];                       // [tile][nt][s]: K steps 0 and 1 of the B operand = the 16 hash-grid features of this half
#pragma unroll
        for (int t = 0; t < kStagedTiles; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) hb[t][k >> 1][k & 1] = make_uint4(0u, 0u, 0u, 0u);
        // Level L = 2 g + k: group g belongs to half g & 1 as its (g >> 1)-th group, slots 4 (g >> 1) + 2 k, + 1 (encode_half), i.e. word
        // pos = 2 (g >> 1) + k of the half's eight hash words (K step pos >> 2, word pos & 3).  Levels L and L + 2 (bit 1 of L clear) fill
        // the same word position, one for each half: the loop runs over the positions, two levels each.
        for (int pos = 0; pos < 8; ++pos) {
            uint32_t w0[kStagedTiles], w1[kStagedTiles];    // the new word of tile half 0 / 1 of every tile
#pragma unroll
            for (int t = 0; t < kStagedTiles; ++t) { w0[t] = 0u; w1[t] = 0u; }
#pragma unroll
            for (int hL = 0; hL < 2; ++hL) {
                const int L = 4 * (pos >> 1) + (pos & 1) + 2 * hL;
                NrcLevel lv = d.levels[L];
                GFX_CYC(1);                                 // waiting for the block's other waves to finish the level before
                __syncthreads();                            // the table (or the weights / staging area) of before is no longer read
                GFX_CYC(2);                                 // the level's table: global -> LDS DMA, issued, landed, seen by every wave
                {
                    const char* src = reinterpret_cast<const char*>(grid + lv.offset);
                    const uint32_t bytes = lv.entries * 4u; // a multiple of 32
                    for (uint32_t off = static_cast<uint32_t>(wave) * 1024u; off < bytes; off += (kStagedBlock / 64) * 1024u) {
                        if (off + static_cast<uint32_t>(lane) * 16u < bytes) {
                            typedef const __attribute__((address_space(1))) void* GlobalPtr;
                            typedef __attribute__((address_space(3))) void* LdsPtr;
                            __builtin_amdgcn_global_load_lds((GlobalPtr)(src + off + lane * 16), (LdsPtr)(reinterpret_cast<char*>(ldsAll) + off), 16, 0, 0);
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                GFX_CYC(3);                                 // the level's two features of this wave's queries (index arithmetic, eight ds_read_b32, blend, exchange)
                lv.offset = 0u;                             // indices into the LDS copy
                // (the kind of the level decides the index arithmetic once for all queries: a mask for the power-of-two tables of the
                // reference's configuration, hashed or dense; anything else takes the general form)
                const bool dense = static_cast<unsigned long long>(lv.res) * lv.res * lv.res <= lv.entries, pow2 = (lv.entries & (lv.entries - 1)) == 0;
                if (pow2 && !dense) staged_level<0, 1>(lv, ldsTable, px, py, pz, h, hL, lastSlotUsed, w0, w1);
                else if (pow2) staged_level<1, 1>(lv, ldsTable, px, py, pz, h, hL, lastSlotUsed, w0, w1);
                else staged_level<-1, -1>(lv, ldsTable, px, py, pz, h, hL, lastSlotUsed, w0, w1);
            }
            switch (pos) {                                  // block-uniform
#define GFX_STAGED_CASE(P) case P: _Pragma("unroll") for (int t = 0; t < kStagedTiles; ++t) { \
                staged_set_word(hb[t][0][(P) >> 2], (P) & 3, w0[t]); staged_set_word(hb[t][1][(P) >> 2], (P) & 3, w1[t]); } break;
            GFX_STAGED_CASE(0) GFX_STAGED_CASE(1) GFX_STAGED_CASE(2) GFX_STAGED_CASE(3)
            GFX_STAGED_CASE(4) GFX_STAGED_CASE(5) GFX_STAGED_CASE(6) GFX_STAGED_CASE(7)
#undef GFX_STAGED_CASE
            default: break;
            }
        }
        GFX_CYC(4);                                         // weights into LDS (two barriers)
        __syncthreads();                                    // the last table is no longer read
        for (uint32_t i = threadIdx.x; i < fwdElems / 8; i += kStagedBlock) ldsAll[i] = reinterpret_cast<const uint4*>(fwd)[i];
        __syncthreads();
        const uint4* ldsW = ldsAll;
        float* ldsX = reinterpret_cast<float*>(ldsAll + fwdElems / 8) + wave * (64 * kNrcIn);
        for (int t = 0; t < kStagedTiles; ++t) {           // not unrolled: the tile in turn is hb[0], the others move up behind it
            const uint32_t tile = tile0 + t * kWaves;
            if (tile < tileEnd) {                           // wave-uniform
            GFX_CYC(5);                                     // the tile's inputs through LDS, one-blob / identity features, operands
            {
                const size_t base = static_cast<size_t>(tile) * 64 * kNrcIn;
                const size_t limit = static_cast<size_t>(numData) * kNrcIn;
                float v[kNrcIn];
#pragma unroll
                for (int j = 0; j < kNrcIn; ++j) { const size_t e = base + 64 * j + lane; v[j] = e < limit ? inputs[e] : 0.0f; }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < kNrcIn; ++j) ldsX[64 * j + lane] = v[j];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            uint4 b[2][4];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float x[kNrcIn];
#pragma unroll
                for (int k = 0; k < kNrcIn; ++k) x[k] = ldsX[(32 * nt + n) * kNrcIn + k];
                // groups 8 .. 15 of this half (slots 16 .. 31): one-blob, identity, padding -- encode_half's plain branch, with the two halves
                // side by side instead of one after the other: half h owns groups 8 + h, 10 + h, 12 + h, 14 + h; groups 8 .. 12 are the one-blob
                // encodings of inputs 3 .. 7 (four features each), 13 and 14 the six identity inputs + two ones, 15 ones
                float enc[16];
#pragma unroll
                for (int i = 0; i < 2; ++i) {               // groups 8 + h and 10 + h: one-blob of input 3 + h / 5 + h
                    float ob[4];
                    oneblob4(h ? x[4 + 2 * i] : x[3 + 2 * i], ob);
#pragma unroll
                    for (int r = 0; r < 4; ++r) enc[4 * i + r] = ob[r];
                }
                {                                           // group 12 (h = 0): one-blob of input 7; group 13 (h = 1): inputs 8 .. 11
                    float ob[4];
                    oneblob4(x[7], ob);
#pragma unroll
                    for (int r = 0; r < 4; ++r) enc[8 + r] = h ? x[8 + r] : ob[r];
                }
                enc[12] = h ? 1.0f : x[12]; enc[13] = h ? 1.0f : x[13]; enc[14] = 1.0f; enc[15] = 1.0f;   // group 14 (h = 0): inputs 12, 13, ones; 15: ones
                b[nt][0] = hb[0][nt][0]; b[nt][1] = hb[0][nt][1];
                b[nt][2] = pack8(enc); b[nt][3] = pack8(enc + 8);
            }
            GFX_CYC(6);                                     // the layers (MFMA, ReLU, pack) and the output
            for (int layer = 0; layer < d.numHidden; ++layer) {
                const uint4* frags = ldsW + layer * (kMatFwdElems / 8);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x16 acc[2];
                    layer64(frags, lane, b[nt], acc);
                    relu_to_operand(acc, b[nt]);
                }
            }
            const uint4* fragsOut = ldsW + d.numHidden * (kMatFwdElems / 8);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x16 c;
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = 0.0f;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fragsOut[s * 64 + lane]),
                                                                __builtin_bit_cast(bf16x8, b[nt][s]), c, 0, 0, 0);
                const uint32_t col = tile * 64 + 32 * nt + n;
                if (h == 0 && col < numData) {
                    float* o = predictions + static_cast<size_t>(col) * kNrcOut;
                    o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
                }
            }
            }
#pragma unroll
            for (int k = 0; k + 1 < kStagedTiles; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) hb[k][j >> 1][j & 1] = hb[k + 1][j >> 1][j & 1];
        }
    }
    GFX_CYC_END;
}

// ---------------------------------------------------------------- inference, staged levels + the table-free half under the table copies
// k_nrc_infer_staged spends a third of a wave's cycles waiting for the level tables (the DMA of a 128-KiB table and its two barriers, sixteen
// times per pass: every wave of the block waits at the same time and one block fills the CU's LDS) and another third on work that needs no
// table (one-blob features, the layers, the output) -- profiles/r06_nrc_infer_profile.json.  This kernel runs the second under the first: a
// software pipeline across passes.  The hash half of a pass's operands is parked in an L2-resident scratch instead of registers (one K step
// of 4 words per tile half at a time: after the 8th and the 16th level); the table-free half of pass p - 1 is cut into sixteen slices -- tile
// 0..3 x batch half 0..1 x {one-blob + first layer, remaining layers + output} -- and slice i runs between the DMA issue and the DMA wait of the
// i-th level of pass p.  A slice's own loads (8 parked words, 11 inputs) are issued BEFORE the DMA instructions: vmcnt retires in order, a
// load issued behind the DMA would wait for it.  The weights stay in the 32 KiB of LDS the table leaves free (2 hidden layers: 20 KiB; deeper
// networks keep k_nrc_infer_staged).  Per query the same operations in the same order as k_nrc_infer: bit-equal outputs.
constexpr uint32_t kPipedWordsPerWave = 2u * kStagedTiles * 2u * 2u * 4u * 64u;    // [pass parity][tile][batch half][K step][word][lane]
GFX_DEV uint32_t piped_index(uint32_t parity, int t, int nt, int s, int w, int lane) {
    return ((((parity * kStagedTiles + static_cast<uint32_t>(t)) * 2u + static_cast<uint32_t>(nt)) * 2u + static_cast<uint32_t>(s)) * 4u + static_cast<uint32_t>(w)) * 64u + static_cast<uint32_t>(lane);
}
// One slice of the table-free half of a tile: part 0 = k_nrc_infer's features 32 .. 63 of this lane's half (one-blob, identity, ones) beside the
// parked hash features, then the first layer (its ReLU-packed output stays in bLive); part 1 = the remaining layers and the output.
GFX_DEV void piped_slice_features_and_first_layer(const uint4* ldsW, int lane, int h, const uint4& hp0, const uint4& hp1, const float (&x)[kNrcIn], uint4 (&bLive)[4]) {
    float enc[16];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float ob[4];
        oneblob4(h ? x[4 + 2 * i] : x[3 + 2 * i], ob);
#pragma unroll
        for (int r = 0; r < 4; ++r) enc[4 * i + r] = ob[r];
    }
    {
        float ob[4];
        oneblob4(x[7], ob);
#pragma unroll
        for (int r = 0; r < 4; ++r) enc[8 + r] = h ? x[8 + r] : ob[r];
    }
    enc[12] = h ? 1.0f : x[12]; enc[13] = h ? 1.0f : x[13]; enc[14] = 1.0f; enc[15] = 1.0f;
    uint4 b[4];
    b[0] = hp0; b[1] = hp1; b[2] = pack8(enc); b[3] = pack8(enc + 8);
    f32x16 acc[2];
    layer64(ldsW, lane, b, acc);
    relu_to_operand(acc, bLive);
}
GFX_DEV void piped_slice_rest(const NrcDev& d, const uint4* ldsW, int lane, int h, uint4 (&bLive)[4], uint32_t col, uint32_t numData, float* __restrict__ predictions) {
    for (int layer = 1; layer < d.numHidden; ++layer) {
        f32x16 acc[2];
        layer64(ldsW + layer * (kMatFwdElems / 8), lane, bLive, acc);
        relu_to_operand(acc, bLive);
    }
    const uint4* fragsOut = ldsW + d.numHidden * (kMatFwdElems / 8);
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fragsOut[s * 64 + lane]), __builtin_bit_cast(bf16x8, bLive[s]), c, 0, 0, 0);
    if (h == 0 && col < numData) {
        float* o = predictions + static_cast<size_t>(col) * kNrcOut;
        o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
    }
}
// The slice's own operands: the two parked K steps of (tile st, batch half snt) and inputs 3 .. 13 of the lane's query -- eleven loads issued
// as inline assembly, so that the COMPILER does not know they are in flight: it would wait for them with vmcnt(0) at their first use, behind
// the DMA instructions that are issued after them (it does not count across the blocks of this loop).  piped_slice_loads_wait() is the wait
// that belongs to them: all but the `dmaBehind` newer instructions (vmcnt retires in order).  Parked words are read past the L1 (sc1): this
// wave wrote them one pass ago and an L1 line from two passes ago may still be resident.
typedef float PipedF4 __attribute__((ext_vector_type(4)));
typedef float PipedF3 __attribute__((ext_vector_type(3)));
struct PipedLoads { uint32_t p[8]; PipedF4 xa, xb; PipedF3 xc; };
GFX_DEV void piped_slice_loads(const uint32_t* park, uint32_t parity, int st, int snt, int lane, const float* __restrict__ inputs, uint32_t col, uint32_t numData, PipedLoads& v) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t* src = park + piped_index(parity, st, snt, k >> 2, k & 3, lane);
        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v.p[k]) : "v"(src) : "memory");
    }
    const float* xs = inputs + static_cast<size_t>(col < numData ? col : numData - 1u) * kNrcIn + 3;       // (a column past the batch reads the last query's inputs; its output is not stored)
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v.xa) : "v"(xs) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v.xb) : "v"(xs + 4) : "memory");
    asm volatile("global_load_dwordx3 %0, %1, off" : "=v"(v.xc) : "v"(xs + 8) : "memory");
}
#define GFX_STR2(x) #x
#define GFX_STR(x) GFX_STR2(x)
template <int DMA_BEHIND>
GFX_DEV void piped_slice_loads_wait(PipedLoads& v) {
    asm volatile("s_waitcnt vmcnt(%11)"
                 : "+v"(v.p[0]), "+v"(v.p[1]), "+v"(v.p[2]), "+v"(v.p[3]), "+v"(v.p[4]), "+v"(v.p[5]), "+v"(v.p[6]), "+v"(v.p[7]), "+v"(v.xa), "+v"(v.xb),
                   "+v"(v.xc)
                 : "n"(DMA_BEHIND));          // (no "memory" clobber: behind one the compiler waits for the DMA before the slice's first LDS read)
}
GFX_DEV void piped_unpack(const PipedLoads& v, uint4& hp0, uint4& hp1, float (&x)[kNrcIn]) {
    hp0 = make_uint4(v.p[0], v.p[1], v.p[2], v.p[3]); hp1 = make_uint4(v.p[4], v.p[5], v.p[6], v.p[7]);
    x[0] = x[1] = x[2] = 0.0f;
    x[3] = v.xa.x; x[4] = v.xa.y; x[5] = v.xa.z; x[6] = v.xa.w; x[7] = v.xb.x; x[8] = v.xb.y; x[9] = v.xb.z; x[10] = v.xb.w;
    x[11] = v.xc[0]; x[12] = v.xc[1]; x[13] = v.xc[2];
}
__global__ __launch_bounds__(kStagedBlock) __attribute__((amdgpu_waves_per_eu(kStagedBlock / 256, kStagedBlock / 256)))
void k_nrc_infer_piped(NrcDev d, const uint16_t* __restrict__ fwd, const uint32_t* __restrict__ grid, const float* __restrict__ inputs,
                       uint32_t numDataArg, const uint32_t* __restrict__ numDataPtr, float* __restrict__ predictions, uint32_t* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) uint4 ldsAll[];         // kStagedTableBytes: a level's table
    // the weight fragments in an LDS object of their own: the compiler must be able to tell a slice's ds_reads from the table the DMA is
    // filling (reads of the array the DMA writes wait for it: vmcnt(0) before the first of them, and the overlap is gone)
    __shared__ __attribute__((aligned(16))) uint4 ldsWeights[(2 * kMatFwdElems + kOutFwdElems) / 8];
    const uint32_t numData = numDataPtr ? min(*numDataPtr, numDataArg) : numDataArg;
    const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t numTiles = (numData + 63) / 64;
    constexpr uint32_t kWaves = kStagedBlock / 64, kTilesPerPass = kWaves * kStagedTiles;
    constexpr int kCopiesPerWave = static_cast<int>((kStagedTableBytes / 1024u + kWaves - 1u) / kWaves);    // 1-KiB DMA instructions of a full table per wave
    const uint32_t rounds = (numTiles + gridDim.x * kTilesPerPass - 1) / (gridDim.x * kTilesPerPass);
    const uint32_t numPasses = gridDim.x * rounds;
    const uint32_t passTiles = (numTiles + numPasses - 1) / numPasses;
    const uint32_t fwdElems = d.numHidden * kMatFwdElems + kOutFwdElems;
    const uint32_t* ldsTable = reinterpret_cast<const uint32_t*>(ldsAll);
    uint4* ldsW = ldsWeights;
    for (uint32_t i = threadIdx.x; i < fwdElems / 8; i += kStagedBlock) ldsW[i] = reinterpret_cast<const uint4*>(fwd)[i];   // (the first level's barriers publish them)
    uint32_t* park = scratch + static_cast<size_t>(blockIdx.x * kWaves + static_cast<uint32_t>(wave)) * kPipedWordsPerWave;
    bool havePrev = false;
    uint32_t prevTile0 = 0, prevTileEnd = 0, parity = 0;
    uint4 bLive[4] = { make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u) };   // a slice's first-layer output on its way to its second part
    for (uint32_t pass = blockIdx.x; pass < numPasses && pass * passTiles < numTiles; pass += gridDim.x) {
        const uint32_t tile0 = pass * passTiles + wave, tileEnd = min((pass + 1) * passTiles, numTiles);
        const bool lastSlotUsed = tile0 + (kStagedTiles - 1) * kWaves < tileEnd;
        float px[kStagedTiles], py[kStagedTiles], pz[kStagedTiles];
#pragma unroll
        for (int t = 0; t < kStagedTiles; ++t) {
            const uint32_t tile = tile0 + t * kWaves;
            const size_t col = static_cast<size_t>(tile) * 64 + lane;
            const bool ok = tile < tileEnd && col < numData;
            px[t] = ok ? inputs[col * kNrcIn] : 0.0f; py[t] = ok ? inputs[col * kNrcIn + 1] : 0.0f; pz[t] = ok ? inputs[col * kNrcIn + 2] : 0.0f;
        }
        uint4 hb[kStagedTiles][2];                            // [tile][batch half]: the K step being filled (four word positions)
#pragma unroll
        for (int t = 0; t < kStagedTiles; ++t) { hb[t][0] = make_uint4(0u, 0u, 0u, 0u); hb[t][1] = make_uint4(0u, 0u, 0u, 0u); }
        for (int pos = 0; pos < 8; ++pos) {
            uint32_t w0[kStagedTiles], w1[kStagedTiles];
#pragma unroll
            for (int t = 0; t < kStagedTiles; ++t) { w0[t] = 0u; w1[t] = 0u; }
#pragma unroll
            for (int hL = 0; hL < 2; ++hL) {
                const int L = 4 * (pos >> 1) + (pos & 1) + 2 * hL;
                // ---- slice 2 pos + hL of the previous pass: tile st, batch half snt, part 0 or 1; its loads go out ahead of the DMA
                const int li = 2 * pos + hL, st = li >> 2, snt = (li >> 1) & 1, part = li & 1;
                const uint32_t ptile = prevTile0 + static_cast<uint32_t>(st) * kWaves;
                const bool slice = havePrev && ptile < prevTileEnd;                 // wave-uniform
                const uint32_t col = ptile * 64u + 32u * static_cast<uint32_t>(snt) + static_cast<uint32_t>(n);
                PipedLoads pl;
#pragma unroll
                for (int k = 0; k < 8; ++k) pl.p[k] = 0u;
                pl.xa = PipedF4{ 0.0f, 0.0f, 0.0f, 0.0f }; pl.xb = pl.xa; pl.xc = PipedF3{ 0.0f, 0.0f, 0.0f };
                if (slice && part == 0) piped_slice_loads(park, parity ^ 1u, st, snt, lane, inputs, col, numData, pl);
                NrcLevel lv = d.levels[L];
                // the table of the level before is no longer read: every wave's ds_reads of it have returned (their values were used).  A bare
                // s_barrier, not __syncthreads(): its fence would wait for the slice's loads here, ahead of the DMA they are meant to fly beside
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                {
                    // the level's table: a FIXED number of 1-KiB DMA instructions per wave (chunks past the table's end re-copy its last chunk:
                    // the same bytes to the same place), so that the compiler can count them -- the wait for the slice's loads above is then
                    // "all but the DMA instructions behind them", not "everything"
                    const char* src = reinterpret_cast<const char*>(grid + lv.offset);
                    const uint32_t lastChunk = lv.entries * 4u / 1024u - 1u;         // tables are whole KiB (nrc_infer checks before it picks this kernel)
#pragma unroll
                    for (int j = 0; j < kCopiesPerWave; ++j) {
                        const uint32_t chunk = min(static_cast<uint32_t>(wave) + static_cast<uint32_t>(j) * kWaves, lastChunk);
                        typedef const __attribute__((address_space(1))) void* GlobalPtr;
                        typedef __attribute__((address_space(3))) void* LdsPtr;
                        __builtin_amdgcn_global_load_lds((GlobalPtr)(src + chunk * 1024u + lane * 16), (LdsPtr)(reinterpret_cast<char*>(ldsAll) + chunk * 1024u), 16, 0, 0);
                    }
                }
                if (slice) {
                    if (part == 0) {
                        piped_slice_loads_wait<kCopiesPerWave>(pl);             // the slice's loads have landed; the table's DMA may still be in flight
                        uint4 hp0, hp1; float x[kNrcIn];
                        piped_unpack(pl, hp0, hp1, x);
                        piped_slice_features_and_first_layer(ldsW, lane, h, hp0, hp1, x, bLive);
                    }
                    else piped_slice_rest(d, ldsW, lane, h, bLive, col, numData, predictions);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                {
                    lv.offset = 0u;
                    const bool dense = static_cast<unsigned long long>(lv.res) * lv.res * lv.res <= lv.entries, pow2 = (lv.entries & (lv.entries - 1)) == 0;
                    if (pow2 && !dense) staged_level<0, 1>(lv, ldsTable, px, py, pz, h, hL, lastSlotUsed, w0, w1);
                    else if (pow2) staged_level<1, 1>(lv, ldsTable, px, py, pz, h, hL, lastSlotUsed, w0, w1);
                    else staged_level<-1, -1>(lv, ldsTable, px, py, pz, h, hL, lastSlotUsed, w0, w1);
                }
            }
            switch (pos & 3) {                              // block-uniform: word position of the K step being filled
#define GFX_PIPED_CASE(P) case P: _Pragma("unroll") for (int t = 0; t < kStagedTiles; ++t) { \
                staged_set_word(hb[t][0], (P), w0[t]); staged_set_word(hb[t][1], (P), w1[t]); } break;
            GFX_PIPED_CASE(0) GFX_PIPED_CASE(1) GFX_PIPED_CASE(2) GFX_PIPED_CASE(3)
#undef GFX_PIPED_CASE
            default: break;
            }
            if ((pos & 3) == 3) {                           // a K step is complete: park it (coalesced 4-byte planes), start the next
#pragma unroll
                for (int t = 0; t < kStagedTiles; ++t)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        park[piped_index(parity, t, nt, pos >> 2, 0, lane)] = hb[t][nt].x; park[piped_index(parity, t, nt, pos >> 2, 1, lane)] = hb[t][nt].y;
                        park[piped_index(parity, t, nt, pos >> 2, 2, lane)] = hb[t][nt].z; park[piped_index(parity, t, nt, pos >> 2, 3, lane)] = hb[t][nt].w;
                        hb[t][nt] = make_uint4(0u, 0u, 0u, 0u);
                    }
            }
        }
        havePrev = true; prevTile0 = tile0; prevTileEnd = tileEnd; parity ^= 1u;
    }
    // ---- the last pass's table-free half: nothing left to run it under
    if (havePrev) {
        for (int li = 0; li < 4 * kStagedTiles; ++li) {
            const int st = li >> 2, snt = (li >> 1) & 1, part = li & 1;
            const uint32_t ptile = prevTile0 + static_cast<uint32_t>(st) * kWaves;
            if (ptile >= prevTileEnd) continue;                                      // wave-uniform
            const uint32_t col = ptile * 64u + 32u * static_cast<uint32_t>(snt) + static_cast<uint32_t>(n);
            if (part == 0) {
                PipedLoads pl;
                piped_slice_loads(park, parity ^ 1u, st, snt, lane, inputs, col, numData, pl);
                piped_slice_loads_wait<0>(pl);
                uint4 hp0, hp1; float x[kNrcIn];
                piped_unpack(pl, hp0, hp1, x);
                piped_slice_features_and_first_layer(ldsW, lane, h, hp0, hp1, x, bLive);
            }
            else piped_slice_rest(d, ldsW, lane, h, bLive, col, numData, predictions);
        }
    }
}

// ---------------------------------------------------------------- training step
// One wave per block, 64 batch columns per block.  Weight fragments are read straight from the packed
// global images (each is used once per block); LDS keeps every layer's activations twice: in operand
// order (ReLU masks) and as a [feature][batch] matrix (the K = batch contraction of dL/dW).
struct NrcTrainArgs {
    NrcDev d;
    const uint16_t* fwd; const uint16_t* bwd; const uint32_t* grid;
    const float* inputs; const float* targets; uint32_t numData;
    float* gradPartials;      // [numBlocks][mlpParams]
    float* gridGrad;          // [gridParams] fp32 atomics, or (mode 1) one fp16 pair per entry in the first half
    int gridGradMode;         // kGridGradF32Atomics / kGridGradF16Atomics / kGridGradLdsTables
    float2* gridDelta;        // mode 2: [16 levels][numData] dL/d(the level's two features), loss-scaled, for k_nrc_grid_scatter
    float* lossSum;
};
GFX_DEV void store_transposed(uint16_t* ldsT, int nt, int n, int h, const uint4 b[4]) {   // operand -> [feature][batch]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint32_t w[4] = { b[s].x, b[s].y, b[s].z, b[s].w };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = nrc_feature_of_slot(s, h, i);
            ldsT[f * kTStride + 32 * nt + n] = static_cast<uint16_t>((w[i >> 1] >> (16 * (i & 1))) & 0xFFFFu);
        }
    }
}
// dW [outRows x 64] of one layer: A = delta^T (M = out feature, K = batch), B = act^T (K = batch, N = in feature)
template <int KSTEPS>   // K = batch: 16 records per MFMA step
GFX_DEV void weight_gradient(const uint16_t* ldsDelta, const uint16_t* ldsAct, int lane, int outTiles, int outRows, float* gradOut) {
    const int m = lane & 31, h = lane >> 5;
    for (int mt = 0; mt < outTiles; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x16 c;
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const uint4 a = *reinterpret_cast<const uint4*>(ldsDelta + (32 * mt + m) * kTStride + 16 * ks + 8 * h);
                const uint4 b = *reinterpret_cast<const uint4*>(ldsAct + (32 * nt + m) * kTStride + 16 * ks + 8 * h);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < outRows) gradOut[row * 64 + 32 * nt + m] = c[r];
            }
        }
}

// Precision contract of the packed grid gradient (the default; GFX_NRC_GRID_GRAD=f32 selects the fp32 atomics): every
// contribution w_c * dL/dfeature (loss-scaled by 128) is clamped to the fp16 range, rounded to fp16 (nearest even) and
// added by the L2 atomic unit in fp16, so an entry that receives K contributions carries a relative error of the order
// sqrt(K) * 2^-11 in its gradient sum (order-dependent, like the fp32 atomics) and contributions below 2^-24 * 128 vanish.
// How the hash-grid gradient is summed (NrcNet::gridGradMode; GFX_NRC_GRID_GRAD = f32 | f16atomic | lds):
//   0  two fp32 atomics per corner (global_atomic_add_f32)
//   1  one packed-fp16 atomic per corner (global_atomic_pk_add_f16; what tiny-cuda-nn does with __half2) -- rounds 1-3
//   2  the default: no global atomics.  The L2s retire these atomics at ~20 G/s in this access pattern whatever the launch shape
//      (tools/microbench/pk_atomics.hip: 2.1 M atomics of a training step = 104 us, 138-307 us when the records cluster), which was
//      72-76 % of k_nrc_train (profiles/r04_nrc_train_profile_before.jsonl).  Instead k_nrc_train writes dL/dfeature per record and
//      level (coalesced), and k_nrc_grid_scatter -- one block per (level, chunk of records) -- sums a whole level table (<= 32 768
//      packed-fp16 words = 128 KiB) in LDS with ds_pk_add_f16 and writes it out as one coalesced run; the optimizer adds the chunks'
//      tables in fp32.  Same rounding points as mode 1 per contribution (clamp, fp16), fp16 sums inside a chunk, fp32 across chunks.
constexpr int kGridGradF32Atomics = 0, kGridGradF16Atomics = 1, kGridGradLdsTables = 2;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
GFX_DEV void grid_grad_add_f16x2(uint32_t* word, float g0, float g1) {
    f16x2 v;
    v.x = static_cast<_Float16>(fmin2(fmax2(g0, -65504.0f), 65504.0f));
    v.y = static_cast<_Float16>(fmin2(fmax2(g1, -65504.0f), 65504.0f));
    typedef __attribute__((address_space(1))) f16x2* GlobalF16x2;
    (void)__builtin_amdgcn_global_atomic_fadd_v2f16((GlobalF16x2)word, v);
}

// NT = N tiles (of 32 records) per wave: 2 = 64 records per block (rounds 1-3), 1 = 32 records per block -- twice the blocks, two waves
// per CU for the reference's 16 384-record step, half the dependent gathers per wave (the step is a latency chain of one wave per CU).
template <int NT>
__global__ __launch_bounds__(64) void k_nrc_train(NrcTrainArgs a) {
    constexpr int kTile = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) uint4 ldsT4[];
    const NrcDev& d = a.d;
    const int numLayers = d.numHidden + 1;                       // activation sets: encoded input + hidden outputs
    uint4* ldsOp = ldsT4;                                        // [numLayers][nt][s][lane] operand order
    uint16_t* ldsActT = reinterpret_cast<uint16_t*>(ldsOp + numLayers * NT * 4 * 64);   // [numLayers][64][kTStride]
    uint16_t* ldsDeltaT = ldsActT + numLayers * 64 * kTStride;   // [64][kTStride]
    const int lane = threadIdx.x, h = lane >> 5, n = lane & 31;
    const uint32_t tile = blockIdx.x;
    const uint4* fwd4 = reinterpret_cast<const uint4*>(a.fwd);
    const uint4* bwd4 = reinterpret_cast<const uint4*>(a.bwd);
    const uint32_t mlpParams = d.numHidden * 4096 + kNrcOutPad * 64;
    float* gradOut = a.gradPartials + static_cast<size_t>(tile) * mlpParams;

    // ---- forward
    GFX_CYC_BEGIN
    GFX_CYC(0);   // inputs + encoding (hash-grid gathers, one-blob)
    uint4 b[NT][4];
    float xpos[NT][3];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const uint32_t col = tile * kTile + 32 * nt + n;
        float x[kNrcIn];
#pragma unroll
        for (int k = 0; k < kNrcIn; ++k) x[k] = col < a.numData ? a.inputs[static_cast<size_t>(col) * kNrcIn + k] : 0.0f;
        xpos[nt][0] = x[0]; xpos[nt][1] = x[1]; xpos[nt][2] = x[2];
        float enc[32];
        encode_half(d, a.grid, x, h, enc);
        to_operand(enc, b[nt]);
#pragma unroll
        for (int s = 0; s < 4; ++s) ldsOp[((0 * NT + nt) * 4 + s) * 64 + lane] = b[nt][s];
        store_transposed(ldsActT, nt, n, h, b[nt]);
    }
    GFX_CYC(1);   // hidden layers forward (weight fragments from L2, MFMA, activations to LDS twice)
    for (int layer = 0; layer < d.numHidden; ++layer) {
        const uint4* frags = fwd4 + layer * (kMatFwdElems / 8);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc[2];
            layer64(frags, lane, b[nt], acc);
            relu_to_operand(acc, b[nt]);
#pragma unroll
            for (int s = 0; s < 4; ++s) ldsOp[(((layer + 1) * NT + nt) * 4 + s) * 64 + lane] = b[nt][s];
            store_transposed(ldsActT + (layer + 1) * 64 * kTStride, nt, n, h, b[nt]);
        }
    }
    // ---- output layer + loss gradient (RelativeL2Luminance)
    GFX_CYC(2);   // output layer, loss, loss gradient
    uint4 delta[NT][4];
    float lossLocal = 0.0f;
    {
        const uint4* fragsOut = fwd4 + d.numHidden * (kMatFwdElems / 8);
        const float nTotal = static_cast<float>(a.numData) * kNrcOut;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 c;
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fragsOut[s * 64 + lane]),
                                                            __builtin_bit_cast(bf16x8, b[nt][s]), c, 0, 0, 0);
            float dv[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
            const uint32_t col = tile * kTile + 32 * nt + n;
            if (h == 0 && col < a.numData) {
                const float* t = a.targets + static_cast<size_t>(col) * kNrcOut;
                const float lum = 0.299f * c[0] + 0.587f * c[1] + 0.114f * c[2];
                const float denom = lum * lum + 0.01f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float diff = c[k] - t[k];
                    lossLocal += (diff * diff / denom) / nTotal;
                    dv[k] = kLossScale * 2 * diff / denom / nTotal;
                }
            }
            delta[nt][0] = pack8(dv);                        // features 0..2 = slots (s 0, h 0, i 0..2)
            delta[nt][1] = make_uint4(0, 0, 0, 0); delta[nt][2] = make_uint4(0, 0, 0, 0); delta[nt][3] = make_uint4(0, 0, 0, 0);
            store_transposed(ldsDeltaT, nt, n, h, delta[nt]);
        }
    }
    for (int off = 32; off >= 1; off >>= 1) lossLocal += __shfl_xor(lossLocal, off);
    if (lane == 0) atomicAdd(a.lossSum, lossLocal);
    __syncthreads();

    // ---- backward
    // output matrix: dWout = delta_out . h_last^T; delta_last = (Wout^T . delta_out) * relu'(h_last)
    GFX_CYC(3);   // backward through the layers: dW (MFMA over the batch, partials to HBM), delta (MFMA, ReLU masks from LDS)
    weight_gradient<2 * NT>(ldsDeltaT, ldsActT + d.numHidden * 64 * kTStride, lane, 1, kNrcOutPad, gradOut + d.numHidden * 4096);
    __syncthreads();
    {
        const uint4* fragsT = bwd4 + d.numHidden * (kMatFwdElems / 8);     // WoutT: fragments (mt, s), s in {0, 1}
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x16 c;
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = 0.0f;
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fragsT[(mt * 2 + s) * 64 + lane]),
                                                                __builtin_bit_cast(bf16x8, delta[nt][s]), c, 0, 0, 0);
                acc[mt] = c;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float act[8], v[8];
                unpack8(ldsOp[((d.numHidden * NT + nt) * 4 + s) * 64 + lane], act);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = act[i] > 0.0f ? acc[s >> 1][8 * (s & 1) + i] : 0.0f;
                delta[nt][s] = pack8(v);
            }
            store_transposed(ldsDeltaT, nt, n, h, delta[nt]);
        }
    }
    __syncthreads();
    for (int layer = d.numHidden - 1; layer >= 0; --layer) {
        // dW_layer = delta_{layer+1} . act_layer^T
        weight_gradient<2 * NT>(ldsDeltaT, ldsActT + layer * 64 * kTStride, lane, 2, 64, gradOut + layer * 4096);
        __syncthreads();
        if (layer == 0 && d.posEnc != 1) break;
        const uint4* fragsT = bwd4 + layer * (kMatFwdElems / 8);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc[2];
            layer64(fragsT, lane, delta[nt], acc);
            if (layer > 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float act[8], v[8];
                    unpack8(ldsOp[((layer * NT + nt) * 4 + s) * 64 + lane], act);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = act[i] > 0.0f ? acc[s >> 1][8 * (s & 1) + i] : 0.0f;
                    delta[nt][s] = pack8(v);
                }
                store_transposed(ldsDeltaT, nt, n, h, delta[nt]);
            }
            else {
                GFX_CYC(4);   // hash-grid gradient scatter (grid_corners again + one packed atomic per corner)
                // dL/d(encoded input), fp32: scatter the hash-grid part.  Owned group q < 4 (canonical
                // features 8 q + 4 h .. + 3 = levels 2 g, 2 g + 1 with g = 2 q + h) sits in M tile 0,
                // registers 8 (q >> 1) + 4 (q & 1) + r.
                const uint32_t col = tile * kTile + 32 * nt + n;
                if (col < a.numData) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int g = 2 * q + h;
                        const int base = 8 * (q >> 1) + 4 * (q & 1);
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const NrcLevel lv = d.levels[2 * g + k];
                            const float d0 = acc[0][base + 2 * k], d1 = acc[0][base + 2 * k + 1];
                            if (a.gridGradMode == kGridGradLdsTables) {      // consecutive lanes = consecutive records: coalesced 8-byte stores
                                a.gridDelta[static_cast<size_t>(2 * g + k) * a.numData + col] = make_float2(d0, d1);
                                continue;
                            }
                            if (d0 == 0.0f && d1 == 0.0f) continue;
                            uint32_t idx[8]; float w[8];
                            grid_corners(lv, xpos[nt][0], xpos[nt][1], xpos[nt][2], idx, w);
                            if (a.gridGradMode == kGridGradF16Atomics) {
                                // one packed-fp16 atomic per corner (global_atomic_pk_add_f16; tiny-cuda-nn scatters __half2
                                // the same way): the two features of an entry share a 32-bit word
#pragma unroll
                                for (int c = 0; c < 8; ++c) grid_grad_add_f16x2(reinterpret_cast<uint32_t*>(a.gridGrad) + idx[c], w[c] * d0, w[c] * d1);
                            }
                            else {
#pragma unroll
                                for (int c = 0; c < 8; ++c) {
                                    atomicAdd(a.gridGrad + 2ull * idx[c], w[c] * d0);
                                    atomicAdd(a.gridGrad + 2ull * idx[c] + 1, w[c] * d1);
                                }
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    GFX_CYC_END;
}
#ifdef GFX_LANE_PROFILE   // experiment builds only (gm_math.hip.h GFX_CYC, tools/nrc_train_profile.py)
extern "C" int gfx_debug_nrc_profile(unsigned long long* out64, int reset) {
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_laneProfile), sizeof(g_laneProfile)) != hipSuccess) return 1;
    if (reset) { unsigned long long zero[64] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_laneProfile), zero, sizeof(zero)) != hipSuccess) return 1; }
    return 0;
}
#endif

// ---------------------------------------------------------------- hash-grid gradient: one level table per block, in LDS
// grid (numChunks, 16 levels) x 256 threads; dynamic LDS = the level's entries x 4 B + a staging area of kScatterBatch records x 20 B.
// partials: [numChunks][all entries] packed fp16 pairs.
// The sum is made in a DEFINED order since round 5: the block's four waves copy a batch of records (the level's two deltas and the
// position) into the staging area, coalesced; then ONE wave adds them to the table, 64 records per step in record order, corner 0 .. 7 of
// all 64 after one another -- LDS atomics of one wave execute in program order and the lanes of one instruction that meet in an entry are
// served in a fixed order, so the fp16 sums no longer depend on how four waves happened to interleave.  Two runs of a training step
// from the same state give the same parameters bit for bit (tests/test_gpu_nrc_net.py), which is what lets every rank of a band-split
// NRC frame train its own copy of the network on the gathered batch instead of waiting for rank 0's (nrc_driver.cpp).  The adding wave's
// part is ~3 us per block next to the 128-KiB table it zeroes and writes out; the step's time does not change.
constexpr int kScatterBlock = 256, kScatterBatch = 1024;
__global__ __launch_bounds__(kScatterBlock) void k_nrc_grid_scatter(NrcDev d, const float* __restrict__ inputs, const float2* __restrict__ gridDelta,
                                                                    uint32_t numData, uint32_t chunkRecords, uint32_t totalEntries, uint32_t tableWords,
                                                                    uint32_t* __restrict__ partials) {
    extern __shared__ uint32_t ldsTable[];
    float* stage = reinterpret_cast<float*>(ldsTable + tableWords);          // [5][kScatterBatch]: delta.x, delta.y, position
    const NrcLevel lv = d.levels[blockIdx.y];
    for (uint32_t e = threadIdx.x; e < lv.entries; e += kScatterBlock) ldsTable[e] = 0u;
    const uint32_t begin = blockIdx.x * chunkRecords, end = min(begin + chunkRecords, numData);
    for (uint32_t batch = begin; batch < end; batch += kScatterBatch) {
        const uint32_t count = min(static_cast<uint32_t>(kScatterBatch), end - batch);
        __syncthreads();                                   // the table is zero / the previous batch has been added
        for (uint32_t i = threadIdx.x; i < count; i += kScatterBlock) {
            const float2 dl = gridDelta[static_cast<size_t>(blockIdx.y) * numData + batch + i];
            const float* x = inputs + static_cast<size_t>(batch + i) * kNrcIn;
            stage[i] = dl.x; stage[kScatterBatch + i] = dl.y;
            stage[2 * kScatterBatch + i] = x[0]; stage[3 * kScatterBatch + i] = x[1]; stage[4 * kScatterBatch + i] = x[2];
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            for (uint32_t i = threadIdx.x; i < count; i += 64) {     // record order: 64 consecutive records per step
                const float dx = stage[i], dy = stage[kScatterBatch + i];
                if (dx == 0.0f && dy == 0.0f) continue;
                uint32_t idx[8]; float w[8];
                grid_corners(lv, stage[2 * kScatterBatch + i], stage[3 * kScatterBatch + i], stage[4 * kScatterBatch + i], idx, w);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    f16x2 v;
                    v.x = static_cast<_Float16>(fmin2(fmax2(w[c] * dx, -65504.0f), 65504.0f));
                    v.y = static_cast<_Float16>(fmin2(fmax2(w[c] * dy, -65504.0f), 65504.0f));
                    typedef __attribute__((address_space(3))) f16x2* LdsF16x2;
                    (void)__builtin_amdgcn_ds_atomic_fadd_v2f16((LdsF16x2)(ldsTable + (idx[c] - lv.offset)), v);
                }
            }
        }
    }
    __syncthreads();
    uint32_t* out = partials + static_cast<size_t>(blockIdx.x) * totalEntries + lv.offset;
    for (uint32_t e = threadIdx.x; e < lv.entries; e += kScatterBlock) out[e] = ldsTable[e];
}

// ---------------------------------------------------------------- dW partials -> one gradient
// The training blocks leave one dW partial each (256-512 per step); summing them per parameter inside the optimizer was a chain of
// that many dependent-latency loads in the 9 216 threads that own an MLP weight while the million grid threads had long finished.
// Here 16 threads share a parameter (each sums every 16th partial), the 16 sums are added in slice order through LDS: a defined fp32 order.
constexpr int kReduceBlock = 256, kReduceSlices = 16;
__global__ __launch_bounds__(kReduceBlock) void k_nrc_reduce_partials(const float* __restrict__ partials, uint32_t numPartials, uint32_t mlpParams, float* __restrict__ gradSum) {
    __shared__ float lds[kReduceBlock];
    const uint32_t pi = threadIdx.x & (kReduceBlock / kReduceSlices - 1), slice = threadIdx.x / (kReduceBlock / kReduceSlices);
    const uint32_t p = blockIdx.x * (kReduceBlock / kReduceSlices) + pi;
    float g = 0.0f;
    if (p < mlpParams)
        for (uint32_t k = slice; k < numPartials; k += kReduceSlices) g += partials[static_cast<size_t>(k) * mlpParams + p];
    lds[threadIdx.x] = g;
    __syncthreads();
    if (slice == 0 && p < mlpParams) {
        float sum = 0.0f;
#pragma unroll
        for (int sl = 0; sl < kReduceSlices; ++sl) sum += lds[sl * (kReduceBlock / kReduceSlices) + pi];
        gradSum[p] = sum;
    }
}

// ---------------------------------------------------------------- optimizer: Adam + EMA
struct NrcOptArgs {
    NrcDev d;
    float* params; float* adamM; float* adamV; float* ema;
    const float* gradPartials; uint32_t mlpParams;      // the summed dW (k_nrc_reduce_partials)
    float* gridGrad;
    int gridGradMode;         // mode 1: cleared by the caller afterwards (two parameters share a word)
    const uint32_t* gridPartials; uint32_t numGridChunks, totalEntries;   // mode 2: k_nrc_grid_scatter's tables
    float lrT, beta1, beta2, eps, l2Reg, emaDecay, debiasOld, debiasNew;
};
__global__ void k_nrc_optimizer(NrcOptArgs a) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.d.total) return;
    const bool isGrid = p >= a.d.gridOff;
    float g;
    if (isGrid && a.gridGradMode == kGridGradLdsTables) {
        const uint32_t q = p - a.d.gridOff;
        g = 0.0f;
        for (uint32_t c = 0; c < a.numGridChunks; ++c) {        // chunk order: a defined fp32 sum
            const uint32_t word = a.gridPartials[static_cast<size_t>(c) * a.totalEntries + (q >> 1)];
            const uint16_t bits = static_cast<uint16_t>((q & 1u) ? word >> 16 : word & 0xFFFFu);
            g += static_cast<float>(__builtin_bit_cast(_Float16, bits));
        }
    }
    else if (isGrid && a.gridGradMode == kGridGradF16Atomics) {
        const uint32_t q = p - a.d.gridOff;
        const uint32_t word = reinterpret_cast<const uint32_t*>(a.gridGrad)[q >> 1];
        const uint16_t bits = static_cast<uint16_t>((q & 1u) ? word >> 16 : word & 0xFFFFu);
        g = static_cast<float>(__builtin_bit_cast(_Float16, bits));
    }
    else if (isGrid) { g = a.gridGrad[p - a.d.gridOff]; a.gridGrad[p - a.d.gridOff] = 0.0f; }
    else g = a.gradPartials[p];                                  // k_nrc_reduce_partials' sum
    float grad = g / kLossScale;
    float w = a.params[p];
    if (!(isGrid && grad == 0.0f)) {       // untouched hash-grid entries keep their moments
        if (!isGrid) grad = grad + a.l2Reg * w;
        const float m = a.beta1 * a.adamM[p] + (1 - a.beta1) * grad;
        const float v = a.beta2 * a.adamV[p] + (1 - a.beta2) * grad * grad;
        a.adamM[p] = m; a.adamV[p] = v;
        w = w - a.lrT * m / (sqrtf(v) + a.eps);
        a.params[p] = w;
    }
    a.ema[p] = ((1 - a.emaDecay) * w + a.emaDecay * a.debiasOld * a.ema[p]) * a.debiasNew;
}

// ---------------------------------------------------------------- host side
struct NrcNet {
    NrcDev d;
    float learningRate;
    uint32_t step = 0;
    uint32_t mlpParams = 0, gridParams = 0;
    DevBuf params, adamM, adamV, ema, gradPartials, gradSum, gridGrad, lossSum, gridDelta, gridPartials;
    DevBuf packTrainFwd, packTrainBwd, packInferFwd, gridTrain, gridInfer;
    DevBuf pipeScratch;                      // k_nrc_infer_piped: the parked hash operands of a pass (32 KiB per wave)
    uint32_t partialCapacity = 0;
    // the inference images (bf16 fragments + grid of the EMA weights) are packed when somebody asks for them, not after every step:
    // a frame trains four steps and infers once
    bool inferDirty = false;
    hipEvent_t trained = nullptr;            // recorded behind the last training step (its optimizer): whoever packs the inference images waits for it
    hipStream_t packStream = nullptr;        // gfx_nrc_inference_image without a stream packs here and waits for it on the host
    int gridGradMode = kGridGradLdsTables;   // GFX_NRC_GRID_GRAD = f32 | f16atomic | lds at creation
    size_t scatterLdsConfigured = 0;
};

static void nrc_levels(NrcDev& d) {
    uint32_t offset = 0;
    for (int l = 0; l < kHashLevels; ++l) {
        const float scale = exp2f(static_cast<float>(l) * log2f(2.0f)) * kBaseRes - 1.0f;
        const uint32_t res = static_cast<uint32_t>(ceilf(scale)) + 1;
        unsigned long long n = static_cast<unsigned long long>(res) * res * res;
        n = (n + 7) / 8 * 8;
        if (n > (1ull << kLog2Hashmap)) n = 1ull << kLog2Hashmap;
        d.levels[l] = NrcLevel{ scale, res, static_cast<uint32_t>(n), offset };
        offset += static_cast<uint32_t>(n);
    }
    d.total = d.gridOff + (d.posEnc == 1 ? offset * 2 : 0);
}

static void nrc_pack(Context& ctx, hipStream_t stream, NrcNet& net, bool training) {
    const uint32_t fwdElems = net.d.numHidden * kMatFwdElems + kOutFwdElems;
    const uint32_t bwdElems = net.d.numHidden * kMatFwdElems + kOutBwdElems;
    const uint32_t threads = std::max(fwdElems, bwdElems);
    const float* src = training ? net.params.as<float>() : net.ema.as<float>();
    uint16_t* fwd = training ? net.packTrainFwd.as<uint16_t>() : net.packInferFwd.as<uint16_t>();
    uint16_t* bwd = training ? net.packTrainBwd.as<uint16_t>() : nullptr;
    uint32_t* grid = training ? net.gridTrain.as<uint32_t>() : net.gridInfer.as<uint32_t>();
    ScopedKernelTimer timer(ctx, stream, "nrc_pack");
    hipLaunchKernelGGL(k_nrc_pack, dim3((threads + 255) / 256), dim3(256), 0, stream, net.d, src, fwd, bwd, grid);
    GFX_HIP(hipGetLastError());
}

void nrc_set_params(Context& ctx, hipStream_t stream, NrcNet* net, const float* hostParams, uint32_t count);

NrcNet* nrc_create(Context& ctx, int posEnc, uint32_t numHiddenLayers, float learningRate) {
    if (numHiddenLayers < 1 || numHiddenLayers > kMaxHidden) throw HipError("gfx_nrc_create: numHiddenLayers must be 1..5");
    if (posEnc != 0 && posEnc != 1) throw HipError("gfx_nrc_create: unknown position encoding");
    NrcNet* net = new NrcNet();
    std::memset(&net->d, 0, sizeof(net->d));
    net->d.posEnc = posEnc; net->d.numHidden = static_cast<int>(numHiddenLayers);
    net->learningRate = learningRate;
    net->mlpParams = numHiddenLayers * 4096 + kNrcOutPad * 64;
    net->d.gridOff = net->mlpParams;
    nrc_levels(net->d);
    net->gridParams = net->d.total - net->d.gridOff;
    if (const char* e = getenv("GFX_NRC_GRID_GRAD"))
        net->gridGradMode = std::strcmp(e, "f32") == 0 ? kGridGradF32Atomics : std::strcmp(e, "f16atomic") == 0 ? kGridGradF16Atomics : kGridGradLdsTables;
    const size_t bytes = sizeof(float) * net->d.total;
    net->params.reserve(bytes); net->adamM.reserve(bytes); net->adamV.reserve(bytes); net->ema.reserve(bytes);
    net->gridGrad.reserve(sizeof(float) * std::max<uint32_t>(net->gridParams, 4));
    net->lossSum.reserve(16);
    const uint32_t fwdElems = numHiddenLayers * kMatFwdElems + kOutFwdElems;
    const uint32_t bwdElems = numHiddenLayers * kMatFwdElems + kOutBwdElems;
    net->packTrainFwd.reserve(2 * fwdElems); net->packInferFwd.reserve(2 * fwdElems); net->packTrainBwd.reserve(2 * bwdElems);
    net->gridTrain.reserve(std::max<uint32_t>(2 * net->gridParams, 16)); net->gridInfer.reserve(std::max<uint32_t>(2 * net->gridParams, 16));
    GFX_HIP(hipMemset(net->adamM.p, 0, bytes)); GFX_HIP(hipMemset(net->adamV.p, 0, bytes));
    GFX_HIP(hipMemset(net->params.p, 0, bytes)); GFX_HIP(hipMemset(net->ema.p, 0, bytes));
    GFX_HIP(hipMemset(net->gridGrad.p, 0, sizeof(float) * std::max<uint32_t>(net->gridParams, 4)));
    // default initialisation: Xavier-uniform MLP weights, U(-1e-4, 1e-4) grid features from one PCG32
    // stream (state 1337, increment 1) -- the same stream as oracle/nrc_net.py init_params
    std::vector<float> init(net->d.total);
    uint64_t state = 1337;
    auto uniform = [&state]() {
        const uint64_t old = state;
        state = old * 6364136223846793005ull + 1ull;
        const uint32_t xorshifted = static_cast<uint32_t>(((old >> 18u) ^ old) >> 27u);
        const uint32_t rot = static_cast<uint32_t>(old >> 59u);
        const uint32_t v = (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
        const uint32_t bits = (v >> 9) | 0x3F800000u;
        float f; std::memcpy(&f, &bits, 4);
        return f - 1.0f;
    };
    for (uint32_t mat = 0; mat <= numHiddenLayers; ++mat) {
        const uint32_t rows = mat == numHiddenLayers ? kNrcOutPad : 64;
        const float bound = static_cast<float>(std::sqrt(6.0 / (64 + rows)));
        for (uint32_t k = 0; k < rows * 64; ++k) init[mat * 4096 + k] = (uniform() * 2.0f - 1.0f) * bound;
    }
    for (uint32_t k = net->d.gridOff; k < net->d.total; ++k) init[k] = (uniform() * 2.0f - 1.0f) * 1e-4f;
    nrc_set_params(ctx, nullptr, net, init.data(), net->d.total);
    return net;
}

void nrc_destroy(NrcNet* net) {
    if (!net) return;
    DevBuf* all[] = { &net->params, &net->adamM, &net->adamV, &net->ema, &net->gradPartials, &net->gradSum, &net->gridGrad, &net->lossSum, &net->gridDelta, &net->gridPartials,
                      &net->packTrainFwd, &net->packTrainBwd, &net->packInferFwd, &net->gridTrain, &net->gridInfer, &net->pipeScratch };
    for (DevBuf* b : all) b->release();
    if (net->trained) (void)hipEventDestroy(net->trained);
    if (net->packStream) (void)hipStreamDestroy(net->packStream);
    delete net;
}
uint32_t nrc_num_params(const NrcNet* net) { return net->d.total; }

// which: 0 = training parameters (also resets the EMA copy, the Adam moments and the step counter), 1 = EMA
void nrc_set_params(Context& ctx, hipStream_t stream, NrcNet* net, const float* hostParams, uint32_t count) {
    if (count != net->d.total) throw HipError("gfx_nrc_set_params: parameter count mismatch");
    const size_t bytes = sizeof(float) * count;
    GFX_HIP(hipMemcpyAsync(net->params.p, hostParams, bytes, hipMemcpyHostToDevice, stream));
    GFX_HIP(hipMemcpyAsync(net->ema.p, hostParams, bytes, hipMemcpyHostToDevice, stream));
    GFX_HIP(hipMemsetAsync(net->adamM.p, 0, bytes, stream));
    GFX_HIP(hipMemsetAsync(net->adamV.p, 0, bytes, stream));
    net->step = 0;
    nrc_pack(ctx, stream, *net, true);
    nrc_pack(ctx, stream, *net, false);
    net->inferDirty = false;
    GFX_HIP(hipStreamSynchronize(stream));
}
// The inference images brought up to date on `stream`, behind the training that made them stale (the event recorded after its optimizer): no
// stream handle of an earlier call is kept.
static void nrc_refresh_inference_images(Context& ctx, hipStream_t stream, NrcNet* net) {
    if (!net->inferDirty) return;
    if (net->trained) GFX_HIP(hipStreamWaitEvent(stream, net->trained, 0));
    nrc_pack(ctx, stream, *net, false);
    net->inferDirty = false;
}
void nrc_inference_image(Context& ctx, NrcNet* net, int which, void** dPtr, uint64_t* bytes, hipStream_t stream, bool onStream) {
    if (onStream) nrc_refresh_inference_images(ctx, stream, net);      // the caller uses the images in `stream` order
    else if (net->inferDirty) {                                        // no stream to order with: the images are complete when this returns
        if (!net->packStream) GFX_HIP(hipStreamCreateWithFlags(&net->packStream, hipStreamNonBlocking));
        nrc_refresh_inference_images(ctx, net->packStream, net);
        GFX_HIP(hipStreamSynchronize(net->packStream));
    }
    const uint32_t fwdElems = net->d.numHidden * kMatFwdElems + kOutFwdElems;
    if (which == 0) { *dPtr = net->packInferFwd.p; *bytes = 2ull * fwdElems; }
    else if (which == 1) { *dPtr = net->d.posEnc == 1 ? net->gridInfer.p : nullptr; *bytes = net->d.posEnc == 1 ? 2ull * net->gridParams : 0; }
    else throw HipError("gfx_nrc_inference_image: which must be 0 (MLP fragments) or 1 (hash grid)");
}

// Sum of the 32-bit words of both inference images, added to *dOut (integer adds: any order gives the same sum).
__global__ __launch_bounds__(256) void k_nrc_words_sum(const uint32_t* __restrict__ words, size_t n, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256) acc += words[i] * (static_cast<uint32_t>(i) | 1u);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
void nrc_params_checksum(Context& ctx, hipStream_t stream, NrcNet* net, uint32_t* dOut) {
    for (int which = 0; which < 2; ++which) {
        void* p = nullptr; uint64_t bytes = 0;
        nrc_inference_image(ctx, net, which, &p, &bytes, stream, true);
        if (!p || bytes < 4) continue;
        const size_t n = bytes / 4;
        const uint32_t grid = static_cast<uint32_t>(std::min<size_t>((n + 255) / 256, static_cast<size_t>(ctx.numCUs) * 4));
        hipLaunchKernelGGL(k_nrc_words_sum, dim3(grid), dim3(256), 0, stream, static_cast<const uint32_t*>(p), n, dOut);
        GFX_HIP(hipGetLastError());
    }
}

void nrc_get_params(NrcNet* net, int which, float* hostOut, uint32_t count) {
    if (count != net->d.total) throw HipError("gfx_nrc_get_params: parameter count mismatch");
    GFX_HIP(hipDeviceSynchronize());
    const DevBuf& src = which == 0 ? net->params : which == 1 ? net->ema : which == 2 ? net->adamM : net->adamV;
    GFX_HIP(hipMemcpy(hostOut, src.p, sizeof(float) * count, hipMemcpyDeviceToHost));
}

void nrc_infer(Context& ctx, hipStream_t stream, NrcNet* net, const float* dInputs, uint32_t numData, float* dPredictions, const uint32_t* dNumData) {
    if (numData & 0x7F) throw HipError("gfx_nrc_infer: numData must be a multiple of 128");   // network_interface.cu:143
    if (numData == 0) return;
    nrc_refresh_inference_images(ctx, stream, net);
    const int numCUs = ctx.numCUs;
    const uint32_t numTiles = numData / 64;
    // A large hash-grid batch is encoded level by level out of LDS copies of the level tables (k_nrc_infer_staged): worth it when every CU
    // gets at least one pass of 3 072 queries ("nrc_staged_infer": 0 by batch size, 1 never, 2 always)
    const uint32_t stagedPasses = (numTiles + (kStagedBlock / 64) * kStagedTiles - 1) / ((kStagedBlock / 64) * kStagedTiles);
    const bool staged = net->d.posEnc == 1 && ctx.tune.nrcStagedInfer != 1 && (ctx.tune.nrcStagedInfer >= 2 || stagedPasses >= static_cast<uint32_t>(numCUs));
    // "nrc_staged_infer" 3: the software-pipelined form (k_nrc_infer_piped; networks of up to two hidden layers: the weights share the LDS with the table)
    bool wholeKiB = true;
    for (int l = 0; l < kHashLevels; ++l) wholeKiB = wholeKiB && net->d.levels[l].entries % 256u == 0 && net->d.levels[l].entries >= 256u;
    const bool piped = staged && ctx.tune.nrcStagedInfer == 3 && net->d.numHidden <= 2 && wholeKiB;
    if (piped) {
        const uint32_t lds = kStagedTableBytes;              // + 20 KiB of static LDS for the weight fragments
        const uint32_t blocks = std::min<uint32_t>(stagedPasses, static_cast<uint32_t>(numCUs));
        if (ctx.nrcInferPipedLds < lds) {
            GFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_nrc_infer_piped), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
            ctx.nrcInferPipedLds = lds;
        }
        net->pipeScratch.reserve(sizeof(uint32_t) * static_cast<size_t>(numCUs) * (kStagedBlock / 64) * kPipedWordsPerWave);
        ScopedKernelTimer timer(ctx, stream, "nrc_infer");
        hipLaunchKernelGGL(k_nrc_infer_piped, dim3(blocks), dim3(kStagedBlock), lds, stream, net->d, net->packInferFwd.as<uint16_t>(), net->gridInfer.as<uint32_t>(),
                           dInputs, numData, dNumData, dPredictions, net->pipeScratch.as<uint32_t>());
        GFX_HIP(hipGetLastError());
        return;
    }
    if (staged) {
        if (!ctx.nrcInferStagedConfigured) {
            GFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_nrc_infer_staged), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kStagedTableBytes)));
            ctx.nrcInferStagedConfigured = true;
        }
        ScopedKernelTimer timer(ctx, stream, "nrc_infer");
        hipLaunchKernelGGL(k_nrc_infer_staged, dim3(std::min<uint32_t>(stagedPasses, static_cast<uint32_t>(numCUs))), dim3(kStagedBlock), kStagedTableBytes, stream, net->d,
                           net->packInferFwd.as<uint16_t>(), net->gridInfer.as<uint32_t>(), dInputs, numData, dNumData, dPredictions);
        GFX_HIP(hipGetLastError());
        return;
    }
    const uint32_t wavesPerBlock = kInferBlock / 64;
    uint32_t grid = std::min<uint32_t>((numTiles + wavesPerBlock - 1) / wavesPerBlock, static_cast<uint32_t>(numCUs) * 4);
    const size_t lds = 2ull * (net->d.numHidden * kMatFwdElems + kOutFwdElems) + wavesPerBlock * 64 * kNrcIn * sizeof(float);
    ScopedKernelTimer timer(ctx, stream, "nrc_infer");
    hipLaunchKernelGGL(k_nrc_infer, dim3(grid), dim3(kInferBlock), lds, stream, net->d, net->packInferFwd.as<uint16_t>(),
                       net->gridInfer.as<uint32_t>(), dInputs, numData, dNumData, dPredictions);
    GFX_HIP(hipGetLastError());
}

void nrc_train(Context& ctx, hipStream_t stream, NrcNet* net, const float* dInputs, const float* dTargets, uint32_t numData, float* lossOnCPU) {
    if (numData & 0x7F) throw HipError("gfx_nrc_train: numData must be a multiple of 128");   // network_interface.cu:151
    if (numData == 0) return;
    // records per one-wave block: 32 while that still leaves the GPU short of blocks (the reference's 16 384-record step: 512 blocks on 256 CUs),
    // 64 for large batches (half the dW partials)
    static const int forcedTile = [] { const char* e = getenv("GFX_NRC_TRAIN_TILE"); return e ? atoi(e) : 0; }();
    const uint32_t tileRecords = forcedTile == 32 || forcedTile == 64 ? static_cast<uint32_t>(forcedTile) : (numData / 64 >= 4u * static_cast<uint32_t>(ctx.numCUs) ? 64u : 32u);
    const uint32_t numBlocks = numData / tileRecords;
    net->gradPartials.reserve(sizeof(float) * static_cast<size_t>(numBlocks) * net->mlpParams);
    GFX_HIP(hipMemsetAsync(net->lossSum.p, 0, sizeof(float), stream));
    NrcTrainArgs a;
    a.d = net->d;
    a.fwd = net->packTrainFwd.as<uint16_t>(); a.bwd = net->packTrainBwd.as<uint16_t>(); a.grid = net->gridTrain.as<uint32_t>();
    a.inputs = dInputs; a.targets = dTargets; a.numData = numData;
    a.gradPartials = net->gradPartials.as<float>(); a.gridGrad = net->gridGrad.as<float>(); a.lossSum = net->lossSum.as<float>();
    a.gridGradMode = net->gridGradMode;
    a.gridDelta = nullptr;
    const bool ldsTables = net->d.posEnc == 1 && net->gridGradMode == kGridGradLdsTables;
    // chunks of records per level table: sixteen for the reference's 16 384-record step (256 blocks), never fewer than 256 records
    const uint32_t chunkRecords = std::max<uint32_t>(256u, ((numData + 15u) / 16u + 63u) / 64u * 64u);
    const uint32_t numChunks = (numData + chunkRecords - 1u) / chunkRecords;
    const uint32_t totalEntries = net->gridParams / 2;
    if (ldsTables) {
        net->gridDelta.reserve(sizeof(float2) * static_cast<size_t>(kHashLevels) * numData);
        net->gridPartials.reserve(sizeof(uint32_t) * static_cast<size_t>(numChunks) * totalEntries);
        a.gridDelta = net->gridDelta.as<float2>();
    }
    const int numLayers = net->d.numHidden + 1;
    const uint32_t nTiles = tileRecords / 32;
    const size_t lds = static_cast<size_t>(numLayers) * nTiles * 4 * 64 * 16 + (static_cast<size_t>(numLayers) + 1) * 64 * kTStride * 2;
    const size_t ldsMax = static_cast<size_t>(kMaxHidden + 1) * 2 * 4 * 64 * 16 + (static_cast<size_t>(kMaxHidden) + 2) * 64 * kTStride * 2;
    if (ldsMax > ctx.nrcTrainLdsConfigured) {   // per device: the attribute belongs to the device's copy of the kernels
        GFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_nrc_train<1>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsMax)));
        GFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_nrc_train<2>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsMax)));
        ctx.nrcTrainLdsConfigured = ldsMax;
    }
    {
        ScopedKernelTimer timer(ctx, stream, "nrc_train_fwd_bwd");
        if (nTiles == 1) hipLaunchKernelGGL(k_nrc_train<1>, dim3(numBlocks), dim3(64), lds, stream, a);
        else hipLaunchKernelGGL(k_nrc_train<2>, dim3(numBlocks), dim3(64), lds, stream, a);
        GFX_HIP(hipGetLastError());
    }
    if (ldsTables) {
        uint32_t maxEntries = 0;
        for (int l = 0; l < kHashLevels; ++l) maxEntries = std::max(maxEntries, net->d.levels[l].entries);
        const size_t scatterLds = sizeof(uint32_t) * maxEntries + 5 * sizeof(float) * kScatterBatch;
        if (scatterLds > net->scatterLdsConfigured) {
            GFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_nrc_grid_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(scatterLds)));
            net->scatterLdsConfigured = scatterLds;
        }
        ScopedKernelTimer timer(ctx, stream, "nrc_grid_scatter");
        hipLaunchKernelGGL(k_nrc_grid_scatter, dim3(numChunks, kHashLevels), dim3(kScatterBlock), scatterLds, stream, net->d, dInputs, net->gridDelta.as<float2>(),
                           numData, chunkRecords, totalEntries, maxEntries, net->gridPartials.as<uint32_t>());
        GFX_HIP(hipGetLastError());
    }
    // Adam (beta1 0.9, beta2 0.99, l2_reg 1e-6) inside EMA(0.99): network_interface.cu:53-64, 91, 118
    ++net->step;
    NrcOptArgs o;
    o.d = net->d;
    o.params = net->params.as<float>(); o.adamM = net->adamM.as<float>(); o.adamV = net->adamV.as<float>(); o.ema = net->ema.as<float>();
    net->gradSum.reserve(sizeof(float) * net->mlpParams);
    {
        ScopedKernelTimer timer(ctx, stream, "nrc_reduce_partials");
        const uint32_t perBlock = kReduceBlock / kReduceSlices;
        hipLaunchKernelGGL(k_nrc_reduce_partials, dim3((net->mlpParams + perBlock - 1) / perBlock), dim3(kReduceBlock), 0, stream,
                           net->gradPartials.as<float>(), numBlocks, net->mlpParams, net->gradSum.as<float>());
        GFX_HIP(hipGetLastError());
    }
    o.gradPartials = net->gradSum.as<float>(); o.mlpParams = net->mlpParams;
    o.gridGrad = net->gridGrad.as<float>(); o.gridGradMode = net->gridGradMode;
    o.gridPartials = net->gridPartials.as<uint32_t>(); o.numGridChunks = ldsTables ? numChunks : 0; o.totalEntries = totalEntries;
    o.beta1 = 0.9f; o.beta2 = 0.99f; o.l2Reg = 1e-6f; o.emaDecay = 0.99f;
    o.eps = net->d.posEnc == 1 ? 1e-15f : 1e-8f;
    const double t = net->step;
    o.lrT = static_cast<float>(static_cast<double>(net->learningRate) * std::sqrt(1.0 - std::pow(static_cast<double>(o.beta2), t)) /
                               (1.0 - std::pow(static_cast<double>(o.beta1), t)));
    o.debiasOld = static_cast<float>(1.0 - std::pow(static_cast<double>(o.emaDecay), t - 1.0));
    o.debiasNew = static_cast<float>(1.0 / (1.0 - std::pow(static_cast<double>(o.emaDecay), t)));
    {
        ScopedKernelTimer timer(ctx, stream, "nrc_optimizer");
        hipLaunchKernelGGL(k_nrc_optimizer, dim3((net->d.total + 255) / 256), dim3(256), 0, stream, o);
        GFX_HIP(hipGetLastError());
        if (net->gridGradMode == kGridGradF16Atomics && net->gridParams) GFX_HIP(hipMemsetAsync(net->gridGrad.p, 0, sizeof(uint32_t) * (net->gridParams / 2), stream));
    }
    nrc_pack(ctx, stream, *net, true);
    if (!net->trained) GFX_HIP(hipEventCreateWithFlags(&net->trained, hipEventDisableTiming));
    GFX_HIP(hipEventRecord(net->trained, stream));
    net->inferDirty = true;
    if (lossOnCPU) {
        GFX_HIP(hipStreamSynchronize(stream));
        GFX_HIP(hipMemcpy(lossOnCPU, net->lossSum.p, sizeof(float), hipMemcpyDeviceToHost));
    }
}

} // namespace gfx
