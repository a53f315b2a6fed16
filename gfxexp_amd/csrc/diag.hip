// diag.hip -- measurement utilities behind include/gfxexp.h that no renderer calls.
#include <algorithm>
#include "internal.h"

namespace gfx {

// Streaming copy: every lane moves 16 bytes per access, four accesses in flight per thread, grid-stride; the grid is sized to fill
// the GPU (8 blocks of 256 per CU) and each wave walks contiguous 1-KiB lines.
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_copy(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const v4f a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        const v4f c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride);
        __builtin_nontemporal_store(c, dst + i + 2 * stride); __builtin_nontemporal_store(d, dst + i + 3 * stride);
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

void stream_copy(Context& ctx, hipStream_t stream, void* dDst, const void* dSrc, size_t bytes) {
    if (bytes % 16 != 0 || (reinterpret_cast<uintptr_t>(dDst) | reinterpret_cast<uintptr_t>(dSrc)) % 16 != 0)
        throw HipError("gfx_stream_copy: pointers and byte count must be multiples of 16");
    if (!bytes) return;
    const size_t n = bytes / 16;
    const uint32_t grid = static_cast<uint32_t>(std::min<size_t>((n + 255) / 256, static_cast<size_t>(ctx.numCUs) * 8));
    hipLaunchKernelGGL(k_stream_copy, dim3(grid), dim3(256), 0, stream, static_cast<const v4f*>(dSrc), static_cast<v4f*>(dDst), n);
    GFX_HIP(hipGetLastError());
}

} // namespace gfx
