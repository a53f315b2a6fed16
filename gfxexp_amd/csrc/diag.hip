// diag.hip -- measurement utilities behind include/gfxexp.h that no renderer calls.
#include <algorithm>
#include "internal.h"

namespace gfx {

// Streaming copy: every block owns one contiguous chunk, every lane moves 16 bytes per access with four non-temporal accesses in
// flight, 16 blocks of 256 per CU -- the best of the shapes tools/microbench/stream_copy.hip tries on this part (5.9 TB/s read + write
// over 1 GiB, 5.5 over 4 GiB; a grid-stride loop reaches 4.7-5.2, hipMemcpyAsync 4.6-5.5) -- profiles/r04_stream_copy.jsonl.
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_copy(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t b = blockIdx.x * per, e = b + per < n ? b + per : n;
    size_t i = b + threadIdx.x;
    for (; i + 3 * 256 < e; i += 4 * 256) {
        const v4f v0 = __builtin_nontemporal_load(src + i), v1 = __builtin_nontemporal_load(src + i + 256);
        const v4f v2 = __builtin_nontemporal_load(src + i + 512), v3 = __builtin_nontemporal_load(src + i + 768);
        __builtin_nontemporal_store(v0, dst + i); __builtin_nontemporal_store(v1, dst + i + 256);
        __builtin_nontemporal_store(v2, dst + i + 512); __builtin_nontemporal_store(v3, dst + i + 768);
    }
    for (; i < e; i += 256) dst[i] = src[i];
}
// Read-only pass over the same bytes (6.3-6.5 TB/s here): the rate a kernel that only reads can be held against.
__global__ __launch_bounds__(256) void k_stream_read(const v4f* __restrict__ src, size_t n, float* __restrict__ sink) {
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    v4f acc = { 0.0f, 0.0f, 0.0f, 0.0f };
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) acc += src[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) *sink = 1.0f;   // never true for the buffers bench.py passes; keeps the loads alive
}

void stream_copy(Context& ctx, hipStream_t stream, void* dDst, const void* dSrc, size_t bytes) {
    if (bytes % 16 != 0 || (reinterpret_cast<uintptr_t>(dDst) | reinterpret_cast<uintptr_t>(dSrc)) % 16 != 0)
        throw HipError("gfx_stream_copy: pointers and byte count must be multiples of 16");
    if (!bytes) return;
    const size_t n = bytes / 16;
    if (!dDst) {
        ctx.smallCounters.reserve(kSmallCountersBytes);
        const uint32_t grid = static_cast<uint32_t>(std::min<size_t>((n + 255) / 256, static_cast<size_t>(ctx.numCUs) * 8));
        hipLaunchKernelGGL(k_stream_read, dim3(grid), dim3(256), 0, stream, static_cast<const v4f*>(dSrc), n, ctx.smallCounters.as<float>() + 200);
    }
    else {
        const uint32_t grid = static_cast<uint32_t>(std::min<size_t>((n + 1023) / 1024, static_cast<size_t>(ctx.numCUs) * 16));
        hipLaunchKernelGGL(k_stream_copy, dim3(grid), dim3(256), 0, stream, static_cast<const v4f*>(dSrc), static_cast<v4f*>(dDst), n);
    }
    GFX_HIP(hipGetLastError());
}

} // namespace gfx
