// stream_copy.hip -- what a streaming copy reaches on this box, by access shape ("measure peak with a streaming-copy
// microbenchmark on the box").  hipcc --offload-arch=gfx950 -O3 stream_copy.hip -o stream_copy && ./stream_copy
// One JSON line per variant: read + write bytes over the HIP-event time of 10 launches, 1-GiB and 4-GiB buffers (the Infinity Cache is
// 256 MiB).  Variants: grid-stride / per-block contiguous chunks, temporal / non-temporal, 1-8 accesses in flight per lane, and the
// read-only and write-only rates.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_stride(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        v4f v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) v[k] = NT ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) { if (NT) __builtin_nontemporal_store(v[k], dst + i + k * stride); else dst[i + k * stride] = v[k]; }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
// every block owns one contiguous chunk
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_chunk(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t b = blockIdx.x * per, e = b + per < n ? b + per : n;
    size_t i = b + threadIdx.x;
    for (; i + (UNROLL - 1) * 256 < e; i += UNROLL * 256) {
        v4f v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) v[k] = NT ? __builtin_nontemporal_load(src + i + k * 256) : src[i + k * 256];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) { if (NT) __builtin_nontemporal_store(v[k], dst + i + k * 256); else dst[i + k * 256] = v[k]; }
    }
    for (; i < e; i += 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_read(const v4f* __restrict__ src, float* __restrict__ out, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    v4f acc = { 0, 0, 0, 0 };
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) acc += src[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.0f;
}
__global__ __launch_bounds__(256) void k_write(v4f* __restrict__ dst, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    const v4f v = { 1, 2, 3, 4 };
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
}

template <typename F> static double timed(F launch) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) launch();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) launch();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 10 * 1e-3;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    for (size_t gib : { size_t(1), size_t(4) }) {
        const size_t bytes = gib << 30, n = bytes / 16;
        v4f *src, *dst; float* out;
        CHECK(hipMalloc(&src, bytes)); CHECK(hipMalloc(&dst, bytes)); CHECK(hipMalloc(&out, 4));
        CHECK(hipMemset(src, 1, bytes)); CHECK(hipMemset(dst, 0, bytes));
        for (int bpc : { 4, 8, 16, 32 }) {
            const dim3 g(cus * bpc), blk(256);
#define RUN(NAME, KERNEL) do { const double s = timed([&] { hipLaunchKernelGGL(KERNEL, g, blk, 0, 0, src, dst, n); }); \
            printf("{\"variant\": \"%s\", \"GiB\": %zu, \"blocks_per_cu\": %d, \"GBps_read_plus_write\": %.1f}\n", NAME, gib, bpc, 2.0 * bytes / s / 1e9); } while (0)
            RUN("stride x1", (k_stride<1, false>)); RUN("stride x4", (k_stride<4, false>)); RUN("stride x8", (k_stride<8, false>));
            RUN("stride x4 nt", (k_stride<4, true>)); RUN("stride x8 nt", (k_stride<8, true>));
            RUN("chunk x4", (k_chunk<4, false>)); RUN("chunk x4 nt", (k_chunk<4, true>)); RUN("chunk x8 nt", (k_chunk<8, true>));
            const double sr = timed([&] { hipLaunchKernelGGL(k_read, g, blk, 0, 0, src, out, n); });
            const double sw = timed([&] { hipLaunchKernelGGL(k_write, g, blk, 0, 0, dst, n); });
            printf("{\"variant\": \"read only\", \"GiB\": %zu, \"blocks_per_cu\": %d, \"GBps\": %.1f}\n", gib, bpc, bytes / sr / 1e9);
            printf("{\"variant\": \"write only\", \"GiB\": %zu, \"blocks_per_cu\": %d, \"GBps\": %.1f}\n", gib, bpc, bytes / sw / 1e9);
        }
        const double sm = timed([&] { CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0)); });
        printf("{\"variant\": \"hipMemcpyAsync d2d\", \"GiB\": %zu, \"GBps_read_plus_write\": %.1f}\n", gib, 2.0 * bytes / sm / 1e9);
        CHECK(hipFree(src)); CHECK(hipFree(dst)); CHECK(hipFree(out));
    }
    return 0;
}
