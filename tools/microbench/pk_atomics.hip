// pk_atomics.hip -- the rate at which the L2s retire packed-fp16 atomics (global_atomic_pk_add_f16) in the access pattern of k_nrc_train's
// hash-grid gradient scatter: 16 384 records x 16 levels x 8 corners = 2.1 M atomics per step into 16 tables of <= 32 768 words.
// Question (profiles/r04_experiments.txt): is a training step (0.15 ms uniform records, 0.25 ms in the frame) waiting for these atomics,
// or for the single wave per CU that issues them?   hipcc --offload-arch=gfx950 -O3 pk_atomics.hip -o pk_atomics && ./pk_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ inline uint32_t hash3(uint32_t a, uint32_t b, uint32_t c) { uint32_t h = a * 2654435761u ^ b * 805459861u ^ c * 3674653429u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; return h; }

// one thread = one record: 16 levels x 8 corners.  `spread`: records draw their cell from [0, spread) per level (clustered records share
// cells); ATOMIC 0 = plain stores to the same addresses (the traffic without the read-modify-write), 1 = packed fp16, 2 = two fp32
template <int ATOMIC>
__global__ void k_scatter(uint32_t* table, uint32_t numRecords, uint32_t spread, uint32_t recordsPerThreadShift) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= numRecords) return;
    const uint32_t cell = hash3(t, 17u, 3u) % spread;
    for (int level = 0; level < 16; ++level) {
        const uint32_t entries = level == 0 ? 4920u : 32768u;
        const uint32_t base = hash3(cell >> (15 - level > 0 ? (15 - level) / 2 : 0), level, 7u);
        for (int c = 0; c < 8; ++c) {
            const uint32_t idx = level * 32768u + (base + hash3(c, level, 1u)) % entries;
            f16x2 v; v.x = (_Float16)0.001f; v.y = (_Float16)0.002f;
            typedef __attribute__((address_space(1))) f16x2* G;
            if (ATOMIC == 1) (void)__builtin_amdgcn_global_atomic_fadd_v2f16((G)(table + idx), v);
            else if (ATOMIC == 2) { atomicAdd(reinterpret_cast<float*>(table) + 2 * idx, 0.001f); atomicAdd(reinterpret_cast<float*>(table) + 2 * idx + 1, 0.002f); }
            else table[idx] = t;
        }
    }
}

template <typename F> static double timed(F launch) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 20; ++i) launch();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 20 * 1e3;   // microseconds
}

int main() {
    uint32_t* table; CHECK(hipMalloc(&table, 16 * 32768 * 8)); CHECK(hipMemset(table, 0, 16 * 32768 * 8));
    const uint32_t n = 16384;
    for (uint32_t spread : { 1u << 30, 1u << 16, 1u << 12, 1u << 8 })
        for (int block : { 64, 256 }) {
            const dim3 g((n + block - 1) / block), b(block);
            const double a1 = timed([&] { hipLaunchKernelGGL(k_scatter<1>, g, b, 0, 0, table, n, spread, 0u); });
            const double a2 = timed([&] { hipLaunchKernelGGL(k_scatter<2>, g, b, 0, 0, table, n, spread, 0u); });
            const double a0 = timed([&] { hipLaunchKernelGGL(k_scatter<0>, g, b, 0, 0, table, n, spread, 0u); });
            printf("{\"records\": %u, \"cells\": %u, \"block\": %d, \"blocks\": %u, \"pk_f16_us\": %.1f, \"two_f32_us\": %.1f, \"plain_store_us\": %.1f, \"G_pk_atomics_per_s\": %.2f}\n",
                   n, spread, block, g.x, a1, a2, a0, n * 128.0 / a1 / 1e3);
        }
    // the same 2.1 M atomics from a launch that fills the GPU (one atomic per thread)
    return 0;
}
