// l2_gather.hip -- how many scattered 64-byte sectors per second do the L2 caches of an MI355X deliver to the CUs?
// The candidate pass (restir.hip k_initial_candidates) reads ~3 random sectors per light candidate out of ~3.4 MB of tables
// (guide cell, emitter record, normal matrix): this is the roof it runs against.  Three access shapes:
//   lane16   every lane loads 16 B from its own random sector               (64 line requests per wave instruction)
//   lane64   every lane loads its whole sector with four 16-byte loads      (4 x 64 requests, one sector per lane)
//   coop64   four lanes share a sector, 16 B each, global -> LDS DMA        (16 requests per instruction; coop_fetch.hip.h)
//   lane128  every lane loads a whole 128-byte line (two adjacent sectors) with eight 16-byte loads: is the unit of cost the
//            sector or the line?  (Reported in sectors: two per lane and iteration.)
//   pair64   every lane loads two sectors from two unrelated places (what record + normal matrix cost today)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_gather l2_gather.hip ; prints one JSON line per (shape, table size).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ table, uint32_t sectorMask, int iters, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint4 buf[4 * 256];
    const int lane = threadIdx.x & 63;
    uint4* waveBuf = buf + 256 * (threadIdx.x >> 6);
    uint32_t seed = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        seed = hash32(seed + i);
        const uint32_t sector = seed & sectorMask;
        if (SHAPE == 0) {
            const uint4 v = table[sector * 4u + (seed >> 30)];
            acc ^= v.x + v.w;
        } else if (SHAPE == 1) {
            const uint4 a = table[sector * 4u], b = table[sector * 4u + 1], c = table[sector * 4u + 2], d = table[sector * 4u + 3];
            acc ^= a.x + b.y + c.z + d.w;
        } else if (SHAPE == 3) {
            const uint4* q = table + (sector & ~1u) * 4u;
            const uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6], h = q[7];
            acc ^= a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
        } else if (SHAPE == 4) {
            const uint32_t sector2 = hash32(seed ^ 0x9e3779b9u) & sectorMask;
            const uint4* q = table + sector * 4u;
            const uint4* r = table + sector2 * 4u;
            const uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = r[0], f = r[1], g = r[2], h = r[3];
            acc ^= a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
        } else {
            typedef const __attribute__((address_space(1))) void* GlobalPtr;
            typedef __attribute__((address_space(3))) void* LdsPtr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t s = __shfl(sector, 16 * r + (lane >> 2));
                __builtin_amdgcn_global_load_lds((GlobalPtr)(table + s * 4u + (lane & 3)), (LdsPtr)(waveBuf + 64 * r), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint4* mine = waveBuf + 4 * lane;
            acc ^= mine[0].x + mine[1].y + mine[2].z + mine[3].w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int SHAPE>
static void run(const char* name, const uint4* table, size_t bytes, uint32_t* out, int cus) {
    const uint32_t sectors = static_cast<uint32_t>(bytes / 64);
    const int iters = 2000, grid = cus * 8;   // 8 blocks of 4 waves per CU
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, table, sectors - 1, 50, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, table, sectors - 1, iters, out);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = double(grid) * 256.0 * iters * (SHAPE >= 3 ? 2.0 : 1.0);
    printf("{\"shape\": \"%s\", \"table_MB\": %.2f, \"ms\": %.3f, \"Gsectors_per_s\": %.1f, \"TB_per_s_of_64B_sectors\": %.2f}\n",
           name, bytes / 1048576.0, ms, n / ms * 1e-6, n * 64.0 / ms * 1e-9);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const size_t maxBytes = 256ull << 20;
    uint4* table; uint32_t* out;
    (void)hipMalloc(&table, maxBytes); (void)hipMemset(table, 1, maxBytes);
    (void)hipMalloc(&out, size_t(p.multiProcessorCount) * 8 * 256 * 4);
    const size_t sizes[] = { 256ull << 10, 1ull << 20, 2ull << 20, 4ull << 20, 8ull << 20, 32ull << 20, 256ull << 20 };   // powers of two (mask)
    for (size_t s : sizes) {
        run<0>("lane16", table, s, out, p.multiProcessorCount);
        run<1>("lane64", table, s, out, p.multiProcessorCount);
        run<2>("coop64", table, s, out, p.multiProcessorCount);
        run<3>("lane128", table, s, out, p.multiProcessorCount);
        run<4>("pair64", table, s, out, p.multiProcessorCount);
    }
    return 0;
}
