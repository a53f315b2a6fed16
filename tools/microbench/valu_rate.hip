// valu_rate.hip -- how many wave64 VALU instructions does one CDNA4 SIMD issue per clock, as a function of the number of
// resident waves and of the instruction kind?  (Decides whether k_trace, at ~4.9 SIMD cycles per VALU instruction with four
// waves per SIMD, is issue-bound or latency-bound.)   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned u = threadIdx.x * 2654435761u;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {          // one dependent chain of v_fma_f32
            asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(x0) : "v"(a), "v"(b));
        } else if (KIND == 1) {   // eight independent chains
            asm volatile(REP16("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == 2) {   // the slab-test mix: cvt ubyte, fma, max, min (independent pairs)
            asm volatile(REP16("v_cvt_f32_ubyte0 %0, %4\n v_fma_f32 %1, %0, %5, %6\n v_max_f32 %2, %2, %1\n v_min_f32 %3, %3, %1\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(u), "v"(a), "v"(b));
        } else if (KIND == 3) {   // integer / bit ops
            asm volatile(REP16("v_and_b32 %0, %0, %2\n v_lshl_or_b32 %1, %0, 3, %1\n v_bfe_u32 %0, %1, 2, 8\n v_xor_b32 %1, %1, %0\n")
                         : "+v"(u), "+v"(x0) : "v"(x1));
        } else if (KIND == 4) {   // v_cvt_f32_ubyte only
            asm volatile(REP16("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(u));
        } else if (KIND == 5) {   // packed f16 fma, independent
            asm volatile(REP16("v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %4, %5\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %4, %5\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
        } else if (KIND == 6) {   // packed f32 fma, independent (register pairs)
            double d0 = x0, d1 = x1, d2 = x2, d3 = x3, da = a, db = b;
            asm volatile(REP16("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(da), "v"(db));
            x0 += (float)d0; x1 += (float)d1; x2 += (float)d2; x3 += (float)d3;
        } else if (KIND == 7) {   // v_cndmask + v_cmp
            asm volatile(REP16("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_gt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "vcc");
        } else if (KIND == 8) {   // 1/x, IEEE sqrt building blocks
            asm volatile(REP16("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_rsq_f32 %3, %3\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        } else if (KIND == 9) {   // v_max3 / v_min3
            asm volatile(REP16("v_max3_f32 %0, %0, %1, %2\n v_min3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_min3_f32 %3, %3, %0, %1\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        } else if (KIND == 10) {  // v_fma_mix_f32: fp16 half of a register as the first source, fp32 fma -- the conversion folded into the fma
            asm volatile(REP16("v_fma_mix_f32 %0, %4, %5, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %5, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %2, %4, %6, %5 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %4, %6, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(u), "v"(a), "v"(b));
        } else if (KIND == 11) {  // the slab-test mix with it: fma_mix, max, min (what cvt + fma + max + min would become)
            asm volatile(REP16("v_fma_mix_f32 %1, %4, %5, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_max_f32 %2, %2, %1\n v_fma_mix_f32 %0, %4, %5, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_min_f32 %3, %3, %0\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(u), "v"(a), "v"(b));
        } else if (KIND == 12) {  // v_cvt_f32_f16 (plain and SDWA high half) for comparison
            asm volatile(REP16("v_cvt_f32_f16 %0, %4\n v_cvt_f32_f16 %1, %4\n v_cvt_f32_f16 %2, %4\n v_cvt_f32_f16 %3, %4\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(u));
        } else if (KIND == 13) {  // fma + max + min without any conversion (the floor of a slab test)
            asm volatile(REP16("v_fma_f32 %1, %4, %5, %6\n v_max_f32 %2, %2, %1\n v_fma_f32 %0, %4, %6, %5\n v_min_f32 %3, %3, %0\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(u), "v"(a), "v"(b));
        } else if (KIND == 14) {  // v_pk_max_f16 / v_pk_min_f16 / v_pk_fma_f16: a slab test on packed halves
            asm volatile(REP16("v_pk_fma_f16 %1, %4, %5, %6\n v_pk_max_f16 %2, %2, %1\n v_pk_fma_f16 %0, %4, %6, %5\n v_pk_min_f16 %3, %3, %0\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(u), "v"(a), "v"(b));
        } else if (KIND == 15) {  // 32-bit integer multiplies: the 64-bit LCG step of PCG32 is built from these
            asm volatile(REP16("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(u));
        } else if (KIND == 16) {
            asm volatile(REP16("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(u));
        } else if (KIND == 17) {  // 24-bit multiplies
            asm volatile(REP16("v_mul_u32_u24 %0, %0, %4\n v_mad_u32_u24 %1, %1, %4, %0\n v_mul_hi_u32_u24 %2, %2, %4\n v_mad_u32_u24 %3, %3, %4, %2\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(u));
        } else if (KIND == 18) {  // 32 x 32 + 64 -> 64
            unsigned long long w0 = u, w1 = u + 1, w2 = u + 2, w3 = u + 3;
            asm volatile(REP16("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n")
                         : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(u), "v"(a) : "vcc");
            u += (unsigned)(w0 + w1 + w2 + w3);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)u;
}

template <int KIND>
static void run(const char* name, float* d, int numCUs, double ghz) {
    const int iters = 2000;
    for (int wavesPerSimd = 1; wavesPerSimd <= 8; wavesPerSimd *= 2) {
        const int grid = numCUs * 4 * wavesPerSimd;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d, 10, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double instPerSimd = 64.0 * iters * wavesPerSimd;
        const double cycles = ms * 1e-3 * ghz * 1e9;
        printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_wave64_inst_per_simd\": %.3f}\n", name, wavesPerSimd, ms, cycles / instPerSimd);
    }
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f}\n", p.name, p.multiProcessorCount, ghz);
    float* d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4 * 2);
    run<0>("fma_dependent", d, p.multiProcessorCount, ghz);
    run<1>("fma_8_chains", d, p.multiProcessorCount, ghz);
    run<2>("cvt_fma_max_min", d, p.multiProcessorCount, ghz);
    run<3>("int_and_lshlor_bfe_xor", d, p.multiProcessorCount, ghz);
    run<4>("cvt_f32_ubyte", d, p.multiProcessorCount, ghz);
    run<5>("pk_fma_f16", d, p.multiProcessorCount, ghz);
    run<6>("pk_fma_f32", d, p.multiProcessorCount, ghz);
    run<7>("cmp_cndmask", d, p.multiProcessorCount, ghz);
    run<8>("rcp_sqrt_rsq", d, p.multiProcessorCount, ghz);
    run<9>("max3_min3", d, p.multiProcessorCount, ghz);
    run<10>("fma_mix_f32_from_f16", d, p.multiProcessorCount, ghz);
    run<11>("fmamix_max_fmamix_min", d, p.multiProcessorCount, ghz);
    run<12>("cvt_f32_f16", d, p.multiProcessorCount, ghz);
    run<13>("fma_max_fma_min", d, p.multiProcessorCount, ghz);
    run<14>("pk_fma_max_min_f16", d, p.multiProcessorCount, ghz);
    run<15>("mul_lo_u32", d, p.multiProcessorCount, ghz);
    run<16>("mul_hi_u32", d, p.multiProcessorCount, ghz);
    run<17>("mul_u32_u24_mad_u32_u24_mul_hi_u32_u24", d, p.multiProcessorCount, ghz);
    run<18>("mad_u64_u32", d, p.multiProcessorCount, ghz);
    return 0;
}
