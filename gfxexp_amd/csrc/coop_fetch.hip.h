// coop_fetch.hip.h -- cooperative gather of ONE 64-byte item per lane (BVH nodes / triangle records in trace.hip, emitter
// records and normal-matrix rows in restir.hip).
//
// A per-lane gather of 64 B costs four dwordx4 instructions whose 64 lanes all touch different
// cache lines: 256 line requests per wave, and the CU's texture-addresser accepts about one line
// request per clock -- that, not arithmetic, is what gather-heavy kernels here are bound by
// (profiles/r01b: k_trace took the same time at 2 and 6 blocks per CU; profiles/r03_experiments.txt:
// k_initial_candidates at 11 gathers per lane and candidate ran at the 1-request-per-clock time).  Here four
// neighbouring lanes fetch the four 16-byte quarters of ONE item with a
// global->LDS DMA (global_load_lds_dwordx4, no VGPR round trip), so each instruction issues 16
// line requests instead of 64, and every lane then reads its own item back from LDS with four
// ds_read_b128.  LDS-DMA writes lane-linearly (base + lane * 16), so the quarter a lane fetches is
// XOR-swizzled with (item >> 2) & 3 to make those reads bank-conflict free.
#pragma once
#include "device_types.h"
#include "gm_math.hip.h"

namespace gfx {

constexpr uint32_t kCoopNone = 0xFFFFFFFFu;   // this lane needs no item

// Issues the loads for the wave: `code` = item index (bits 26-31 are shifted out: a tag may ride there) or kCoopNone;
// item i lives at itemBase + 64 i, i < 2^26.  waveBuf = 256 x 16 B of LDS private to the wave.  Every lane of the wave must call.
GFX_DEV void coop_fetch64_issue(uint32_t code, const char* itemBase, uint4* waveBuf, int lane) {
    // item 16 k + (lane >> 2) of round k: its swizzle ((item >> 2) & 3) = (lane >> 4) & 3 does not depend on k
    const uint32_t quarterOff = static_cast<uint32_t>((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    uint32_t c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = __shfl(code, 16 * k + (lane >> 2));   // whose item this lane helps to fetch
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (c[k] != kCoopNone) {
            const uint32_t off = (c[k] << 6) | quarterOff;   // the byte offset fits 32 bits: the loads use the scalar-base form
            typedef const __attribute__((address_space(1))) void* GlobalPtr;
            typedef __attribute__((address_space(3))) void* LdsPtr;
            __builtin_amdgcn_global_load_lds((GlobalPtr)(itemBase + off), (LdsPtr)(waveBuf + 64 * k), 16, 0, 0);
        }
    }
}
GFX_DEV void coop_fetch64_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// The four quarters of this lane's own item (garbage for a lane that passed kCoopNone).
GFX_DEV void coop_fetch64_read(const uint4* waveBuf, int lane, uint4& q0, uint4& q1, uint4& q2, uint4& q3) {
    const int sw = (lane >> 2) & 3;
    const uint4* mine = waveBuf + 4 * lane;
    q0 = mine[0 ^ sw]; q1 = mine[1 ^ sw]; q2 = mine[2 ^ sw]; q3 = mine[3 ^ sw];
}

GFX_DEV float4 as_float4(uint4 q) { return make_float4(bits2f(q.x), bits2f(q.y), bits2f(q.z), bits2f(q.w)); }

// k_trace's form: nodes and triangle records share one allocation (internal.h Accel): item = code & 0x7FFFFFFF.  The builder
// keeps the item count below 2^26.
GFX_DEV void fetch_items(uint32_t code, const DevAccel& acc, uint4* waveBuf /* 256 x 16 B, wave-private */, int lane,
                         uint4& q0, uint4& q1, uint4& q2, uint4& q3) {
    coop_fetch64_issue(code, reinterpret_cast<const char*>(acc.nodes), waveBuf, lane);
    coop_fetch64_wait();
    coop_fetch64_read(waveBuf, lane, q0, q1, q2, q3);
}

} // namespace gfx
