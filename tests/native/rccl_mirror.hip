// rccl_mirror.hip -- a librccl stand-in for TIMING the band frame loop on ONE GPU (test / measurement infrastructure; loaded through
// GFX_RCCL_LIBRARY like tests/native/rccl_stub.cpp).
//
// A one-GPU box cannot run two RCCL ranks, so what an exchange costs a band renderer -- how much of a link's latency the frame
// schedule hides -- cannot be measured with the real library there.  This stand-in gives gfxh_rccl_exchange (the production C++
// callback) a transport with the same shape on the device's streams:
//   ncclSend / ncclRecv inside a group   at ncclGroupEnd: a spin kernel of `strip latency` microseconds on the group's stream, then
//                                        every Recv is filled from the Send posted to the same peer (k-th with k-th): the rows a
//                                        rank would send across a seam come back as the rows it receives across that seam -- real
//                                        reservoirs / G-buffer texels of adjacent rows, so the passes that read them do real work
//   ncclAllGather                        a spin kernel of `gather latency` microseconds, then the caller's slab into every rank's
//                                        slot (the same bytes written into this GPU's memory as the real collective writes)
//   ncclAllReduce / ncclBroadcast        the spin kernel only
// The latency is a constant per operation: wire time + transfer time as one number (tools/band_host_overhead.py sweeps it).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {
struct Comm { int rank, world; };
struct Op { bool send; char* buf; size_t bytes; int peer; };
std::vector<Op> g_group;
hipStream_t g_groupStream = nullptr;
int g_depth = 0;
float g_stripUs = 0.0f, g_gatherUs = 0.0f;
bool g_copy = true;   // rccl_mirror_set_copy(0): the calls return at once (what the host spends in the callback, no transport behind it)
uint64_t g_ops = 0, g_bytes = 0;

__global__ void k_spin(uint64_t ticks) {            // wall_clock64: the constant 100-MHz counter
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
void spin(float us, hipStream_t stream) {
    if (us > 0.0f) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, stream, static_cast<uint64_t>(us * 100.0f));
}
size_t dtype_size(int t) { return t == 3 ? 4 : 1; }   // ncclUint8 = 1, ncclUint32 = 3
void flush_group() {
    if (g_group.empty()) return;
    if (!g_copy) { ++g_ops; g_group.clear(); return; }
    spin(g_stripUs, g_groupStream);
    std::vector<bool> used(g_group.size(), false);
    for (size_t i = 0; i < g_group.size(); ++i) {
        if (g_group[i].send) continue;
        for (size_t j = 0; j < g_group.size(); ++j) {
            if (used[j] || !g_group[j].send || g_group[j].peer != g_group[i].peer || g_group[j].bytes != g_group[i].bytes) continue;
            used[j] = true;
            (void)hipMemcpyAsync(g_group[i].buf, g_group[j].buf, g_group[i].bytes, hipMemcpyDeviceToDevice, g_groupStream);
            g_bytes += g_group[i].bytes;
            break;
        }
    }
    ++g_ops;
    g_group.clear();
}
}

extern "C" {
int ncclGetUniqueId(void* id) { memset(id, 0x3C, 128); return 0; }
struct Id128 { char b[128]; };
int ncclCommInitRank(void** comm, int world, Id128, int rank) { *comm = new Comm{ rank, world }; return 0; }
int ncclCommDestroy(void* comm) { delete static_cast<Comm*>(comm); return 0; }
int ncclGroupStart() { ++g_depth; return 0; }
int ncclGroupEnd() { if (--g_depth == 0) flush_group(); return 0; }
int ncclSend(const void* buf, size_t count, int dtype, int peer, void*, void* stream) {
    g_group.push_back({ true, static_cast<char*>(const_cast<void*>(buf)), count * dtype_size(dtype), peer });
    g_groupStream = static_cast<hipStream_t>(stream);
    if (g_depth == 0) flush_group();
    return 0;
}
int ncclRecv(void* buf, size_t count, int dtype, int peer, void*, void* stream) {
    g_group.push_back({ false, static_cast<char*>(buf), count * dtype_size(dtype), peer });
    g_groupStream = static_cast<hipStream_t>(stream);
    if (g_depth == 0) flush_group();
    return 0;
}
int ncclAllReduce(const void*, void*, size_t, int, int, void*, void* stream) { spin(g_stripUs, static_cast<hipStream_t>(stream)); ++g_ops; return 0; }
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, void* stream) {
    const Comm* c = static_cast<const Comm*>(comm);
    const size_t bytes = count * dtype_size(dtype);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!g_copy) { ++g_ops; return 0; }
    spin(g_gatherUs, s);
    for (int r = 0; r < c->world; ++r) {
        char* dst = static_cast<char*>(recv) + bytes * r;
        if (dst != send) { (void)hipMemcpyAsync(dst, send, bytes, hipMemcpyDeviceToDevice, s); g_bytes += bytes; }
    }
    ++g_ops;
    return 0;
}
int ncclBroadcast(const void*, void*, size_t, int, int, void*, void* stream) { spin(g_stripUs, static_cast<hipStream_t>(stream)); ++g_ops; return 0; }

void rccl_mirror_set_latency_us(float strips, float gather) { g_stripUs = strips; g_gatherUs = gather; }
void rccl_mirror_set_copy(int on) { g_copy = on != 0; }
void rccl_mirror_stats(uint64_t* ops, uint64_t* bytes, int reset) { *ops = g_ops; *bytes = g_bytes; if (reset) { g_ops = 0; g_bytes = 0; } }
}
