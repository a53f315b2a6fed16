// rccl_stub.cpp -- a recording stand-in for librccl (test infrastructure; loaded through GFX_RCCL_LIBRARY).
//
// gfxh_rccl_exchange (gfxexp_amd/csrc/host/rccl_exchange.cpp) issues ncclSend / ncclRecv / ncclAllReduce / ncclAllGather /
// ncclBroadcast with peers, pointers and byte counts derived from a gfxh_exchange_desc.  On a box with one GPU only
// world = 1 ever executes against the real library, so the rank +- 1 arithmetic would never run.  This stub implements the
// ten entry points the product resolves, RECORDS every call, and emulates the data movement of a collective with the one
// rank it has: ncclAllGather puts the caller's slab into EVERY rank's slot of the receive buffer, so the test can see by
// content which rows of the frame each slot was scattered to.  Copies go through memcpy (host pointers, CPU tests) or
// hipMemcpyAsync resolved from the process (device pointers, RCCL_STUB_DEVICE=1 on the GPU box).
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {
struct Comm { int rank, world; };
struct Call { int32_t op, dtype, peer, inGroup; uint64_t a, b, count; };
enum { kSend = 1, kRecv = 2, kAllReduce = 3, kAllGather = 4, kBroadcast = 5, kGroupStart = 6, kGroupEnd = 7 };
std::vector<Call> g_calls;
int g_groupDepth = 0;
size_t dtype_size(int t) { return t == 3 ? 4 : 1; }   // ncclUint8 = 1, ncclUint32 = 3: the two types the product uses
void record(int op, const void* a, void* b, size_t count, int dtype, int peer) {
    Call c; c.op = op; c.dtype = dtype; c.peer = peer; c.inGroup = g_groupDepth;
    c.a = reinterpret_cast<uint64_t>(a); c.b = reinterpret_cast<uint64_t>(b); c.count = count;
    g_calls.push_back(c);
}
void copy_bytes(void* dst, const void* src, size_t n, void* stream) {
    if (dst == src || n == 0) return;
    const char* dev = getenv("RCCL_STUB_DEVICE");
    if (dev && dev[0] == '1') {
        // stream-ordered like the real collective: the HIP runtime the process already loaded (ctypes loads libraries
        // RTLD_LOCAL, so the symbol is looked up in that library's handle, not in the global scope)
        typedef int (*MemcpyAsync)(void*, const void*, size_t, int, void*);
        static MemcpyAsync f = nullptr;
        if (!f) {
            f = reinterpret_cast<MemcpyAsync>(dlsym(RTLD_DEFAULT, "hipMemcpyAsync"));
            const char* names[] = { "libamdhip64.so.7", "libamdhip64.so.6", "libamdhip64.so" };
            for (int k = 0; k < 3 && !f; ++k)
                if (void* h = dlopen(names[k], RTLD_NOW | RTLD_NOLOAD)) f = reinterpret_cast<MemcpyAsync>(dlsym(h, "hipMemcpyAsync"));
        }
        if (!f) { fprintf(stderr, "rccl_stub: RCCL_STUB_DEVICE=1 but no loaded HIP runtime exports hipMemcpyAsync\n"); abort(); }
        f(dst, src, n, 3 /* hipMemcpyDeviceToDevice */, stream);
        return;
    }
    memmove(dst, src, n);
}
}

extern "C" {
int ncclGetUniqueId(void* id) { memset(id, 0x5A, 128); return 0; }
struct Id128 { char b[128]; };
int ncclCommInitRank(void** comm, int world, Id128, int rank) { Comm* c = new Comm{ rank, world }; *comm = c; return 0; }
int ncclCommDestroy(void* comm) { delete static_cast<Comm*>(comm); return 0; }
int ncclGroupStart() { ++g_groupDepth; record(kGroupStart, nullptr, nullptr, 0, 0, -1); return 0; }
int ncclGroupEnd() { record(kGroupEnd, nullptr, nullptr, 0, 0, -1); --g_groupDepth; return 0; }
int ncclSend(const void* buf, size_t count, int dtype, int peer, void*, void*) { record(kSend, buf, nullptr, count, dtype, peer); return 0; }
int ncclRecv(void* buf, size_t count, int dtype, int peer, void*, void*) { record(kRecv, nullptr, buf, count, dtype, peer); return 0; }
int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void*, void*) { record(kAllReduce, send, recv, count, dtype, op); return 0; }
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, void* stream) {
    record(kAllGather, send, recv, count, dtype, -1);
    const Comm* c = static_cast<const Comm*>(comm);
    const size_t bytes = count * dtype_size(dtype);
    for (int r = 0; r < c->world; ++r) copy_bytes(static_cast<char*>(recv) + bytes * r, send, bytes, stream);
    return 0;
}
int ncclBroadcast(const void* send, void* recv, size_t count, int dtype, int root, void*, void*) { record(kBroadcast, send, recv, count, dtype, root); return 0; }

// ---- what the tests read back
uint32_t rccl_stub_num_calls() { return static_cast<uint32_t>(g_calls.size()); }
void rccl_stub_get(uint32_t i, Call* out) { *out = g_calls[i]; }
void rccl_stub_reset() { g_calls.clear(); }
}
