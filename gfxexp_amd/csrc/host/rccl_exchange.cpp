// rccl_exchange.cpp -- the exchange callback of a band renderer (gfxh_restir_set_exchange) over RCCL, for C++ host
// programs that run one process per GPU of a node (xGMI).  librccl is loaded with dlopen on first use, so libgfxexp.so
// has no link-time dependency on it and single-GPU users never touch it.
//
//   strips           ncclSend / ncclRecv of every (buffer, plane) row range, one group per exchange point: the two
//                    neighbours of a rank sit on direct xGMI links, a strip is 2-4 MB at 1080p
//   counters         ncclAllReduce(sum, u32) in place (ReGIR cell-access counters: 32 KB)
//   HDR bands        ncclAllGather of slabs sized for the tallest band into a staging buffer, then one hipMemcpyAsync
//                    per remote band into the frame (bands differ by at most 8 rows)
// GFX_RCCL_LIBRARY names the library to load instead of librccl.so.1 (another RCCL build; tests/native/rccl_stub.cpp, a
// recording stand-in that lets the send / receive plan of ANY rank of ANY world size be checked in one process).
// All operations are enqueued on the caller's stream.  bench.py and the tests use torch.distributed for the same
// descriptors (gfxexp_amd/tilesplit.py StripExchange); both are driven by the same gfxh_exchange_desc.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../../include/gfxexp_host.h"

namespace {

typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum { kNcclUint8 = 1, kNcclUint32 = 3, kNcclSum = 0 };

struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    bool load(std::string& err) {
        if (lib) return true;
        void* h = nullptr;
        const char* named = std::getenv("GFX_RCCL_LIBRARY");
        if (named && *named) h = dlopen(named, RTLD_NOW | RTLD_LOCAL);
        else {
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        }
        if (!h) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        bool complete = true;
        auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) { err = std::string("librccl lacks ") + n; complete = false; } return p; };
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
        if (!complete) { dlclose(h); return false; }   // `lib` stays null: the next call tries again instead of running on null pointers
        lib = h;
        return true;
    }
};
RcclApi g_rccl;
thread_local std::string g_rcclError;

} // namespace

struct gfxh_rccl {
    // one communicator per lane of the renderer (gfxexp_host.h gfxh_lane) when created with gfxh_rccl_create_lanes: operations of
    // different lanes are enqueued on different streams and must not queue behind each other inside one communicator; with a single
    // communicator (gfxh_rccl_create) every lane shares it and the operations run in issue order
    ncclComm_t comms[GFXH_NUM_LANES] = { nullptr, nullptr, nullptr, nullptr };
    uint32_t numComms = 0;
    ncclComm_t lane(uint32_t l) const { return comms[l < numComms ? l : 0]; }
    int rank = 0, world = 1;
    std::vector<uint32_t> bandBegin, bandEnd;   // of every rank: whole 8-row tiles, remainder spread from rank 0
    void* staging = nullptr; size_t stagingBytes = 0;
};

extern "C" {

const char* gfxh_rccl_last_error(void) { return g_rcclError.c_str(); }

int gfxh_rccl_unique_id(void* id128) {
    if (!g_rccl.load(g_rcclError)) return 1;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id)) { g_rcclError = "ncclGetUniqueId failed"; return 1; }
    std::memcpy(id128, &id, sizeof(id));
    return 0;
}

int gfxh_rccl_create_lanes(const void* ids, uint32_t numLanes, int rank, int world, uint32_t height, gfxh_rccl** out) {
    *out = nullptr;
    if (!g_rccl.load(g_rcclError)) return 1;
    if (numLanes < 1 || numLanes > GFXH_NUM_LANES) { g_rcclError = "gfxh_rccl_create_lanes: 1 to " + std::to_string(GFXH_NUM_LANES) + " communicators"; return 1; }
    gfxh_rccl* c = new gfxh_rccl();
    if (world < 1 || rank < 0 || rank >= world) { g_rcclError = "gfxh_rccl_create: rank outside the world"; delete c; return 1; }
    c->rank = rank; c->world = world;
    for (int r = 0; r < world; ++r) {
        uint32_t b = 0, e = 0;
        gfxh_band_rows(height, static_cast<uint32_t>(world), static_cast<uint32_t>(r), &b, &e);
        if (e <= b) {   // more ranks than 8-row tiles: an empty band would enter the strip / gather collectives with nothing to send
            g_rcclError = "gfxh_rccl_create: " + std::to_string(world) + " ranks for " + std::to_string(height) + " rows leave rank " + std::to_string(r) + " without a band";
            delete c; return 1;
        }
        c->bandBegin.push_back(b); c->bandEnd.push_back(e);
    }
    for (uint32_t l = 0; l < numLanes; ++l) {
        ncclUniqueId id; std::memcpy(&id, static_cast<const char*>(ids) + sizeof(id) * l, sizeof(id));
        if (g_rccl.CommInitRank(&c->comms[l], world, id, rank)) {
            g_rcclError = "ncclCommInitRank failed (lane " + std::to_string(l) + ")";
            gfxh_rccl_destroy(c); return 1;
        }
        c->numComms = l + 1;
    }
    *out = c;
    return 0;
}

int gfxh_rccl_create(const void* id128, int rank, int world, uint32_t height, gfxh_rccl** out) {
    return gfxh_rccl_create_lanes(id128, 1, rank, world, height, out);
}

int gfxh_rccl_set_bands(gfxh_rccl* c, const uint32_t* bandBegin) {
    if (!c || !bandBegin) { g_rcclError = "gfxh_rccl_set_bands: null argument"; return 1; }
    for (int r = 0; r < c->world; ++r)
        if (bandBegin[r + 1] <= bandBegin[r] || (r == 0 && bandBegin[0] != 0)) { g_rcclError = "gfxh_rccl_set_bands: the partition must ascend from row 0"; return 1; }
    for (int r = 0; r < c->world; ++r) { c->bandBegin[r] = bandBegin[r]; c->bandEnd[r] = bandBegin[r + 1]; }
    return 0;
}

void gfxh_rccl_destroy(gfxh_rccl* c) {
    if (!c) return;
    for (uint32_t l = 0; l < c->numComms; ++l) if (c->comms[l]) g_rccl.CommDestroy(c->comms[l]);
    if (c->staging) (void)hipFree(c->staging);
    delete c;
}

int gfxh_rccl_exchange(void* user, void* streamPtr, const gfxh_exchange_desc* d) {
    gfxh_rccl* c = static_cast<gfxh_rccl*>(user);
    hipStream_t stream = static_cast<hipStream_t>(streamPtr);
    int err = 0;
    struct { ncclComm_t comm; } lane = { c->lane(d->lane) };   // the communicator of the lane `stream` belongs to
    if (d->kind == GFXH_EXCHANGE_ALLREDUCE_SUM_U32)
        return g_rccl.AllReduce(d->counters, d->counters, d->numCounters, kNcclUint32, kNcclSum, lane.comm, stream) ? 1 : 0;
    if (d->kind == GFXH_EXCHANGE_STRIPS) {
        err |= g_rccl.GroupStart();
        for (uint32_t k = 0; k < d->numBuffers; ++k) {
            const gfxh_exchange_buffer& b = d->buffers[k];
            const size_t rowBytes = static_cast<size_t>(b.bytesPerPixel) * d->width;
            for (uint32_t plane = 0; plane < b.numPlanes; ++plane) {
                char* base = static_cast<char*>(b.base) + plane * b.planeStride;
                auto rows = [&](const uint32_t r[2]) { return static_cast<size_t>(r[1] - r[0]) * rowBytes; };
                if (c->rank > 0) {
                    if (rows(d->sendAbove)) err |= g_rccl.Send(base + d->sendAbove[0] * rowBytes, rows(d->sendAbove), kNcclUint8, c->rank - 1, lane.comm, stream);
                    if (rows(d->recvAbove)) err |= g_rccl.Recv(base + d->recvAbove[0] * rowBytes, rows(d->recvAbove), kNcclUint8, c->rank - 1, lane.comm, stream);
                }
                if (c->rank + 1 < c->world) {
                    if (rows(d->sendBelow)) err |= g_rccl.Send(base + d->sendBelow[0] * rowBytes, rows(d->sendBelow), kNcclUint8, c->rank + 1, lane.comm, stream);
                    if (rows(d->recvBelow)) err |= g_rccl.Recv(base + d->recvBelow[0] * rowBytes, rows(d->recvBelow), kNcclUint8, c->rank + 1, lane.comm, stream);
                }
            }
        }
        err |= g_rccl.GroupEnd();
        return err ? 1 : 0;
    }
    if (d->kind == GFXH_EXCHANGE_GATHER_BANDS) {
        // the communicator was created for one partition of the frame; a renderer configured with other bands would
        // have its rows gathered into the wrong place
        if (d->bandBegin != c->bandBegin[c->rank] || d->bandEnd != c->bandEnd[c->rank]) {
            g_rcclError = "gfxh_rccl_exchange: the renderer's band [" + std::to_string(d->bandBegin) + ", " + std::to_string(d->bandEnd) + ") is not rank " +
                          std::to_string(c->rank) + "'s band [" + std::to_string(c->bandBegin[c->rank]) + ", " + std::to_string(c->bandEnd[c->rank]) + ") of gfxh_band_rows / gfxh_rccl_set_bands";
            return 1;
        }
        const gfxh_exchange_buffer& b = d->buffers[0];
        const size_t rowBytes = static_cast<size_t>(b.bytesPerPixel) * d->width;
        uint32_t maxRows = 0;
        for (int r = 0; r < c->world; ++r) maxRows = std::max(maxRows, c->bandEnd[r] - c->bandBegin[r]);
        const size_t slab = static_cast<size_t>(maxRows) * rowBytes;
        if (c->stagingBytes < slab * c->world) {
            if (c->staging) (void)hipFree(c->staging);
            if (hipMalloc(&c->staging, slab * c->world) != hipSuccess) { g_rcclError = "hipMalloc of the gather staging buffer failed"; return 1; }
            c->stagingBytes = slab * c->world;
        }
        char* frame = static_cast<char*>(b.base);
        // in-place form: every rank's own slab inside the staging buffer is its send buffer
        char* own = static_cast<char*>(c->staging) + slab * c->rank;
        if (hipMemcpyAsync(own, frame + c->bandBegin[c->rank] * rowBytes, (c->bandEnd[c->rank] - c->bandBegin[c->rank]) * rowBytes,
                           hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
        if (g_rccl.AllGather(own, c->staging, slab, kNcclUint8, lane.comm, stream)) return 1;
        for (int r = 0; r < c->world; ++r) {
            if (r == c->rank) continue;
            if (hipMemcpyAsync(frame + c->bandBegin[r] * rowBytes, static_cast<char*>(c->staging) + slab * r,
                               (c->bandEnd[r] - c->bandBegin[r]) * rowBytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
        }
        return 0;
    }
    if (d->kind == GFXH_EXCHANGE_BROADCAST) {
        for (uint32_t k = 0; k < d->numBuffers; ++k)
            if (g_rccl.Broadcast(d->buffers[k].base, d->buffers[k].base, d->buffers[k].planeStride, kNcclUint8, 0, lane.comm, stream)) return 1;
        return 0;
    }
    if (d->kind == GFXH_EXCHANGE_GATHER_RECORDS) {
        // counts of every rank (one all-gather of a u32 + a host round trip: the caller needs the total on the host anyway),
        // then per array an all-gather of slabs sized for the largest count and a compaction into rank order
        uint32_t* hostCounts = static_cast<uint32_t*>(d->counters);
        // staging layout: [0, headBytes) the counts (world gathered + world own, u32), then the record slabs
        const size_t headBytes = ((2 * sizeof(uint32_t) * static_cast<size_t>(c->world) + 255) / 256) * 256;
        const size_t need = headBytes;
        if (c->stagingBytes < need) {
            if (c->staging) (void)hipFree(c->staging);
            c->staging = nullptr; c->stagingBytes = 0;
            const size_t first = std::max<size_t>(need, 1 << 20);
            if (hipMalloc(&c->staging, first) != hipSuccess) { c->staging = nullptr; return 1; }
            c->stagingBytes = first;
        }
        uint32_t* dCounts = static_cast<uint32_t*>(c->staging);
        if (hipMemcpyAsync(dCounts + c->world + c->rank, &hostCounts[0], 4, hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
        if (g_rccl.AllGather(dCounts + c->world + c->rank, dCounts, 1, kNcclUint32, lane.comm, stream)) return 1;
        std::vector<uint32_t> counts(c->world);
        if (hipMemcpyAsync(counts.data(), dCounts, 4 * c->world, hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
        if (hipStreamSynchronize(stream) != hipSuccess) return 1;
        uint32_t most = 0, total = 0, first = 0;
        for (int r = 0; r < c->world; ++r) { most = std::max(most, counts[r]); if (r < c->rank) first += counts[r]; total += counts[r]; }
        if (total > d->numCounters) { g_rcclError = "gfxh_rccl_exchange: more records than the arrays hold"; return 1; }
        for (uint32_t k = 0; k < d->numBuffers && most; ++k) {
            const size_t rec = d->buffers[k].bytesPerPixel, slab = rec * most;
            const size_t bytes = headBytes + slab * c->world;
            if (c->stagingBytes < bytes) {
                if (hipStreamSynchronize(stream) != hipSuccess) return 1;
                (void)hipFree(c->staging);
                if (hipMalloc(&c->staging, bytes) != hipSuccess) { c->staging = nullptr; c->stagingBytes = 0; return 1; }
                c->stagingBytes = bytes;
            }
            char* slabs = static_cast<char*>(c->staging) + headBytes;
            char* base = static_cast<char*>(d->buffers[k].base);
            if (hipMemcpyAsync(slabs + slab * c->rank, base, rec * counts[c->rank], hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
            if (g_rccl.AllGather(slabs + slab * c->rank, slabs, slab, kNcclUint8, lane.comm, stream)) return 1;
            size_t at = 0;
            for (int r = 0; r < c->world; ++r) {
                if (counts[r] && hipMemcpyAsync(base + at * rec, slabs + slab * r, rec * counts[r], hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
                at += counts[r];
            }
        }
        hostCounts[0] = total; hostCounts[1] = first;
        return 0;
    }
    g_rcclError = "gfxh_rccl_exchange: unknown exchange kind";
    return 1;
}

} // extern "C"
