"""The send / receive plan of gfxh_rccl_exchange (gfxexp_amd/csrc/host/rccl_exchange.cpp) for EVERY rank of a world, in
one process, against the recording librccl stand-in (tests/native/rccl_stub.cpp, loaded through GFX_RCCL_LIBRARY).

Run as a script by the tests (the library is chosen once per process, so the stub must be named before libgfxexp loads
librccl):  python tests/rccl_plan.py cpu|gpu  -> prints "ok ..." or raises.

  cpu   strips, counter all-reduce and broadcast for all 8 ranks of a 1920x1080 frame (7 x 136 + 128 rows), every exchange
        step of every renderer's frame program (gfxh_restir_frame_program + gfxh_frame_step_exchange_desc): peer, pointer and
        byte count of every ncclSend / ncclRecv against an independent statement of the strip geometry, and every Send
        matched by the Recv its peer posts.  Pointers are never dereferenced on this path: the "buffers" are address ranges.
  gpu   + the band gather and the NRC record gather with device memory, as rank 3 of 8 and rank 7 of 8: the stub's all-gather
        puts the caller's slab into every rank's slot, so the rows each slot is scattered to are visible in the frame.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STUB = os.path.join(ROOT, "tests", "native", "librccl_stub.so")
SEND, RECV, ALLREDUCE, ALLGATHER, BROADCAST, GROUP_START, GROUP_END = 1, 2, 3, 4, 5, 6, 7
W, H, WORLD = 1920, 1080, 8


class Call(C.Structure):
    _fields_ = [("op", C.c_int32), ("dtype", C.c_int32), ("peer", C.c_int32), ("inGroup", C.c_int32),
                ("a", C.c_uint64), ("b", C.c_uint64), ("count", C.c_uint64)]


def calls(stub):
    out = []
    for i in range(stub.rccl_stub_num_calls()):
        c = Call()
        stub.rccl_stub_get(C.c_uint32(i), C.byref(c))
        out.append(c)
    stub.rccl_stub_reset()
    return out


def fake_static_params(api):
    """gfx_restir_static_params whose buffers are disjoint address ranges (never dereferenced by the strip path)."""
    s = api.GfxRestirStaticParams()
    s.imageSizeX, s.imageSizeY = W, H
    base = [0x100000000]

    def take(nbytes):
        p = base[0]
        base[0] += (nbytes + 0xFFFF) & ~0xFFFF
        return p
    n = W * H
    for i in range(2):
        s.gbuffer0[i], s.gbuffer1[i], s.gbuffer2[i], s.gbuffer3[i] = take(16 * n), take(8 * n), take(16 * n), take(16 * n)
        s.reservoirBuffer[i], s.reservoirInfoBuffer[i], s.sampleVisibilityBuffer[i] = take(48 * n), take(8 * n), take(4 * n)
    s.beautyAccumBuffer = take(16 * n)
    s.rngBuffer = take(8 * n)
    return s


def expected_strips(api, d, rank, bands, rows):
    """Independent statement of the strip geometry: a rank sends its first / last `rows` rows to the rank above / below
    and receives the `rows` rows adjacent to its band from them; every (buffer, plane) is one message per direction."""
    b, e = bands[rank]
    want = []
    for k in range(d.numBuffers):
        buf = d.buffers[k]
        row_bytes = buf.bytesPerPixel * W
        for plane in range(buf.numPlanes):
            base = buf.base + plane * buf.planeStride
            if rank > 0:
                want.append((SEND, rank - 1, base + b * row_bytes, rows * row_bytes))
                want.append((RECV, rank - 1, base + (b - rows) * row_bytes, rows * row_bytes))
            if rank < WORLD - 1:
                want.append((SEND, rank + 1, base + (e - rows) * row_bytes, rows * row_bytes))
                want.append((RECV, rank + 1, base + e * row_bytes, rows * row_bytes))
    return want


def cpu_plan(api, L, stub):
    bands = [api.band_rows(H, WORLD, r) for r in range(WORLD)]
    assert bands[0] == (0, 136) and bands[6] == (816, 952) and bands[7] == (952, 1080)
    ident = (C.c_uint8 * 128)()
    assert L.gfxh_rccl_unique_id(ident) == 0
    comms = []
    for r in range(WORLD):
        comm = C.c_void_p()
        assert L.gfxh_rccl_create(ident, r, WORLD, C.c_uint32(H), C.byref(comm)) == 0, L.gfxh_rccl_last_error()
        comms.append(comm)
    bad = C.c_void_p()
    assert L.gfxh_rccl_create(ident, 8, WORLD, C.c_uint32(H), C.byref(bad)) == 1      # rank outside the world
    assert L.gfxh_rccl_create(ident, 0, WORLD, C.c_uint32(40), C.byref(bad)) == 1     # 40 rows = 5 tiles for 8 ranks: three ranks without a band
    assert b"without a band" in L.gfxh_rccl_last_error()
    sp = fake_static_params(api)
    regir = api.GfxRegirParams()
    regir.perCellNumAccesses = 0x7000000000
    regir.gridDimension[0], regir.gridDimension[1], regir.gridDimension[2] = 32, 8, 32
    checked = 0
    cases = [(api.RENDERER_BIASED, 0), (api.RENDERER_BIASED, 16), (api.RENDERER_UNBIASED, 8), (api.RENDERER_REARCH_BIASED, 0),
             (api.RENDERER_REARCH_UNBIASED, 24), (api.RENDERER_PATH_TRACE_REGIR, 0), (api.RENDERER_PATH_TRACE, 0)]
    for renderer, motion in cases:
        for new_sequence in (True, False):
            per_rank = {}
            for rank in range(WORLD):
                cfg = api.RestirRenderer.default_config(W, H, renderer)
                cfg.rowBegin, cfg.rowEnd = bands[rank]
                unbiased = renderer in (api.RENDERER_UNBIASED, api.RENDERER_REARCH_UNBIASED)
                # stripMode 3 (what the driver runs: one reservoir exchange of radius x passes rows + RNG states) and 1 (one per spatial pass)
                steps = api.frame_program(cfg, 3, motion, new_sequence, 1, 0, unbiased)[0] + api.frame_program(cfg, 1, motion, new_sequence, 1, 0, unbiased)[0]
                log = []
                for k, st in enumerate(steps):
                    if st.op == api.STEP_EXCHANGE_STRIPS:
                        d = api.exchange_desc(cfg, st, k, sp, regir, 1)
                        assert L.gfxh_rccl_exchange(comms[rank], None, C.byref(d)) == 0, L.gfxh_rccl_last_error()
                        got = calls(stub)
                        assert got[0].op == GROUP_START and got[-1].op == GROUP_END, "strips of one exchange point are ONE group"
                        msgs = [(c.op, c.peer, c.a if c.op == SEND else c.b, c.count) for c in got[1:-1]]
                        assert all(c.dtype == 1 and c.inGroup == 1 for c in got[1:-1])
                        want = expected_strips(api, d, rank, bands, st.exchangeRows)
                        assert msgs == want, (renderer, motion, rank, k, msgs[:4], want[:4])
                        assert all(0 <= m[1] < WORLD and abs(m[1] - rank) == 1 for m in msgs)
                        if 0 < rank < WORLD - 1:
                            assert {m[1] for m in msgs} == {rank - 1, rank + 1}        # an interior rank talks to both neighbours
                        log.append((k, msgs))
                        checked += len(msgs)
                    elif st.op == api.STEP_ALLREDUCE_CELL_ACCESSES:
                        d = api.exchange_desc(cfg, st, k, sp, regir, 1)
                        assert L.gfxh_rccl_exchange(comms[rank], None, C.byref(d)) == 0
                        (c,) = calls(stub)
                        assert (c.op, c.a, c.b, c.count, c.dtype) == (ALLREDUCE, 0x7000000000, 0x7000000000, 32 * 8 * 32, 3)
                        checked += 1
                per_rank[rank] = log
            # every Send has the Recv its peer posts at the same exchange point: same byte count, same (buffer, plane) order
            for rank in range(WORLD - 1):
                for (k0, down), (k1, up) in zip(per_rank[rank], per_rank[rank + 1]):
                    assert k0 == k1
                    s_down = [m[3] for m in down if m[0] == SEND and m[1] == rank + 1]
                    r_up = [m[3] for m in up if m[0] == RECV and m[1] == rank]
                    s_up = [m[3] for m in up if m[0] == SEND and m[1] == rank]
                    r_down = [m[3] for m in down if m[0] == RECV and m[1] == rank + 1]
                    assert s_down == r_up and s_up == r_down and s_down
    # broadcast of the NRC inference images: root 0, the byte count of planeStride, in place
    d = api.GfxhExchangeDesc()
    d.kind, d.numBuffers = api.EXCHANGE_BROADCAST, 2
    d.buffers[0].base, d.buffers[0].planeStride = 0x9000000000, 20480
    d.buffers[1].base, d.buffers[1].planeStride = 0x9100000000, 2 << 20
    assert L.gfxh_rccl_exchange(comms[3], None, C.byref(d)) == 0
    got = calls(stub)
    assert [(c.op, c.a, c.b, c.count, c.peer) for c in got] == [(BROADCAST, 0x9000000000, 0x9000000000, 20480, 0),
                                                              (BROADCAST, 0x9100000000, 0x9100000000, 2 << 20, 0)]
    for c in comms:
        L.gfxh_rccl_destroy(c)
    return checked


def gpu_plan(api, L, stub):
    import torch
    ident = (C.c_uint8 * 128)()
    assert L.gfxh_rccl_unique_id(ident) == 0
    bands = [api.band_rows(H, WORLD, r) for r in range(WORLD)]
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rank in (0, 3, 7):
        comm = C.c_void_p()
        assert L.gfxh_rccl_create(ident, rank, WORLD, C.c_uint32(H), C.byref(comm)) == 0
        # ---- band gather: the frame holds row numbers; after the gather (every slot = this rank's slab) row r of band q
        # must hold row (own band begin + r - band q begin), for as many rows as band q has
        frame = torch.arange(H, dtype=torch.float32, device="cuda").repeat_interleave(W * 4).contiguous()
        d = api.GfxhExchangeDesc()
        d.kind, d.width, d.height, d.numBuffers = api.EXCHANGE_GATHER_BANDS, W, H, 1
        d.bandBegin, d.bandEnd = bands[rank]
        d.buffers[0].base, d.buffers[0].bytesPerPixel, d.buffers[0].numPlanes, d.buffers[0].planeStride = frame.data_ptr(), 16, 1, 16 * W * H
        assert L.gfxh_rccl_exchange(comm, stream, C.byref(d)) == 0, L.gfxh_rccl_last_error()
        torch.cuda.synchronize()
        got = calls(stub)
        (ag,) = [c for c in got if c.op == ALLGATHER]
        slab = 136 * W * 16                                            # the tallest band
        assert ag.count == slab and ag.dtype == 1 and ag.a == ag.b + slab * rank, "in-place all-gather of max-band slabs"
        rows = frame.view(H, W * 4)[:, 0].cpu().numpy()
        b0 = bands[rank][0]
        own = bands[rank][1] - bands[rank][0]
        for q, (qb, qe) in enumerate(bands):
            # a slot holds as many valid rows as the CALLER's band has (the stub echoes its slab); a taller band q also
            # receives the slab's padding rows, which carry no data in this emulation
            m = min(qe - qb, own)
            want = np.arange(qb, qb + m) if q == rank else b0 + np.arange(m)
            assert np.array_equal(rows[qb:qb + m], want.astype(np.float32)), (rank, q)
        # a renderer whose band is not this rank's band of the partition is refused
        d.bandBegin += 8
        assert L.gfxh_rccl_exchange(comm, stream, C.byref(d)) == 1
        assert b"gfxh_band_rows" in L.gfxh_rccl_last_error()
        calls(stub)
        # ---- record gather: 2 arrays; every rank "has" this rank's count (the stub echoes it), so the total is 8 x count,
        # this rank's records start at rank x count, and the arrays hold 8 copies of its records in rank order
        count = 700 + rank
        q = torch.arange(count * 14, dtype=torch.float32, device="cuda")
        t = torch.arange(count * 3, dtype=torch.float32, device="cuda") + 0.5
        cap = 1 << 14
        qbuf = torch.zeros(cap * 14, device="cuda"); qbuf[:count * 14] = q
        tbuf = torch.zeros(cap * 3, device="cuda"); tbuf[:count * 3] = t
        host_counts = (C.c_uint32 * 2)(count, 0)
        d = api.GfxhExchangeDesc()
        d.kind, d.width, d.height, d.numBuffers = api.EXCHANGE_GATHER_RECORDS, W, H, 2
        d.buffers[0].base, d.buffers[0].bytesPerPixel, d.buffers[0].numPlanes = qbuf.data_ptr(), 56, 1
        d.buffers[1].base, d.buffers[1].bytesPerPixel, d.buffers[1].numPlanes = tbuf.data_ptr(), 12, 1
        d.counters, d.numCounters = C.addressof(host_counts), cap
        assert L.gfxh_rccl_exchange(comm, stream, C.byref(d)) == 0, L.gfxh_rccl_last_error()
        torch.cuda.synchronize()
        assert list(host_counts) == [WORLD * count, rank * count]
        got = [c for c in calls(stub) if c.op == ALLGATHER]
        assert [(c.count, c.dtype) for c in got] == [(1, 3), (count * 56, 1), (count * 12, 1)]
        assert torch.equal(qbuf[:WORLD * count * 14], q.repeat(WORLD)) and torch.equal(tbuf[:WORLD * count * 3], t.repeat(WORLD))
        L.gfxh_rccl_destroy(comm)
    return 3


def main(mode):
    os.environ["GFX_RCCL_LIBRARY"] = STUB
    if mode == "gpu":
        os.environ["RCCL_STUB_DEVICE"] = "1"
    from gfxexp_amd import api
    L = api.lib()
    L.gfxh_rccl_last_error.restype = C.c_char_p
    stub = C.CDLL(STUB)                  # the same handle the product dlopens
    stub.rccl_stub_num_calls.restype = C.c_uint32
    n = cpu_plan(api, L, stub)
    if mode == "gpu":
        n += gpu_plan(api, L, stub)
    print("ok: %d rccl calls checked (%s)" % (n, mode))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "cpu")
