"""Strip exchange between band renderers that share ONE GPU (test infrastructure).

A 1-GPU box cannot run two RCCL ranks, so the -m gpu tests drive N band renderers from N host threads of one process;
their exchange callbacks meet at a barrier and copy the rows straight out of the neighbour's buffers (every buffer is a
full-frame buffer with the same row indexing, so "rows [a, b) from the rank above" is rows [a, b) of its buffer).  The
descriptors are the ones the production callbacks (tilesplit.StripExchange over RCCL, gfxh_rccl_exchange) receive."""
import ctypes
import threading

import torch

from gfxexp_amd import api, tilesplit


class LoopbackExchange:
    def __init__(self, world, timeout=120.0):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=timeout)
        self.descs = [None] * world
        self.staged = [None] * world
        self.calls = [[] for _ in range(world)]

    def callback(self, rank):
        return lambda stream, d: self._exchange(rank, d)

    @staticmethod
    def _rows(d, k, plane, r):
        b = d.buffers[k]
        row_bytes = b.bytesPerPixel * d.width
        n = (int(r[1]) - int(r[0])) * row_bytes
        return tilesplit.device_bytes(b.base + plane * b.planeStride + int(r[0]) * row_bytes, n) if n > 0 else None

    def _exchange(self, rank, d):
        torch.cuda.synchronize()
        mine = api.GfxhExchangeDesc()
        ctypes.memmove(ctypes.byref(mine), ctypes.byref(d), ctypes.sizeof(mine))
        self.descs[rank] = mine
        self.calls[rank].append((mine.kind, mine.stage))
        self.barrier.wait()
        kinds = {x.kind for x in self.descs}
        assert len(kinds) == 1, "the ranks are at different exchange points"
        if mine.kind == api.EXCHANGE_STRIPS:
            for k in range(mine.numBuffers):
                for plane in range(mine.buffers[k].numPlanes):
                    for peer, recv, send in ((rank - 1, mine.recvAbove, "sendBelow"), (rank + 1, mine.recvBelow, "sendAbove")):
                        dst = self._rows(mine, k, plane, recv)
                        if dst is None:
                            continue
                        other = self.descs[peer]
                        src_rows = getattr(other, send)
                        assert list(src_rows) == list(recv), "the neighbour sends other rows than this rank expects"
                        dst.copy_(self._rows(other, k, plane, src_rows))
        elif mine.kind == api.EXCHANGE_ALLREDUCE_SUM_U32:
            total = sum(tilesplit.device_bytes(x.counters, 4 * x.numCounters).view(torch.int32).clone() for x in self.descs)
            torch.cuda.synchronize()
            self.barrier.wait()          # everyone has read everyone's counters
            tilesplit.device_bytes(mine.counters, 4 * mine.numCounters).view(torch.int32).copy_(total)
        elif mine.kind == api.EXCHANGE_GATHER_RECORDS:
            counts_host = (ctypes.c_uint32 * 2).from_address(mine.counters)
            n = int(counts_host[0])
            self.staged[rank] = (n, [tilesplit.device_bytes(mine.buffers[k].base, max(1, n * mine.buffers[k].bytesPerPixel)).clone()
                                     for k in range(mine.numBuffers)])
            torch.cuda.synchronize()
            self.barrier.wait()          # everyone's records are staged and counted
            counts = [self.staged[r][0] for r in range(self.world)]
            for k in range(mine.numBuffers):
                rec = mine.buffers[k].bytesPerPixel
                whole = tilesplit.device_bytes(mine.buffers[k].base, mine.numCounters * rec)
                at = 0
                for r, c in enumerate(counts):
                    if c:
                        whole[at * rec:(at + c) * rec].copy_(self.staged[r][1][k][:c * rec])
                    at += c
            counts_host[0], counts_host[1] = sum(counts), sum(counts[:rank])
        elif mine.kind == api.EXCHANGE_BROADCAST:
            if rank != 0:
                for k in range(mine.numBuffers):
                    n = mine.buffers[k].planeStride
                    tilesplit.device_bytes(mine.buffers[k].base, n).copy_(tilesplit.device_bytes(self.descs[0].buffers[k].base, n))
        elif mine.kind == api.EXCHANGE_GATHER_BANDS:
            for peer, other in enumerate(self.descs):
                if peer != rank:
                    r = (other.bandBegin, other.bandEnd)
                    self._rows(mine, 0, 0, r).copy_(self._rows(other, 0, 0, r))
        torch.cuda.synchronize()
        self.barrier.wait()


def run_bands(renderers, frames, before_frame=None):
    """Render `frames` frames on every band renderer, one host thread per band.  `before_frame(frame, rank, renderer)`
    runs on the band's thread before each frame (camera moves)."""
    errors = []

    def work(rank, r):
        try:
            for frame in range(frames):
                if before_frame is not None:
                    before_frame(frame, rank, r)
                r.render_frame()
            torch.cuda.synchronize()
        except BaseException as e:      # collected here and raised by the caller below (the band's thread just ends)
            errors.append((rank, e))
    threads = [threading.Thread(target=work, args=(rank, r), daemon=True) for rank, r in enumerate(renderers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads), "a band renderer is stuck"
