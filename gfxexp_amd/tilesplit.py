"""Pixel-band split of one frame across the GPUs of a node.

The reference is single-GPU (restir_di/restir_di_main.cpp:130-133).  Here the frame is cut into
`world` horizontal bands whose heights are multiples of 8 rows (so the 8x8 tiles of the
rearchitected per-pixel RIS never straddle ranks); each process renders its band (plus the halo the
reuse passes read) and the float4 HDR bands are all-gathered once per frame over RCCL/xGMI
(torch.distributed backend "nccl" is RCCL on ROCm; "gloo" on CPU for tests).
"""
import numpy as np


def band_rows(height, world):
    """Row ranges [(begin, end)] for every rank: multiples of 8 rows, remainder spread from rank 0."""
    tiles = (height + 7) // 8
    base, extra = divmod(tiles, world)
    out, row = [], 0
    for r in range(world):
        h = (base + (1 if r < extra else 0)) * 8
        end = min(height, row + h)
        out.append((row, end))
        row = end
    assert row == height and all(e >= b for b, e in out)
    return out


def band_for_rank(height, world, rank):
    if world <= 1:
        return (0, 0)            # 0,0 = whole frame
    return band_rows(height, world)[rank]


def halo_rows(radius, num_spatial_passes, max_motion=0):
    """Rows of neighbour state a band needs beyond its own rows before the first reuse pass
    (radius x passes for spatial reuse, plus the largest motion vector for temporal reuse)."""
    return int(np.ceil(radius)) * int(num_spatial_passes) + int(np.ceil(max_motion))


class BandGather:
    """All-gather of the float4 HDR bands into the full frame on every rank.

    Bands can differ by 8 rows, so every rank contributes a max-band-sized slab and the slabs are
    scattered back to their rows.  One collective per frame: W*H*16 bytes total (33 MB at 1080p),
    4.15 MB per rank at 8 ranks.  The collective is asynchronous: nothing of the next frame depends on the other
    ranks' pixels, so it overlaps that frame's rendering."""

    def __init__(self, beauty_view, width, height, world, rank, dist, bands=None):
        import torch
        self.torch = torch
        self.dist = dist
        self.w, self.h, self.world, self.rank = width, height, world, rank
        self.bands = [(int(b), int(e)) for b, e in bands] if bands is not None else band_rows(height, world)
        self.max_rows = max(e - b for b, e in self.bands)
        self.beauty = beauty_view                      # tensor view of the full-frame beauty buffer [H*W*4]
        device = beauty_view.device
        self.send = torch.zeros(self.max_rows * width * 4, dtype=torch.float32, device=device)
        self.recv = torch.zeros(world * self.max_rows * width * 4, dtype=torch.float32, device=device)

    def all_gather(self):
        """Issue this frame's collective and return: it runs on the communicator's stream underneath the next
        frame's kernels.  The received bands of the PREVIOUS call are put into place first (`finish`), so every
        rank holds the complete frame one call later -- or right after `finish()`."""
        self.finish()
        b, e = self.bands[self.rank]
        n = (e - b) * self.w * 4
        self.send[:n].copy_(self.beauty[b * self.w * 4:e * self.w * 4])
        self.work = self.dist.all_gather_into_tensor(self.recv, self.send, async_op=True)
        return self.work

    def finish(self):
        """Wait (stream-ordered on RCCL, blocking on gloo) for the outstanding collective and scatter the other
        ranks' bands into the full-frame buffer.  Call once after the last frame."""
        work = getattr(self, "work", None)
        if work is None:
            return self.beauty
        work.wait()
        self.work = None
        slab = self.max_rows * self.w * 4
        for r, (rb, re) in enumerate(self.bands):
            if r == self.rank:
                continue
            m = (re - rb) * self.w * 4
            self.beauty[rb * self.w * 4:re * self.w * 4].copy_(self.recv[r * slab:r * slab + m])
        return self.beauty


class HaloExchange:
    """Per-frame refresh of the halo rows' FINAL temporal state from their owners (gfxh_band_plan).

    `state` maps names to flat torch tensors viewing this rank's full-frame buffers:
        "rng"   int64  [H*W]          PCG32 state per pixel
        "info0" "info1"  float32 [H*W*2]   ReservoirInfo ping-pong
        "res0"  "res1"   float32 [3*H*W*4] reservoir planes ping-pong
    Only the buffers of `last_res_index` (the frame's final reservoirs) and the RNG are exchanged:
    64 bytes per halo pixel per neighbour (~4.9 MB per neighbour at 1080p, 40 halo rows).
    Works with any torch.distributed backend (RCCL on the GPUs, gloo in the CPU tests)."""

    def __init__(self, state, plan, width, height, rank, world, dist):
        self.state, self.plan, self.w, self.h = state, plan, width, height
        self.rank, self.world, self.dist = rank, world, dist

    def _slices(self, rows, last):
        b, e = int(rows[0]), int(rows[1])
        if e <= b:
            return []
        n = self.w * self.h
        out = [self.state["rng"][b * self.w:e * self.w], self.state["info%d" % last][b * self.w * 2:e * self.w * 2]]
        res = self.state["res%d" % last]
        for plane in range(3):
            out.append(res[plane * n * 4 + b * self.w * 4:plane * n * 4 + e * self.w * 4])
        return out

    def exchange(self, last_res_index):
        dist, ops = self.dist, []
        p = self.plan
        if self.rank > 0:
            for t in self._slices(p.sendAbove, last_res_index):
                ops.append(dist.P2POp(dist.isend, t, self.rank - 1))
            for t in self._slices(p.recvAbove, last_res_index):
                ops.append(dist.P2POp(dist.irecv, t, self.rank - 1))
        if self.rank < self.world - 1:
            for t in self._slices(p.sendBelow, last_res_index):
                ops.append(dist.P2POp(dist.isend, t, self.rank + 1))
            for t in self._slices(p.recvBelow, last_res_index):
                ops.append(dist.P2POp(dist.irecv, t, self.rank + 1))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()


class StripExchange:
    """The exchange callback of a band renderer (gfxh_restir_set_exchange / gfxh_exchange_desc) over torch.distributed:
    RCCL on the GPUs (backend "nccl"), gloo in the CPU tests.  `view(ptr, nbytes)` turns an address into a flat uint8
    tensor (device_bytes for device memory, host_view for the oracle's numpy buffers); views are cached per address.

      strips        batch_isend_irecv of every (buffer, plane) row range with the rank above / below: one RCCL group
                    per exchange point, stream-ordered after the pass that produced the rows (the collective stream
                    waits for torch's current stream, which is the stream the renderer launches on)
      counters      all_reduce(sum) on the u32 array (as int32: wrap-around addition is the same)
      HDR bands     all_gather_into_tensor of slabs sized for the tallest band.  Synchronous by default, like the C++
                    twin: when render_frame returns, the frame buffer holds every rank's rows of THIS frame (stream-ordered
                    on RCCL), so any reader -- tone map, save, MSE, copy_to_linear -- sees a complete frame.
                    async_gather=True (bench.py): nothing of the next frame reads the other ranks' pixels, so the
                    collective is left running underneath the next frame's kernels and the received bands are put into the
                    frame at the NEXT gather -- until then the other ranks' rows are one frame old -- or by finish(), which
                    a caller in this mode must invoke (with the renderer's stream current) before it reads the frame.
    Lanes (gfxexp_host.h gfxh_lane): the driver hands every exchange the stream of the lane it belongs to -- the G-buffer strips on
    the renderer's G-buffer stream, the band gather (gfxh_restir_set_async_gather) on its gather stream.  On a device the callback
    makes that stream torch's current stream for the call (the collective is ordered with it), and `lane_groups` = {lane: process
    group} (make_lane_groups) gives each lane its own communicator so that the lanes do not queue behind each other.
    The C++ twin is gfxh_rccl_exchange (csrc/host/rccl_exchange.cpp); both consume the same descriptors."""

    def __init__(self, dist, rank, world, height, view, device="cpu", async_gather=False, bands=None, lane_groups=None):
        self.dist, self.rank, self.world, self.device = dist, rank, world, device
        self.async_gather = bool(async_gather)
        self.lane_groups = dict(lane_groups or {})
        self._view, self._views = view, {}
        # `bands`: an explicit partition [(begin, end)] per rank (cost-balanced bands, api.balance_bands) instead of the equal one;
        # every rank must pass the same list.  RestirRenderer.set_exchange then checks THAT partition (api.check_bands).
        self.custom_bands = [(int(b), int(e)) for b, e in bands] if bands is not None else None
        self.bands = self.custom_bands if bands is not None else band_rows(height, world)
        assert len(self.bands) == world and self.bands[0][0] == 0 and self.bands[-1][1] == height
        self.bytes_moved = 0
        self._stage = None
        self._pending = None

    def view(self, ptr, nbytes):
        key = (int(ptr), int(nbytes))
        t = self._views.get(key)
        if t is None:
            t = self._views[key] = self._view(ptr, nbytes)
        return t

    def __call__(self, stream, d):
        import torch
        if self.device != "cpu" and stream:
            # the lane's stream (a raw hipStream_t from the driver) is torch's current stream for the call
            with torch.cuda.stream(torch.cuda.ExternalStream(int(stream))):
                return self._exchange(d)
        return self._exchange(d)

    def _exchange(self, d):
        from gfxexp_amd import api
        import torch
        dist = self.dist
        group = self.lane_groups.get(int(d.lane))
        kw = {"group": group} if group is not None else {}
        if d.kind == api.EXCHANGE_ALLREDUCE_SUM_U32:
            t = self.view(d.counters, 4 * d.numCounters).view(torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, **kw)
            return
        if d.kind == api.EXCHANGE_STRIPS:
            ops = []
            for k in range(d.numBuffers):
                b = d.buffers[k]
                row_bytes = b.bytesPerPixel * d.width
                for plane in range(b.numPlanes):
                    base = b.base + plane * b.planeStride

                    def rows(r):
                        n = (int(r[1]) - int(r[0])) * row_bytes
                        return self.view(base + int(r[0]) * row_bytes, n) if n > 0 else None
                    if self.rank > 0:
                        for t, op in ((rows(d.sendAbove), dist.isend), (rows(d.recvAbove), dist.irecv)):
                            if t is not None:
                                ops.append(dist.P2POp(op, t, self.rank - 1, **kw))
                    if self.rank < self.world - 1:
                        for t, op in ((rows(d.sendBelow), dist.isend), (rows(d.recvBelow), dist.irecv)):
                            if t is not None:
                                ops.append(dist.P2POp(op, t, self.rank + 1, **kw))
            self.bytes_moved += sum(op.tensor.numel() for op in ops)
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            return
        if d.kind == api.EXCHANGE_GATHER_BANDS:
            self.finish()
            b = d.buffers[0]
            row_bytes = b.bytesPerPixel * d.width
            max_rows = max(e - s for s, e in self.bands)
            slab = max_rows * row_bytes
            if self._stage is None or self._stage[0].numel() != slab:
                self._stage = (torch.zeros(slab, dtype=torch.uint8, device=self.device),
                               torch.zeros(slab * self.world, dtype=torch.uint8, device=self.device))
            send, recv = self._stage
            s0, e0 = self.bands[self.rank]
            frame = self.view(b.base, row_bytes * d.height)
            send[:(e0 - s0) * row_bytes].copy_(frame[s0 * row_bytes:e0 * row_bytes])
            work = dist.all_gather_into_tensor(recv, send, async_op=True, **kw)
            self._pending = (work, frame, row_bytes, slab)
            if not self.async_gather:
                self.finish()
            return
        if d.kind == api.EXCHANGE_GATHER_RECORDS:
            # NRC band renderers: variable-length record arrays of every rank, concatenated in rank order on every rank
            import ctypes
            counts_host = (ctypes.c_uint32 * 2).from_address(d.counters)
            mine = int(counts_host[0])
            counts = torch.zeros(self.world, dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(counts, torch.tensor([mine], dtype=torch.int64, device=self.device), **kw)
            counts = [int(c) for c in counts.cpu()]
            total = sum(counts)
            if total > d.numCounters:
                raise ValueError("gathered %d records, capacity %d" % (total, d.numCounters))
            most, first = max(counts), sum(counts[:self.rank])
            for k in range(d.numBuffers):
                rec = d.buffers[k].bytesPerPixel
                if most == 0:
                    continue
                send = torch.zeros(most * rec, dtype=torch.uint8, device=self.device)
                recv = torch.zeros(most * rec * self.world, dtype=torch.uint8, device=self.device)
                whole = self.view(d.buffers[k].base, d.numCounters * rec)
                send[:mine * rec].copy_(whole[:mine * rec])
                dist.all_gather_into_tensor(recv, send, **kw)
                at = 0
                for r, c in enumerate(counts):
                    whole[at * rec:(at + c) * rec].copy_(recv[r * most * rec:r * most * rec + c * rec])
                    at += c
                self.bytes_moved += most * rec * self.world
            counts_host[0], counts_host[1] = total, first
            return
        if d.kind == api.EXCHANGE_BROADCAST:
            for k in range(d.numBuffers):
                dist.broadcast(self.view(d.buffers[k].base, d.buffers[k].planeStride), src=0, **kw)
            return
        raise ValueError("unknown exchange kind %d" % d.kind)

    def finish(self):
        """Wait (stream-ordered on RCCL, blocking on gloo) for the outstanding band gather and put the other ranks' bands
        into the frame.  Called by the next gather; call it once after the last frame."""
        if self._pending is None:
            return
        work, frame, row_bytes, slab = self._pending
        self._pending = None
        work.wait()
        recv = self._stage[1]
        for r, (rs, re) in enumerate(self.bands):
            if r != self.rank:
                frame[rs * row_bytes:re * row_bytes].copy_(recv[r * slab:r * slab + (re - rs) * row_bytes])


def make_lane_groups(dist, lanes=(1, 2, 3)):
    """One extra process group (= one more RCCL communicator and stream) per lane beyond MAIN; every rank must call this at the same
    point.  {lane: group} for StripExchange(lane_groups=...)."""
    return {int(lane): dist.new_group() for lane in lanes}


class HostStaged:
    """torch.distributed (gloo) with every CUDA tensor staged through host memory, behind the handful of calls StripExchange and
    bench.py make.  RCCL refuses two ranks on one device, so this is how the multi-rank frame loop (band partition, cost balancing,
    strip exchange between the passes, band gather, max-over-ranks timing) is exercised end to end on a ONE-GPU box:
    `GFX_BENCH_ONE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` puts every rank on device 0.
    A functional check (tests/test_gpu_strip_exchange.py), not a transport to measure: every call drains the device first."""

    class _Op:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Done:
        def wait(self):
            pass

    def __init__(self, dist):
        self._d = dist
        self.ReduceOp = dist.ReduceOp

    def isend(self, *a, **k):      # only ever used as tags of P2POp
        raise NotImplementedError

    def irecv(self, *a, **k):
        raise NotImplementedError

    def P2POp(self, op, tensor, peer, group=None):
        return HostStaged._Op("send" if op == self.isend else "recv", tensor, peer)

    def batch_isend_irecv(self, ops):
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        pending = []
        for o in ops:
            if o.op == "send":
                h = o.tensor.cpu()
                pending.append((self._d.isend(h, o.peer), None, h))
            else:
                h = torch.empty(o.tensor.shape, dtype=o.tensor.dtype)
                pending.append((self._d.irecv(h, o.peer), o.tensor, h))

        class _Req:
            def wait(self_inner):
                for work, dst, h in pending:
                    work.wait()
                    if dst is not None:
                        dst.copy_(h)
        return [_Req()]

    def all_reduce(self, t, op=None, async_op=False, group=None):
        h = t.cpu()
        self._d.all_reduce(h, op=op if op is not None else self.ReduceOp.SUM)
        t.copy_(h)
        return HostStaged._Done()

    def all_gather_into_tensor(self, out, inp, async_op=False, group=None):
        import torch
        ho = torch.empty(out.shape, dtype=out.dtype)
        self._d.all_gather_into_tensor(ho, inp.cpu())
        out.copy_(ho)
        return HostStaged._Done()

    def broadcast(self, t, src=0, group=None):
        h = t.cpu()
        self._d.broadcast(h, src=src)
        t.copy_(h)

    def barrier(self):
        self._d.barrier()

    def destroy_process_group(self):
        self._d.destroy_process_group()


def host_view(ptr, nbytes):
    """Flat uint8 torch tensor over host memory at `ptr` (the oracle's numpy buffers in the CPU tests)."""
    import ctypes
    import torch
    return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_uint8 * int(nbytes)).from_address(int(ptr))))


def device_bytes(ptr, nbytes):
    return device_view(ptr, nbytes, "|u1")


def device_view(ptr, count, typestr="<f4"):
    """Wrap a raw device pointer as a flat torch tensor without copying (CUDA array interface)."""
    import torch

    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device="cuda")


def renderer_state_views(renderer, width, height):
    """Tensor views of a gfxexp RestirRenderer's device buffers for HaloExchange / BandGather."""
    s, _, _, _, _ = renderer.params()
    n = width * height
    state = {"rng": device_view(s.rngBuffer, n, "<i8"), "beauty": device_view(s.beautyAccumBuffer, n * 4)}
    for i in range(2):
        state["info%d" % i] = device_view(s.reservoirInfoBuffer[i], n * 2)
        state["res%d" % i] = device_view(s.reservoirBuffer[i], n * 12)
    return state
