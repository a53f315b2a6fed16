"""The frame program of the host driver (gfxh_restir_frame_program) executed with the CPU oracle.

Test infrastructure: the multi-process CPU tests run the SAME step list the GPU driver executes (passes with their
row ranges, exchange points with their descriptors from gfxh_frame_step_exchange_desc) with the oracle standing in for
the kernels, so the band logic and the communication code are the production code and only the arithmetic differs.
The bookkeeping around the program mirrors gfxh_restir_render_frame (csrc/host/restir_driver.cpp)."""
import numpy as np

from gfxexp_amd import api
from oracle import oracle as O
from tests import util


class OracleBandRenderer:
    def __init__(self, osc, cfg, regir=None, seed=util.PIXEL_RNG_SEED):
        self.osc, self.cfg = osc, cfg
        self.W, self.H = cfg.width, cfg.height
        self.pb = util.PixelBuffers(self.W, self.H, seed)
        self.sp = self.pb.host_static_params()
        self.regir = regir
        self.regir_params = regir.host_params() if regir is not None else None
        self.frame_index = 0
        self.last_res, self.last_base = 1, 0
        self.prev_cam = None
        self.exchange = None
        self.max_motion_rows = 0
        self.strip_mode = 3           # what gfxh_restir_render_frame runs by default (GFX_STRIP_MODE)
        self.log = []

    def set_exchange(self, fn, max_motion_rows=0):
        self.exchange, self.max_motion_rows = fn, max_motion_rows

    def render_frame(self, cam):
        cfg, W, H = self.cfg, self.W, self.H
        frame = self.frame_index
        new_sequence = frame == 0
        unbiased = cfg.renderer in (api.RENDERER_UNBIASED, api.RENDERER_REARCH_UNBIASED)
        ocam = util.copy_struct(O.GfxCamera, cam)
        f = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, W, H, ocam, prev_cam=self.prev_cam if frame else None,
                              frameIndex=frame, bufferIndex=frame % 2, resetFlowBuffer=int(new_sequence), numAccumFrames=0,
                              numSpatialNeighbors=cfg.numSpatialNeighbors, spatialNeighborRadius=cfg.spatialNeighborRadius,
                              log2NumCandidateSamples=cfg.log2NumCandidateSamples, enableTemporalReuse=cfg.enableTemporalReuse,
                              enableSpatialReuse=cfg.enableSpatialReuse, useUnbiasedEstimator=int(unbiased),
                              useLowDiscrepancyNeighbors=cfg.useLowDiscrepancyNeighbors, reuseVisibility=cfg.reuseVisibility)
        whole = cfg.rowBegin == 0 and cfg.rowEnd == 0
        strips = self.exchange is not None and not whole
        # stripMode 2: what gfxh_restir_render_frame runs -- the seam rows of a spatial pass that another pass follows go first
        steps, new_res, new_base = api.frame_program(cfg, self.strip_mode if strips else 0, self.max_motion_rows, new_sequence, self.last_res, self.last_base, unbiased)
        if self.regir_params is not None:
            self.osc.regir_set_params(self.regir_params)
        for k, st in enumerate(steps):
            rect = None if (st.rowBegin == 0 and st.rowEnd == 0) else (0, st.rowBegin, W, st.rowEnd)
            if st.op == api.STEP_RESTIR_PASS and st.gapEnd > st.gapBegin:      # rows [rowBegin, gapBegin) + [gapEnd, rowEnd)
                for rb, re in ((st.rowBegin, st.gapBegin), (st.gapEnd, st.rowEnd)):
                    if re > rb:
                        self.osc.restir_launch(self.sp, f, st.currentReservoirIndex, st.spatialNeighborBaseIndex, st.pass_, rect=(0, rb, W, re))
            elif st.op == api.STEP_RESTIR_PASS:
                self.osc.restir_launch(self.sp, f, st.currentReservoirIndex, st.spatialNeighborBaseIndex, st.pass_, rect=rect)
            elif st.op == api.STEP_PT_PASS:
                self.osc.pt_launch(self.sp, f, st.pass_, cfg.maxPathLength, rect=rect)
            elif st.op in (api.STEP_EXCHANGE_STRIPS, api.STEP_ALLREDUCE_CELL_ACCESSES, api.STEP_GATHER_BANDS):
                d = api.exchange_desc(cfg, st, k, self.sp, self.regir_params, frame % 2)
                self.log.append((frame, st.op, st.exchangeRows, st.buffers))
                self.exchange(0, d)
        self.last_res, self.last_base = new_res, new_base
        self.prev_cam = ocam
        self.frame_index += 1


def small_config(W, H, renderer, band=(0, 0), radius=5.0, passes=2, neighbors=3):
    cfg = api.RestirRenderer.default_config(W, H, renderer)
    cfg.spatialNeighborRadius = radius
    cfg.numSpatialReusePasses = passes
    cfg.numSpatialNeighbors = neighbors
    cfg.enableAccumulation = 0
    cfg.rowBegin, cfg.rowEnd = band
    return cfg


class OracleNrcBandRenderer:
    """gfxh_nrc_render_frame's band sequence (csrc/host/nrc_driver.cpp) with the oracle as the kernels and oracle/nrc_net.py as
    the network: path tracing / inference / accumulation on the band, records gathered in rank order, every rank trains its own copy
    of the network on the gathered batch (train_on_rank0: rank 0 trains, the inference parameters are broadcast -- the driver's
    GFX_NRC_TRAIN_ON_RANK0=1).  band = (0, 0) and exchange = None: the whole-frame loop."""

    def __init__(self, osc, hs, W, H, band=(0, 0), rank=0, exchange=None, max_len=3, train_on_rank0=False):
        self.train_on_rank0 = train_on_rank0
        from oracle import nrc_net as N
        self.N = N
        self.osc, self.W, self.H, self.band, self.rank, self.exchange, self.max_len = osc, W, H, band, rank, exchange, max_len
        self.pb = util.PixelBuffers(W, H)
        self.nb = util.NrcBuffers(W, H, hs.bounds())
        self.net = N.NrcNet(N.POS_TRIANGLEWAVE, 2, 1e-2)
        self.frame = 0
        self.counts = np.zeros(2, np.uint32)
        self.last_loss = None

    def _desc(self, kind):
        d = api.GfxhExchangeDesc()
        d.kind, d.width, d.height, d.bandBegin, d.bandEnd = kind, self.W, self.H, self.band[0], self.band[1]
        return d

    def render_frame(self, cam):
        W, H, frame, nb, pb = self.W, self.H, self.frame, self.nb, self.pb
        b = frame % 2
        banded = self.band != (0, 0)
        rect = (0, self.band[0], W, self.band[1]) if banded else None
        s = pb.host_static_params()
        ocam = util.copy_struct(O.GfxCamera, cam)
        f = util.frame_params(O.GfxRestirFrameParams, O.GfxCamera, W, H, ocam, frameIndex=frame, bufferIndex=b,
                              resetFlowBuffer=int(frame == 0), numAccumFrames=0)
        self.osc.nrc_set_render_params(nb.host_params(3 + 7 * frame, 5 + 11 * frame, frame == 0))
        self.osc.pt_launch(s, f, api.PT_SETUP_GBUFFERS, self.max_len, rect=rect)
        self.osc.pt_launch(s, f, api.PT_NRC_PREPROCESS, self.max_len)
        self.osc.pt_launch(s, f, api.PT_PATH_TRACE_NRC, self.max_len, rect=rect)
        a = nb.a
        tile = a[f"nrc_tile_{b}"]
        tiles = ((W + int(tile[0]) - 1) // int(tile[0])) * ((H + int(tile[1]) - 1) // int(tile[1]))
        n = W * H
        if banded:
            rows = slice(self.band[0] * W, self.band[1] * W)
            a["nrc_inferred"][rows] = self.net.infer(a["nrc_queries"][rows])
            a["nrc_inferred"][n:n + tiles] = self.net.infer(a["nrc_queries"][n:n + tiles])
        else:
            a["nrc_inferred"][:n + tiles] = self.net.infer(a["nrc_queries"][:n + tiles])
        self.osc.pt_launch(s, f, api.PT_NRC_ACCUMULATE, self.max_len, rect=rect)
        self.osc.pt_launch(s, f, api.PT_NRC_PROPAGATE, self.max_len)
        if banded:
            d = self._desc(api.EXCHANGE_GATHER_RECORDS)
            d.numBuffers = 2
            d.buffers[0].base, d.buffers[0].bytesPerPixel, d.buffers[0].numPlanes = a["nrc_trainq_0"].ctypes.data, 56, 1
            d.buffers[1].base, d.buffers[1].bytesPerPixel, d.buffers[1].numPlanes = a["nrc_traint_0"].ctypes.data, 12, 1
            self.counts[0], self.counts[1] = a[f"nrc_num_{b}"][0], 0
            d.counters, d.numCounters = self.counts.ctypes.data, nb.TRAIN
            self.exchange(0, d)
            a[f"nrc_num_{b}"][0] = self.counts[0]
        self.osc.pt_launch(s, f, api.PT_NRC_SHUFFLE, self.max_len)
        rank0_trains = banded and self.train_on_rank0
        if not rank0_trains or self.rank == 0:
            for step in range(4):
                sl = slice(step * 16384, (step + 1) * 16384)
                self.last_loss = self.net.train(a["nrc_trainq_1"][sl], a["nrc_traint_1"][sl])
        if rank0_trains:
            d = self._desc(api.EXCHANGE_BROADCAST)
            d.numBuffers = 1
            d.buffers[0].base, d.buffers[0].bytesPerPixel, d.buffers[0].numPlanes, d.buffers[0].planeStride = self.net.ema.ctypes.data, 1, 1, self.net.ema.nbytes
            self.exchange(0, d)
        if banded:
            d = self._desc(api.EXCHANGE_GATHER_BANDS)
            d.numBuffers = 1
            d.buffers[0].base, d.buffers[0].bytesPerPixel, d.buffers[0].numPlanes, d.buffers[0].planeStride = pb.beauty.ctypes.data, 16, 1, 16 * n
            self.exchange(0, d)
        self.frame += 1
