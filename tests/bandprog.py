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
        steps, new_res, new_base = api.frame_program(cfg, strips, self.max_motion_rows, new_sequence, self.last_res, self.last_base, unbiased)
        if self.regir_params is not None:
            self.osc.regir_set_params(self.regir_params)
        for k, st in enumerate(steps):
            rect = None if (st.rowBegin == 0 and st.rowEnd == 0) else (0, st.rowBegin, W, st.rowEnd)
            if st.op == api.STEP_RESTIR_PASS:
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
