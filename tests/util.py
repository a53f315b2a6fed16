"""Helpers shared by the tests: build the same scene in the product (gfxexp_amd, through the C ABI)
and in the CPU oracle, allocate ReSTIR pixel buffers on either side, compare buffers."""
import ctypes as C
import contextlib
import os

import numpy as np

from gfxexp_amd import api
from oracle import oracle as O

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "assets")
PIXEL_RNG_SEED = 591842031321323413  # restir_di/restir_di_main.cpp:1317


def to_oracle_material(m):
    om = O.GfxMaterial()
    C.memmove(C.byref(om), C.byref(m), C.sizeof(om))
    return om


_oracle_threads_override = [None]


@contextlib.contextmanager
def every_host_thread():
    """Oracle scenes fed inside the block run their passes on every host thread, whatever the caller asked for (whole frames at
    1920x1080; the oracle's results do not depend on the thread count -- tests/test_oracle_*.py)."""
    import pytest
    if (os.cpu_count() or 1) < 32:
        pytest.skip("a whole 1920x1080 frame on the oracle wants a many-core host (the GPU boxes have 128 threads); the window tests cover the same passes")
    _oracle_threads_override[0] = "max"
    try:
        yield
    finally:
        _oracle_threads_override[0] = None


def feed_oracle(host_scene, threads=None, brute_force=False, config=None, library=None):
    """Push every array of a HostScene into an OracleScene (same slots, same order).  `library`: another build of the
    oracle sources (O.lib_fast(), bench.py's speed-mode CPU baseline); default = the parity build."""
    osc = O.OracleScene(threads=threads, library=library)
    if _oracle_threads_override[0] == "max":
        osc.set_threads(osc.L.orc_max_threads())
    for slot, w, h, fmt, data in host_scene.textures():
        osc.set_texture(slot, w, h, fmt, data)
    for i, m in enumerate(host_scene.materials()):
        osc.set_material(i, to_oracle_material(m))
    for v, t, mat in host_scene.geoms():
        osc.add_geom(v, t, mat)
    for g in host_scene.groups():
        osc.add_group(g)
    for g, x in host_scene.instances():
        osc.add_instance(g, x)
    secs = osc.commit(brute_force=brute_force, config=config)
    osc.build_seconds = secs
    return osc


from gfxexp_amd import scenes as _scenes  # noqa: E402


def bunny_scene(with_light=True, with_ground=True):
    return _scenes.bunny_scene(os.path.join(ASSETS, "stanford_bunny_309_faces.obj"), with_light, with_ground)


def teapot_scene(emissive=False):
    """teapot.obj as one instance.  emissive=True is BASELINE configs[0] part (ii): teapot.mtl
    has Ke 0, so the "emitter set" is built here -- every one of the 15 704 triangles on a material with emittance
    RGB(1, 1, 1) (importance = luminance(1,1,1) * area = area, compute_light_probs.cu:33-43)."""
    src = api.HostScene()
    g = src.load_obj(os.path.join(ASSETS, "teapot.obj"))
    if not emissive:
        src.add_instance(g, api.make_transform())
        return src
    s = api.HostScene()
    mat = s.add_material_traditional((0.01, 0.01, 0.01), (0, 0, 0), 0.3, (1.0, 1.0, 1.0))
    geoms = [s.add_geom(v, t, mat) for v, t, _ in src.geoms()]
    s.add_instance(s.add_group(geoms), api.make_transform())
    return s


from gfxexp_amd.scenes import bench_street, small_street  # noqa: E402,F401  (scene definitions live with the product)


def pathological_light_scene(seed=11, instances=420):
    """Emitters whose importance spans ~12 decades, runs of non-emissive instances between them, and groups
    with the emissive geometry first / last / in the middle: the instance-level guide table gets crowded
    and empty cells, and the geometry-instance search sees zero-weight entries on either side."""
    rng = np.random.default_rng(seed)
    s = api.HostScene()

    def quad(mat, n=1):
        # n x n grid of two-triangle cells in the xz plane, 1 x 1 overall
        m = n + 1
        v = np.zeros(m * m, api.VERTEX_DTYPE)
        xs, zs = np.meshgrid(np.linspace(-0.5, 0.5, m), np.linspace(-0.5, 0.5, m))
        v["position"] = np.stack([xs.ravel(), np.zeros(m * m), zs.ravel()], 1)
        v["normal"] = (0, 1, 0)
        v["texCoord0Dir"] = (1, 0, 0)
        v["texCoord"] = np.stack([xs.ravel() + 0.5, zs.ravel() + 0.5], 1)
        t = []
        for j in range(n):
            for i in range(n):
                a = j * m + i
                t += [(a, a + m + 1, a + 1), (a, a + m, a + m + 1)]
        return s.add_geom(v, t, mat)

    dark = s.add_material_traditional((0.6, 0.6, 0.6), (0.04, 0.04, 0.04), 0.2)
    lit = [s.add_material_traditional((0.01, 0.01, 0.01), (0, 0, 0), 0.3, e)
           for e in ((30, 20, 10), (0.5, 2, 8), (1e-3, 1e-3, 1e-3), (400, 400, 380))]
    g_dark, g_dark5 = quad(dark), quad(dark, 5)
    g_lit = [quad(m, n) for m, n in zip(lit, (1, 3, 2, 7))]
    groups = [s.add_group([g_lit[0]]), s.add_group([g_dark]), s.add_group([g_dark5, g_lit[1]]),
              s.add_group([g_lit[2], g_dark]), s.add_group([g_dark, g_lit[3], g_dark5]),
              s.add_group([g_lit[1], g_lit[0], g_lit[3]]), s.add_group([g_dark5])]
    ground = s.add_group([quad(dark, 8)])
    s.add_instance(ground, api.make_transform(scale=60.0))
    i = 0
    while i < instances:
        run = int(rng.integers(1, 40))                     # runs of one kind: long stretches of zero weight
        kind = int(rng.integers(0, len(groups)))
        for _ in range(min(run, instances - i)):
            scale = float(10.0 ** rng.uniform(-2.5, 1.2))
            pos = (float(rng.uniform(-25, 25)), float(rng.uniform(0.5, 12)), float(rng.uniform(-25, 25)))
            s.add_instance(groups[kind], api.make_transform(scale=scale, roll=float(rng.uniform(0, 360)),
                                                            pitch=float(rng.uniform(0, 360)), yaw=float(rng.uniform(0, 360)), pos=pos))
            i += 1
    return s


def pinhole_rays(width, height, cam_pos, look_at, fov_y_deg=45.0, tmax=np.float32(3.0e38)):
    """Pinhole camera rays (the shape of the reference's testBvhBuilder harness, nrtdsm_sandbox.cpp:3376-3410)."""
    cam_pos = np.asarray(cam_pos, np.float64)
    fwd = np.asarray(look_at, np.float64) - cam_pos
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, (0, 1, 0))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    h = 2 * np.tan(np.radians(fov_y_deg) / 2)
    w = h * width / height
    xs = ((np.arange(width) + 0.5) / width - 0.5) * w
    ys = (0.5 - (np.arange(height) + 0.5) / height) * h
    X, Y = np.meshgrid(xs, ys)
    d = fwd[None, None, :] + X[..., None] * right + Y[..., None] * up
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    n = width * height
    org = np.zeros((n, 4), np.float32)
    org[:, :3] = cam_pos
    dirs = np.zeros((n, 4), np.float32)
    dirs[:, :3] = d.reshape(n, 3)
    dirs[:, 3] = tmax
    return org, dirs


PRESAMPLED_LIGHTS = 128 * 1024   # numLightSubsets * lightSubsetSize, restir_di_shared.h:8-9


class PixelBuffers:
    """Host-side (numpy) copies of every ReSTIR per-pixel buffer in the ABI layouts."""

    def __init__(self, width, height, seed=PIXEL_RNG_SEED):
        n = width * height
        self.w, self.h, self.n = width, height, n
        self.rng = O.seed_rngs(n, seed)
        self.gb0 = [np.zeros(n, api.GBUFFER0_DTYPE) for _ in range(2)]
        self.gb1 = [np.zeros((n, 2), np.float32) for _ in range(2)]
        self.gb2 = [np.zeros(n, api.GBUFFER2_DTYPE) for _ in range(2)]
        self.gb3 = [np.zeros(n, api.GBUFFER3_DTYPE) for _ in range(2)]
        self.res = [np.zeros((3, n, 4), np.float32) for _ in range(2)]
        self.info = [np.zeros((n, 2), np.float32) for _ in range(2)]
        self.vis = [np.zeros(n, np.uint32) for _ in range(2)]
        self.beauty = np.zeros((n, 4), np.float32)
        self.albedo = np.zeros((n, 4), np.float32)
        self.normal = np.zeros((n, 4), np.float32)
        self.deltas = O.spatial_neighbor_deltas()
        self.env = None
        # rearchitected ReSTIR: restir_di_main.cpp:1210-1222 (mt19937_64(894213312210))
        self.presample_rngs = O.seed_rngs(PRESAMPLED_LIGHTS, 894213312210)
        self.presampled = np.zeros((PRESAMPLED_LIGHTS, 12), np.float32)

    def set_env(self, texels, w, h, oracle_side=False):
        """Attach a lat-long environment map with its importance map.  The GPU side gets the PRODUCT's host builder
        (gfxh_env_build_importance + guide tables); oracle_side=True -- what every harness passes for the buffers it hands to
        the oracle -- builds it with the ORACLE's own restatement (orc_env_build), so a host-builder error shows up as a
        GPU-vs-oracle mismatch instead of cancelling out (the two builders are also compared directly in test_env_light.py)."""
        t = np.ascontiguousarray(texels, np.float32).copy()
        e = O.env_build(t, w, h) if oracle_side else api.env_build_importance(t, w, h)
        e.update(texels=t, w=w, h=h)
        self.env = e

    def _fill_env(self, s, ptr):
        if self.env is None:
            return
        e = self.env
        s.envLightTexture = ptr(e["texels"]); s.envWidth, s.envHeight = e["w"], e["h"]
        s.envRowPDF = ptr(e["rowPDF"]); s.envRowCDF = ptr(e["rowCDF"]); s.envRowIntegrals = ptr(e["rowIntegrals"])
        s.envTopPDF = ptr(e["topPDF"]); s.envTopCDF = ptr(e["topCDF"]); s.envTopIntegral = e["topIntegral"]

    def arrays(self):
        out = {"rng": self.rng, "beauty": self.beauty, "albedo": self.albedo, "normal": self.normal,
               "presample_rngs": self.presample_rngs, "presampled": self.presampled}
        for i in range(2):
            out[f"vis_{i}"] = self.vis[i]
            out.update({f"gb0_{i}": self.gb0[i], f"gb1_{i}": self.gb1[i], f"gb2_{i}": self.gb2[i], f"gb3_{i}": self.gb3[i],
                        f"res_{i}": self.res[i], f"info_{i}": self.info[i]})
        return out

    def static_params(self, cls, ptr):
        s = cls()
        s.imageSizeX, s.imageSizeY = self.w, self.h
        s.rngBuffer = ptr(self.rng)
        for i in range(2):
            s.gbuffer0[i] = ptr(self.gb0[i]); s.gbuffer1[i] = ptr(self.gb1[i])
            s.gbuffer2[i] = ptr(self.gb2[i]); s.gbuffer3[i] = ptr(self.gb3[i])
            s.reservoirBuffer[i] = ptr(self.res[i]); s.reservoirInfoBuffer[i] = ptr(self.info[i])
            s.sampleVisibilityBuffer[i] = ptr(self.vis[i])
        s.spatialNeighborDeltas = ptr(self.deltas)
        s.beautyAccumBuffer = ptr(self.beauty); s.albedoAccumBuffer = ptr(self.albedo); s.normalAccumBuffer = ptr(self.normal)
        s.numTilesX, s.numTilesY = (self.w + 7) // 8, (self.h + 7) // 8
        s.lightPreSamplingRngs = ptr(self.presample_rngs); s.preSampledLights = ptr(self.presampled)
        self._fill_env(s, ptr)
        return s

    def host_static_params(self):
        return self.static_params(O.GfxRestirStaticParams, lambda a: a.ctypes.data)


class DeviceBuffers:
    """torch-allocated device mirrors of a PixelBuffers (PyTorch = device memory plumbing only)."""

    def __init__(self, pb):
        import torch
        self.torch = torch
        self.pb = pb
        self.t = {}
        for k, a in pb.arrays().items():
            self.t[k] = torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda()
        self.t["deltas"] = torch.from_numpy(pb.deltas.view(np.uint8).reshape(-1).copy()).cuda()
        self.env_t = {}
        if pb.env is not None:
            for k in ("texels", "rowPDF", "rowCDF", "rowIntegrals", "topPDF", "topCDF"):
                self.env_t[k] = torch.from_numpy(pb.env[k].reshape(-1)).cuda()
            if pb.env.get("guidesUsable") and getattr(pb, "use_env_guides", True):
                for k in ("rowGuide", "topGuide"):      # uint16 tables (torch has no uint16: ship the bytes)
                    self.env_t[k] = torch.from_numpy(pb.env[k].view(np.uint8).reshape(-1).copy()).cuda()
                if "rowTable" in pb.env and getattr(pb, "use_env_row_table", True):      # the interleaved rows (gfx_restir_static_params::envRowTable)
                    self.env_t["rowTable"] = torch.from_numpy(pb.env["rowTable"].view(np.uint8).reshape(-1).copy()).cuda()
                    if "rowSketch" in pb.env and getattr(pb, "use_env_row_sketch", True):   # ... and their inverse-CDF sketches (envRowSketch)
                        self.env_t["rowSketch"] = torch.from_numpy(pb.env["rowSketch"].view(np.uint8).reshape(-1).copy()).cuda()

    def static_params(self):
        pb, t = self.pb, self.t
        s = api.GfxRestirStaticParams()
        s.imageSizeX, s.imageSizeY = pb.w, pb.h
        s.rngBuffer = t["rng"].data_ptr()
        for i in range(2):
            s.gbuffer0[i] = t[f"gb0_{i}"].data_ptr(); s.gbuffer1[i] = t[f"gb1_{i}"].data_ptr()
            s.gbuffer2[i] = t[f"gb2_{i}"].data_ptr(); s.gbuffer3[i] = t[f"gb3_{i}"].data_ptr()
            s.reservoirBuffer[i] = t[f"res_{i}"].data_ptr(); s.reservoirInfoBuffer[i] = t[f"info_{i}"].data_ptr()
            s.sampleVisibilityBuffer[i] = t[f"vis_{i}"].data_ptr()
        s.spatialNeighborDeltas = t["deltas"].data_ptr()
        s.beautyAccumBuffer = t["beauty"].data_ptr(); s.albedoAccumBuffer = t["albedo"].data_ptr()
        s.normalAccumBuffer = t["normal"].data_ptr()
        s.numTilesX, s.numTilesY = (pb.w + 7) // 8, (pb.h + 7) // 8
        s.lightPreSamplingRngs = t["presample_rngs"].data_ptr(); s.preSampledLights = t["presampled"].data_ptr()
        if pb.env is not None:
            e, et = pb.env, self.env_t
            s.envLightTexture = et["texels"].data_ptr(); s.envWidth, s.envHeight = e["w"], e["h"]
            s.envRowPDF = et["rowPDF"].data_ptr(); s.envRowCDF = et["rowCDF"].data_ptr()
            s.envRowIntegrals = et["rowIntegrals"].data_ptr()
            s.envTopPDF = et["topPDF"].data_ptr(); s.envTopCDF = et["topCDF"].data_ptr(); s.envTopIntegral = e["topIntegral"]
            if "rowGuide" in et:
                s.envRowGuide = et["rowGuide"].data_ptr(); s.envTopGuide = et["topGuide"].data_ptr()
            if "rowTable" in et:
                s.envRowTable = et["rowTable"].data_ptr()
            if "rowSketch" in et:
                s.envRowSketch = et["rowSketch"].data_ptr()
        return s

    def download(self):
        """Return {name: numpy array} with the dtypes/shapes of PixelBuffers.arrays()."""
        self.torch.cuda.synchronize()
        out = {}
        for k, a in self.pb.arrays().items():
            raw = self.t[k].cpu().numpy()
            out[k] = raw.view(a.dtype).reshape(a.shape)
        return out


# Frame-parameter overrides applied to BOTH sides by every harness (e.g. {"enableBumpMapping": 1}); use frame_overrides().
FRAME_OVERRIDES = {}


class frame_overrides:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.saved = dict(FRAME_OVERRIDES)
        FRAME_OVERRIDES.update(self.kw)

    def __exit__(self, *exc):
        FRAME_OVERRIDES.clear()
        FRAME_OVERRIDES.update(self.saved)


def frame_params(cls_frame, cls_cam, width, height, cam, prev_cam=None, **kw):
    f = cls_frame()
    C.memmove(C.byref(f.camera), C.byref(cam), C.sizeof(cam))
    pc = prev_cam if prev_cam is not None else cam
    C.memmove(C.byref(f.prevCamera), C.byref(pc), C.sizeof(pc))
    f.envLightPowerCoeff = 1.0
    f.spatialNeighborRadius = 20.0
    f.radiusThresholdForSpatialVisReuse = 10.0
    f.log2NumCandidateSamples = 5
    f.numSpatialNeighbors = 5
    f.useLowDiscrepancyNeighbors = 1
    f.reuseVisibility = 1
    f.reuseVisibilityForTemporal = 1
    f.enableTemporalReuse = 1
    f.enableSpatialReuse = 1
    for k, v in kw.items():
        setattr(f, k, v)
    for k, v in FRAME_OVERRIDES.items():
        setattr(f, k, v)
    return f


def copy_struct(dst_cls, src):
    d = dst_cls()
    assert C.sizeof(d) == C.sizeof(src)
    C.memmove(C.byref(d), C.byref(src), C.sizeof(src))
    return d


def assert_same_bits(name, a, b):
    """Bit-exact comparison with a helpful message (NaNs with equal payload compare equal)."""
    a8 = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    b8 = np.ascontiguousarray(b).view(np.uint8).reshape(-1)
    assert a8.shape == b8.shape, f"{name}: shape {a8.shape} vs {b8.shape}"
    if not np.array_equal(a8, b8):
        item = np.ascontiguousarray(a).dtype.itemsize if np.ascontiguousarray(a).dtype.fields is None else 1
        bad = np.nonzero(a8 != b8)[0]
        first = bad[0] // max(item, 1)
        raise AssertionError(f"{name}: {len(bad)} differing bytes, first at element {first}: "
                             f"{np.ascontiguousarray(a).reshape(-1)[first] if item > 1 else a8[bad[0]]} vs "
                             f"{np.ascontiguousarray(b).reshape(-1)[first] if item > 1 else b8[bad[0]]}")


class RegirBuffers:
    """Host (numpy) ReGIR grid state in the ABI layout; device mirrors via to_device()."""

    def __init__(self, bounds, dims=(8, 4, 8), log2_slot=3, log2_cell=2, randomize=1):
        self.dims = tuple(int(d) for d in dims)
        self.cells = self.dims[0] * self.dims[1] * self.dims[2]
        self.slots = self.cells * 512
        lo, hi = np.asarray(bounds[:3], np.float32), np.asarray(bounds[3:], np.float32)
        self.origin = lo
        self.cell_size = ((hi - lo) / np.asarray(self.dims, np.float32)).astype(np.float32)   # regir_main.cpp:1079
        self.res = [np.zeros((3, self.slots, 4), np.float32) for _ in range(2)]
        self.info = [np.zeros((self.slots, 2), np.float32) for _ in range(2)]
        self.rngs = O.seed_rngs(self.slots, PIXEL_RNG_SEED)                                 # regir_main.cpp:1086-1092
        self.accesses = np.zeros(self.cells, np.uint32)
        self.last_access = np.full(self.cells, 0xFFFFFFFF, np.uint32)                       # fill(-1), :1096, :1112
        self.active = [np.zeros(1, np.uint32) for _ in range(2)]
        self.cfg = (log2_slot, log2_cell, randomize)
        self.t = None

    def arrays(self):
        out = {"regir_rngs": self.rngs, "regir_accesses": self.accesses, "regir_last_access": self.last_access}
        for i in range(2):
            out.update({f"regir_res_{i}": self.res[i], f"regir_info_{i}": self.info[i], f"regir_active_{i}": self.active[i]})
        return out

    def _params(self, cls, ptr):
        g = cls()
        for i in range(2):
            g.reservoirs[i] = ptr(f"regir_res_{i}"); g.reservoirInfos[i] = ptr(f"regir_info_{i}")
            g.numActiveCells[i] = ptr(f"regir_active_{i}")
        g.lightSlotRngs = ptr("regir_rngs"); g.perCellNumAccesses = ptr("regir_accesses")
        g.lastAccessFrameIndices = ptr("regir_last_access")
        for k in range(3):
            g.gridOrigin[k] = float(self.origin[k]); g.gridCellSize[k] = float(self.cell_size[k]); g.gridDimension[k] = self.dims[k]
        g.log2NumCandidatesPerLightSlot, g.log2NumCandidatesPerCell, g.enableCellRandomization = self.cfg
        return g

    def host_params(self):
        arrs = self.arrays()
        return self._params(O.GfxRegirParams, lambda k: arrs[k].ctypes.data)

    def device_params(self):
        import torch
        self.t = {k: torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda() for k, a in self.arrays().items()}
        return self._params(api.GfxRegirParams, lambda k: self.t[k].data_ptr())

    def download(self):
        import torch
        torch.cuda.synchronize()
        return {k: self.t[k].cpu().numpy().view(a.dtype).reshape(a.shape) for k, a in self.arrays().items()}


def lcg_shufflers():
    """dataShufflerBuffer initialisation, neural_radiance_caching_main.cpp:1186-1194."""
    out = np.zeros(1 << 16, np.uint32)
    state = 471313181
    for i in range(1 << 16):
        state = (state * 1103515245 + 12345) % (1 << 31)
        out[i] = state
    return out


class NrcBuffers:
    """Host (numpy) NRC render-side state in the ABI layout (gfx_nrc_params); device mirrors on demand."""
    TRAIN = 1 << 17

    def __init__(self, width, height, bounds, radiance_scale=1.0):
        n = width * height
        self.n, self.suffixes = n, n // 16
        cap = ((n + self.suffixes + 255) // 256) * 256
        self.bounds = np.asarray(bounds, np.float32)
        self.radiance_scale = radiance_scale
        a = {}
        for i in range(2):
            a[f"nrc_num_{i}"] = np.zeros(1, np.uint32)
            a[f"nrc_tile_{i}"] = np.full(2, 8, np.uint32)
            a[f"nrc_minmax_{i}"] = np.zeros(6, np.int32)
            a[f"nrc_avg_{i}"] = np.zeros(3, np.float32)
            a[f"nrc_trainq_{i}"] = np.zeros((self.TRAIN, 14), np.float32)
            a[f"nrc_traint_{i}"] = np.zeros((self.TRAIN, 3), np.float32)
        a["nrc_off_unbiased"] = np.zeros(1, np.uint32)
        a["nrc_off_training"] = np.zeros(1, np.uint32)
        a["nrc_queries"] = np.zeros((cap, 14), np.float32)
        a["nrc_terminal"] = np.zeros((n, 4), np.float32)
        a["nrc_inferred"] = np.zeros((cap, 3), np.float32)
        a["nrc_contribution"] = np.zeros((n, 3), np.float32)
        a["nrc_vertex"] = np.zeros((self.TRAIN, 4), np.float32)
        a["nrc_suffix"] = np.zeros(self.suffixes, np.uint32)
        a["nrc_shuffler"] = lcg_shufflers()
        self.a = a
        self.t = None

    def arrays(self):
        return self.a

    def _params(self, cls, ptr, off_unbiased, off_training, new_sequence):
        g = cls()
        for k in range(3):
            g.sceneAabbMin[k] = float(self.bounds[k]); g.sceneAabbMax[k] = float(self.bounds[3 + k])
        g.maxNumTrainingSuffixes = self.suffixes
        for i in range(2):
            g.numTrainingData[i] = ptr(f"nrc_num_{i}"); g.tileSize[i] = ptr(f"nrc_tile_{i}")
            g.targetMinMax[i] = ptr(f"nrc_minmax_{i}"); g.targetAvg[i] = ptr(f"nrc_avg_{i}")
            g.trainRadianceQueryBuffer[i] = ptr(f"nrc_trainq_{i}"); g.trainTargetBuffer[i] = ptr(f"nrc_traint_{i}")
        g.offsetToSelectUnbiasedTile = ptr("nrc_off_unbiased"); g.offsetToSelectTrainingPath = ptr("nrc_off_training")
        g.inferenceRadianceQueryBuffer = ptr("nrc_queries"); g.inferenceTerminalInfoBuffer = ptr("nrc_terminal")
        g.inferredRadianceBuffer = ptr("nrc_inferred"); g.perFrameContributionBuffer = ptr("nrc_contribution")
        g.trainVertexInfoBuffer = ptr("nrc_vertex"); g.trainSuffixTerminalInfoBuffer = ptr("nrc_suffix")
        g.dataShufflerBuffer = ptr("nrc_shuffler")
        g.radianceScale = self.radiance_scale
        g.preprocessOffsetToSelectUnbiasedTile, g.preprocessOffsetToSelectTrainingPath = off_unbiased, off_training
        g.isNewSequence = int(new_sequence)
        return g

    def host_params(self, off_unbiased, off_training, new_sequence):
        return self._params(O.GfxNrcParams, lambda k: self.a[k].ctypes.data, off_unbiased, off_training, new_sequence)

    def to_device(self):
        import torch
        self.t = {k: torch.from_numpy(v.view(np.uint8).reshape(-1).copy()).cuda() for k, v in self.a.items()}

    def device_params(self, off_unbiased, off_training, new_sequence):
        if self.t is None:
            self.to_device()
        return self._params(api.GfxNrcParams, lambda k: self.t[k].data_ptr(), off_unbiased, off_training, new_sequence)

    def download(self):
        import torch
        torch.cuda.synchronize()
        return {k: self.t[k].cpu().numpy().view(v.dtype).reshape(v.shape) for k, v in self.a.items()}

    def upload(self, name, array):
        import torch
        self.t[name].copy_(torch.from_numpy(np.ascontiguousarray(array).view(np.uint8).reshape(-1)))


def nrc_chains(arrays, buf_idx=0):
    """Canonical form of the training records: {tile: (suffix bits without the index, [(query, target,
    localThroughput, pathLength), ...] from the suffix end to the first vertex)} -- independent of the
    order in which records were allocated."""
    out = {}
    suffix = arrays["nrc_suffix"]
    vinfo = arrays["nrc_vertex"]
    q, t = arrays["nrc_trainq_0"], arrays["nrc_traint_0"]
    for tile in np.nonzero((suffix & 0x7FFFFF) != 0x7FFFFF)[0]:
        bits = int(suffix[tile])
        last = bits & 0x7FFFFF
        chain = []
        while last != 0x7FFFFF and len(chain) < 256:
            vb = int(vinfo[last, 3].view(np.uint32))
            chain.append((q[last].tobytes(), t[last].tobytes(), vinfo[last, :3].tobytes(), vb >> 23))
            last = vb & 0x7FFFFF
        out[int(tile)] = (bits >> 23, chain)
    return out
