"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle.so, the CPU restatement of the GfxExp ReSTIR-DI / BVH hot path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (gfxexp_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc only, no GPU needed)."""
    # make decides (the Makefile lists every header): a library older than a source is rebuilt, an up-to-date one costs nothing.
    # Where there is no compiler (a GPU box that received the prebuilt library) the library is used as it is.
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    except (OSError, subprocess.CalledProcessError):
        if not os.path.exists(_LIB_PATH):
            raise
    return _LIB_PATH


class GfxMaterial(C.Structure):
    _fields_ = [("bsdfType", C.c_uint32), ("a", C.c_float * 3), ("b", C.c_float * 3),
                ("smoothness", C.c_float), ("emittance", C.c_float * 3), ("hasEmittance", C.c_uint32),
                ("texA", C.c_uint32), ("texB", C.c_uint32), ("texSmoothness", C.c_uint32), ("texNormal", C.c_uint32),
                ("texEmittance", C.c_uint32), ("bumpMapType", C.c_uint32), ("pad", C.c_uint32 * 2)]


class GfxCamera(C.Structure):
    _fields_ = [("aspect", C.c_float), ("fovY", C.c_float), ("position", C.c_float * 3),
                ("orientation", C.c_float * 9)]


class GfxRestirStaticParams(C.Structure):
    _fields_ = [
        ("imageSizeX", C.c_int32), ("imageSizeY", C.c_int32),
        ("rngBuffer", C.c_void_p),
        ("gbuffer0", C.c_void_p * 2), ("gbuffer1", C.c_void_p * 2),
        ("gbuffer2", C.c_void_p * 2), ("gbuffer3", C.c_void_p * 2),
        ("reservoirBuffer", C.c_void_p * 2), ("reservoirInfoBuffer", C.c_void_p * 2),
        ("sampleVisibilityBuffer", C.c_void_p * 2),
        ("spatialNeighborDeltas", C.c_void_p),
        ("beautyAccumBuffer", C.c_void_p), ("albedoAccumBuffer", C.c_void_p), ("normalAccumBuffer", C.c_void_p),
        ("numTilesX", C.c_int32), ("numTilesY", C.c_int32),
        ("lightPreSamplingRngs", C.c_void_p), ("preSampledLights", C.c_void_p),
        ("envLightTexture", C.c_void_p), ("envWidth", C.c_int32), ("envHeight", C.c_int32),
        ("envRowPDF", C.c_void_p), ("envRowCDF", C.c_void_p), ("envRowIntegrals", C.c_void_p),
        ("envTopPDF", C.c_void_p), ("envTopCDF", C.c_void_p), ("envTopIntegral", C.c_float),
        ("envRowGuide", C.c_void_p), ("envTopGuide", C.c_void_p), ("envRowTable", C.c_void_p), ("envRowSketch", C.c_void_p),   # accepted and ignored: the oracle always searches the plain arrays
    ]


class GfxRestirFrameParams(C.Structure):
    _fields_ = [
        ("travHandle", C.c_uint64), ("numAccumFrames", C.c_uint32), ("frameIndex", C.c_uint32),
        ("camera", GfxCamera), ("prevCamera", GfxCamera),
        ("envLightPowerCoeff", C.c_float), ("envLightRotation", C.c_float),
        ("spatialNeighborRadius", C.c_float), ("radiusThresholdForSpatialVisReuse", C.c_float),
        ("log2NumCandidateSamples", C.c_uint32), ("numSpatialNeighbors", C.c_uint32),
        ("useLowDiscrepancyNeighbors", C.c_uint32), ("reuseVisibility", C.c_uint32),
        ("reuseVisibilityForTemporal", C.c_uint32), ("reuseVisibilityForSpatiotemporal", C.c_uint32),
        ("enableTemporalReuse", C.c_uint32), ("enableSpatialReuse", C.c_uint32),
        ("useUnbiasedEstimator", C.c_uint32), ("bufferIndex", C.c_uint32),
        ("resetFlowBuffer", C.c_uint32), ("enableJittering", C.c_uint32),
        ("enableEnvLight", C.c_uint32), ("enableBumpMapping", C.c_uint32), ("useSolidAngleSampling", C.c_uint32),
    ]


class GfxRegirParams(C.Structure):
    _fields_ = [("reservoirs", C.c_void_p * 2), ("reservoirInfos", C.c_void_p * 2), ("lightSlotRngs", C.c_void_p),
                ("perCellNumAccesses", C.c_void_p), ("lastAccessFrameIndices", C.c_void_p),
                ("numActiveCells", C.c_void_p * 2), ("gridOrigin", C.c_float * 3), ("gridCellSize", C.c_float * 3),
                ("gridDimension", C.c_uint32 * 3), ("log2NumCandidatesPerLightSlot", C.c_uint32),
                ("log2NumCandidatesPerCell", C.c_uint32), ("enableCellRandomization", C.c_uint32)]


class GfxNrcParams(C.Structure):
    _fields_ = [("sceneAabbMin", C.c_float * 3), ("sceneAabbMax", C.c_float * 3), ("maxNumTrainingSuffixes", C.c_uint32),
                ("numTrainingData", C.c_void_p * 2), ("tileSize", C.c_void_p * 2), ("targetMinMax", C.c_void_p * 2),
                ("targetAvg", C.c_void_p * 2), ("offsetToSelectUnbiasedTile", C.c_void_p),
                ("offsetToSelectTrainingPath", C.c_void_p), ("inferenceRadianceQueryBuffer", C.c_void_p),
                ("inferenceTerminalInfoBuffer", C.c_void_p), ("inferredRadianceBuffer", C.c_void_p),
                ("perFrameContributionBuffer", C.c_void_p), ("trainRadianceQueryBuffer", C.c_void_p * 2),
                ("trainTargetBuffer", C.c_void_p * 2), ("trainVertexInfoBuffer", C.c_void_p),
                ("trainSuffixTerminalInfoBuffer", C.c_void_p), ("dataShufflerBuffer", C.c_void_p),
                ("radianceScale", C.c_float), ("preprocessOffsetToSelectUnbiasedTile", C.c_uint32),
                ("preprocessOffsetToSelectTrainingPath", C.c_uint32), ("isNewSequence", C.c_uint32)]


GFX_HIT_DTYPE = np.dtype([("dist", "<f4"), ("bcB", "<f4"), ("bcC", "<f4"), ("triIndex", "<u4")])
GFX_TRI_IDS_DTYPE = np.dtype([("instSlot", "<u4"), ("geomInstSlot", "<u4"), ("primIndex", "<u4")])
GFX_VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("texCoord0Dir", "<f4", 3),
                             ("texCoord", "<f4", 2)])
assert GFX_VERTEX_DTYPE.itemsize == 44

_lib = None
_fast_lib = None
PARITY_FLAGS = "-std=c++17 -O2 -march=x86-64-v3 -ffp-contract=off -fno-fast-math -fopenmp"      # oracle/Makefile
FAST_FLAGS = "-std=c++17 -O3 -march=native -fopenmp"      # BASELINE.md section 2 "speed mode": contraction allowed (GNU default)


def _declare(L):
    L.orc_scene_create.restype = C.c_void_p
    L.orc_version.restype = C.c_char_p
    L.orc_max_threads.restype = C.c_int
    return L


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _declare(C.CDLL(_LIB_PATH))
    return _lib


def _host_id():
    try:
        with open("/proc/cpuinfo") as f:
            lines = [ln for ln in f if ln.startswith(("model name", "flags"))][:2]
        return "".join(lines)
    except OSError:
        return "unknown"


def build_fast(force=False):
    """The SPEED build of the same sources for bench.py's cpu_baseline: g++ -O3 -march=native, contraction allowed.  Its
    results are NOT the parity contract (fused multiply-adds round differently) -- nothing compares against it.
    -march=native binds the binary to the machine that compiled it, so it is built where it runs (5 s) and rebuilt when
    the host CPU differs from the one in its stamp."""
    path = os.path.join(_HERE, "liboracle_fast.so")
    stamp = path + ".host"
    want = FAST_FLAGS + "\n" + _host_id()
    if not force and os.path.exists(path) and os.path.exists(stamp) and open(stamp).read() == want \
            and os.path.getmtime(path) >= max(os.path.getmtime(os.path.join(_HERE, f)) for f in os.listdir(_HERE) if f.endswith((".h", ".cpp"))):
        return path
    subprocess.check_call(["g++"] + FAST_FLAGS.split() + ["-fPIC", "-Wall", "-Wno-unused-function", "-shared", "-o", path,
                                                         os.path.join(_HERE, "orc_capi.cpp")])
    with open(stamp, "w") as f:
        f.write(want)
    return path


def lib_fast():
    global _fast_lib
    if _fast_lib is None:
        _fast_lib = _declare(C.CDLL(build_fast()))
    return _fast_lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleScene:
    """Scene + world BVH + passes of the CPU restatement."""

    def __init__(self, threads=None, library=None):
        self.L = library if library is not None else lib()
        self.h = C.c_void_p(self.L.orc_scene_create())
        self.keep = []
        self.set_threads(threads if threads else min(8, self.L.orc_max_threads()))

    def close(self):
        if self.h:
            self.L.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_texture(self, slot, width, height, fmt, texel_bytes):
        t = np.ascontiguousarray(texel_bytes)
        self.L.orc_texture_set(self.h, C.c_uint32(slot), C.c_uint32(width), C.c_uint32(height), C.c_uint32(fmt), _p(t))

    def texture_sample(self, slot, uv, gather=False):
        """tex2DLod (or tex2Dgather of component 0) of texture `slot` at uv (n, 2) -> (n, 4) float32."""
        c = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        out = np.zeros((len(c), 4), np.float32)
        self.L.orc_texture_sample(self.h, C.c_uint32(slot), _p(c), C.c_uint32(len(c)), _p(out), C.c_int(1 if gather else 0))
        return out

    def set_threads(self, n):
        self.threads = n
        self.L.orc_set_num_threads(self.h, C.c_int(n))

    def set_material(self, slot, mat):
        self.L.orc_material_set(self.h, C.c_uint32(slot), C.byref(mat))

    def add_geom(self, vertices, triangles, mat_slot):
        v = np.ascontiguousarray(vertices)
        assert v.dtype == GFX_VERTEX_DTYPE
        t = np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 3)
        out = C.c_uint32()
        self.L.orc_geom_create(self.h, _p(v), C.c_uint32(44), C.c_uint32(len(v)), _p(t), C.c_uint32(len(t)),
                               C.c_uint32(mat_slot), C.byref(out))
        return out.value

    def add_group(self, geom_slots):
        g = np.ascontiguousarray(geom_slots, dtype=np.uint32)
        out = C.c_uint32()
        self.L.orc_group_create(self.h, _p(g), C.c_uint32(len(g)), C.byref(out))
        return out.value

    def add_instance(self, group, xfm12):
        x = np.ascontiguousarray(xfm12, dtype=np.float32).reshape(12)
        out = C.c_uint32()
        self.L.orc_instance_create(self.h, C.c_uint32(group), _p(x), C.byref(out))
        return out.value

    def set_instance_transform(self, inst_slot, xfm12, normal_matrix9=None):
        """InstanceController::update for one instance; call commit() again before tracing / rendering."""
        x = np.ascontiguousarray(xfm12, dtype=np.float32).reshape(12)
        nm = None if normal_matrix9 is None else _p(np.ascontiguousarray(normal_matrix9, dtype=np.float32).reshape(9))
        if self.L.orc_instance_set_transform(self.h, C.c_uint32(inst_slot), _p(x), nm):
            raise RuntimeError("orc_instance_set_transform failed")

    def commit(self, brute_force=False, config=None):
        secs = C.c_double()
        cfg = None
        if config is not None:
            cfg = np.ascontiguousarray(config, dtype=np.float32)
        self.L.orc_scene_commit(self.h, C.c_int(1 if brute_force else 0), _p(cfg), C.byref(secs))
        return secs.value

    def accel_stats(self):
        s = (C.c_uint32 * 4)()
        self.L.orc_accel_stats(self.h, s)
        return list(s)

    def accel_validate(self):
        return self.L.orc_accel_validate(self.h)

    def lights_read(self, level, index=0):
        n = C.c_uint32()
        integ = C.c_float()
        self.L.orc_lights_read(self.h, C.c_uint32(level), C.c_uint32(index), None, None, C.c_uint32(0),
                               C.byref(n), C.byref(integ))
        w = np.zeros(n.value, np.float32)
        c = np.zeros(n.value, np.float32)
        self.L.orc_lights_read(self.h, C.c_uint32(level), C.c_uint32(index), _p(w), _p(c), C.c_uint32(n.value),
                               C.byref(n), C.byref(integ))
        return w, c, integ.value

    def trace(self, mode, ray_org_tmin, ray_dir_tmax, want_stats=False):
        o = np.ascontiguousarray(ray_org_tmin, dtype=np.float32).reshape(-1, 4)
        d = np.ascontiguousarray(ray_dir_tmax, dtype=np.float32).reshape(-1, 4)
        n = len(o)
        out = np.zeros(n, np.uint32) if mode == 1 else np.zeros(n, GFX_HIT_DTYPE)
        stats = np.zeros(4, np.uint64)
        self.L.orc_trace(self.h, C.c_int(mode), _p(o), _p(d), C.c_uint32(n), _p(out), _p(stats))
        return (out, stats) if want_stats else out

    def tri_ids(self):
        n = C.c_uint32()
        self.L.orc_tri_ids(self.h, None, C.c_uint32(0), C.byref(n))
        ids = np.zeros(n.value, GFX_TRI_IDS_DTYPE)
        self.L.orc_tri_ids(self.h, _p(ids), C.c_uint32(n.value), C.byref(n))
        return ids

    def world_triangles(self):
        n = C.c_uint32()
        self.L.orc_world_triangles(self.h, None, C.c_uint32(0), C.byref(n))
        t = np.zeros((n.value, 3, 3), np.float32)
        self.L.orc_world_triangles(self.h, _p(t), C.c_uint32(n.value), C.byref(n))
        return t

    def restir_launch(self, static_params, frame_params, cur_res_index, base_index, pass_id, rect=None):
        x0, y0, x1, y1 = rect if rect else (0, 0, 0, 0)
        self.L.orc_restir_launch(self.h, C.byref(static_params), C.byref(frame_params),
                                 C.c_uint32(cur_res_index), C.c_uint32(base_index), C.c_int(pass_id),
                                 C.c_int(x0), C.c_int(y0), C.c_int(x1), C.c_int(y1))

    def nrc_set_render_params(self, params):
        self.L.orc_nrc_set_render_params(self.h, C.byref(params))

    def regir_set_params(self, params):
        self.L.orc_regir_set_params(self.h, C.byref(params))

    def pt_set_reservoir_index(self, index):
        """The reservoir buffer the ReSTIR passes of this frame finished in: read by GFX_PT_PATH_TRACE_NRC_RESTIR (pass 14)."""
        self.L.orc_pt_set_reservoir_index(self.h, C.c_uint32(index))

    def pt_launch(self, static_params, frame_params, pass_id, max_path_length, rect=None):
        x0, y0, x1, y1 = rect if rect else (0, 0, 0, 0)
        self.L.orc_pt_launch(self.h, C.byref(static_params), C.byref(frame_params), C.c_int(pass_id),
                             C.c_uint32(max_path_length), C.c_int(x0), C.c_int(y0), C.c_int(x1), C.c_int(y1))

    def sample_light(self, shading_point, u3):
        u = np.ascontiguousarray(u3, dtype=np.float32).reshape(-1, 3)
        sp = np.ascontiguousarray(shading_point, dtype=np.float32)
        ls = np.zeros((len(u), 10), np.float32)
        pd = np.zeros(len(u), np.float32)
        self.L.orc_sample_light(self.h, _p(sp), _p(u), C.c_uint32(len(u)), _p(ls), _p(pd))
        return ls, pd


def seed_rngs(count, seed):
    out = np.zeros(count, np.uint64)
    lib().orc_seed_rngs(_p(out), C.c_uint64(count), C.c_uint64(seed))
    return out


def spatial_neighbor_deltas():
    out = np.zeros((1024, 2), np.float32)
    lib().orc_spatial_neighbor_deltas(_p(out))
    return out


def pcg32_floats(state, n):
    st = C.c_uint64(state)
    out = np.zeros(n, np.float32)
    lib().orc_pcg32_floats(C.byref(st), _p(out), C.c_uint32(n))
    return out, st.value


def pcg32_uints(state, n):
    st = C.c_uint64(state)
    out = np.zeros(n, np.uint32)
    lib().orc_pcg32_uints(C.byref(st), _p(out), C.c_uint32(n))
    return out, st.value


def math_sincos(x):
    x = np.ascontiguousarray(x, np.float32)
    s = np.zeros_like(x)
    c = np.zeros_like(x)
    lib().orc_math_sincos(_p(x), _p(s), _p(c), C.c_uint32(x.size))
    return s, c


def math_acos(x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros_like(x)
    lib().orc_math_acos(_p(x), _p(y), C.c_uint32(x.size))
    return y


def math_atan2(y, x):
    y = np.ascontiguousarray(y, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    r = np.zeros_like(x)
    lib().orc_math_atan2(_p(y), _p(x), _p(r), C.c_uint32(x.size))
    return r


def encode_normal(v):
    v = np.ascontiguousarray(v, np.float32).reshape(-1, 3)
    q = np.zeros(len(v), np.uint32)
    lib().orc_encode_normal(_p(v), _p(q), C.c_uint32(len(v)))
    return q


def decode_normal(q):
    q = np.ascontiguousarray(q, np.uint32)
    v = np.zeros((len(q), 3), np.float32)
    lib().orc_decode_normal(_p(q), _p(v), C.c_uint32(len(q)))
    return v


def offset_ray_origin(p, n):
    p = np.ascontiguousarray(p, np.float32).reshape(-1, 3)
    n = np.ascontiguousarray(n, np.float32).reshape(-1, 3)
    out = np.zeros_like(p)
    lib().orc_offset_ray_origin(_p(p), _p(n), _p(out), C.c_uint32(len(p)))
    return out


def discrete_sample(weights, us):
    w = np.ascontiguousarray(weights, np.float32)
    u = np.ascontiguousarray(us, np.float32)
    idx = np.zeros(len(u), np.uint32)
    prob = np.zeros(len(u), np.float32)
    rem = np.zeros(len(u), np.float32)
    integ = C.c_float()
    lib().orc_discrete_sample(_p(w), C.c_uint32(len(w)), _p(u), C.c_uint32(len(u)), _p(idx), _p(prob), _p(rem),
                              C.byref(integ))
    return idx, prob, rem, integ.value


def bsdf_eval(mat, mode, v_given, v_sampled):
    vg = np.ascontiguousarray(v_given, np.float32).reshape(-1, 3)
    vs = np.ascontiguousarray(v_sampled, np.float32).reshape(-1, 3)
    out = np.zeros((len(vg), 7), np.float32)
    lib().orc_bsdf_eval(C.byref(mat), C.c_int(mode), _p(vg), _p(vs), _p(out), C.c_uint32(len(vg)))
    return out


def reservoir_stream(weights, us):
    w = np.ascontiguousarray(weights, np.float32)
    u = np.ascontiguousarray(us, np.float32)
    M, K = w.shape
    sel = np.zeros(K, np.int32)
    sw = np.zeros(K, np.float32)
    sl = np.zeros(K, np.uint32)
    lib().orc_reservoir_stream(_p(w), _p(u), C.c_uint32(M), C.c_uint32(K), _p(sel), _p(sw), _p(sl))
    return sel, sw, sl


def env_build(texels, w, h):
    """-> dict(rowPDF,rowCDF,rowIntegrals,topPDF,topCDF,topIntegral); texels (h*w,4) float32 clamped in place."""
    t = texels
    assert t.dtype == np.float32 and t.flags.c_contiguous
    out = dict(rowPDF=np.zeros(h * w, np.float32), rowCDF=np.zeros(h * (w + 1), np.float32), rowIntegrals=np.zeros(h, np.float32),
               topPDF=np.zeros(h, np.float32), topCDF=np.zeros(h + 1, np.float32))
    integ = C.c_float()
    lib().orc_env_build(_p(t), C.c_uint32(w), C.c_uint32(h), _p(out["rowPDF"]), _p(out["rowCDF"]), _p(out["rowIntegrals"]),
                        _p(out["topPDF"]), _p(out["topCDF"]), C.byref(integ))
    out["topIntegral"] = integ.value
    return out


def env_sample(env, w, h, u2):
    u = np.ascontiguousarray(u2, np.float32).reshape(-1, 2)
    out = np.zeros((len(u), 3), np.float32)
    lib().orc_env_sample(_p(env["rowPDF"]), _p(env["rowCDF"]), _p(env["rowIntegrals"]), _p(env["topPDF"]), _p(env["topCDF"]),
                         C.c_float(env["topIntegral"]), C.c_uint32(w), C.c_uint32(h), _p(u), C.c_uint32(len(u)), _p(out))
    return out
