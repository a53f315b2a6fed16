"""restir_di_headless (gfxexp_amd/csrc/host/restir_di_headless.cpp): the reference's command line
(restir_di/restir_di_main.cpp:1-26, parseCommandline :593-878) driving the host API without a window.

CPU: the option state machine (-name / -emittance / -rectangle / -obj / -begin-* / -end-* / -inst), the camera and
instance transforms, error behaviour (-dry-run prints the scene it built and stops before touching a GPU).
GPU: the frames it renders are bit-identical to the same scene driven through the Python bindings."""
import json
import os
import subprocess

import numpy as np
import pytest

from gfxexp_amd import api, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUNNY = os.path.join(ROOT, "tests", "golden", "assets", "stanford_bunny_309_faces.obj")


def _run(args, check=True):
    r = subprocess.run([build.CLI] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    if check:
        assert r.returncode == 0, r.stderr
        return json.loads(r.stdout)
    return r


def _scene_args():
    return ["-cam-pos", 1.5, 5.0, 14.0, "-cam-yaw", 180,
            "-name", "a_bunny", "-obj", BUNNY, 0.1, "trad",
            "-name", "c_small", "-emittance", 50, 50, 50, "-rectangle", 1.0, 1.0,
            "-name", "b_wide", "-emittance", 10, 20, 40, "-rectangle", 2.0, 1.0,
            "-inst", "a_bunny",
            "-begin-pos", 0.0, 12.0, 2.0, "-inst", "c_small",
            "-begin-pos", -6.0, 6.0, 6.0, "-end-pos", -4.0, 6.0, 6.0, "-freq", 2, "-inst", "b_wide"]


def test_dry_run_builds_the_scene_the_options_describe(built_lib):
    d = _run(_scene_args() + ["-size", 160, 96, "-frames", 2, "-dry-run"])
    assert (d["materials"], d["geometries"], d["groups"], d["instances"], d["textures"]) == (3, 3, 3, 3, 0)
    assert d["triangles"] == 309 + 2 + 2 and d["animated_instances"] == 1 and d["size"] == [160, 96] and d["frames"] == 2
    # meshes are created in NAME order (std::map, restir_di_main.cpp:1125), instances in -inst order
    groups = [t[0] for t in d["instance_transforms"]]
    assert groups == [0, 2, 1]
    bunny, small, wide = (np.array(t[1:], np.float32).reshape(3, 4) for t in d["instance_transforms"])
    assert np.array_equal(bunny, np.array(api.make_transform(scale=0.1)).reshape(3, 4))      # the OBJ pre-scale
    assert np.array_equal(small[:, 3], [0, 12, 2]) and np.array_equal(small[:, :3], np.eye(3, dtype=np.float32))
    assert np.array_equal(wide[:, 3], [-6, 6, 6])
    cam = api.make_camera(160, 96, (1.5, 5.0, 14.0), yaw=180.0)
    assert np.allclose(d["camera_orientation"], list(cam.orientation), atol=1e-7)


def test_rotation_options_compose_like_the_reference(built_lib):
    """-roll / -pitch / -yaw pre-multiply in the order given (qRotateZ / X / Y * ori, :612-640); given as roll, pitch, yaw that
    is qFromEulerAngles' Rz * ... order reversed: yaw * pitch * roll applied to the identity."""
    d = _run(["-cam-roll", 10, "-cam-pitch", -20, "-cam-yaw", 75, "-name", "r", "-emittance", 1, 1, 1, "-rectangle", 1, 1,
              "-begin-pitch", -90, "-begin-yaw", 150, "-begin-scale", 2, "-inst", "r", "-dry-run"])

    def rot(axis, deg):
        c, s = np.cos(np.radians(deg)), np.sin(np.radians(deg))
        return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
                "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[axis]
    want_cam = rot("y", 75) @ rot("x", -20) @ rot("z", 10)
    assert np.allclose(np.array(d["camera_orientation"]).reshape(3, 3), want_cam, atol=1e-6)
    xfm = np.array(d["instance_transforms"][0][1:]).reshape(3, 4)
    assert np.allclose(xfm[:, :3], 2 * (rot("y", 150) @ rot("x", -90)), atol=1e-6)


def test_option_errors(built_lib):
    r = _run(["-frobnicate"], check=False)
    assert r.returncode != 0 and "unknown option" in r.stderr
    r = _run(["-inst", "nothing", "-dry-run"], check=False)
    assert r.returncode != 0 and "unknown mesh" in r.stderr
    r = _run(["-name", "m", "-obj", "/nonexistent.obj", "1", "trad", "-inst", "m", "-dry-run"], check=False)
    assert r.returncode != 0 and "cannot open" in r.stderr
    r = _run(["-name", "m", "-obj", BUNNY, "1", "fancy", "-dry-run"], check=False)
    assert r.returncode != 0 and "material convention" in r.stderr
    r = _run(["-cam-pos", "1", "2"], check=False)
    assert r.returncode != 0 and "more arguments" in r.stderr


def test_nrc_options_of_the_reference_command_line(built_lib):
    """-position-encoding / -num-hidden-layers / -learning-rate as neural_radiance_caching_main.cpp:755-790 parses them, with its
    defaults (:458-460: hash grid, 2 hidden layers, 1e-2)."""
    base = _scene_args() + ["-size", 64, 48, "-renderer", "nrc", "-dry-run"]
    d = _run(base)
    assert d["renderer"] == -1 and d["nrc"] == {"position_encoding": "hash-grid", "num_hidden_layers": 2, "learning_rate": 0.00999999978,
                                                "max_path_length": 5, "train": True, "nee": "lights"}
    d = _run(base + ["-position-encoding", "tri-wave", "-num-hidden-layers", 5, "-learning-rate", "1e-3", "-max-path-length", 0, "-no-train"])
    assert d["nrc"]["position_encoding"] == "tri-wave" and d["nrc"]["num_hidden_layers"] == 5 and abs(d["nrc"]["learning_rate"] - 1e-3) < 1e-9
    assert d["nrc"]["max_path_length"] == 0 and d["nrc"]["train"] is False
    assert _run(base + ["-nee", "regir"])["nrc"]["nee"] == "regir"
    assert _run(base + ["-nee", "restir"])["nrc"]["nee"] == "restir"
    r = _run(base + ["-nee", "lightcuts"], check=False)
    assert r.returncode != 0 and "NEE sampler" in r.stderr
    r = _run(base + ["-position-encoding", "fourier"], check=False)
    assert r.returncode != 0 and "position encoding" in r.stderr
    r = _run(base + ["-learning-rate", "nan"], check=False)
    assert r.returncode != 0 and "invalid value" in r.stderr
    r = _run(base + ["-num-hidden-layers"], check=False)
    assert r.returncode != 0 and "more arguments" in r.stderr


def _python_scene():
    s = api.HostScene()
    bunny = s.load_obj(BUNNY)
    wide = s.add_rectangle(2.0, 1.0, (10, 20, 40))        # "b_wide" sorts before "c_small"
    small = s.add_rectangle(1.0, 1.0, (50, 50, 50))
    s.add_instance(bunny, api.make_transform(scale=0.1))
    s.add_instance(small, api.make_transform(pos=(0.0, 12.0, 2.0)))
    s.add_instance(wide, api.make_transform(pos=(-6.0, 6.0, 6.0)))
    return s


def _read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(x) for x in f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return data[::-1]      # PFM rows run bottom-up


@pytest.mark.gpu
@pytest.mark.parametrize("renderer,rid", [("restir-biased", api.RENDERER_BIASED), ("rearch-unbiased", api.RENDERER_REARCH_UNBIASED), ("pt", api.RENDERER_PATH_TRACE)])
def test_cli_frames_match_the_python_driver(built_lib, tmp_path, renderer, rid):
    import torch
    W, H, frames = 160, 96, 3
    out = str(tmp_path / "frame.pfm")
    d = _run(_scene_args() + ["-size", W, H, "-frames", frames, "-renderer", renderer, "-out", out])
    ctx = api.Context(0)
    _python_scene().upload(ctx)
    cfg = api.RestirRenderer.default_config(W, H, rid)
    cam = api.make_camera(W, H, (1.5, 5.0, 14.0))
    for k in range(9):
        cam.orientation[k] = d["camera_orientation"][k]
    cfg.camera = cam
    r = api.RestirRenderer(ctx, cfg)
    for _ in range(frames):
        r.render_frame()
    torch.cuda.synchronize()
    want = ctx.read_device(r.beauty_ptr(), W * H * 16).view(np.float32).reshape(H, W, 4)
    assert np.abs(want[..., :3]).sum() > 0
    got = _read_pfm(out)
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(want[..., :3]).view(np.uint32))
    assert np.allclose(d["mean_rgb"], want[..., :3].reshape(-1, 3).astype(np.float64).mean(0), rtol=1e-9)


@pytest.mark.gpu
def test_cli_animation_moves_the_light(built_lib):
    """-animate advances the instance controllers by 1/60 s per frame (InstanceController::updateBody, common_host.h:825-831)
    and rebuilds the acceleration structure: the picture changes; without it the begin placement stays."""
    base = _scene_args() + ["-size", 96, 64, "-frames", 20]
    still, moved = _run(base), _run(base + ["-animate"])
    assert still["mean_rgb"] != moved["mean_rgb"] and all(np.isfinite(moved["mean_rgb"])) and min(moved["mean_rgb"]) > 0


def _write_textured_asset(tmp):
    """A small "real asset": OBJ + MTL with diffuse / normal maps (PPM, TGA) and an emitter image for -rect-emitter-tex."""
    import struct
    rng = np.random.default_rng(9)
    w = h = 16
    rgb = rng.integers(30, 226, (h, w, 3), dtype=np.uint8)
    with open(os.path.join(tmp, "albedo.ppm"), "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h) + rgb.tobytes())
    nrm = np.zeros((h, w, 4), np.uint8)
    nrm[..., 0] = 128 + rng.integers(-30, 31, (h, w)); nrm[..., 1] = 128 + rng.integers(-30, 31, (h, w)); nrm[..., 2] = 235; nrm[..., 3] = 255
    with open(os.path.join(tmp, "normal.tga"), "wb") as f:
        f.write(struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, w, h, 32, 0x28) + nrm[..., [2, 1, 0, 3]].tobytes())
    glow = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)
    with open(os.path.join(tmp, "glow.ppm"), "wb") as f:
        f.write(b"P6\n8 8\n255\n" + glow.tobytes())
    with open(os.path.join(tmp, "room.mtl"), "w") as f:
        f.write("newmtl floor\nKd 0.6 0.6 0.6\nKs 0.05 0.05 0.05\nNs 30\nmap_Kd albedo.ppm\nmap_bump normal.tga\n"
                "newmtl wall\nKd 0.7 0.3 0.2\nKs 0.2 0.2 0.2\nNs 80\n")
    with open(os.path.join(tmp, "room.obj"), "w") as f:
        f.write("mtllib room.mtl\n"
                "v -6 0 -6\nv 6 0 -6\nv 6 0 6\nv -6 0 6\nv -6 5 -6\nv 6 5 -6\n"
                "vt 0 0\nvt 4 0\nvt 4 4\nvt 0 4\nvn 0 1 0\nvn 0 0 1\n"
                "usemtl floor\nf 1/1/1 4/4/1 3/3/1 2/2/1\n"
                "usemtl wall\nf 1/1/2 2/2/2 6/3/2 5/4/2\n")
    return os.path.join(tmp, "room.obj"), os.path.join(tmp, "glow.ppm")


@pytest.mark.gpu
def test_cli_with_a_textured_asset(built_lib, tmp_path):
    """-obj with MTL texture maps, -rect-emitter-tex and -bump through the command line equal the same scene built through the
    bindings (whose textured kernels are checked against the oracle in tests/test_gpu_textures.py)."""
    import torch
    obj, glow = _write_textured_asset(str(tmp_path))
    W, H, frames = 128, 96, 3
    out = str(tmp_path / "room.pfm")
    d = _run(["-cam-pos", 0, 2.5, 9, "-cam-yaw", 180, "-name", "room", "-obj", obj, 1.0, "trad",
              "-name", "panel", "-emittance", 40, 40, 40, "-rect-emitter-tex", glow, "-rectangle", 2.0, 2.0,
              "-inst", "room", "-begin-pos", 0, 4.5, 0, "-inst", "panel",
              "-size", W, H, "-frames", frames, "-bump", "-out", out])
    assert d["textures"] == 3 and d["materials"] == 3
    lib = api.lib()
    h2 = api.HostScene()                                             # "panel" sorts before "room"
    g_panel = lib.gfxh_scene_add_rectangle_textured(h2.h, api.C.c_float(2.0), api.C.c_float(2.0), (api.C.c_float * 3)(40, 40, 40), glow.encode())
    g_room = h2.load_obj(obj)
    h2.add_instance(g_room, api.make_transform())
    h2.add_instance(g_panel, api.make_transform(pos=(0.0, 4.5, 0.0)))
    assert h2.materials()[0].texEmittance == 1 and h2.materials()[1].texA != 0 and h2.materials()[1].texNormal != 0
    ctx = api.Context(0)
    h2.upload(ctx)
    cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_BIASED)
    cam = api.make_camera(W, H, (0.0, 2.5, 9.0))
    for k in range(9):
        cam.orientation[k] = d["camera_orientation"][k]
    cfg.camera = cam
    cfg.enableBumpMapping = 1
    r = api.RestirRenderer(ctx, cfg)
    for _ in range(frames):
        r.render_frame()
    torch.cuda.synchronize()
    want = ctx.read_device(r.beauty_ptr(), W * H * 16).view(np.float32).reshape(H, W, 4)
    assert np.abs(want[..., :3]).sum() > 0
    assert np.array_equal(_read_pfm(out).view(np.uint32), np.ascontiguousarray(want[..., :3]).view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("encoding,layers,lr", [("hash-grid", 2, 1e-2), ("tri-wave", 5, 5e-3)])
def test_cli_nrc_renderer_matches_the_python_driver(built_lib, tmp_path, encoding, layers, lr):
    """-renderer nrc with the reference's network options (neural_radiance_caching_main.cpp:755-790) over gfxh_nrc_*: frame 0
    (inference with the initial parameters, no training step has run) is bit-identical to the NRC renderer driven through
    the bindings; after three frames the pictures agree to the training's own noise (the order of the gradient atomics)."""
    import torch
    W, H = 160, 96
    enc = api.NRC_HASH_GRID if encoding == "hash-grid" else api.NRC_TRIANGLE_WAVE
    opts = ["-size", W, H, "-renderer", "nrc", "-position-encoding", encoding, "-num-hidden-layers", layers, "-learning-rate", lr]

    def python_frames(frames):
        ctx = api.Context(0)
        hs = _python_scene()
        hs.upload(ctx)
        cfg = api.NrcRenderer.default_config(W, H, hs.bounds())
        cfg.positionEncoding, cfg.numHiddenLayers, cfg.learningRate = enc, layers, lr
        cam = api.make_camera(W, H, (1.5, 5.0, 14.0))
        for k in range(9):
            cam.orientation[k] = d["camera_orientation"][k]
        cfg.camera = cam
        r = api.NrcRenderer(ctx, cfg)
        for _ in range(frames):
            r.render_frame()
        r.network()                                    # joins the training stream
        torch.cuda.synchronize()
        out = ctx.read_device(r.beauty_ptr(), W * H * 16).view(np.float32).reshape(H, W, 4)[..., :3].copy()
        stats = r.stats()
        r.close()
        return out, stats

    out = str(tmp_path / "nrc0.pfm")
    d = _run(_scene_args() + opts + ["-frames", 1, "-out", out])
    want, stats = python_frames(1)
    assert np.abs(want).sum() > 0
    assert np.array_equal(_read_pfm(out).view(np.uint32), np.ascontiguousarray(want).view(np.uint32))
    assert d["nrc_last_frame"]["training_records"] == stats["numTrainingData"] > 20
    assert d["nrc_last_frame"]["tile_size"] == list(stats["tileSize"]) and np.isfinite(d["nrc_last_frame"]["loss"])
    out3 = str(tmp_path / "nrc3.pfm")
    d3 = _run(_scene_args() + opts + ["-frames", 4, "-out", out3])
    want3, _ = python_frames(4)
    got3 = _read_pfm(out3)
    assert np.isfinite(got3).all()
    assert abs(got3.mean() - want3.mean()) < 0.02 * want3.mean()
    assert not np.array_equal(got3, _read_pfm(out))                    # the cache is being trained: the picture moves


def _write_pfm(path, rgb):
    """rgb: (H, W, 3) float32, top row first (PFM stores bottom-up, little-endian when the scale is negative)."""
    h, w = rgb.shape[:2]
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(np.ascontiguousarray(rgb[::-1], "<f4").tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("renderer,kind", [("restir-unbiased", "pfm"), ("nrc", "pfm"), ("restir-unbiased", "exr")])
def test_cli_env_texture(built_lib, tmp_path, renderer, kind):
    """-env-texture (restir_di_main.cpp:1188-1197; the NRC sample takes the same option): a float lat-long image through the scene
    builder's loader into gfxh_restir_set_env / gfxh_nrc_set_env.  Frame 0 of the command line is bit-identical to the renderer driven
    through the bindings with the texels the same loader returns, and differs from the frame without the map."""
    import torch
    W, H = 160, 96
    ew, eh = 128, 64
    sky = api.env_make_sky(ew, eh).reshape(eh, ew, 4)
    env_path = str(tmp_path / ("sky." + kind))
    if kind == "pfm":
        _write_pfm(env_path, sky[..., :3])
    else:                                              # OpenEXR, what the reference's loadEnvTexture reads (common_host.cpp:2674)
        api.save_image_hdr(env_path, sky.reshape(-1, 4), ew, eh, 1.0)
    opts = ["-size", W, H, "-renderer", renderer, "-frames", 1]
    out, out_dark = str(tmp_path / "env.pfm"), str(tmp_path / "dark.pfm")
    d = _run(_scene_args() + opts + ["-env-texture", env_path, "-out", out])
    _run(_scene_args() + opts + ["-out", out_dark])

    tmp = api.HostScene()
    tmp.load_texture(env_path)
    (_, w, h, fmt, data), = tmp.textures()
    assert (w, h, fmt) == (ew, eh, api.TEX_RGBA32F)
    texels = data.view(np.float32).copy()
    ctx = api.Context(0)
    hs = _python_scene()
    hs.upload(ctx)
    cam = api.make_camera(W, H, (1.5, 5.0, 14.0))
    for k in range(9):
        cam.orientation[k] = d["camera_orientation"][k]
    if renderer == "nrc":
        cfg = api.NrcRenderer.default_config(W, H, hs.bounds())
        cfg.camera = cam
        r = api.NrcRenderer(ctx, cfg)
    else:
        cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_UNBIASED)
        cfg.camera = cam
        r = api.RestirRenderer(ctx, cfg)
    r.set_env(texels, ew, eh, 1.0, 0.0)
    r.render_frame()
    if renderer == "nrc":
        r.network()
    torch.cuda.synchronize()
    want = ctx.read_device(r.beauty_ptr(), W * H * 16).view(np.float32).reshape(H, W, 4)[..., :3]
    got = _read_pfm(out)
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(want).view(np.uint32))
    assert got.mean() > 1.05 * _read_pfm(out_dark).mean()
    r.close()
