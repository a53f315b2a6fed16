"""Builds gfxexp_amd/libgfxexp.so in-tree with hipcc for gfx950 (no JIT cache, the .so travels
with the repository snapshot).  Flags that are part of the numerical contract:
  -ffp-contract=off                              no fused multiply-add unless written as fmaf
  -fhip-fp32-correctly-rounded-divide-sqrt       IEEE division / sqrt on the device
and one that is not (measured, profiles/r02_initial_candidates.txt):
  -fno-slp-vectorize                             the SLP vectoriser packs adjacent fp32 ops into v_pk_*_f32, which issue at
                                                 half rate on gfx950 and cost extra v_mov to pair registers: -2..3 % frame time
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libgfxexp.so")
CLI = os.path.join(HERE, "restir_di_headless")       # host/restir_di_headless.cpp: the reference's command line, windowless
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SOURCES = ["capi.cpp", "scene.cpp", "lights.hip", "lbvh.hip", "trace.hip", "restir.hip", "pathtrace.hip", "nrc.hip", "textures.hip", "diag.hip",
           "host/scene_builder.cpp", "host/restir_driver.cpp", "host/nrc_driver.cpp", "host/rccl_exchange.cpp", "host/abi_layout.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(HERE, "..", "include")]


def _deps_hash(src, flags):
    h = hashlib.sha256()
    h.update(" ".join(flags).encode())
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for dp, _, fns in sorted(os.walk(root)):
            for fn in sorted(fns):
                if fn.endswith((".h", ".hip", ".cpp")) and (fn.endswith(".h") or os.path.join(dp, fn) == src):
                    with open(os.path.join(dp, fn), "rb") as f:
                        h.update(f.read())
    return h.hexdigest()


def _compile(rel, obj_dir=OBJ, flags=FLAGS):
    src = os.path.join(CSRC, rel)
    if not os.path.exists(src):
        return None
    obj = os.path.join(obj_dir, rel.replace("/", "_") + ".o")
    stamp = obj + ".stamp"
    want = _deps_hash(src, flags)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return obj
    cmd = [HIPCC] + flags + (["-x", "hip"] if rel.endswith(".hip") else []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (rel, r.stdout[-4000:], r.stderr[-8000:]))
    if r.stderr.strip():
        sys.stderr.write(r.stderr[-3000:])
    with open(stamp, "w") as f:
        f.write(want)
    return obj


def build(force=False):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for fn in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, fn))
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = [o for o in ex.map(_compile, SOURCES) if o]
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        subprocess.check_call(cmd)
    cli_src = os.path.join(CSRC, "host", "restir_di_headless.cpp")
    if force or not os.path.exists(CLI) or os.path.getmtime(CLI) < max(os.path.getmtime(cli_src), os.path.getmtime(LIB)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(HERE, "..", "include"), cli_src, "-o", CLI,
                               "-L" + HERE, "-lgfxexp", "-Wl,-rpath,$ORIGIN"])
    return LIB


def build_variant(name, defines):
    """Experiment builds (tools/sessions): the same sources with extra -D switches -> gfxexp_amd/variants/libgfxexp_<name>.so;
    api.py loads it when GFX_LIB names the file.  Not part of the product build."""
    vdir = os.path.join(HERE, "variants")
    odir = os.path.join(HERE, "build_variants", "obj_" + name)      # objects stay here (.gpurunignore); only the .so travels
    os.makedirs(vdir, exist_ok=True)
    os.makedirs(odir, exist_ok=True)
    flags = FLAGS + [d if d.startswith("-") else "-D" + d for d in defines]
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = [o for o in ex.map(lambda r: _compile(r, odir, flags), SOURCES) if o]
    lib = os.path.join(vdir, "libgfxexp_%s.so" % name)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":
        print(build_variant(sys.argv[2], sys.argv[3:]))
    else:
        print(build(force="--force" in sys.argv))
