"""Builds tests/shim/shim_main (hipcc, host code only) against include/ and gfxexp_amd/libgfxexp.so.
The point of the build is the compile: the shim of INTEGRATION.md is real translation units checked against the C ABI."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXE = os.path.join(HERE, "shim_main")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build(force=False):
    srcs = [os.path.join(HERE, n) for n in ("shim_main.cpp", "network_interface_hip.cpp")]
    deps = srcs + [os.path.join(HERE, n) for n in ("hip_backend.h", "network_interface.h")] + \
        [os.path.join(ROOT, "include", n) for n in ("gfxexp.h", "gfxexp_host.h")]
    if not force and os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return EXE
    lib_dir = os.path.join(ROOT, "gfxexp_amd")
    cmd = [HIPCC, "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + HERE] + srcs + \
          ["-L" + lib_dir, "-lgfxexp", "-Wl,-rpath," + lib_dir, "-o", EXE]
    subprocess.check_call(cmd)
    return EXE


if __name__ == "__main__":
    print(build(force=True))
