"""Builds tests/native/librccl_stub.so (g++, host only): the recording stand-in for librccl that tests/rccl_plan.py loads
through GFX_RCCL_LIBRARY -- and tests/native/librccl_mirror.so (hipcc, gfx950): the timing stand-in of tools/band_host_overhead.py
(strips mirrored across the seams on the device, a spin kernel per operation as the link's latency).  Test / measurement
infrastructure, built in-tree by __graft_entry__.build() so it travels to the GPU box."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
STUB = os.path.join(HERE, "librccl_stub.so")
MIRROR = os.path.join(HERE, "librccl_mirror.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build(force=False):
    msrc = os.path.join(HERE, "rccl_mirror.hip")
    if force or not os.path.exists(MIRROR) or os.path.getmtime(MIRROR) < os.path.getmtime(msrc):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wall", "-shared", "-fPIC", msrc, "-o", MIRROR])
    src = os.path.join(HERE, "rccl_stub.cpp")
    if not force and os.path.exists(STUB) and os.path.getmtime(STUB) >= os.path.getmtime(src):
        return STUB
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-shared", "-fPIC", src, "-o", STUB, "-ldl"])
    return STUB


if __name__ == "__main__":
    print(build(force=True))
