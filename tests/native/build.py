"""Builds tests/native/librccl_stub.so (g++, host only): the recording stand-in for librccl that tests/rccl_plan.py loads
through GFX_RCCL_LIBRARY.  Test infrastructure, built in-tree by __graft_entry__.build() so it travels to the GPU box."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
STUB = os.path.join(HERE, "librccl_stub.so")


def build(force=False):
    src = os.path.join(HERE, "rccl_stub.cpp")
    if not force and os.path.exists(STUB) and os.path.getmtime(STUB) >= os.path.getmtime(src):
        return STUB
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-shared", "-fPIC", src, "-o", STUB, "-ldl"])
    return STUB


if __name__ == "__main__":
    print(build(force=True))
