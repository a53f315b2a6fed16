"""The drop-in boundary as compiled code: tests/shim/ holds the shim a maintainer of the
reference adds -- hipbackend::GPUEnvironment / Pipeline / updateASs and a NeuralRadianceCache
with the reference's member signatures (network_interface.h:14-28) implemented over gfx_nrc_* -- plus a miniature host
program using them.  CPU: it compiles and links against include/gfxexp.h + libgfxexp.so.  GPU: the program runs three
frames of the reference's frame loop and a train / infer cycle through the shim, and an ABI failure arrives as
std::runtime_error."""
import subprocess

import pytest

from tests.shim import build as shim_build


def test_shim_compiles_against_the_c_abi(built_lib):
    exe = shim_build.build(force=True)
    out = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
    for sym in ("gfx_ctx_create", "gfx_restir_launch", "gfx_accel_build", "gfx_nrc_infer", "gfx_nrc_train"):
        assert sym in out, f"shim_main does not reference {sym}"


@pytest.mark.gpu
def test_shim_program_runs(built_lib):
    exe = shim_build.build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("shim ok"), r.stdout + r.stderr
