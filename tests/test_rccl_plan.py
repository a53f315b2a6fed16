"""gfxh_rccl_exchange beyond one rank: the send / receive plan of every rank of an 8-rank split against the recording
librccl stand-in (tests/rccl_plan.py, tests/native/rccl_stub.cpp).  One box has one GPU, so against the real library only
world = 1 ever runs; the peer / pointer / byte arithmetic for rank +- 1 is executed here instead.  Each check runs in its own
process: the product loads ONE librccl per process (GFX_RCCL_LIBRARY names it)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode):
    from tests.native import build as native_build
    native_build.build()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_plan.py"), mode], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_strip_allreduce_and_broadcast_plan_of_all_eight_ranks(built_lib):
    out = _run("cpu")
    assert int(out.split()[1]) > 2000          # messages of 7 renderer configurations x 2 sequence states x 8 ranks


@pytest.mark.gpu
def test_band_and_record_gather_plan_as_ranks_0_3_and_7_of_8(built_lib):
    _run("gpu")


def test_a_library_without_the_rccl_entry_points_is_refused_every_time(built_lib):
    """A failed load must not leave the function table half filled: the second call reports the same error instead of
    jumping through null pointers."""
    code = ("import ctypes as C, os; os.environ['GFX_RCCL_LIBRARY'] = 'libm.so.6'\n"
            "from gfxexp_amd import api; L = api.lib(); L.gfxh_rccl_last_error.restype = C.c_char_p\n"
            "ident = (C.c_uint8 * 128)()\n"
            "for _ in range(2):\n"
            "    assert L.gfxh_rccl_unique_id(ident) == 1\n"
            "    assert b'lacks' in L.gfxh_rccl_last_error(), L.gfxh_rccl_last_error()\n"
            "print('refused twice')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode == 0 and "refused twice" in r.stdout, r.stdout + r.stderr
