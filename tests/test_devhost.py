"""CPU: the product's scalar (one-lane) device code -- the CABAC writer / bit counter of x264_vs2008_amd/csrc/cabac_dev.h and the
trellis quantiser of trellis_dev.h -- compiled for the host (oracle/devcheck.cpp) and replayed on EVERY call the CPU twin makes
while it encodes a chain (oracle/slice_oracle.c built with -DX264O_DEVCHECK): the same context states, bit counts, coder
registers, bytes and quantised levels.  The twin itself is pinned to the reference by tests/test_oracle_slice.py."""
import ctypes
import os
import subprocess

import pytest

from conftest import ROOT
from oracle import refslice as rs
from oracle.gen_golden_slice import CASES2, case_inputs


@pytest.fixture(scope="module")
def devcheck_lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "devcheck"])
    return ctypes.CDLL(os.path.join(ROOT, "oracle", "libdevcheck.so"))


@pytest.mark.parametrize("name,size,frames,kind,kw,ekw", CASES2, ids=[c[0] for c in CASES2])
def test_device_cabac_and_trellis_text_on_host(devcheck_lib, name, size, frames, kind, kw, ekw):
    before_calls, before_bad = devcheck_lib.x264o_devcheck_calls(), devcheck_lib.x264o_devcheck_bad()
    p = rs.make_params(size[0], size[1], frames, **kw)
    y, u, v = case_inputs(size, frames, kind)
    rs.run2(devcheck_lib, "x264o_encode_chain2", p, rs.make_ext(**ekw), y, u, v)
    assert devcheck_lib.x264o_devcheck_calls() - before_calls > 100      # the replay really ran
    assert devcheck_lib.x264o_devcheck_bad() == before_bad
