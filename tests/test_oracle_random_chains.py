"""The twin (oracle/liboracle.so) against the reference's own loop (oracle/_ref/libx264ref.so, built here from the reference's
sources) on the seeded random chains of tests/fuzz_b.py -- payload bytes of every frame.  This is the first hop of the GPU's
random-chain parity (kernel == twin in tests/test_gpu_fuzz_cases.py, twin == reference here); it found a twin bug in round 2
(`--nr` during analysis).  Skipped where the reference library is absent (the GPU box: /root/reference does not travel, but the
built library does, so it runs there too)."""
import ctypes as C
import os

import pytest

import fuzz_b
from oracle import refslice as rs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libx264ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libx264ref.so not built (needs /root/reference)")

# the seeds the GPU test runs (so that both hops cover the same configurations) and a spread of others
SEEDS = [0, 45, 3, 7, 11, 19, 23, 58, 59, 101, 137, 1002, 1019, 1040, 1071, 1153, 1234, 1300, 1411, 1502, 1507, 1511, 1520,
         1, 2, 64, 77, 1600, 1777, 2001, 2500,
         # subme 8-9 (RD refinement), sub-8x8 partitions under the RD levels: 30156 is the chain that showed the cache entries a macroblock inherits
         30000, 30001, 30003, 30006, 30011, 30013, 30019, 30021, 30023, 30156, 31003, 31404]


@pytest.mark.parametrize("seed", SEEDS)
def test_twin_equals_reference_on_random_chain(oracle_lib, seed):
    w, h, frames, kind, kw, ekw, y, u, v = fuzz_b.config(seed)
    a = rs.run2(oracle_lib, "x264o_encode_chain2", rs.make_params(w, h, frames, **kw), rs.make_ext(**ekw), y, u, v)
    b = rs.run_reference2(rs.make_params(w, h, frames, **kw), rs.make_ext(**ekw), y, u, v)
    bad = [f for f in range(frames) if bytes(a["payload"][f, :a["payload_len"][f]]) != bytes(b["payload"][f, :b["payload_len"][f]])]
    assert not bad, "%dx%d x%d %s %s %s: payload of frames %s differs between the twin and the reference" % (w, h, frames, kind, kw, ekw, bad)
    for k in ("mb_type", "mv", "ref", "qp", "cbp"):
        assert (a[k] == b[k]).all(), k
