"""Seeded random I P B chains (tests/fuzz_b.py) through the raster sweep against the CPU twin, payload bytes of every frame: the
configurations that once differed (adaptive quantisation + a motion search reaching more than 64 pixels from its predictor read the
SLICE QP's vector-cost table instead of the macroblock's: seeds 0 and 45) and a spread of B-slice options incl. temporal direct
prediction with P frames whose last macroblocks end intra; seeds from 1000 draw from a wider option space (I / P chains, UMH / ESA,
ranges, `--nr` with the RD levels -- where the comparison found the TWIN wrong: it denoised the trial encodes)."""
import ctypes as C
import os

import pytest

from fuzz_b import compare

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# seeds from 30000: subme 6..9 (tests/fuzz_b.py: config_refine) -- the RD refinement of subme 8-9 in I / P chains, B chains at subme 8; what the sweep
# refuses (subme 9 with B slices, sub-8x8 partitions with the RD levels) counts as refused, not as a difference
@pytest.mark.parametrize("seed", [0, 45, 3, 7, 11, 19, 23, 58, 59, 101, 137, 1002, 1019, 1040, 1071, 1153, 1234, 1300, 1411, 1502, 1507, 1511, 1520,
                                  30000, 30002, 30004, 30005, 30007, 30008, 30014, 30018, 30024, 30022, 30029, 30016, 30020])
def test_random_chain_matches_twin(hip_lib, cqm, seed):
    twin = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    what, diffs, _ = compare(hip_lib, twin, cqm, seed)
    assert not diffs, "%s: payload differs in %s" % (what, diffs)
