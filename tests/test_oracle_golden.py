"""CPU: the oracle (oracle/x264_oracle.c) against the committed golden vectors.

The golden vectors are outputs of the reference's own C build of the six DSP
tables (oracle/_ref/libx264ref.so, R/common/*.c compiled where they lie),
produced by oracle/gen_golden.py with the checkasm-style driver
oracle/harness.py.  Equality here is what pins the oracle to the reference;
the GPU parity tests then compare the HIP path with this oracle.
"""
import ctypes as C

import numpy as np
import pytest

import os

from conftest import GOLDEN, load_golden
from oracle import harness
from x264_vs2008_amd.tables import TableSet

SEEDS = (1234, 20090216)


@pytest.mark.parametrize("seed", SEEDS)
def test_inputs_reproducible(seed, cqm):
    """The input pool regenerated from the seed equals the stored inputs."""
    ins, _ = load_golden(seed, 0)
    again = harness.make_inputs(seed, cqm)
    for k, v in ins.items():
        assert np.array_equal(v, again[k]), k


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("family", ["pixel", "dct", "quant", "mc", "predict", "deblock"])
def test_oracle_matches_reference_vectors(oracle_lib, cqm, seed, family):
    ins, gold = load_golden(seed, 0)
    inp = dict(ins)
    inp.update({"cqm." + k: v for k, v in cqm.items()})
    got = harness.run_all(TableSet(oracle_lib, "oracle"), inp, (family,))
    assert got, "no cases ran"
    for k, v in got.items():
        assert k in gold, k
        assert v.shape == gold[k].shape and np.array_equal(v, gold[k]), "oracle != reference: " + k
    prefixes = {"dct": ("dct.", "zigzag.")}.get(family, (family + ".",))
    expected = [k for k in gold if k.startswith(prefixes)]
    assert sorted(expected) == sorted(got.keys())


@pytest.mark.parametrize("seed", SEEDS)
def test_oracle_field_scans(oracle_lib, cqm, seed):
    ins, _ = load_golden(seed, 0)
    _, gold = load_golden(seed, 1)
    inp = dict(ins)
    inp.update({"cqm." + k: v for k, v in cqm.items()})
    got = harness.run_all(TableSet(oracle_lib, "oracle", 1), inp, ("dct",))
    for k, v in gold.items():
        assert np.array_equal(got[k], v), k


def test_oracle_cqm_tables(oracle_lib, cqm):
    """x264o_cqm_flat restates x264_cqm_init for flat matrices (R/common/set.c:68-168)."""
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for is8, (mfk, bk, dk, ncat, n) in enumerate((("quant4_mf", "quant4_bias", "dequant4_mf", 4, 16),
                                                  ("quant8_mf", "quant8_bias", "dequant8_mf", 2, 64))):
        for cat in range(ncat):
            for qp in range(52):
                mf = np.zeros(n, np.uint16); b = np.zeros(n, np.uint16); dq = np.zeros(6 * n, np.int32)
                oracle_lib.x264o_cqm_flat(cat, qp, is8, p(mf), p(b), p(dq))
                assert np.array_equal(mf, cqm[mfk][cat][qp])
                assert np.array_equal(b, cqm[bk][cat][qp])
                assert np.array_equal(dq.reshape(6, n), cqm[dk][cat])


def test_oracle_cqm_jvt_tables(oracle_lib):
    """x264o_cqm with the H.264 default scaling lists (--cqm jvt) against the tables the reference's x264_cqm_init built for them
    (tests/golden/cqm_jvt.npz, oracle/gen_golden_cqm.py), every category and qp including the low ones whose multipliers wrap."""
    with np.load(os.path.join(GOLDEN, "cqm_jvt.npz")) as z:
        jvt = {k: z[k] for k in z.files}
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for is8, (mfk, bk, dk, ncat, n) in enumerate((("quant4_mf", "quant4_bias", "dequant4_mf", 4, 16),
                                                  ("quant8_mf", "quant8_bias", "dequant8_mf", 2, 64))):
        for cat in range(ncat):
            for qp in range(52):
                mf = np.zeros(n, np.uint16); b = np.zeros(n, np.uint16); dq = np.zeros(6 * n, np.int32)
                oracle_lib.x264o_cqm(1, cat, qp, is8, p(mf), p(b), p(dq))
                assert np.array_equal(mf, jvt[mfk][cat][qp]) and np.array_equal(b, jvt[bk][cat][qp]) and np.array_equal(dq.reshape(6, n), jvt[dk][cat])
