"""CPU: the twin of the reference's per-macroblock loop (oracle/slice_oracle.c: cache_load ->
x264_macroblock_analyse -> x264_macroblock_encode -> cache_save over chains of I and P frames)
against golden arrays produced by the reference's own functions (oracle/ref_slice.c, vectors by
oracle/gen_golden_slice.py).  Every decision (types, modes, vectors, references, cbp, nnz), every
coefficient level and every reconstructed pixel, before and after the loop filter."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import refslice as rs
from oracle.gen_golden_slice import CASES, CASES2, CASES2_TWIN, case_inputs, masked, masked2


def load_case(name):
    with np.load(os.path.join(GOLDEN, "slice_%s.npz" % name)) as z:
        return {k: z[k] for k in z.files}


def assert_same(got, want):
    for k in want:
        assert np.array_equal(got[k], want[k]), "%s differs first at %s" % (k, np.argwhere(got[k] != want[k])[:3].tolist())


@pytest.mark.parametrize("name,size,frames,kind,kw", CASES, ids=[c[0] for c in CASES])
def test_sweep_twin_matches_reference_loop(oracle_lib, name, size, frames, kind, kw):
    want = load_case(name)
    p = rs.make_params(size[0], size[1], frames, **kw)
    y, u, v = case_inputs(size, frames, kind)
    got = masked(rs.run(oracle_lib, "x264o_encode_chain", p, y, u, v))
    assert_same(got, want)
    # the vectors exercise what they claim to
    t = want["mb_type"][1:]
    assert (t == rs.P_L0).any() and (want["cbp"] != 0).any()
    if kind == "static":
        assert (t == rs.P_SKIP).sum() > 20
    if kw.get("intra", 0) & 1:
        assert (want["mb_type"] == rs.I_4x4).any()
    if kw.get("transform8x8"):
        assert (want["mb_type"] == rs.I_8x8).any() and want["t8"][1:].any()
    if kw.get("n_refs", 1) > 1:
        assert (want["ref"] > 0).any()


def load_case2(name):
    with np.load(os.path.join(GOLDEN, "slice2_%s.npz" % name)) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name,size,frames,kind,kw,ekw", CASES2 + CASES2_TWIN, ids=[c[0] for c in CASES2 + CASES2_TWIN])
def test_twin_with_entropy_writer_matches_reference_loop(oracle_lib, name, size, frames, kind, kw, ekw):
    """Round 2: the reference's loop with x264_macroblock_write_cabac in it (so the RD levels, trellis, psy-rd and adaptive
    quantisation see the coder state they see in the encoder) against the twin: every array as above AND the slice payload bytes."""
    want = load_case2(name)
    p = rs.make_params(size[0], size[1], frames, **kw)
    y, u, v = case_inputs(size, frames, kind)
    got = masked2(rs.run2(oracle_lib, "x264o_encode_chain2", p, rs.make_ext(**ekw), y, u, v))
    n = want["payload_len"]
    assert np.array_equal(got["payload_len"], n)
    for f in range(frames):
        assert np.array_equal(got["payload"][f, :n[f]], want["payload"][f, :n[f]]), "payload of frame %d differs" % f
    assert_same({k: v for k, v in got.items() if k != "payload"}, {k: v for k, v in want.items() if k != "payload"})
    assert (n > 16).all()
    if ekw.get("aq_mode"):
        assert len(np.unique(want["qp"][1])) > 2 and np.abs(want["qp_offset"]).max() > 0.5
    if kw["subme"] >= 6:
        assert (want["mb_type"][1:] == rs.P_8x8).any() or (want["mb_type"][1:] == rs.P_L0).any()
    if ekw.get("bframes") and kw["subme"] < 8:
        # B slices: the fixture is in coding order (frame_info2 holds the display index); every family of B types occurs, list 1 is used
        t = want["mb_type"]
        assert (want["frame_info"][:, 0] == rs.SLICE_B).sum() >= 3 and not np.array_equal(want["frame_info2"][:, 0], np.arange(frames))
        assert ((t == rs.B_SKIP).any() or kw["qp"] < 24) and (t == rs.B_DIRECT).any() and (t == rs.B_8x8).any() and (t == 16).any() and (t == 12).any()   # B_BI_BI, B_L1_L1
        assert ((t > rs.B_L0_L0) & (t < 16) & (t != 12)).any()                     # 16x8 / 8x16 with mixed lists
        assert (want["ref1"] == 0).any() and np.abs(want["mv1"]).max() > 0
