"""The library's host half of the lookahead and rate control (x264_vs2008_amd/csrc/lookahead_host.hip through x264_vs2008_amd.lookahead:
x264_encoder_encode's frame queue, x264_slicetype_decide with b-adapt 1 / 2 and the pre-encode scene cut, x264_rc_analyse_slice, CQP and
CRF) against the REFERENCE's own encoder run on the same clips -- frame order, slice types, QPs, i_satd and the lowres vectors handed to
the main encode.  No GPU: the per-frame costs the decisions read come from the oracle's restatement (oracle/look_oracle.c), which this
pins at the same time; tests/test_gpu_lookahead.py then holds the kernel to the oracle.

  * golden: tests/golden/look_host.npz, made by oracle/gen_golden_look.py from the reference (runs everywhere);
  * live: further seeds against oracle/_ref/libx264ref.so where it is built."""
import os

import numpy as np
import pytest

import look_cases as K
import look_util as U
from oracle import refslice as rs
from x264_vs2008_amd import lib as L
from x264_vs2008_amd import lookahead as LA

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libx264ref.so")
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "look_host.npz"))


def run_mine(c, log=None, speculative=True):
    lib = L.open_library()
    y, u, v = K.clip(c["w"], c["h"], c["frames"], c["cut"], c["t0"], c["slow"])
    look = U.CpuLook(lib, c["w"], c["h"], c["me"], 16, c["weightb"], c["bframe_bias"], c["bframes"])
    return U.run_chain(lib, K.lookahead_params(c), look, y, u, v, c["frames"], log, speculative)


@pytest.mark.parametrize("seed", [int(s) for s in GOLD["seeds"]])
def test_host_lookahead_equals_reference_fixture(seed):
    c = K.config(seed)
    head, qavg, mv = GOLD["s%d_head" % seed], GOLD["s%d_qavg" % seed], GOLD["s%d_mv" % seed]
    ref = [dict(frame=int(h[0]), slice=int(h[1]), poc=int(h[2]), qp=int(h[3]), satd=int(h[4]), f_qp_avg=float(q),
                mv0=m[0] if h[5] else None, mv1=m[1] if h[6] else None) for h, q, m in zip(head, qavg, mv)]
    bad = K.compare(run_mine(c), ref)
    assert not bad, "%s: %s" % (c, bad[:5])


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libx264ref.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed", list(range(200, 236)))
def test_host_lookahead_equals_reference_live(seed):
    c = K.config(seed)
    ref = K.records_of_reference(K.reference_records(c), c["frames"])
    bad = K.compare(run_mine(c), ref)
    assert not bad, "%s: %s" % (c, bad[:5])


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libx264ref.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed", [11, 13, 14, 15, 19, 20, 26, 36, 37, 201, 208, 209, 214, 226, 229])
def test_host_post_encode_scenecut_equals_reference_live(seed):
    """Without --pre-scenecut: the reference's run tells which attempts its post-encode scene cut gave up (refslice_out.stat[..][3]); with the same
    verdicts x264hip_lookahead_scenecut must rearrange the queues, re-type the pictures and run the rate control (x264_ratecontrol_start twice for
    a picture coded twice) exactly as x264_encoder_encode does (encoder.c:1645-1699): order, types, QPs, i_satd, the vectors offered to the encode."""
    c = dict(K.config(seed), pre_scenecut=0)
    if c["scenecut_threshold"] < 0:
        c["scenecut_threshold"] = 40
    a = K.reference_records(c)
    give = [int(a["stat"][f][3]) for f in range(c["frames"])]
    lib = L.open_library()
    y, u, v = K.clip(c["w"], c["h"], c["frames"], c["cut"], c["t0"], c["slow"])
    look = U.CpuLook(lib, c["w"], c["h"], c["me"], 16, c["weightb"], c["bframe_bias"], c["bframes"])
    mine = U.run_chain(lib, K.lookahead_params(c), look, y, u, v, c["frames"], giveups=give)
    bad = K.compare(mine, K.records_of_reference(a, c["frames"]))
    assert not bad, "%s (given up: %s): %s" % (c, give, bad[:5])
    assert sum(give) > 0


@pytest.mark.parametrize("seed", [0, 3, 6, 9, 17])
def test_speculative_tasks_change_nothing(seed):
    """The batch of independent costs get() offers beside the one it asked for is an optimisation: with and without it the decisions, the
    QPs and the vectors the main encode sees are the same, and it shortens the question / answer rounds."""
    c = K.config(seed)
    la, lb = [], []
    a, b = run_mine(c, la, True), run_mine(c, lb, False)
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x[:6] == y[:6] and x[8] == y[8]
        for k in (6, 7):
            assert (x[k] is None) == (y[k] is None) and (x[k] is None or np.array_equal(x[k], y[k]))
    assert all(t[5] == 0 for t in lb)


def test_refusals():
    lib = LA.bind(L.open_library())
    # the scene cut that looks at the coded P frame: the queue itself decides without cuts then (the look is the caller's: StreamEncoder.check_scenecut)
    LA.Lookahead(lib, LA.make_params(8, 6, bframes=3, pre_scenecut=0, scenecut_threshold=40)).close()
    with pytest.raises(ValueError):
        LA.Lookahead(lib, LA.make_params(8, 6, bframes=17, scenecut_threshold=-1, pre_scenecut=0))


def test_queue_protocol():
    """put / get / end in the encoder's rhythm: the B buffer fills first (NONE), a frame must be ended before the next get, END after the flush."""
    lib = LA.bind(L.open_library())
    la = LA.Lookahead(lib, LA.make_params(8, 6, bframes=2, b_adapt=0, scenecut_threshold=-1, pre_scenecut=0, qp=30))
    kinds = []
    for _ in range(5):
        la.put()
        kind, fr, needs = la.get(False)
        kinds.append(kind)
        if kind == LA.FRAME:
            with pytest.raises(RuntimeError):
                la.get(False)
            la.end()
    assert kinds == [LA.NONE, LA.NONE, LA.FRAME, LA.FRAME, LA.FRAME]
    coded = 3
    while True:
        kind, fr, needs = la.get(True)
        if kind == LA.END:
            break
        assert kind == LA.FRAME
        la.end()
        coded += 1
    assert coded == 5
