"""The lookahead's cost kernel (x264hip_lookahead_cost_frames: x264_slicetype_frame_cost with x264_slicetype_mb_cost on the GPU, one task
per wavefront) against the oracle's restatement (oracle/look_oracle.c, itself pinned to the reference by tests/test_lookahead_host.py):
every task's score / intra count / intra cost, every vector and vector cost it leaves in HBM, and -- through the same host state
machine -- the frame types, QPs and offered vectors of whole chains.  Several chains per launch, each with its own clip."""
import ctypes as C

import numpy as np
import pytest

import look_cases as K
import look_util as U
from x264_vs2008_amd import lookahead as LA
from x264_vs2008_amd.frame import FrameCtx

pytestmark = pytest.mark.gpu


def _chains(seeds, w, h, frames):
    cs = []
    for s in seeds:
        c = K.config(s)
        c.update(w=w, h=h, frames=frames)
        cs.append(c)
    return cs


@pytest.mark.parametrize("w,h,seeds,opts", [
    (128, 96, [0, 3, 9], dict(bframes=3, b_adapt=1, crf=23.0, weightb=1, me=1)),
    (144, 112, [4, 7], dict(bframes=2, b_adapt=2, crf=27.5, weightb=0, me=2, bframe_bias=20)),        # odd number of macroblock columns
    (96, 80, [1, 5, 6, 8], dict(bframes=1, b_adapt=1, crf=None, weightb=1, me=0)),
    (352, 288, [2], dict(bframes=3, b_adapt=1, crf=18.0, weightb=1, me=1)),
])
def test_chains_on_gpu_equal_cpu(hip_lib, w, h, seeds, opts):
    frames = 12
    cs = _chains(seeds, w, h, frames)
    for c in cs:                                # one encoder configuration for the batch; the clips (content, scene change, pace) differ
        c.update(bframe_bias=0, qp=cs[0]["qp"])
        c.update(opts)
        c.update(pre_scenecut=1, scenecut_threshold=40, keyint=250, keyint_min=0)
    clips = [K.clip(c["w"], c["h"], frames, c["cut"], c["t0"], c["slow"]) for c in cs]
    # CPU: each chain alone, costs from the oracle
    want, want_tasks = [], []
    for c, (y, u, v) in zip(cs, clips):
        look = U.CpuLook(hip_lib, w, h, c["me"], 16, c["weightb"], c["bframe_bias"], c["bframes"])
        log = []
        want.append(U.run_chain(hip_lib, K.lookahead_params(c), look, y, u, v, frames, log))
        want_tasks.append(log)
    # GPU: all chains together
    c0 = cs[0]
    ctx = FrameCtx(hip_lib, w, h, batch=len(cs))
    dev = LA.LookaheadDevice(ctx, n_slots=2 * c0["bframes"] + 4 if c0["b_adapt"] != 2 else 40, bframes=c0["bframes"], me_method=c0["me"], me_range=16,
                             weightb=c0["weightb"], bframe_bias=c0["bframe_bias"], subme=5)
    lb = LA.LookaheadBatch(ctx, K.lookahead_params(c0), dev)
    got = [[] for _ in cs]
    fed = 0

    def fill(pic, frame):
        for b, (y, u, v) in enumerate(clips):
            ctx.upload(pic, y[frame], u[frame], v[frame], b=b)

    while True:
        flushing = fed >= frames
        if not flushing:
            lb.put(fill)
            fed += 1
        out = lb.get(flushing)
        if all(o is None for o in out):
            if flushing:
                break
            continue
        for ci, fr in enumerate(out):
            if fr is None:
                continue
            mv0 = dev.mv_host(ci, fr.frame, 0, fr.frame - fr.ref0_frame).copy() if fr.lowres_l0 else None
            mv1 = dev.mv_host(ci, fr.frame, 1, fr.ref1_frame - fr.frame).copy() if fr.lowres_l1 else None
            got[ci].append((fr.frame, fr.type, fr.qp, fr.f_qpm, fr.ref0_frame, fr.ref1_frame, mv0, mv1, fr.i_satd))
        lb.end()
    for ci in range(len(cs)):
        assert len(got[ci]) == len(want[ci]) == frames
        for g, wnt in zip(got[ci], want[ci]):
            assert g[:6] == wnt[:6] and g[8] == wnt[8], "chain %d: %s vs %s" % (ci, g[:6] + (g[8],), wnt[:6] + (wnt[8],))
            for k in (6, 7):
                assert (g[k] is None) == (wnt[k] is None)
                assert g[k] is None or np.array_equal(g[k], wnt[k]), "chain %d frame %d list %d: %d vectors differ" % (ci, g[0], k - 6, int((g[k] != wnt[k]).any(1).sum()))
    assert dev.n_tasks_run >= sum(len(t) for t in want_tasks) * 0 + 1
    lb.close(); dev.close()


def test_single_tasks_equal_oracle(hip_lib):
    """Task by task: P, B and intra-only costs of fixed frame triples, with the searched vectors and their costs read back."""
    w, h = 160, 128
    y, u, v = K.clip(w, h, 6, 0, 40, 1)
    look = U.CpuLook(hip_lib, w, h, 1, 16, 1, 0, 3)
    ctx = FrameCtx(hip_lib, w, h, batch=2)
    dev = LA.LookaheadDevice(ctx, n_slots=6, bframes=3, me_method=1, me_range=16, weightb=1, subme=5)
    for f in range(6):
        look.add(f, y[f], u[f], v[f])
        pic = dev.begin_frame(f)
        ctx.upload(pic, y[f], u[f], v[f], b=0)
        ctx.upload(pic, y[5 - f], u[5 - f], v[5 - f], b=1)           # chain 1: the clip backwards
        dev.prepare(f)
    look1 = U.CpuLook(hip_lib, w, h, 1, 16, 1, 0, 3)
    for f in range(6):
        look1.add(f, y[5 - f], u[5 - f], v[5 - f])
    seq = [(0, 0, 0, 0, 0), (3, 0, 3, 1, 0), (1, 0, 3, 1, 1), (2, 0, 3, 1, 1), (1, 0, 1, 0, 0), (2, 1, 2, 1, 0), (4, 3, 4, 1, 0), (5, 3, 5, 1, 0), (4, 3, 5, 0, 1)]
    for (b, p0, p1, ds0, ds1) in seq:
        res = dev.run([(0, b, p0, p1, ds0, ds1), (1, b, p0, p1, ds0, ds1)])
        for ci, lk in enumerate((look, look1)):
            assert tuple(int(x) for x in res[ci]) == lk.cost(b, p0, p1, ds0, ds1), "task %s chain %d" % ((b, p0, p1), ci)
            for lst, dist in ((0, b - p0), (1, p1 - b)):
                if dist:
                    mv, cost = lk.arrays(b, lst, dist)
                    assert np.array_equal(dev.mv_host(ci, b, lst, dist), mv)
                    assert np.array_equal(dev.mv_cost[dev.slot(b)].get()[ci, lst, dist - 1], cost)
    dev.close()


def test_refusals(hip_lib):
    ctx = FrameCtx(hip_lib, 32, 32, batch=1)                # two macroblock columns: the reference scores the edges too
    dev = LA.LookaheadDevice(ctx, n_slots=3, bframes=1)
    for f in range(2):
        dev.begin_frame(f)
    with pytest.raises(RuntimeError, match="two macroblock"):
        dev.run([(0, 1, 0, 1, 1, 0)])
    dev.close()
    ctx = FrameCtx(hip_lib, 128, 96, batch=1)
    dev = LA.LookaheadDevice(ctx, n_slots=3, bframes=1, subme=1)
    for f in range(2):
        dev.begin_frame(f)
    with pytest.raises(RuntimeError, match="SATD"):
        dev.run([(0, 1, 0, 1, 1, 0)])
    dev.close()
