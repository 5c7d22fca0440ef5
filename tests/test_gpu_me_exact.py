"""GPU: x264hip_me_search16_frame -- the reference's own x264_me_search_ref + refine_subpel for the
16x16 block over every macroblock and three references with the half-pel threshold chain -- against
golden vectors produced by the reference's real functions (oracle/ref_shim.c via
oracle/gen_golden_frames.py) and against the CPU twin on fresh predictors."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN
from frame_util import make_clip_frame
from oracle import hostpic
from test_oracle_frame_golden import ME_CASES
from x264_vs2008_amd.frame import DeviceArray, FrameCtx, Me16Params, cost_mv_table
from x264_vs2008_amd.pipeline import LAMBDA_TAB

pytestmark = pytest.mark.gpu

SPAN = 4 * 2048


def _setup(hip_lib, oracle_lib, size):
    ctx = FrameCtx(hip_lib, *size)
    cur = ctx.new_picture()
    hc = make_clip_frame(ctx, cur, 7, oracle_lib)
    refs, hrefs = [], []
    for t in (6, 5, 4):
        pic = ctx.new_picture()
        hp = make_clip_frame(ctx, pic, t, oracle_lib)
        assert hip_lib.x264hip_expand_border(ctx.h, C.byref(pic), 0) == 0
        assert hip_lib.x264hip_hpel_filter_frame(ctx.h, C.byref(pic)) == 0
        hostpic.make_reference(oracle_lib, "x264o_", hp)
        refs.append(pic); hrefs.append(hp)
    return ctx, cur, refs, hc, hrefs


def _gpu_search(hip_lib, ctx, cur, refs, method, me_range, subme, chroma_me, qp, mvp, mvc, n_mvc, ref_cost):
    d = ctx.dims
    n = d.mb_w * d.mb_h
    nr = len(refs)
    tab = cost_mv_table(LAMBDA_TAB[qp], SPAN)
    bufs = [DeviceArray(hip_lib, tab.shape, np.uint16, tab), DeviceArray(hip_lib, (n, nr, 2), np.int16, mvp),
            DeviceArray(hip_lib, (n, nr, 8, 2), np.int16, mvc), DeviceArray(hip_lib, (n, nr), np.uint8, n_mvc)]
    out_mv = DeviceArray(hip_lib, (n, nr, 2), np.int16)
    out_cost = DeviceArray(hip_lib, (n, nr), np.int32)
    best = DeviceArray(hip_lib, (n, 4), np.int32)
    p = Me16Params(me_method=method, me_range=me_range, subme=subme, chroma_me=chroma_me, mv_range=512,
                   cost_mv=bufs[0].ptr, cost_mv_range=SPAN, mvp=bufs[1].ptr, mvc=bufs[2].ptr, n_mvc=bufs[3].ptr)
    for i in range(nr):
        p.ref_cost[i] = int(ref_cost[i])
    arr = (C.c_void_p * nr)(*[C.addressof(r) for r in refs])
    rc = hip_lib.x264hip_me_search16_frame(ctx.h, C.byref(cur), arr, nr, C.byref(p), out_mv.p, out_cost.p, best.p)
    assert rc == 0, hip_lib.x264hip_last_error()
    ctx.sync()
    got = out_mv.get(), out_cost.get(), best.get()
    for b in bufs + [out_mv, out_cost, best]:
        b.free()
    return got


@pytest.mark.parametrize("size,method,me_range,subme,chroma_me,qp", ME_CASES)
def test_me_search16_matches_reference_golden(hip_lib, oracle_lib, size, method, me_range, subme, chroma_me, qp):
    name = "me16_%dx%d_m%d_r%d_s%d_c%d_qp%d.npz" % (size[0], size[1], method, me_range, subme, chroma_me, qp)
    with np.load(os.path.join(GOLDEN, name)) as z:
        gold = {k: z[k] for k in z.files}
    ctx, cur, refs, hc, hrefs = _setup(hip_lib, oracle_lib, size)
    try:
        mv, cost, best = _gpu_search(hip_lib, ctx, cur, refs, method, me_range, subme, chroma_me, qp,
                                     gold["mvp"], gold["mvc"], gold["n_mvc"], gold["ref_cost"])
    finally:
        ctx.close()
    assert np.array_equal(mv, gold["out_mv"]), "vectors differ at %s" % np.argwhere(mv != gold["out_mv"])[:5]
    assert np.array_equal(cost, gold["out_cost"]), "costs differ at %s" % np.argwhere(cost != gold["out_cost"])[:5]
    assert np.array_equal(best, gold["best"]), "best reference differs"
    assert len(np.unique(best[:, 0])) > 1, "every reference index should win somewhere"


@pytest.mark.parametrize("method,subme,chroma_me,n_refs", [(1, 6, 1, 3), (0, 4, 0, 1), (1, 9, 1, 2), (2, 5, 1, 2), (2, 1, 0, 1), (3, 5, 1, 2), (3, 1, 0, 1)])
def test_me_search16_matches_twin_on_wild_predictors(hip_lib, oracle_lib, method, subme, chroma_me, n_refs):
    """Predictors up to the vector limits (clipped starts, border reads), 0..8 candidates, one to three references."""
    size, qp, me_range = (208, 144), 30, 16
    ctx, cur, refs, hc, hrefs = _setup(hip_lib, oracle_lib, size)
    refs, hrefs = refs[:n_refs], hrefs[:n_refs]
    d = ctx.dims
    n = d.mb_w * d.mb_h
    rng = np.random.default_rng(77 + subme)
    mvp = rng.integers(-64, 65, (n, n_refs, 2)).astype(np.int16)
    mvp[::7] = rng.integers(-1200, 1201, (len(mvp[::7]), n_refs, 2))
    mvc = rng.integers(-96, 97, (n, n_refs, 8, 2)).astype(np.int16)
    mvc[::5] = rng.integers(-2000, 2001, (len(mvc[::5]), n_refs, 8, 2))
    n_mvc = rng.integers(0, 9, (n, n_refs)).astype(np.uint8)
    lam = LAMBDA_TAB[qp]
    ref_cost = np.zeros(8, np.int32); ref_cost[:3] = [lam, 3 * lam, 3 * lam]
    try:
        mv, cost, best = _gpu_search(hip_lib, ctx, cur, refs, method, me_range, subme, chroma_me, qp, mvp, mvc, n_mvc, ref_cost)
    finally:
        ctx.close()
    g = hc.g
    vp = hostpic.vp
    tab = np.ascontiguousarray(cost_mv_table(lam, SPAN).view(np.int16))
    planes = (hostpic.u8p * (6 * n_refs))(*[hp.ptr(nm) for hp in hrefs for nm in ("y", "h", "vv", "c", "u", "v")])
    w_mv = np.zeros((n, n_refs, 2), np.int16); w_cost = np.zeros((n, n_refs), np.int32); w_best = np.zeros((n, 4), np.int32)
    oracle_lib.x264o_frame_me_search16(hc.ptr("y"), hc.ptr("u"), hc.ptr("v"), planes, n_refs, g.mb_w, g.mb_h, g.stride_y, g.stride_c,
                                       method, me_range, subme, chroma_me, 512, vp(tab), SPAN, vp(mvp), vp(mvc), vp(n_mvc),
                                       vp(ref_cost), vp(w_mv), vp(w_cost), vp(w_best))
    assert np.array_equal(mv, w_mv), "vectors differ at %s" % np.argwhere(mv != w_mv)[:5]
    assert np.array_equal(cost, w_cost), "costs differ at %s" % np.argwhere(cost != w_cost)[:5]
    assert np.array_equal(best, w_best)
