"""CPU: the per-frame DRIVER twins (oracle/frame_oracle.c) against golden planes
produced by the reference's own x264_frame_expand_border_mod16,
x264_frame_init_lowres, x264_frame_deblock_row, x264_frame_expand_border,
x264_frame_filter and x264_frame_expand_border_filtered
(oracle/gen_golden_frames.py).  Whole padded planes, byte for byte."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import hostpic

SIZES = [(352, 288), (200, 120)]
ENC_CASES = [((352, 288), 24, 0, 0), ((352, 288), 27, 1, 0), ((200, 120), 33, 1, 1), ((200, 120), 14, 0, 0)]


ME_CASES = [((352, 288), 1, 16, 7, 1, 26), ((352, 288), 0, 16, 5, 1, 30), ((200, 120), 1, 16, 2, 0, 22), ((200, 120), 1, 8, 1, 0, 36),
            ((352, 288), 1, 16, 3, 1, 40), ((200, 120), 0, 16, 0, 0, 26)]


def me_setup(lib, prefix, size):
    """cur = synthetic frame 7; references = frames 6, 5, 4 with borders and half-pel planes."""
    from x264_vs2008_amd import synth
    g = hostpic.Geometry(*size)
    pics = []
    for t in (7, 6, 5, 4):
        hp = hostpic.HostPic(g)
        hp.load_yuv(lib, prefix, *synth.frame(size[0], size[1], t))
        if t != 7:
            hostpic.make_reference(lib, prefix, hp)
        pics.append(hp)
    return g, pics[0], pics[1:]


@pytest.mark.parametrize("size,method,me_range,subme,chroma_me,qp", ME_CASES)
def test_me_search_twin_matches_reference_me_search_ref(oracle_lib, size, method, me_range, subme, chroma_me, qp):
    """Twin vs the reference's own x264_me_search_ref + refine_subpel on whole frames, three
    references with the half-pel threshold chain (oracle/ref_shim.c, gen_golden_frames.py)."""
    from x264_vs2008_amd.frame import cost_mv_table
    from x264_vs2008_amd.pipeline import LAMBDA_TAB
    name = "me16_%dx%d_m%d_r%d_s%d_c%d_qp%d.npz" % (size[0], size[1], method, me_range, subme, chroma_me, qp)
    with np.load(os.path.join(GOLDEN, name)) as z:
        gold = {k: z[k] for k in z.files}
    g, cur, refs = me_setup(oracle_lib, "x264o_", size)
    n = g.mb_w * g.mb_h
    vp = hostpic.vp
    span = 4 * 2048
    tab = np.ascontiguousarray(cost_mv_table(LAMBDA_TAB[qp], span).view(np.int16))
    planes = (hostpic.u8p * 18)(*[hp.ptr(nm) for hp in refs for nm in ("y", "h", "vv", "c", "u", "v")])
    out_mv = np.zeros((n, 3, 2), np.int16); out_cost = np.zeros((n, 3), np.int32); best = np.zeros((n, 4), np.int32)
    a = {k: np.ascontiguousarray(gold[k]) for k in ("mvp", "mvc", "n_mvc", "ref_cost")}
    oracle_lib.x264o_frame_me_search16(cur.ptr("y"), cur.ptr("u"), cur.ptr("v"), planes, 3, g.mb_w, g.mb_h, g.stride_y, g.stride_c,
                                       method, me_range, subme, chroma_me, 512, vp(tab), span, vp(a["mvp"]), vp(a["mvc"]),
                                       vp(a["n_mvc"]), vp(a["ref_cost"]), vp(out_mv), vp(out_cost), vp(best))
    assert np.array_equal(out_mv, gold["out_mv"]), "vectors differ at %s" % np.argwhere(out_mv != gold["out_mv"])[:5]
    assert np.array_equal(out_cost, gold["out_cost"]), "costs differ at %s" % np.argwhere(out_cost != gold["out_cost"])[:5]
    assert np.array_equal(best, gold["best"]), "best reference differs"


@pytest.mark.parametrize("size,qp,t8,field", ENC_CASES)
def test_residual_and_probe_skip_twins_match_reference_encode(oracle_lib, cqm, size, qp, t8, field):
    """Twin vs the reference's own x264_macroblock_encode (general P partitions, two references)
    and x264_macroblock_probe_skip, run by oracle/ref_shim.c (oracle/gen_golden_frames.py)."""
    from x264_vs2008_amd import synth
    from x264_vs2008_amd.frame import chroma_qp
    with np.load(os.path.join(GOLDEN, "encode_%dx%d_qp%d_t%d_f%d.npz" % (size[0], size[1], qp, t8, field))) as z:
        gold = {k: z[k] for k in z.files}
    g = hostpic.Geometry(*size)
    n = g.mb_w * g.mb_h
    vp = hostpic.vp
    pics = []
    for t in (6, 6, 4):
        hp = hostpic.HostPic(g)
        hp.load_yuv(oracle_lib, "x264o_", *synth.frame(size[0], size[1], t))
        pics.append(hp)
    cur, refs = pics[0], pics[1:]
    for hp in refs:
        hostpic.make_reference(oracle_lib, "x264o_", hp)
    planes = (hostpic.u8p * 12)(*[hp.ptr(nm) for hp in refs for nm in ("y", "h", "vv", "c", "u", "v")])
    rec = hostpic.HostPic(g)
    ly = np.zeros((n, 256), np.int16); lc = np.zeros((n, 128), np.int16); dc = np.zeros((n, 8), np.int16)
    cbp = np.zeros(n, np.int32); nnz = np.zeros((n, 26), np.uint8)
    tabs = {k: np.ascontiguousarray(v) for k, v in cqm.items()}
    dq4 = tabs["dequant4_mf"].astype(np.int32); dq8 = tabs["dequant8_mf"].astype(np.int32)
    mv16 = np.ascontiguousarray(gold["mv16"]); ref8 = np.ascontiguousarray(gold["ref8"])
    oracle_lib.x264o_frame_inter_residual_mp(
        cur.ptr("y"), cur.ptr("u"), cur.ptr("v"), planes, 2, rec.ptr("y"), rec.ptr("u"), rec.ptr("v"), g.mb_w, g.mb_h,
        g.stride_y, g.stride_c, qp, chroma_qp(qp), t8, field, vp(tabs["quant4_mf"]), vp(tabs["quant4_bias"]), vp(tabs["quant8_mf"]),
        vp(tabs["quant8_bias"]), vp(dq4), vp(dq8), vp(mv16), 16, vp(ref8), vp(ly), vp(lc), vp(dc), vp(cbp), vp(nnz))
    for nm in ("y", "u", "v"):
        assert np.array_equal(rec.visible(nm), gold["rec_" + nm]), "reconstruction " + nm
    assert np.array_equal(cbp, gold["cbp"]), "cbp"
    assert np.array_equal(nnz, gold["nnz"]), "nnz"
    assert np.array_equal(dc, gold["dc_c"]), "chroma dc"
    assert np.array_equal(lc, gold["levels_c"]), "chroma levels"
    assert np.array_equal(ly, gold["levels_y"]), "luma levels"
    assert (cbp != 0).any()
    pskip = np.ascontiguousarray(gold["pskip"])
    for w, hp in enumerate(refs):
        out = np.zeros(n, np.uint8)
        oracle_lib.x264o_frame_probe_skip(cur.ptr("y"), cur.ptr("u"), cur.ptr("v"), hp.ptr("y"), hp.ptr("h"), hp.ptr("vv"),
                                          hp.ptr("c"), hp.ptr("u"), hp.ptr("v"), g.mb_w, g.mb_h, g.stride_y, g.stride_c, qp,
                                          chroma_qp(qp), int(gold["lambda2"][0]), field, vp(tabs["quant4_mf"]),
                                          vp(tabs["quant4_bias"]), vp(pskip), vp(out))
        assert np.array_equal(out, gold["skip"][w]), "probe_skip vs reference %d" % w


@pytest.mark.parametrize("size", SIZES, ids=lambda s: "%dx%d" % s)
def test_driver_twins_match_reference_planes(oracle_lib, size):
    with np.load(os.path.join(GOLDEN, "frame_drivers_%dx%d.npz" % size)) as z:
        inp = {k[3:]: z[k] for k in z.files if k.startswith("in.")}
        gold = {k[4:]: z[k] for k in z.files if k.startswith("out.")}
        a_off, b_off, c_off = (int(v) for v in z["offsets"])
    g = hostpic.Geometry(*size)
    vp = hostpic.vp
    # source side
    src = hostpic.HostPic(g)
    src.load_yuv(oracle_lib, "x264o_", inp["y"], inp["u"], inp["v"])
    oracle_lib.x264o_frame_lowres(src.ptr("y"), g.stride_y, g.w16, g.h16, src.ptr("l0"), src.ptr("lh"), src.ptr("lv"), src.ptr("lc"),
                                  g.stride_lowres, g.width_lowres, g.lines_lowres)
    for nm in ("y", "u", "v", "l0", "lh", "lv", "lc"):
        want = gold["src." + nm]
        if nm.startswith("l") and g.stride_lowres - 64 != g.width_lowres:
            # odd mb_w: the reference replicates from columns it never wrote (frame.c:298-301); compare the written part
            a, stride, w, h, padh, padv = src.full[nm]
            assert np.array_equal(src.arr(nm)[padv:padv + h, padh:padh + w], want[padv:padv + h, padh:padh + w]), "src." + nm
            continue
        assert np.array_equal(src.arr(nm), want), "src." + nm
    # reconstruction side
    rec = hostpic.HostPic(g)
    rec.load_yuv(oracle_lib, "x264o_", inp["y"], inp["u"], inp["v"])
    mbt = np.where(inp["mb_type"] == 3, 0, inp["mb_type"]).astype(np.uint8)      # four 8x8 vectors: still "inter" for the filter
    oracle_lib.x264o_frame_deblock(rec.ptr("y"), rec.ptr("u"), rec.ptr("v"), g.mb_w, g.mb_h, g.stride_y, g.stride_c,
                                   vp(mbt), vp(inp["qp"]), vp(inp["nnz"]), vp(inp["t8"]), vp(inp["mv"]), vp(inp["ref"]),
                                   a_off, b_off, c_off)
    hostpic.make_reference(oracle_lib, "x264o_", rec)
    for nm in ("y", "u", "v", "h", "vv", "c"):
        got, want = rec.arr(nm), gold["rec." + nm]
        assert np.array_equal(got, want), "rec.%s differs at %s" % (nm, np.argwhere(got != want)[:5])
    assert (rec.visible("y")[:size[1], :size[0]] != inp["y"]).sum() > 1000
