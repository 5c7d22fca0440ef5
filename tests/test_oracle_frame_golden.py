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
