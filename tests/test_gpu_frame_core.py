"""GPU: whole-plane kernels (upload + mod-16 pad, border expansion, half-pel
planes, half-resolution planes, AQ energy, SSD) against the CPU twin built
from the oracle's table entries.  Byte-exact over the FULL padded planes,
padding included, because motion vectors may point into it."""
import ctypes as C

import numpy as np
import pytest

from frame_util import HostPic, make_clip_frame
from x264_vs2008_amd.frame import FrameCtx

pytestmark = pytest.mark.gpu

SIZES = [(352, 288), (200, 120), (1920, 1080)]


@pytest.fixture(params=SIZES, ids=lambda s: "%dx%d" % s)
def ctx(request, hip_lib):
    c = FrameCtx(hip_lib, *request.param)
    yield c
    c.close()


def _eq(got, want, what):
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        raise AssertionError("%s: %d bytes differ, first at (row, col) %s" % (what, len(bad), bad[0]))


def test_upload_and_border(ctx, hip_lib, oracle_lib):
    pic = ctx.new_picture()
    hp = make_clip_frame(ctx, pic, 3, oracle_lib)
    assert hip_lib.x264hip_expand_border(ctx.h, C.byref(pic), 0) == 0
    for name in ("y", "u", "v"):
        _, stride, w, h, padh, padv = hp.full[name]
        oracle_lib.x264o_plane_expand_border(hp.ptr(name), stride, w, h, padh, padv)
        _eq(ctx.download(pic, name), hp.arr(name), "border " + name)


def test_hpel_planes(ctx, hip_lib, oracle_lib):
    pic = ctx.new_picture()
    hp = make_clip_frame(ctx, pic, 5, oracle_lib)
    assert hip_lib.x264hip_expand_border(ctx.h, C.byref(pic), 0) == 0
    assert hip_lib.x264hip_hpel_filter_frame(ctx.h, C.byref(pic)) == 0
    _, stride, w, h, padh, padv = hp.full["y"]
    oracle_lib.x264o_plane_expand_border(hp.ptr("y"), stride, w, h, padh, padv)
    oracle_lib.x264o_frame_hpel(hp.ptr("y"), hp.ptr("h"), hp.ptr("vv"), hp.ptr("c"), stride, w, h, ctx.dims.mb_h)
    for name in ("h", "vv", "c"):
        _eq(ctx.download(pic, name), hp.arr(name), "hpel plane " + name)


def test_lowres_planes(ctx, hip_lib, oracle_lib):
    pic = ctx.new_picture()
    hp = make_clip_frame(ctx, pic, 7, oracle_lib)
    assert hip_lib.x264hip_lowres_init_frame(ctx.h, C.byref(pic)) == 0
    _, stride, w, h, _, _ = hp.full["y"]
    oracle_lib.x264o_frame_lowres(hp.ptr("y"), stride, w, h, hp.ptr("l0"), hp.ptr("lh"), hp.ptr("lv"), hp.ptr("lc"),
                                  pic.stride_lowres, pic.width_lowres, pic.lines_lowres)
    for name in ("l0", "lh", "lv", "lc"):
        got, want = ctx.download(pic, name), hp.arr(name)
        # for odd mb_w the reference pads from columns [width_lowres, stride - 64) that it never
        # writes (frame.c:298-301); both sides start from zeroed planes here, so they still agree
        _eq(got, want, "lowres " + name)
    # the source plane gets its last column / row duplicated (mc.c:314-317)
    _eq(ctx.download(pic, "y"), hp.arr("y"), "source after lowres")
    # lookahead intra cost on the lowres plane (slicetype.c:186-245)
    from x264_vs2008_amd.frame import DeviceArray
    d = ctx.dims
    n = d.mb_w * d.mb_h
    out = DeviceArray(hip_lib, n, np.int32)
    assert hip_lib.x264hip_lookahead_intra_frame(ctx.h, C.byref(pic), out.p) == 0
    ctx.sync()
    want = np.zeros(n, np.int32)
    oracle_lib.x264o_frame_lookahead_intra(hp.ptr("l0"), pic.stride_lowres, d.mb_w, d.mb_h, want.ctypes.data_as(C.c_void_p))
    got = out.get()
    assert np.array_equal(got, want), "lookahead intra cost differs at %s" % np.argwhere(got != want)[:5]
    assert got.min() >= 5 and len(np.unique(got)) > 10
    out.free()


def test_aq_var_and_ssd(ctx, hip_lib, oracle_lib):
    from x264_vs2008_amd.frame import DeviceArray
    a, b = ctx.new_picture(), ctx.new_picture()
    ha = make_clip_frame(ctx, a, 1, oracle_lib)
    hb = make_clip_frame(ctx, b, 2, oracle_lib)
    d = ctx.dims
    n = d.mb_w * d.mb_h
    out = DeviceArray(hip_lib, n, np.int32)
    assert hip_lib.x264hip_aq_var_frame(ctx.h, C.byref(a), out.p) == 0
    ctx.sync()
    want = np.zeros(n, np.int32)
    oracle_lib.x264o_frame_aq_var(ha.ptr("y"), ha.ptr("u"), ha.ptr("v"), d.stride_y, d.stride_c, d.mb_w, d.mb_h,
                                  want.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out.get(), want)
    out.free()
    ssd = (C.c_int64 * 3)()
    assert hip_lib.x264hip_ssd_frame(ctx.h, C.byref(a), C.byref(b), ssd) == 0
    oracle_lib.x264o_frame_ssd.restype = C.c_int64
    for i, name in enumerate(("y", "u", "v")):
        st = d.stride_c if i else d.stride_y
        want = oracle_lib.x264o_frame_ssd(ha.ptr(name), st, hb.ptr(name), st, d.width >> (i > 0), d.height >> (i > 0))
        assert ssd[i] == want, name


@pytest.mark.parametrize("size,batch", [((200, 120), 3), ((352, 288), 2), ((1920, 1080), 1)])
def test_synthetic_source_on_the_device_equals_the_host_generator(hip_lib, size, batch):
    """x264hip_picture_synth (the bench's input: SURVEY 8(d)'s integer generator as a kernel) against x264_vs2008_amd/synth.py,
    every visible pixel and the mod-16 padding, for several batch elements with their own frame numbers."""
    from x264_vs2008_amd import synth
    c = FrameCtx(hip_lib, *size, batch=batch)
    try:
        pic = c.new_picture(source_only=True)
        t0, ts = 37, 12
        c.synth(pic, t0, ts)
        c.sync()
        w, h = size
        for b in range(batch):
            want = synth.frame(w, h, t0 + b * ts)
            for name, pl in zip(("y", "u", "v"), want):
                got = c.download(pic, name, padded=False, b=b)
                hh, ww = pl.shape
                _eq(got[:hh, :ww], pl, "synth %s element %d" % (name, b))
                # x264_frame_expand_border_mod16: the last column / row repeated up to the coded size
                assert (got[:hh, ww:] == pl[:, -1:]).all() and (got[hh:, :ww] == pl[-1:, :]).all() and (got[hh:, ww:] == pl[-1, -1]).all()
    finally:
        c.close()
