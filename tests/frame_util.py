"""Helpers for frame-level parity tests: host images in the device layout
(oracle/hostpic.py) tied to a FrameCtx."""
from oracle import hostpic
from x264_vs2008_amd import synth


class HostPic(hostpic.HostPic):
    """Host twin of x264hip_picture; geometry cross-checked against the device context."""

    def __init__(self, ctx, pic=None):
        d = ctx.dims
        g = hostpic.Geometry(d.width, d.height)
        assert (g.mb_w, g.mb_h, g.stride_y, g.stride_c) == (d.mb_w, d.mb_h, d.stride_y, d.stride_c)
        if pic is not None:
            assert (g.stride_lowres, g.width_lowres, g.lines_lowres) == (pic.stride_lowres, pic.width_lowres, pic.lines_lowres)
        super().__init__(g)
        self.ctx, self.d = ctx, d


def make_clip_frame(ctx, pic, t, ora):
    """Upload synthetic frame t to `pic`; return the HostPic holding the same
    pixels after the reference's mod-16 edge replication."""
    d = ctx.dims
    y, u, v = synth.frame(d.width, d.height, t)
    ctx.upload(pic, y, u, v)
    hp = HostPic(ctx, pic)
    hp.load_yuv(ora, "x264o_", y, u, v)
    return hp
