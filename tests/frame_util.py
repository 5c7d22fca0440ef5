"""Helpers for frame-level parity tests: host images in the device layout and
ctypes access to the CPU twin (oracle/frame_oracle.c)."""
import ctypes as C

import numpy as np

from x264_vs2008_amd import synth

u8p = C.POINTER(C.c_uint8)


class HostPic:
    """Host twin of x264hip_picture with identical padded geometry."""

    def __init__(self, ctx, pic):
        d = ctx.dims
        self.ctx, self.d = ctx, d
        self.w16, self.h16 = d.mb_w * 16, d.lines_y
        self.full = {}
        for name in ("y", "u", "v", "h", "vv", "c", "l0", "lh", "lv", "lc"):
            stride, w, h, padh, padv = ctx.geometry(pic, name)
            self.full[name] = (np.zeros((h + 2 * padv + 1, stride), np.uint8), stride, w, h, padh, padv)

    def arr(self, name):
        """Full padded image, shaped like FrameCtx.download(padded=True)."""
        a, stride, w, h, padh, padv = self.full[name]
        return a[:h + 2 * padv]

    def ptr(self, name, x=0, y=0):
        a, stride, w, h, padh, padv = self.full[name]
        return C.cast(a.ctypes.data + (padv + y) * stride + padh + x, u8p)

    def stride(self, name):
        return self.full[name][1]

    def set_visible(self, name, img):
        a, stride, w, h, padh, padv = self.full[name]
        a[padv:padv + img.shape[0], padh:padh + img.shape[1]] = img


def make_clip_frame(ctx, pic, t, ora):
    """Upload synthetic frame t to `pic`; return the HostPic holding the same
    pixels after the reference's mod-16 edge replication."""
    d = ctx.dims
    y, u, v = synth.frame(d.width, d.height, t)
    ctx.upload(pic, y, u, v)
    hp = HostPic(ctx, pic)
    for name, img in (("y", y), ("u", u), ("v", v)):
        hp.set_visible(name, img)
        _, stride, w16, h16, _, _ = hp.full[name]
        ora.x264o_plane_pad_mod16(hp.ptr(name), stride, img.shape[1], img.shape[0], w16, h16)
    return hp
