"""GPU: whole-frame in-loop deblocking (2:1 anti-diagonal sweep, one wavefront
per macroblock) against the CPU twin that filters macroblocks in raster order
through the oracle's eight edge kernels, exactly as x264_frame_deblock_row.
Planes must be byte-identical -- any ordering mistake between neighbouring
macroblocks changes pixels."""
import ctypes as C

import numpy as np
import pytest

from frame_util import HostPic, make_clip_frame
from x264_vs2008_amd.frame import DeblockParams, DeviceArray, FrameCtx

pytestmark = pytest.mark.gpu


def _blocky(img, r, step, amp):
    """Add per-block DC offsets so block edges exist for the filter to find."""
    h, w = img.shape
    off = r.randint(-amp, amp + 1, ((h + step - 1) // step, (w + step - 1) // step))
    big = np.kron(off, np.ones((step, step), np.int64))[:h, :w]
    return np.clip(img.astype(np.int64) + big, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("size,seed,a_off,b_off,c_off,qlo,qhi", [
    ((352, 288), 1, 0, 0, 0, 20, 40), ((200, 120), 2, 3, -2, 2, 10, 51), ((352, 288), 3, -6, 6, -4, 0, 30),
    ((640, 368), 4, 0, 0, 0, 30, 51)])
def test_deblock_frame(hip_lib, oracle_lib, size, seed, a_off, b_off, c_off, qlo, qhi):
    ctx = FrameCtx(hip_lib, *size)
    try:
        d = ctx.dims
        n = d.mb_w * d.mb_h
        r = np.random.RandomState(seed)
        pic = ctx.new_picture()
        from x264_vs2008_amd import synth
        y, u, v = synth.frame(d.width, d.height, seed)
        y, u, v = _blocky(_blocky(y, r, 4, 3), r, 16, 6), _blocky(u, r, 4, 4), _blocky(v, r, 4, 4)
        ctx.upload(pic, y, u, v)
        hp = HostPic(ctx, pic)
        for name, img in (("y", y), ("u", u), ("v", v)):
            hp.set_visible(name, img)
            _, stride, w16, h16, _, _ = hp.full[name]
            oracle_lib.x264o_plane_pad_mod16(hp.ptr(name), stride, img.shape[1], img.shape[0], w16, h16)
        mb_type = r.choice([0, 0, 0, 1, 2], n).astype(np.uint8)
        qp = r.randint(qlo, qhi + 1, n).astype(np.uint8)
        t8 = (r.rand(n) < 0.4).astype(np.uint8)
        nnz = (r.rand(n, 26) < 0.3).astype(np.uint8)
        nnz[mb_type == 2] = 0
        nnz[r.rand(n) < 0.3] = 0
        for mb in np.nonzero(t8)[0]:         # 8x8 transform: nnz is per 8x8 (cabac storage)
            for b in range(4):
                nnz[mb, 4 * b:4 * b + 4] = nnz[mb, 4 * b]
        mv16 = r.randint(-6, 7, (n, 1, 2)).astype(np.int16)
        mv = np.repeat(mv16, 16, axis=1)
        sub = r.rand(n) < 0.3                 # some macroblocks with four 8x8 vectors
        for mb in np.nonzero(sub)[0]:
            m8 = r.randint(-6, 7, (2, 2, 2))
            for by in range(4):
                for bx in range(4):
                    mv[mb, bx + 4 * by] = m8[by >> 1, bx >> 1]
        ref = r.randint(0, 2, (n, 4)).astype(np.int8)
        ref[~sub] = ref[~sub][:, :1]
        mv[mb_type == 1] = 0; ref[mb_type == 1] = -1
        bufs = [DeviceArray(hip_lib, a.shape, a.dtype, a) for a in (mb_type, qp, nnz, t8, mv, ref)]
        p = DeblockParams(mb_type=bufs[0].ptr, qp=bufs[1].ptr, nnz=bufs[2].ptr, transform8x8=bufs[3].ptr,
                          mv=bufs[4].ptr, ref=bufs[5].ptr, alpha_c0_offset=a_off, beta_offset=b_off, chroma_qp_offset=c_off)
        assert hip_lib.x264hip_deblock_frame(ctx.h, C.byref(pic), C.byref(p)) == 0, hip_lib.x264hip_last_error()
        ctx.sync()
        vp = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
        keep = [np.ascontiguousarray(a) for a in (mb_type, qp, nnz, t8, mv, ref)]
        oracle_lib.x264o_frame_deblock(hp.ptr("y"), hp.ptr("u"), hp.ptr("v"), d.mb_w, d.mb_h, d.stride_y, d.stride_c,
                                       *[a.ctypes.data_as(C.c_void_p) for a in keep], a_off, b_off, c_off)
        changed = 0
        for name, src in (("y", y), ("u", u), ("v", v)):
            got = ctx.download(pic, name, padded=False)
            a, stride, w, h, padh, padv = hp.full[name]
            want = a[padv:padv + h, padh:padh + w]
            assert np.array_equal(got, want), "%s differs at %s" % (name, np.argwhere(got != want)[:4])
            changed += int((got[:src.shape[0], :src.shape[1]] != src).sum())
        assert changed > 1000, "filter barely ran (%d pixels changed)" % changed
    finally:
        ctx.close()
