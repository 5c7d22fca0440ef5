"""GPU: inter residual pipeline (mc -> dct -> quant -> scan -> decimate ->
dequant -> idct -> reconstruction) for every macroblock of a frame against the
CPU twin, which walks the macroblocks one by one through the oracle's table
entries exactly as x264_macroblock_encode does.  Reconstruction planes,
levels, cbp and nnz flags must all be identical."""
import ctypes as C

import numpy as np
import pytest

from frame_util import HostPic, make_clip_frame
from test_gpu_me import _setup
from x264_vs2008_amd.frame import CqmDevice, DeviceArray, chroma_qp

pytestmark = pytest.mark.gpu


def _mvs(ctx, seed, spread):
    d = ctx.dims
    n = d.mb_w * d.mb_h
    r = np.random.RandomState(seed)
    mv = r.randint(-spread, spread + 1, (n, 2)).astype(np.int16)
    # two thirds of the macroblocks get (near-)perfect vectors so decimation and skipped blocks occur
    kind = r.randint(0, 3, n)
    mv[kind == 0] = 0
    mv[kind == 1] = r.randint(-1, 2, (int((kind == 1).sum()), 2))
    # keep vectors inside the reference's sub-pel limits (analyse.c:258-298)
    for mb in range(n):
        mbx, mby = mb % d.mb_w, mb // d.mb_w
        mv[mb, 0] = np.clip(mv[mb, 0], 4 * (-16 * mbx - 24), 4 * (16 * (d.mb_w - mbx - 1) + 24))
        mv[mb, 1] = np.clip(mv[mb, 1], 4 * (-16 * mby - 24), 4 * (16 * (d.mb_h - mby - 1) + 24))
    mv[0] = 0
    return mv


@pytest.mark.parametrize("size,qp,t8,field", [((352, 288), 26, 0, 0), ((352, 288), 26, 1, 0), ((200, 120), 12, 0, 0),
                                             ((200, 120), 38, 1, 1), ((352, 288), 51, 0, 1), ((352, 288), 0, 1, 0)])
def test_inter_residual(hip_lib, oracle_lib, cqm, size, qp, t8, field):
    ctx, cur, ref, hc, hr = _setup(hip_lib, oracle_lib, size[0], size[1], 6, 6)
    try:
        d = ctx.dims
        n = d.mb_w * d.mb_h
        # chroma of the reference needs its border too (mc_chroma reads into the padding)
        for name in ("u", "v"):
            _, stride, w, h, padh, padv = hr.full[name]
            oracle_lib.x264o_plane_expand_border(hr.ptr(name), stride, w, h, padh, padv)
        recon = ctx.new_picture()
        hrec = HostPic(ctx, recon)
        mv = _mvs(ctx, 11 + qp, 40)
        cq = CqmDevice(hip_lib, cqm)
        p = cq.params(qp, t8, field)
        mv_dev = DeviceArray(hip_lib, (n, 2), np.int16, mv)
        ly = DeviceArray(hip_lib, (n, 256), np.int16); lc = DeviceArray(hip_lib, (n, 128), np.int16)
        dc = DeviceArray(hip_lib, (n, 8), np.int16); cbp = DeviceArray(hip_lib, n, np.int32)
        nnz = DeviceArray(hip_lib, (n, 26), np.uint8)
        rc = hip_lib.x264hip_inter_residual_frame(ctx.h, C.byref(cur), C.byref(ref), C.byref(recon), C.byref(p),
                                                  mv_dev.p, ly.p, lc.p, dc.p, cbp.p, nnz.p)
        assert rc == 0, hip_lib.x264hip_last_error()
        ctx.sync()
        w_ly = np.zeros((n, 256), np.int16); w_lc = np.zeros((n, 128), np.int16); w_dc = np.zeros((n, 8), np.int16)
        w_cbp = np.zeros(n, np.int32); w_nnz = np.zeros((n, 26), np.uint8)
        vp = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
        tabs = {k: np.ascontiguousarray(v) for k, v in cqm.items()}
        tabs["dequant4_mf"] = tabs["dequant4_mf"].astype(np.int32); tabs["dequant8_mf"] = tabs["dequant8_mf"].astype(np.int32)
        oracle_lib.x264o_frame_inter_residual(
            hc.ptr("y"), hc.ptr("u"), hc.ptr("v"), hr.ptr("y"), hr.ptr("h"), hr.ptr("vv"), hr.ptr("c"), hr.ptr("u"), hr.ptr("v"),
            hrec.ptr("y"), hrec.ptr("u"), hrec.ptr("v"), d.mb_w, d.mb_h, d.stride_y, d.stride_c, qp, chroma_qp(qp), t8, field,
            tabs["quant4_mf"].ctypes.data_as(C.c_void_p), tabs["quant4_bias"].ctypes.data_as(C.c_void_p),
            tabs["quant8_mf"].ctypes.data_as(C.c_void_p), tabs["quant8_bias"].ctypes.data_as(C.c_void_p),
            tabs["dequant4_mf"].ctypes.data_as(C.c_void_p), tabs["dequant8_mf"].ctypes.data_as(C.c_void_p),
            mv.ctypes.data_as(C.c_void_p), w_ly.ctypes.data_as(C.c_void_p), w_lc.ctypes.data_as(C.c_void_p),
            w_dc.ctypes.data_as(C.c_void_p), w_cbp.ctypes.data_as(C.c_void_p), w_nnz.ctypes.data_as(C.c_void_p))
        for name in ("y", "u", "v"):
            got, want = ctx.download(recon, name, padded=False), None
            a, stride, w, h, padh, padv = hrec.full[name]
            want = a[padv:padv + h, padh:padh + w]
            assert np.array_equal(got, want), "recon %s differs at %s" % (name, np.argwhere(got != want)[:4])
        assert np.array_equal(cbp.get(), w_cbp), "cbp"
        assert np.array_equal(nnz.get(), w_nnz), "nnz"
        assert np.array_equal(ly.get(), w_ly), "luma levels"
        assert np.array_equal(lc.get(), w_lc), "chroma levels"
        assert np.array_equal(dc.get(), w_dc), "chroma dc"
        if 10 < qp < 45:
            assert w_cbp.any() and (w_cbp == 0).any(), "test should cover coded and decimated macroblocks"
        cq.free()
    finally:
        ctx.close()
