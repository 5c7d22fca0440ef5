"""TEST INFRASTRUCTURE -- host images with the device plane layout, and a CPU
run of the whole per-frame hot-path pass through the twin library
(liboracle.so = our restatement, or _ref/libframe_ref.so = the same loops
around the reference's own C table entries).  Used by tests and by bench.py's
cpu_baseline leg only."""
import ctypes as C

import numpy as np

u8p = C.POINTER(C.c_uint8)
PADH = PADV = 32


def align_up(v, a):
    return (v + a - 1) // a * a


def load_lazy(path):
    """ctypes.CDLL with RTLD_LAZY (ctypes itself always binds eagerly): the reference build leaves one
    never-called function undefined (oracle/Makefile)."""
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    ora = C.CDLL(os.path.join(here, "liboracle.so"))
    ora.x264o_dlopen_lazy.restype = C.c_void_p
    ora.x264o_dlerror.restype = C.c_char_p
    h = ora.x264o_dlopen_lazy(path.encode())
    if not h:
        raise OSError("dlopen(%s): %s" % (path, ora.x264o_dlerror().decode()))
    return C.CDLL(path, handle=h)


class Geometry:
    """Same numbers x264hip_frame_ctx_new computes (x264_frame_new, R/common/frame.c:29-152)."""

    def __init__(self, width, height):
        self.width, self.height = width, height
        self.mb_w, self.mb_h = (width + 15) // 16, (height + 15) // 16
        self.w16, self.h16 = self.mb_w * 16, self.mb_h * 16
        self.stride_y = align_up(self.w16 + 2 * PADH, 16)
        self.stride_c = align_up(self.stride_y >> 1, 16)
        self.width_lowres, self.lines_lowres = self.w16 // 2, self.h16 // 2
        self.stride_lowres = align_up(self.width_lowres + 2 * PADH, 16)

    def plane(self, name):
        """(stride, width, lines, padh, padv) of a named plane."""
        if name in ("y", "h", "vv", "c"):
            return self.stride_y, self.w16, self.h16, PADH, PADV
        if name in ("u", "v"):
            return self.stride_c, self.w16 // 2, self.h16 // 2, PADH // 2, PADV // 2
        return self.stride_lowres, self.width_lowres, self.lines_lowres, PADH, PADV


PLANES = ("y", "u", "v", "h", "vv", "c", "l0", "lh", "lv", "lc")


class HostPic:
    """Host twin of x264hip_picture with identical padded geometry."""

    def __init__(self, geom):
        self.g = geom
        self.full = {}
        for name in PLANES:
            stride, w, h, padh, padv = geom.plane(name)
            self.full[name] = (np.zeros((h + 2 * padv + 1, stride), np.uint8), stride, w, h, padh, padv)

    def arr(self, name):
        a, stride, w, h, padh, padv = self.full[name]
        return a[:h + 2 * padv]

    def ptr(self, name, x=0, y=0):
        a, stride, w, h, padh, padv = self.full[name]
        return C.cast(a.ctypes.data + (padv + y) * stride + padh + x, u8p)

    def stride(self, name):
        return self.full[name][1]

    def visible(self, name):
        a, stride, w, h, padh, padv = self.full[name]
        return a[padv:padv + h, padh:padh + w]

    def set_visible(self, name, img):
        a, stride, w, h, padh, padv = self.full[name]
        a[padv:padv + img.shape[0], padh:padh + img.shape[1]] = img

    def load_yuv(self, lib, prefix, y, u, v):
        """x264_frame_copy_picture + x264_frame_expand_border_mod16."""
        for name, img in (("y", y), ("u", u), ("v", v)):
            self.set_visible(name, img)
            stride, w16, h16, _, _ = self.g.plane(name)
            getattr(lib, prefix + "plane_pad_mod16")(self.ptr(name), stride, img.shape[1], img.shape[0], w16, h16)


def vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_reference(lib, prefix, pic):
    """What x264_fdec_filter_row leaves behind for a kept reference: borders + half-pel planes."""
    g = pic.g
    for name in ("y", "u", "v"):
        stride, w, h, padh, padv = g.plane(name)
        getattr(lib, prefix + "plane_expand_border")(pic.ptr(name), stride, w, h, padh, padv)
    getattr(lib, prefix + "frame_hpel")(pic.ptr("y"), pic.ptr("h"), pic.ptr("vv"), pic.ptr("c"), g.stride_y, g.w16, g.h16, g.mb_h)


def cpu_pframe_pass(lib, prefix, g, cur, refs, recon, cqm, qp, qpc, cost_tab, span, me_range=16, t8=1):
    """One P-frame hot-path pass on the CPU: the same sequence bench.py times on the GPU.
    Returns a dict of the produced arrays (also used by tests for end-to-end parity)."""
    f = lambda name: getattr(lib, prefix + name)
    n = g.mb_w * g.mb_h
    f("frame_lowres")(cur.ptr("y"), g.stride_y, g.w16, g.h16, cur.ptr("l0"), cur.ptr("lh"), cur.ptr("lv"), cur.ptr("lc"),
                      g.stride_lowres, g.width_lowres, g.lines_lowres)
    aq = np.zeros(n, np.int32)
    f("frame_aq_var")(cur.ptr("y"), cur.ptr("u"), cur.ptr("v"), g.stride_y, g.stride_c, g.mb_w, g.mb_h, vp(aq))
    out = {"aq": aq, "mv9": [], "cost9": [], "mvq": [], "costq": []}
    for ref in refs:
        mv9 = np.zeros((n, 9, 2), np.int16); c9 = np.zeros((n, 9), np.int32)
        f("frame_me_fullpel")(cur.ptr("y"), ref.ptr("y"), g.mb_w, g.mb_h, g.stride_y, me_range, 512, vp(cost_tab), span,
                              None, None, vp(mv9), vp(c9), None, None)
        mvq = np.zeros((n, 2), np.int16); cq = np.zeros(n, np.int32)
        f("frame_me_subpel")(cur.ptr("y"), ref.ptr("y"), ref.ptr("h"), ref.ptr("vv"), ref.ptr("c"), g.mb_w, g.mb_h, g.stride_y,
                             512, vp(cost_tab), span, None, vp(mv9), vp(mvq), vp(cq))
        out["mv9"].append(mv9); out["cost9"].append(c9); out["mvq"].append(mvq); out["costq"].append(cq)
    ref0, mv = refs[0], out["mvq"][0]
    ly = np.zeros((n, 256), np.int16); lc = np.zeros((n, 128), np.int16); dc = np.zeros((n, 8), np.int16)
    cbp = np.zeros(n, np.int32); nnz = np.zeros((n, 26), np.uint8)
    tabs = {k: np.ascontiguousarray(v) for k, v in cqm.items()}
    dq4 = tabs["dequant4_mf"].astype(np.int32); dq8 = tabs["dequant8_mf"].astype(np.int32)
    f("frame_inter_residual")(cur.ptr("y"), cur.ptr("u"), cur.ptr("v"), ref0.ptr("y"), ref0.ptr("h"), ref0.ptr("vv"), ref0.ptr("c"),
                              ref0.ptr("u"), ref0.ptr("v"), recon.ptr("y"), recon.ptr("u"), recon.ptr("v"), g.mb_w, g.mb_h,
                              g.stride_y, g.stride_c, qp, qpc, t8, 0, vp(tabs["quant4_mf"]), vp(tabs["quant4_bias"]),
                              vp(tabs["quant8_mf"]), vp(tabs["quant8_bias"]), vp(dq4), vp(dq8), vp(mv), vp(ly), vp(lc), vp(dc),
                              vp(cbp), vp(nnz))
    mb_type = np.zeros(n, np.uint8); qpa = np.full(n, qp, np.uint8); t8a = np.full(n, t8, np.uint8)
    mv16 = np.ascontiguousarray(np.repeat(mv[:, None, :], 16, axis=1)); refi = np.zeros((n, 4), np.int8)
    f("frame_deblock")(recon.ptr("y"), recon.ptr("u"), recon.ptr("v"), g.mb_w, g.mb_h, g.stride_y, g.stride_c,
                       vp(mb_type), vp(qpa), vp(nnz), vp(t8a), vp(mv16), vp(refi), 0, 0, 0)
    make_reference(lib, prefix, recon)
    fs = f("frame_ssd"); fs.restype = C.c_int64
    ssd = [fs(cur.ptr(nm), g.plane(nm)[0], recon.ptr(nm), g.plane(nm)[0], g.width >> (i > 0), g.height >> (i > 0))
           for i, nm in enumerate(("y", "u", "v"))]
    out.update(levels_y=ly, levels_c=lc, dc_c=dc, cbp=cbp, nnz=nnz, ssd=np.array(ssd, np.int64))
    return out
