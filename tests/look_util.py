"""TEST INFRASTRUCTURE: the lookahead of one chain driven on the CPU -- the library's host state machine (x264_vs2008_amd.lookahead)
with oracle/look_oracle.c (x264o_look_frame_cost) as the provider of the per-frame costs, on lowres planes made by the oracle
(x264o_frame_lowres, x264o_frame_lookahead_intra)."""
import ctypes as C
import os

import numpy as np

from oracle import hostpic
from x264_vs2008_amd import lookahead as LA

HERE = os.path.dirname(os.path.abspath(__file__))
_ora = None
SPAN = 16384


class Task(C.Structure):
    """x264o_look_task"""
    _fields_ = [("mb_w", C.c_int), ("mb_h", C.c_int), ("stride", C.c_int), ("p0", C.c_int), ("p1", C.c_int), ("b", C.c_int),
                ("do_search", C.c_int * 2), ("me_method", C.c_int), ("me_range", C.c_int), ("weighted_bipred", C.c_int),
                ("bframe_bias", C.c_int)]


def oracle():
    global _ora
    if _ora is None:
        _ora = C.CDLL(os.path.join(os.path.dirname(HERE), "oracle", "liboracle.so"))
    return _ora


class CpuLook:
    """Per-frame lookahead data of one chain on the host: lowres planes, intra costs, the vectors / costs of every (list, distance)."""

    def __init__(self, lib, width, height, me_method, me_range, weightb, bframe_bias, bframes):
        self.ora, self.g = oracle(), hostpic.Geometry(width, height)
        self.n = self.g.mb_w * self.g.mb_h
        self.me_method, self.me_range, self.weightb, self.bias, self.bframes = min(1, me_method), me_range, weightb, bframe_bias, bframes
        self.frames = {}
        self.cost_mv = np.zeros(2 * SPAN + 1, np.int16)
        lib.x264hip_cost_mv_table(C.c_int(1), C.c_int(SPAN), self.cost_mv.ctypes.data_as(C.c_void_p))    # a->i_lambda = x264_lambda_tab[12] = 1

    def add(self, number, y, u, v):
        g, pic = self.g, hostpic.HostPic(self.g)
        pic.load_yuv(self.ora, "x264o_", y, u, v)
        self.ora.x264o_frame_lowres(pic.ptr("y"), g.stride_y, g.w16, g.h16, pic.ptr("l0"), pic.ptr("lh"), pic.ptr("lv"), pic.ptr("lc"),
                                    g.stride_lowres, g.width_lowres, g.lines_lowres)
        intra = np.zeros(self.n, np.int32)
        self.ora.x264o_frame_lookahead_intra(pic.ptr("l0"), g.stride_lowres, g.mb_w, g.mb_h, intra.ctypes.data_as(C.c_void_p))
        self.frames[number] = dict(pic=pic, intra=intra, mv={}, cost={})

    def drop_before(self, number):
        for k in [k for k in self.frames if k < number]:
            del self.frames[k]

    def arrays(self, number, lst, dist):
        f = self.frames[number]
        if (lst, dist) not in f["mv"]:
            f["mv"][(lst, dist)] = np.zeros((self.n, 2), np.int16)
            f["cost"][(lst, dist)] = np.zeros(self.n, np.int32)
        return f["mv"][(lst, dist)], f["cost"][(lst, dist)]

    def cost(self, b, p0, p1, ds0, ds1):
        g = self.g
        t = Task(g.mb_w, g.mb_h, g.stride_lowres, 0, p1 - p0, b - p0, (C.c_int * 2)(ds0, ds1), self.me_method, self.me_range, self.weightb, self.bias)
        planes = (C.c_void_p * 12)()
        for i, fn in enumerate((b, p0, p1)):
            for k, name in enumerate(("l0", "lh", "lv", "lc")):
                planes[4 * i + k] = C.cast(self.frames[fn]["pic"].ptr(name), C.c_void_p)
        mv0, c0 = self.arrays(b, 0, b - p0) if b != p0 else (None, None)
        mv1, c1 = self.arrays(b, 1, p1 - b) if b != p1 else (None, None)
        mvr = self.arrays(p1, 0, p1 - p0)[0] if b < p1 else None
        out = np.zeros(3, np.int32)
        vp = hostpic.vp
        self.ora.x264o_look_frame_cost(C.byref(t), planes, vp(mv0), vp(c0), vp(mv1), vp(c1), vp(mvr), vp(self.frames[b]["intra"]),
                                       C.c_void_p(self.cost_mv.ctypes.data + 2 * SPAN), vp(out))
        return int(out[0]), int(out[1]), int(out[2])


def run_chain(lib, params, look, y, u, v, n_frames, log=None, speculative=True, giveups=None):
    """x264_encoder_encode's loop for one chain: feeds pictures, serves the lookahead's requests from `look`, returns what would be coded:
    [(frame, type, qp, f_qpm, ref0, ref1, lowres vectors of list 0 / 1 or None, i_satd)] in coding order."""
    la = LA.Lookahead(lib, params)
    out, fed = [], 0
    while True:
        flushing = fed >= n_frames
        if not flushing:
            num = la.put()
            look.add(num, y[num], u[num], v[num])
            fed += 1
        left = giveups[len(out)] if giveups is not None and len(out) < len(giveups) else 0     # (the post-encode scene cut's verdicts, from the reference's run)
        while True:
            kind, fr, needs = la.get(flushing, speculative)
            if kind == LA.FRAME and left > 0:
                left -= 1
                la.scenecut()
                continue
            if kind != LA.NEED:
                break
            for (b, p0, p1, ds0, ds1, spec) in needs:
                if log is not None:
                    log.append((b, p0, p1, ds0, ds1, spec))
                la.set_cost(b, p0, p1, *look.cost(b, p0, p1, ds0, ds1), speculative=spec)
        if kind == LA.END:
            break
        if kind == LA.NONE:
            continue
        mv0 = look.arrays(fr.frame, 0, fr.frame - fr.ref0_frame)[0].copy() if fr.lowres_l0 else None
        mv1 = look.arrays(fr.frame, 1, fr.ref1_frame - fr.frame)[0].copy() if fr.lowres_l1 else None
        out.append((fr.frame, fr.type, fr.qp, fr.f_qpm, fr.ref0_frame, fr.ref1_frame, mv0, mv1, fr.i_satd))
        la.end()
        look.drop_before(la.oldest_live())
    la.close()
    return out
