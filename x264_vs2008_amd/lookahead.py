"""Host mirror of the encoder's frame queue for one GOP chain: ctypes over x264hip_lookahead_* (include/x264hip.h; the logic is the
library's host C, csrc/lookahead_host.hip -- x264_slicetype_decide, x264_rc_analyse_slice, the CQP / CRF rate control).

    la = Lookahead(lib, LookaheadParams(...))
    la.put()                      # a picture enters frames.next (R/encoder/encoder.c:1404-1421)
    kind, frame, needs = la.get(flushing)
        NONE  : the B buffer is filling
        NEED  : `needs` = the x264_slicetype_frame_cost tasks whose results the decision lacks (the first one asked for, the rest speculative);
                compute them, la.set_cost(...) each, call get again
        FRAME : `frame` = what x264_encoder_encode would code now (type, QP, which lowres vectors the 16x16 search is offered)
        END   : flushed
    la.end()                      # x264_ratecontrol_end + x264_reference_update for that frame

Nothing here computes a cost: the GPU does (x264hip_lookahead_cost_frames through x264_vs2008_amd.slice), or, in the CPU tests, the oracle."""
import ctypes as C

NONE, FRAME, NEED, END = 0, 1, 2, 3
TYPE_IDR, TYPE_I, TYPE_P, TYPE_B = 1, 2, 3, 5          # R/x264.h:116-121
RC_CQP, RC_CRF = 0, 1
MAX_NEED = 16


class LookaheadParams(C.Structure):
    """x264hip_lookahead_params"""
    _fields_ = [("mb_w", C.c_int), ("mb_h", C.c_int), ("bframes", C.c_int), ("b_adapt", C.c_int), ("bframe_bias", C.c_int),
                ("keyint_max", C.c_int), ("keyint_min", C.c_int), ("scenecut_threshold", C.c_int), ("pre_scenecut", C.c_int),
                ("rc_method", C.c_int), ("qp_constant", C.c_int), ("rf_constant", C.c_float), ("ip_factor", C.c_float),
                ("pb_factor", C.c_float), ("qcompress", C.c_float), ("qp_min", C.c_int), ("qp_max", C.c_int), ("qp_step", C.c_int)]


def make_params(mb_w, mb_h, bframes=0, b_adapt=1, bframe_bias=0, keyint_max=250, keyint_min=0, scenecut_threshold=40, pre_scenecut=1,
                crf=None, qp=26, ip_factor=1.4, pb_factor=1.3, qcompress=0.6, qp_min=10, qp_max=51, qp_step=4):
    """The reference's defaults (x264_param_default, R/common/common.c:46-150) where not given; keyint_min as x264_validate_parameters
    leaves it (R/encoder/encoder.c:455-457).  crf None: constant QP `qp`."""
    if keyint_max <= 0:
        keyint_max = 1 << 30
    if keyint_min <= 0:
        keyint_min = keyint_max // 10
    keyint_min = max(1, min(keyint_min, keyint_max // 2 + 1))
    return LookaheadParams(mb_w, mb_h, bframes, b_adapt, bframe_bias, keyint_max, keyint_min, scenecut_threshold, pre_scenecut,
                           RC_CQP if crf is None else RC_CRF, qp, 0.0 if crf is None else crf, ip_factor, pb_factor, qcompress, qp_min, qp_max, qp_step)


class Need(C.Structure):
    """x264hip_look_need"""
    _fields_ = [("b", C.c_int), ("p0", C.c_int), ("p1", C.c_int), ("do_search", C.c_int * 2), ("speculative", C.c_int)]


class Frame(C.Structure):
    """x264hip_look_frame"""
    _fields_ = [("frame", C.c_int), ("type", C.c_int), ("poc", C.c_int), ("kept_as_ref", C.c_int), ("qp", C.c_int), ("f_qpm", C.c_float),
                ("ref0_frame", C.c_int), ("ref1_frame", C.c_int), ("lowres_l0", C.c_int), ("lowres_l1", C.c_int), ("i_satd", C.c_int)]


def bind(lib):
    lib.x264hip_lookahead_new.restype = C.c_void_p
    lib.x264hip_lookahead_new.argtypes = [C.POINTER(LookaheadParams)]
    lib.x264hip_lookahead_delete.argtypes = [C.c_void_p]
    lib.x264hip_lookahead_put.argtypes = [C.c_void_p]
    lib.x264hip_lookahead_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(Frame), C.POINTER(Need), C.c_int, C.POINTER(C.c_int)]
    lib.x264hip_lookahead_set_cost.argtypes = [C.c_void_p] + [C.c_int] * 7
    lib.x264hip_lookahead_end.argtypes = [C.c_void_p]
    lib.x264hip_lookahead_oldest_live.argtypes = [C.c_void_p]
    return lib


class Lookahead:
    def __init__(self, lib, params):
        self.lib = bind(lib)
        self.params = params
        self.h = lib.x264hip_lookahead_new(C.byref(params))
        if not self.h:
            raise ValueError("x264hip_lookahead_new refused the parameters (bframes > 16, a scene cut that re-encodes: "
                             "pre_scenecut = 0 with a threshold >= 0, ...)")
        self._need = (Need * MAX_NEED)()
        self._n = C.c_int(0)

    def close(self):
        if self.h:
            self.lib.x264hip_lookahead_delete(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def put(self):
        return self.lib.x264hip_lookahead_put(self.h)

    def get(self, flushing=False, speculative=True):
        fr = Frame()
        kind = self.lib.x264hip_lookahead_get(self.h, int(flushing), C.byref(fr), self._need, MAX_NEED, C.byref(self._n))
        if kind < 0:
            raise RuntimeError("x264hip_lookahead_get: end() of the previous frame is missing")
        needs = []
        if kind == NEED:
            for i in range(self._n.value):
                nd = self._need[i]
                if nd.speculative and not speculative:
                    continue
                needs.append((nd.b, nd.p0, nd.p1, nd.do_search[0], nd.do_search[1], nd.speculative))
        return kind, (fr if kind == FRAME else None), needs

    def set_cost(self, b, p0, p1, score, intra_mbs, cost00, speculative=0):
        self.lib.x264hip_lookahead_set_cost(self.h, b, p0, p1, int(score), int(intra_mbs), int(cost00), int(speculative))

    def end(self):
        self.lib.x264hip_lookahead_end(self.h)

    def oldest_live(self):
        return self.lib.x264hip_lookahead_oldest_live(self.h)
