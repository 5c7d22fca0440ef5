"""The Annex B stream around the sweep's payloads: what x264_encoder_encode writes per call (R/encoder/encoder.c:1548-1600) -- the version
SEI with frame 0, SPS + PPS with every IDR, then the slice NAL -- built from the library's host C (include/x264hip_stream.h:
x264hip_validate_parameters, x264hip_sps_write / _pps_write / _sei_version_write, x264hip_slice_nal, x264hip_nal_encode).

    p = mux.encoder_params(lib, width=352, height=288, rc_method=mux.RC_CQP, qp_constant=26, cabac=0, ...)      # validated like x264_encoder_open
    m = mux.AnnexB(lib, p)
    out.write(m.frame(frame=0, ftype=mux.TYPE_IDR, qp=23, n_ref0=0, n_ref1=0, payload=slice_data_bytes))

One AnnexB per stream (chain): it keeps what x264_t keeps between calls -- i_frame_num, i_idr_pic_id, the input number of the last IDR and
the frame_num of the pictures kept as references.  The reference of this version never resets or wraps i_frame_num (encoder.c:684,1541),
so neither does this; beyond 2^log2_max_frame_num reference pictures the reference's own header is corrupt and this one is not the same."""
import ctypes as C

RC_CQP, RC_CRF = 0, 1
TYPE_IDR, TYPE_I, TYPE_P, TYPE_BREF, TYPE_B = 1, 2, 3, 4, 5            # X264_TYPE_*, R/x264.h:131-136
NAL_SLICE, NAL_SLICE_IDR, NAL_SEI, NAL_SPS, NAL_PPS = 1, 5, 6, 7, 8
PRIORITY_DISPOSABLE, PRIORITY_HIGH, PRIORITY_HIGHEST = 0, 2, 3


class EncoderParams(C.Structure):
    """x264hip_encoder_params (include/x264hip_stream.h)"""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("fps_num", C.c_int), ("fps_den", C.c_int), ("level_idc", C.c_int), ("threads", C.c_int),
                ("frame_reference", C.c_int), ("keyint_max", C.c_int), ("keyint_min", C.c_int), ("scenecut_threshold", C.c_int), ("pre_scenecut", C.c_int),
                ("bframe", C.c_int), ("bframe_adaptive", C.c_int), ("bframe_bias", C.c_int), ("bframe_pyramid", C.c_int),
                ("deblocking_filter", C.c_int), ("deblocking_filter_alphac0", C.c_int), ("deblocking_filter_beta", C.c_int),
                ("cabac", C.c_int), ("cabac_init_idc", C.c_int), ("interlaced", C.c_int), ("cqm_preset", C.c_int),
                ("intra", C.c_uint), ("inter", C.c_uint),
                ("transform_8x8", C.c_int), ("weighted_bipred", C.c_int), ("direct_mv_pred", C.c_int), ("chroma_qp_offset", C.c_int),
                ("me_method", C.c_int), ("me_range", C.c_int), ("mv_range", C.c_int), ("subpel_refine", C.c_int), ("chroma_me", C.c_int),
                ("mixed_references", C.c_int), ("trellis", C.c_int), ("fast_pskip", C.c_int), ("dct_decimate", C.c_int), ("noise_reduction", C.c_int),
                ("psy_rd", C.c_float), ("psy_trellis", C.c_float), ("luma_deadzone", C.c_int * 2),
                ("rc_method", C.c_int), ("qp_constant", C.c_int), ("qp_min", C.c_int), ("qp_max", C.c_int), ("qp_step", C.c_int),
                ("rf_constant", C.c_float), ("ip_factor", C.c_float), ("pb_factor", C.c_float), ("qcompress", C.c_float),
                ("aq_mode", C.c_int), ("aq_strength", C.c_float), ("scaling_list", C.c_void_p * 6),
                ("d_valid", C.c_int), ("d_lossless", C.c_int), ("d_profile_idc", C.c_int), ("d_num_ref_frames", C.c_int), ("d_num_reorder_frames", C.c_int),
                ("d_log2_max_frame_num", C.c_int), ("d_log2_max_poc_lsb", C.c_int), ("d_mb_width", C.c_int), ("d_mb_height", C.c_int),
                ("d_pic_init_qp", C.c_int), ("d_log2_max_mv_length", C.c_int), ("d_psy_rd_fix8", C.c_int)]


class SliceHeader(C.Structure):
    """x264hip_slice_header"""
    _fields_ = [("nal_type", C.c_int), ("nal_ref_idc", C.c_int), ("slice_type", C.c_int), ("frame_num", C.c_int), ("idr_pic_id", C.c_int),
                ("poc", C.c_int), ("qp", C.c_int), ("n_ref0", C.c_int), ("n_ref1", C.c_int), ("direct_spatial", C.c_int), ("ref_frame_num", C.c_int * 16)]


def _err(lib):
    lib.x264hip_last_error.restype = C.c_char_p
    return (lib.x264hip_last_error() or b"").decode()


def encoder_params(lib, validate=True, **kw):
    """x264_param_default, then the given fields (names of x264hip_encoder_params), then x264_validate_parameters: the parameters as
    x264_encoder_open leaves them in h->param, with the values x264_sps_init / x264_pps_init derive."""
    p = EncoderParams()
    lib.x264hip_encoder_params_default(C.byref(p))
    for k, v in kw.items():
        if k == "luma_deadzone":
            p.luma_deadzone[0], p.luma_deadzone[1] = v
        elif not hasattr(p, k) or k.startswith("d_"):
            raise TypeError("encoder_params: no field %r" % k)
        else:
            setattr(p, k, v)
    if validate and lib.x264hip_validate_parameters(C.byref(p)) != 0:
        raise ValueError("x264hip_validate_parameters: " + _err(lib))
    return p


def param2string(lib, p):
    buf = C.create_string_buffer(1500)
    n = lib.x264hip_param2string(C.byref(p), buf, 1500)
    if n < 0:
        raise ValueError(_err(lib))
    return buf.raw[:n].decode()


def _nal(lib, ref_idc, typ, rbsp):
    out = C.create_string_buffer(len(rbsp) * 3 // 2 + 16)
    n = lib.x264hip_nal_encode(out, 1, ref_idc, typ, rbsp, len(rbsp))
    return out.raw[:n]


def _rbsp(lib, fn, p):
    buf = C.create_string_buffer(2048)
    n = fn(C.byref(p), buf, 2048)
    if n < 0:
        raise ValueError(_err(lib))
    return buf.raw[:n]


def headers(lib, p, sei=True):
    """SEI (frame 0 only), SPS, PPS as Annex B NALs, in x264_encoder_encode's order."""
    out = b""
    if sei:
        out += _nal(lib, PRIORITY_DISPOSABLE, NAL_SEI, _rbsp(lib, lib.x264hip_sei_version_write, p))
    out += _nal(lib, PRIORITY_HIGHEST, NAL_SPS, _rbsp(lib, lib.x264hip_sps_write, p))
    out += _nal(lib, PRIORITY_HIGHEST, NAL_PPS, _rbsp(lib, lib.x264hip_pps_write, p))
    return out


class AnnexB:
    def __init__(self, lib, params):
        if not params.d_valid:
            raise ValueError("AnnexB: parameters not validated (mux.encoder_params)")
        self.lib, self.p = lib, params
        self.frame_num, self.idr_pic_id, self.last_idr = 0, 0, 0
        self.refs = []                     # [(poc, frame_num)] of the pictures kept as references since the last IDR

    def frame(self, frame, ftype, qp, payload, n_ref0=None, n_ref1=None, direct_spatial=1, frame_num_reset=False):
        """All NALs of one x264_encoder_encode call: `frame` is the picture's input number, `ftype` its X264_TYPE_*, `qp` the slice QP,
        `payload` the sweep's slice_data() bytes (CAVLC: the CAVLC pass's), n_ref0 / n_ref1 the active references (h->i_ref0 / i_ref1;
        None: what x264_reference_build_list finds among the pictures this muxer saw kept, encoder.c:911-981)."""
        lib, p = self.lib, self.p
        out = b""
        sh = SliceHeader()
        if frame_num_reset:                # a scene-cut IDR: the one place the reference restarts i_frame_num (encoder.c:1682)
            self.frame_num = 0
        if ftype == TYPE_IDR:
            self.last_idr = frame
            self.refs = []
            sh.nal_type, sh.nal_ref_idc, sh.slice_type = NAL_SLICE_IDR, PRIORITY_HIGHEST, 2
            sh.idr_pic_id = self.idr_pic_id
            self.idr_pic_id = (self.idr_pic_id + 1) % 65536
            out += headers(lib, p, sei=(frame == 0))
        else:
            sh.nal_type, sh.idr_pic_id = NAL_SLICE, -1
            sh.nal_ref_idc = PRIORITY_DISPOSABLE if ftype == TYPE_B else PRIORITY_HIGH
            sh.slice_type = 2 if ftype == TYPE_I else 0 if ftype == TYPE_P else 1
        sh.frame_num = self.frame_num
        sh.poc = 2 * (frame - self.last_idr)
        if n_ref0 is None:
            n_ref0 = min(sum(1 for r in self.refs if r[0] < sh.poc), p.frame_reference)
        if n_ref1 is None:
            n_ref1 = min(sum(1 for r in self.refs if r[0] > sh.poc), p.d_num_reorder_frames)
        sh.qp, sh.n_ref0, sh.n_ref1, sh.direct_spatial = qp, n_ref0, n_ref1, int(bool(direct_spatial))
        # x264_reference_build_list: list 0 = earlier pictures, nearest (highest POC) first
        l0 = sorted([r for r in self.refs if r[0] < sh.poc], key=lambda r: -r[0])[:max(n_ref0, 0)]
        for i, r in enumerate(l0[:16]):
            sh.ref_frame_num[i] = r[1]
        cap = len(payload) * 3 // 2 + 64
        buf = C.create_string_buffer(cap)
        n = lib.x264hip_slice_nal(C.byref(p), C.byref(sh), payload, len(payload), buf, cap)
        if n < 0:
            raise ValueError("x264hip_slice_nal: " + _err(lib))
        out += buf.raw[:n]
        if sh.nal_ref_idc != PRIORITY_DISPOSABLE:
            self.refs.append((sh.poc, self.frame_num))
            self.refs = sorted(self.refs, key=lambda r: -r[0])[:p.d_num_ref_frames]
            self.frame_num += 1
        return out
