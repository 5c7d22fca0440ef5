"""TEST INFRASTRUCTURE -- ctypes driver for oracle/ref_slice.c (the reference's own
cache_load / analyse / encode / cache_save loop over a chain of frames) and for the
twin of the same sweep in liboracle.so.  Both fill the same dictionary of arrays."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# R/x264.h:190-199
ANALYSE_I4x4, ANALYSE_I8x8, ANALYSE_PSUB16x16, ANALYSE_PSUB8x8 = 0x0001, 0x0002, 0x0010, 0x0020
ME_DIA, ME_HEX, ME_UMH, ME_ESA = 0, 1, 2, 3
# R/common/macroblock.h:78-102 (mb types), :55-76 (partitions)
I_4x4, I_8x8, I_16x16, I_PCM, P_L0, P_8x8, P_SKIP = 0, 1, 2, 3, 4, 5, 6
B_DIRECT, B_L0_L0, B_8x8, B_SKIP = 7, 8, 17, 18
SLICE_P, SLICE_B, SLICE_I = 0, 1, 2


class Params(C.Structure):
    _fields_ = [(k, C.c_int) for k in (
        "width", "height", "n_frames", "qp", "me_method", "me_range", "subme", "n_refs", "inter", "intra",
        "transform8x8", "fast_pskip", "dct_decimate", "chroma_me", "cabac", "mixed_refs",
        "deblock", "alpha_c0", "beta", "chroma_qp_offset", "keyint", "noise_reduction", "mv_range", "cqm_preset")]


def make_params(width, height, n_frames, qp=26, me_method=ME_DIA, me_range=16, subme=0, n_refs=1, inter=0, intra=0,
                transform8x8=0, fast_pskip=1, dct_decimate=1, chroma_me=1, cabac=0, mixed_refs=0, deblock=0,
                alpha_c0=0, beta=0, chroma_qp_offset=0, keyint=0, noise_reduction=0, mv_range=0, cqm_preset=0):
    if not transform8x8:                      # x264_validate_parameters, R/encoder/encoder.c:487-491
        inter &= ~ANALYSE_I8x8
        intra &= ~ANALYSE_I8x8
    return Params(width, height, n_frames, qp, me_method, me_range, subme, n_refs, inter, intra, transform8x8,
                  fast_pskip, dct_decimate, chroma_me, cabac, mixed_refs, deblock, alpha_c0, beta, chroma_qp_offset, keyint, noise_reduction, mv_range, cqm_preset)


OUT_FIELDS = [("mb_type", np.int8, lambda F, n, R, w, h: (F, n)),
              ("partition", np.int8, lambda F, n, R, w, h: (F, n)),
              ("sub_partition", np.int8, lambda F, n, R, w, h: (F, n, 4)),
              ("mv", np.int16, lambda F, n, R, w, h: (F, n, 16, 2)),
              ("ref", np.int8, lambda F, n, R, w, h: (F, n, 4)),
              ("mvr", np.int16, lambda F, n, R, w, h: (F, R, n, 2)),
              ("nnz", np.uint8, lambda F, n, R, w, h: (F, n, 27)),
              ("i4mode", np.int8, lambda F, n, R, w, h: (F, n, 16)),
              ("i16mode", np.int8, lambda F, n, R, w, h: (F, n)),
              ("chroma_mode", np.int8, lambda F, n, R, w, h: (F, n)),
              ("qp", np.int8, lambda F, n, R, w, h: (F, n)),
              ("cbp", np.int16, lambda F, n, R, w, h: (F, n)),
              ("t8", np.int8, lambda F, n, R, w, h: (F, n)),
              ("luma", np.int16, lambda F, n, R, w, h: (F, n, 256)),
              ("luma_dc", np.int16, lambda F, n, R, w, h: (F, n, 16)),
              ("chroma_dc", np.int16, lambda F, n, R, w, h: (F, n, 8)),
              ("chroma_ac", np.int16, lambda F, n, R, w, h: (F, n, 128)),
              ("rec_y", np.uint8, lambda F, n, R, w, h: (F, h, w)),
              ("rec_u", np.uint8, lambda F, n, R, w, h: (F, h // 2, w // 2)),
              ("rec_v", np.uint8, lambda F, n, R, w, h: (F, h // 2, w // 2)),
              ("fin_y", np.uint8, lambda F, n, R, w, h: (F, h, w)),
              ("fin_u", np.uint8, lambda F, n, R, w, h: (F, h // 2, w // 2)),
              ("fin_v", np.uint8, lambda F, n, R, w, h: (F, h // 2, w // 2)),
              ("frame_info", np.int32, lambda F, n, R, w, h: (F, 4)),
              ("stat", np.int64, lambda F, n, R, w, h: (F, 4))]


class Ext(C.Structure):
    """refslice_ext (oracle/ref_slice.c): what refslice_encode_chain2 takes on top of Params."""
    _fields_ = [("trellis", C.c_int), ("psy_rd", C.c_float), ("psy_trellis", C.c_float), ("aq_mode", C.c_int),
                ("aq_strength", C.c_float), ("write", C.c_int), ("payload_cap", C.c_int), ("cabac_init_idc", C.c_int),
                ("bframes", C.c_int), ("weightb", C.c_int), ("direct_pred", C.c_int), ("lowres_mv", C.c_void_p),
                ("b_adapt", C.c_int), ("pre_scenecut", C.c_int), ("scenecut_threshold", C.c_int), ("keyint_min", C.c_int),
                ("crf", C.c_float), ("bframe_bias", C.c_int)]


DIRECT_SPATIAL, DIRECT_TEMPORAL = 1, 2       # R/x264.h:93-96


def make_ext(trellis=0, psy_rd=0.0, psy_trellis=0.0, aq_mode=0, aq_strength=1.0, write=1, payload_cap=0, cabac_init_idc=0,
             bframes=0, weightb=0, direct_pred=DIRECT_SPATIAL, lowres_mv=None, lowres_seed=None,
             b_adapt=0, pre_scenecut=0, scenecut_threshold=-1, keyint_min=0, crf=-1.0, bframe_bias=0):
    """lowres_mv: int16 [frames in coding order][2 lists][n_mb][2], the lookahead's vectors (0x7fff in a frame / list's first component:
    none); the array must outlive the call.  lowres_seed: run2 makes that array itself with lowres_vectors(seed, ...)."""
    e = Ext(trellis, psy_rd, psy_trellis, aq_mode, aq_strength, write, payload_cap, cabac_init_idc, bframes, weightb, direct_pred, None,
            b_adapt, pre_scenecut, scenecut_threshold, keyint_min, crf, bframe_bias)
    if lowres_mv is not None:
        assert lowres_mv.dtype == np.int16 and lowres_mv.flags["C_CONTIGUOUS"]
        e.lowres_mv = lowres_mv.ctypes.data
        e._keep = lowres_mv
    e._lowres_seed = lowres_seed
    return e


def lowres_vectors(seed, n_frames, n_mb):
    """Stand-ins for the lookahead's half-resolution vectors (fenc->lowres_mvs): [frame in coding order][list][n_mb][2] int16, small
    random vectors; about a quarter of the (frame, list) arrays carry the "not searched" marker 0x7fff in their first component."""
    r = np.random.default_rng(4000 + seed)
    lm = r.integers(-14, 15, (n_frames, 2, n_mb, 2)).astype(np.int16)
    for f in range(n_frames):
        for l in range(2):
            if r.random() < 0.25:
                lm[f, l, 0, 0] = 0x7fff
    return np.ascontiguousarray(lm)


OUT2_FIELDS = [("payload", np.uint8, lambda F, n, cap: (F, cap)),
               ("payload_len", np.int32, lambda F, n, cap: (F,)),
               ("mb_bits", np.int32, lambda F, n, cap: (F, n)),
               ("qp_offset", np.float32, lambda F, n, cap: (F, n)),
               ("mv1", np.int16, lambda F, n, cap: (F, n, 16, 2)),
               ("ref1", np.int8, lambda F, n, cap: (F, n, 4)),
               ("frame_info2", np.int32, lambda F, n, cap: (F, 4)),
               # the stream entry only (the real lookahead and rate control in front of the loop)
               ("rc_info", np.float32, lambda F, n, cap: (F, 4)),
               ("look_mv", np.int16, lambda F, n, cap: (F, 2, n, 2)),
               ("look_cost", np.int32, lambda F, n, cap: (F, 8))]


class Out2(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _, _ in OUT2_FIELDS]


class Out(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _, _ in OUT_FIELDS]


def alloc_out(p):
    mb_w, mb_h = (p.width + 15) // 16, (p.height + 15) // 16
    arrs = {name: np.zeros(shape(p.n_frames, mb_w * mb_h, p.n_refs, 16 * mb_w, 16 * mb_h), dt) for name, dt, shape in OUT_FIELDS}
    o = Out(**{k: v.ctypes.data for k, v in arrs.items()})
    return arrs, o


def clip(width, height, n_frames, t0=0):
    from x264_vs2008_amd import synth
    fr = [synth.frame(width, height, t0 + t) for t in range(n_frames)]
    return tuple(np.ascontiguousarray(np.stack([f[i] for f in fr])) for i in range(3))


def run(lib, fn, p, y, u, v):
    arrs, o = alloc_out(p)
    f = getattr(lib, fn)
    f.restype = C.c_int
    rc = f(C.byref(p), y.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), C.byref(o))
    if rc != 0:
        raise RuntimeError("%s failed: %d" % (fn, rc))
    return arrs


def run2(lib, fn, p, e, y, u, v):
    """The round-2 entry (entropy writer in the loop): the same arrays plus payload / payload_len / mb_bits / qp_offset."""
    mb_w, mb_h = (p.width + 15) // 16, (p.height + 15) // 16
    if not e.payload_cap:
        e.payload_cap = mb_w * mb_h * 800 + 4096
    if getattr(e, "_lowres_seed", None) is not None and not e.lowres_mv:
        e._keep = lowres_vectors(e._lowres_seed, p.n_frames, mb_w * mb_h)
        e.lowres_mv = e._keep.ctypes.data
    arrs, o = alloc_out(p)
    arrs2 = {name: np.zeros(shape(p.n_frames, mb_w * mb_h, e.payload_cap), dt) for name, dt, shape in OUT2_FIELDS}
    o2 = Out2(**{k: v.ctypes.data for k, v in arrs2.items()})
    f = getattr(lib, fn)
    f.restype = C.c_int
    rc = f(C.byref(p), C.byref(e), y.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p),
           C.byref(o), C.byref(o2))
    if rc != 0:
        raise RuntimeError("%s failed: %d" % (fn, rc))
    arrs.update(arrs2)
    # keep only the bytes that were written: the fixture stays small
    arrs["payload"] = np.ascontiguousarray(arrs["payload"][:, :max(int(arrs["payload_len"].max()), 1)])
    return arrs


def reference_lib():
    from oracle import hostpic
    return hostpic.load_lazy(os.path.join(HERE, "_ref", "libx264ref.so"))


def run_reference(p, y, u, v):
    return run(reference_lib(), "refslice_encode_chain", p, y, u, v)


def run_reference2(p, e, y, u, v):
    return run2(reference_lib(), "refslice_encode_chain2", p, e, y, u, v)


def run_reference_stream(p, e, y, u, v):
    """x264_encoder_encode's queue: slice types from x264_slicetype_decide, QPs from x264_ratecontrol_start (e.crf >= 0: CRF)."""
    return run2(reference_lib(), "refslice_encode_stream", p, e, y, u, v)
