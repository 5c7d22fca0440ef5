"""The stream writer (include/x264hip_stream.h, x264_vs2008_amd/mux.py) on the host, no GPU:

  * x264hip_param2string against the REFERENCE's own x264_param2string (R/common/common.c is in oracle/_ref; parameters set through the
    reference's x264_param_parse like its command line does);
  * x264hip_validate_parameters: the levels, motion-vector ranges and option interplay x264_encoder_open arrives at for BASELINE's flag sets;
  * the whole Annex B stream -- version SEI, SPS, PPS, every slice header, the CAVLC bit splice, emulation prevention -- around the
    REFERENCE's payloads (committed fixture for config 1; live harness runs for configs 2 and 3 where oracle/_ref is built) against the md5
    of the .264 the reference's command line wrote.  R/encoder/set.c and encoder.c cannot be built here (config.h), so
    these md5s are what pins the header writers; they cover every byte."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import mux_cases as M
from x264_vs2008_amd import mux

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libx264ref.so")
need_ref = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libx264ref.so not built (needs /root/reference)")

CLI = {
    "UF": [("qp", "26"), ("no-cabac", None), ("me", "dia"), ("subme", "0"), ("partitions", "none"), ("no-deblock", None), ("aq-mode", "0"), ("scenecut", "-1"),
           ("ref", "1"), ("bframes", "0"), ("b-adapt", "0")],
    "MED": [("crf", "23"), ("ref", "3"), ("bframes", "3"), ("b-adapt", "1"), ("me", "hex"), ("subme", "7"), ("8x8dct", None), ("partitions", "p8x8,b8x8,i8x8,i4x4"),
            ("trellis", "1"), ("weightb", None), ("mixed-refs", None), ("direct", "spatial")],
    "SLOW": [("crf", "23"), ("ref", "5"), ("bframes", "3"), ("b-adapt", "2"), ("me", "umh"), ("subme", "8"), ("8x8dct", None), ("partitions", "p8x8,b8x8,i8x8,i4x4"),
             ("trellis", "1"), ("weightb", None), ("mixed-refs", None), ("direct", "auto"), ("pre-scenecut", None)],
    "misc": [("qp", "31"), ("ref", "4"), ("bframes", "2"), ("b-bias", "10"), ("me", "esa"), ("merange", "24"), ("subme", "9"), ("psy-rd", "0.4:0.2"), ("trellis", "2"),
             ("deblock", "-1:2"), ("nr", "100"), ("cqm", "jvt"), ("chroma-qp-offset", "3"), ("keyint", "48"), ("min-keyint", "6"), ("scenecut", "30"), ("ipratio", "1.2"),
             ("pbratio", "1.5"), ("no-chroma-me", None), ("no-dct-decimate", None), ("deadzone-inter", "12"), ("deadzone-intra", "7")],
    "crf_misc": [("crf", "18.5"), ("qcomp", "0.75"), ("qpmin", "12"), ("qpmax", "44"), ("qpstep", "6"), ("aq-strength", "0.7"), ("bframes", "1"), ("no-cabac", None)],
}
OURS = {
    "UF": dict(M.UF, intra=3),                   # before validation the reference's param.analyse.intra is still 0x3
    "MED": M.MED,
    "SLOW": dict(rc_method=mux.RC_CRF, rf_constant=23.0, frame_reference=5, bframe=3, bframe_adaptive=2, me_method=2, subpel_refine=8, transform_8x8=1, inter=0x113,
                 intra=3, trellis=1, weighted_bipred=1, mixed_references=1, direct_mv_pred=3, pre_scenecut=1),
    "misc": dict(rc_method=mux.RC_CQP, qp_constant=31, frame_reference=4, bframe=2, bframe_bias=10, me_method=3, me_range=24, subpel_refine=9, psy_rd=0.4, psy_trellis=0.2,
                 trellis=2, deblocking_filter_alphac0=-1, deblocking_filter_beta=2, noise_reduction=100, cqm_preset=1, chroma_qp_offset=3, keyint_max=48, keyint_min=6,
                 scenecut_threshold=30, ip_factor=1.2, pb_factor=1.5, chroma_me=0, dct_decimate=0, luma_deadzone=(12, 7)),
    "crf_misc": dict(rc_method=mux.RC_CRF, rf_constant=18.5, qcompress=0.75, qp_min=12, qp_max=44, qp_step=6, aq_strength=0.7, bframe=1, cabac=0),
}


@need_ref
@pytest.mark.parametrize("name", sorted(CLI))
def test_param2string_equals_the_references(hip_lib_host, name):
    from oracle import hostpic
    ref = hostpic.load_lazy(REF_SO)
    ref.x264_param2string.restype = C.c_void_p
    buf = C.create_string_buffer(16384)
    ref.x264_param_default(buf)
    for k, v in CLI[name]:
        assert ref.x264_param_parse(buf, k.encode(), None if v is None else v.encode()) == 0, (k, v)
    ptr = ref.x264_param2string(buf, 0)
    want = C.string_at(ptr).decode()
    p = mux.encoder_params(hip_lib_host, validate=False, width=352, height=288, **OURS[name])
    assert mux.param2string(hip_lib_host, p) == want


def test_validate_parameters_levels_and_interplay(hip_lib_host):
    lib = hip_lib_host
    p = mux.encoder_params(lib, width=352, height=288, **M.UF)
    assert (p.level_idc, p.mv_range, p.intra, p.inter, p.d_profile_idc, p.d_num_ref_frames, p.qp_min, p.qp_max, p.aq_mode, p.psy_rd) == (13, 128, 1, 0, 66, 1, 23, 29, 0, 0.0)
    p = mux.encoder_params(lib, width=1920, height=1080, **M.MED)
    assert (p.level_idc, p.mv_range, p.d_profile_idc, p.d_num_ref_frames, p.d_num_reorder_frames, p.chroma_qp_offset, p.d_pic_init_qp) == (40, 512, 100, 3, 1, -2, 23)
    assert (p.d_log2_max_frame_num, p.d_log2_max_poc_lsb, p.d_mb_width, p.d_mb_height, p.d_log2_max_mv_length) == (9, 10, 120, 68, 11)
    p = mux.encoder_params(lib, width=1920, height=1080, **dict(M.MED, frame_reference=5))
    assert p.level_idc == 50                                   # five 1080p references do not fit level 4.x's DPB
    p = mux.encoder_params(lib, width=3840, height=2160, **M.MED)
    assert p.level_idc == 51
    p = mux.encoder_params(lib, width=640, height=480, rc_method=mux.RC_CQP, qp_constant=0, cabac=0, transform_8x8=1, bframe=2, trellis=1, psy_rd=1.0)
    assert (p.d_lossless, p.d_profile_idc, p.transform_8x8, p.bframe, p.trellis, p.psy_rd, p.intra) == (1, 244, 0, 0, 0, 0.0, 1)
    with pytest.raises(ValueError):
        mux.encoder_params(lib, width=351, height=288)
    with pytest.raises(ValueError):
        mux.encoder_params(lib, width=352, height=288, threads=4)
    with pytest.raises(ValueError):
        mux.encoder_params(lib, width=352, height=288, level_idc=14)


def test_c1_uf_stream_md5_from_the_reference_payload_fixture(hip_lib_host):
    """BASELINE config 1, every byte: the reference's payloads (tests/golden/mux_uf_cif30.npz, oracle/gen_golden_mux.py) inside this
    library's SEI / SPS / PPS / slice headers (I and P, CAVLC: the payload bits spliced on behind the header's last bit)."""
    lib = hip_lib_host
    g = np.load(os.path.join(ROOT, "tests", "golden", "mux_uf_cif30.npz"))
    p = mux.encoder_params(lib, width=352, height=288, **M.UF)
    m, out = mux.AnnexB(lib, p), []
    for t in range(30):
        out.append(m.frame(frame=t, ftype=mux.TYPE_IDR if t == 0 else mux.TYPE_P, qp=23 if t == 0 else 26, payload=bytes(g["payload"][t, :g["payload_len"][t]])))
    stream = b"".join(out)
    assert stream.startswith(b"\x00\x00\x00\x01\x06\x05")
    assert hashlib.md5(stream).hexdigest() == M.STREAM_MD5["C1_UF_cif30"], len(stream)


@need_ref
@pytest.mark.parametrize("cfg", ["C2_MED_hd24", "C3_MED_umh_uhd8"])
def test_med_stream_md5_around_the_reference_encoder(hip_lib_host, cfg):
    """BASELINE configs 2 and 3: the reference's whole encoder through the harness (frame order, types, QPs, CABAC payloads of I / P / B
    slices) inside this library's headers = the md5 of the reference CLI's file.  Pins the header writers' B / CABAC / CRF paths AND the
    harness itself (oracle/ref_slice.c refslice_encode_stream is what every GPU stream test compares with) to the real x264."""
    lib = hip_lib_host
    w, h, n, kw = (1920, 1080, 24, M.MED) if cfg == "C2_MED_hd24" else (3840, 2160, 8, dict(M.MED, me_method=2))
    assert M.clip_md5(w, h, n) == M.CLIP_MD5["hd24" if n == 24 else "uhd8"]
    p = mux.encoder_params(lib, width=w, height=h, **kw)
    a = M.reference_med(p, w, h, n)
    stream = M.mux_reference_stream(lib, p, a, n)
    assert hashlib.md5(stream).hexdigest() == M.STREAM_MD5[cfg], len(stream)


@need_ref
def test_c4_slow_stream_md5s_around_the_reference_encoder(hip_lib_host):
    """BASELINE config 4's flag set (SLOW: 5 references, b-adapt 2, UMH, subme 8, --direct auto) on hd24, with and without --pre-scenecut: the
    harness (with --direct auto's running scores, x264_encoder_frame_end's part of them restated in ref_slice.c) inside this library's headers
    has the md5 of both files the reference command line wrote (the product's own: tests/test_gpu_mux.py::test_c4_slow_stream_md5): this pins the reference side
    and the B slice header's direct_spatial_mv_pred bit, which changes from frame to frame here."""
    lib = hip_lib_host
    p = mux.encoder_params(lib, width=1920, height=1080, pre_scenecut=1, **M.SLOW)
    assert (p.level_idc, p.d_num_ref_frames) == (50, 5)
    a = M.reference_med(p, 1920, 1080, 24)
    assert len(set(int(a["frame_info2"][f][3]) for f in range(24) if int(a["frame_info"][f][0]) == 1)) == 2          # both direct modes occur
    assert hashlib.md5(M.mux_reference_stream(lib, p, a, 24)).hexdigest() == M.STREAM_MD5["C4_SLOW_pre_scenecut_hd24"]
    # without --pre-scenecut only the SEI text differs on this clip (no cut fires either way)
    p0 = mux.encoder_params(lib, width=1920, height=1080, **M.SLOW)
    assert hashlib.md5(M.mux_reference_stream(lib, p0, a, 24)).hexdigest() == M.STREAM_MD5["C4_SLOW_hd24"]
