"""The whole Annex B stream (x264_vs2008_amd/mux.py: version SEI, SPS, PPS, slice headers around the sweep's payloads) against the md5 of
the .264 file the REFERENCE's command line wrote for BASELINE's configurations -- the known-answer values of SURVEY.md 8(c), produced in
the survey from the real `x264 --no-asm --threads 1 <flags>` on the integer synthetic clips of SURVEY.md 8(d) (whose own md5s are checked
here first).  This is the one pin the header writers have: R/encoder/set.c and encoder.c cannot be built here (they need the
configure-generated config.h), so there is no live comparison; an md5 covers every byte -- parameter sets, SEI text, every slice header,
every payload, the emulation prevention."""
import hashlib

import numpy as np
import pytest

from x264_vs2008_amd import mux, slice as sl, synth
from x264_vs2008_amd.frame import cqm_init

pytestmark = pytest.mark.gpu

import mux_cases as M

CLIP_MD5, STREAM_MD5 = M.CLIP_MD5, M.STREAM_MD5          # SURVEY.md 8(c) / BASELINE.md 2: md5 of the reference CLI's output, --threads 1


def clip(w, h, n):
    fr = [synth.frame(w, h, t) for t in range(n)]
    m = hashlib.md5()
    for y, u, v in fr:
        m.update(y.tobytes()); m.update(u.tobytes()); m.update(v.tobytes())
    return fr, m.hexdigest()


def uf_stream(hip_lib, frames, w=352, h=288):
    """BASELINE's UF flag set (mux_cases.UF) through the product: ChainEncoder (wavefront variant + the CAVLC pass) + the muxer."""
    p = mux.encoder_params(hip_lib, width=w, height=h, **M.UF)
    assert (p.level_idc, p.mv_range, p.intra, p.d_profile_idc) == (13, 128, 1, 66)
    enc = sl.ChainEncoder(hip_lib, w, h, cqm_init(hip_lib), qp=p.qp_constant, me_method=p.me_method, me_range=p.me_range, subme=p.subpel_refine,
                          n_refs=p.frame_reference, inter=p.inter, intra=p.intra, transform8x8=p.transform_8x8, fast_pskip=p.fast_pskip,
                          dct_decimate=p.dct_decimate, chroma_me=p.chroma_me, cabac=0, deblock=p.deblocking_filter, keyint=p.keyint_max,
                          mv_range=p.mv_range, write=1)
    assert enc.cavlc
    m, out = mux.AnnexB(hip_lib, p), []
    try:
        for t, (y, u, v) in enumerate(frames):
            enc.upload(y, u, v)
            stype, qp, _ = enc.encode_frame()
            enc.status()
            pay = enc.payloads()[0]
            enc.finish_frame()
            out.append(m.frame(frame=t, ftype=mux.TYPE_IDR if stype == sl.SLICE_I else mux.TYPE_P, qp=qp, payload=pay))
    finally:
        enc.close()
    return out


def test_c1_uf_cif30_stream_md5(hip_lib):
    frames, md5 = clip(352, 288, 30)
    assert md5 == CLIP_MD5["cif30"]
    nals = uf_stream(hip_lib, frames)
    stream = b"".join(nals)
    got = hashlib.md5(stream).hexdigest()
    assert got == STREAM_MD5["C1_UF_cif30"], "%d bytes, frames %s, head %s" % (len(stream), [len(n) for n in nals], stream[:160].hex())


def med_stream(hip_lib, p, w, h, n, pre_scenecut=0):
    """BASELINE's MED flag set through the product: StreamEncoder (lookahead, b-adapt 1, CRF, the sweep with the entropy coder) + the muxer.
    pre_scenecut = 0 is the CLI's default: the post-encode scene cut is evaluated after every P frame (it must not fire: the re-encode is not built)."""
    from x264_vs2008_amd.stream import StreamEncoder
    enc = StreamEncoder(hip_lib, w, h, cqm_init(hip_lib), batch=1, n_frames=n, crf=p.rf_constant, b_adapt=p.bframe_adaptive, bframe_bias=p.bframe_bias,
                        keyint_min=p.keyint_min, scenecut_threshold=p.scenecut_threshold, pre_scenecut=pre_scenecut, ip_factor=p.ip_factor, pb_factor=p.pb_factor,
                        qcompress=p.qcompress, qp_step=p.qp_step, qp=p.qp_constant, me_method=p.me_method, me_range=p.me_range, subme=p.subpel_refine,
                        n_refs=p.frame_reference, inter=p.inter, intra=p.intra, transform8x8=p.transform_8x8, cabac=1, deblock=p.deblocking_filter,
                        alpha_c0=p.deblocking_filter_alphac0, beta=p.deblocking_filter_beta, keyint=p.keyint_max, mixed_refs=p.mixed_references, chroma_me=p.chroma_me,
                        trellis=p.trellis, psy_rd=p.psy_rd, aq_mode=p.aq_mode, aq_strength=p.aq_strength, bframes=p.bframe, weightb=p.weighted_bipred,
                        direct_pred=p.direct_mv_pred, qp_min=p.qp_min, qp_max=p.qp_max, mv_range=p.mv_range, fast_pskip=p.fast_pskip, dct_decimate=p.dct_decimate)
    m, out, order = mux.AnnexB(hip_lib, p), [], []

    def fill(pic, f):
        y, u, v = synth.frame(w, h, f)
        enc.src_ctx.upload(pic, y, u, v, b=0)

    try:
        idle = 0
        while idle < 2 and len(out) < n:
            coded = enc.step(fill)
            idle = 0 if coded else idle + bool(enc.flushing)
            if coded:
                enc.sync()
                enc.status()
                cd = coded[0]
                out.append(m.frame(frame=cd.frame, ftype=cd.type, qp=cd.qp, payload=enc.payloads()[0], n_ref0=cd.n_ref0, n_ref1=cd.n_ref1, frame_num_reset=cd.frame_num_reset,
                                   direct_spatial=cd.direct_spatial))
                order.append((cd.frame, cd.type, cd.qp))
    finally:
        enc.close()
    return out, order


@pytest.mark.parametrize("cfg", ["C2_MED_hd24", "C3_MED_umh_uhd8"])
def test_med_stream_md5(hip_lib, cfg):
    """BASELINE configs 2 (the metric's) and 3, as BASELINE.md states them -- no --pre-scenecut: every byte of the product's stream equals the
    reference command line's."""
    import mux_cases as M
    w, h, n, kw = (1920, 1080, 24, M.MED) if cfg == "C2_MED_hd24" else (3840, 2160, 8, dict(M.MED, me_method=2))
    p = mux.encoder_params(hip_lib, width=w, height=h, **kw)
    nals, order = med_stream(hip_lib, p, w, h, n)
    stream = b"".join(nals)
    assert len(nals) == n, order
    assert hashlib.md5(stream).hexdigest() == M.STREAM_MD5[cfg], "%d bytes, coded %s" % (len(stream), order)


def test_c4_slow_stream_md5(hip_lib):
    """BASELINE config 4's flag set as stated (SLOW: --ref 5 --b-adapt 2 --me umh --subme 8 --direct auto, the post-encode scene cut) on hd24:
    the product's stream has the md5 of the reference command line's file.  --direct auto: every B macroblock predicts both direct modes, the
    running skip scores pick each B frame's mode (temporal for the first, spatial later on this clip)."""
    import mux_cases as M
    p = mux.encoder_params(hip_lib, width=1920, height=1080, **M.SLOW)
    nals, order = med_stream(hip_lib, p, 1920, 1080, 24)
    stream = b"".join(nals)
    assert len(nals) == 24, order
    assert hashlib.md5(stream).hexdigest() == M.STREAM_MD5["C4_SLOW_hd24"], "%d bytes, coded %s" % (len(stream), order)
