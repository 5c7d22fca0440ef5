"""BASELINE's configurations as the stream writer's tests use them: the flag sets of BASELINE.md 2 as encoder parameters, the md5 of the
reference CLI's .264 (--threads 1), and the reference-side runs (oracle/_ref) that give the payloads."""
import hashlib

from x264_vs2008_amd import mux, synth

CLIP_MD5 = {"cif30": "a5ce9660cf16d66830d27bbfbf69545a", "hd24": "0c526ccaa572e26ba82832ba43d303d1", "uhd8": "f8f0fa654f79120b17dc2dfb11956e98"}
STREAM_MD5 = {"C1_UF_cif30": "02b208eecef842e084cbb9c83bc1a757", "C2_MED_hd24": "b0d54145534a5158d08c9f5fc06db5da",
              "C3_MED_umh_uhd8": "81eb4c4af687d83cc5d9ffb190f5b52c",
              "C4_SLOW_hd24": "77215ac697eac797265f81fc43fd007a", "C4_SLOW_pre_scenecut_hd24": "f0c9914d4e8f730b776206dd50aceb75"}

# UF  = --qp 26 --no-cabac --me dia --subme 0 --partitions none --no-deblock --aq-mode 0 --scenecut -1 --ref 1 --bframes 0 --b-adapt 0
#       (`--partitions none` clears param.analyse.inter only: analyse.intra keeps I4x4, I8x8 goes with 8x8dct off, encoder.c:483-487)
UF = dict(rc_method=mux.RC_CQP, qp_constant=26, cabac=0, me_method=0, subpel_refine=0, inter=0, deblocking_filter=0, aq_mode=0, scenecut_threshold=-1,
          frame_reference=1, bframe=0, bframe_adaptive=0)
# MED = --crf 23 --ref 3 --bframes 3 --b-adapt 1 --me hex --subme 7 --8x8dct --partitions p8x8,b8x8,i8x8,i4x4 --trellis 1 --weightb --mixed-refs --direct spatial
MED = dict(rc_method=mux.RC_CRF, rf_constant=23.0, frame_reference=3, bframe=3, bframe_adaptive=1, me_method=1, subpel_refine=7, transform_8x8=1, inter=0x113, intra=3,
           trellis=1, weighted_bipred=1, mixed_references=1, direct_mv_pred=1)
# SLOW = --crf 23 --ref 5 --bframes 3 --b-adapt 2 --me umh --subme 8 --8x8dct --partitions p8x8,b8x8,i8x8,i4x4 --trellis 1 --weightb --mixed-refs --direct auto
SLOW = dict(rc_method=mux.RC_CRF, rf_constant=23.0, frame_reference=5, bframe=3, bframe_adaptive=2, me_method=2, subpel_refine=8, transform_8x8=1, inter=0x113, intra=3,
            trellis=1, weighted_bipred=1, mixed_references=1, direct_mv_pred=3)


def clip_md5(w, h, n):
    m = hashlib.md5()
    for t in range(n):
        for pl in synth.frame(w, h, t):
            m.update(pl.tobytes())
    return m.hexdigest()


def reference_uf(w, h, n):
    from oracle import refslice as rs
    y, u, v = rs.clip(w, h, n)
    return rs.run_reference2(rs.make_params(w, h, n, qp=26, me_method=rs.ME_DIA, subme=0, n_refs=1, inter=0, intra=1, cabac=0, deblock=0, chroma_me=1, keyint=250,
                                            mv_range=128), rs.make_ext(write=1), y, u, v)


def reference_med(p, w, h, n, t0=0):
    """The reference's whole encoder (oracle/ref_slice.c refslice_encode_stream) with the validated parameters p -- the post-encode scene cut
    included (BASELINE's runs have it: no --pre-scenecut; it never fires on those clips)."""
    from oracle import refslice as rs
    y, u, v = rs.clip(w, h, n, t0)
    rp = rs.make_params(w, h, n, qp=p.qp_constant, me_method=p.me_method, me_range=p.me_range, subme=p.subpel_refine, n_refs=p.frame_reference, inter=p.inter,
                        intra=p.intra, transform8x8=p.transform_8x8, cabac=p.cabac, deblock=p.deblocking_filter, keyint=p.keyint_max, mixed_refs=p.mixed_references,
                        chroma_me=p.chroma_me, mv_range=p.mv_range)
    e = rs.make_ext(bframes=p.bframe, b_adapt=p.bframe_adaptive, pre_scenecut=p.pre_scenecut, scenecut_threshold=p.scenecut_threshold, keyint_min=p.keyint_min, crf=p.rf_constant,
                    bframe_bias=p.bframe_bias, weightb=p.weighted_bipred, aq_mode=p.aq_mode, aq_strength=p.aq_strength, trellis=p.trellis, psy_rd=p.psy_rd,
                    direct_pred=p.direct_mv_pred)
    return rs.run_reference_stream(rp, e, y, u, v)


def mux_reference_stream(lib, p, a, n):
    """The Annex B stream around the harness output `a` (coded order: input number, slice type, QP, POC, payload)."""
    m, out = mux.AnnexB(lib, p), []
    for f in range(n):
        st, qp, _, poc = (int(x) for x in a["frame_info"][f])
        ft = (mux.TYPE_IDR if poc == 0 else mux.TYPE_I) if st == 2 else mux.TYPE_P if st == 0 else mux.TYPE_B
        out.append(m.frame(frame=int(a["frame_info2"][f][0]), ftype=ft, qp=qp, payload=bytes(a["payload"][f, :a["payload_len"][f]]),
                           direct_spatial=int(a["frame_info2"][f][3]) if st == 1 else 1))       # (B slices: sh.b_direct_spatial_mv_pred as the harness recorded it)
    return b"".join(out)
