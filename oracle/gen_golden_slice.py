"""TEST INFRASTRUCTURE -- golden vectors for the per-macroblock sweep, produced by the REFERENCE's
own x264_macroblock_cache_load / _analyse / _encode / _cache_save loop (oracle/ref_slice.c inside
oracle/_ref/libx264ref.so, built from the sources where they lie).  Runs only where
/root/reference exists; the .npz files under tests/golden/ are what travels.

    python -m oracle.gen_golden_slice
"""
import os
import sys

import numpy as np

from oracle import refslice as rs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# (name, size, frames, clip kind, parameters)
CASES = [
    ("uf", (208, 144), 4, "moving", dict(qp=30, subme=0)),
    ("uf_static", (200, 120), 4, "static", dict(qp=38, subme=1)),
    ("hex2", (200, 120), 4, "static", dict(qp=30, subme=2, me_method=rs.ME_HEX)),
    ("refs3", (208, 144), 5, "static", dict(qp=30, subme=5, me_method=rs.ME_HEX, n_refs=3, cabac=1, deblock=1)),
    ("intra_all", (208, 144), 4, "static", dict(qp=26, subme=4, me_method=rs.ME_HEX, n_refs=2, inter=0x3, intra=0x3,
                                                 transform8x8=1, cabac=1, deblock=1)),
    ("nodecimate", (200, 120), 4, "moving", dict(qp=34, subme=3, intra=0x1, inter=0x1, n_refs=2, deblock=1, dct_decimate=0,
                                                  fast_pskip=0)),
    ("parts", (200, 120), 4, "static", dict(qp=30, subme=2, me_method=rs.ME_HEX, inter=0x10, n_refs=2)),
    ("medium_ip", (208, 144), 5, "static", dict(qp=26, subme=5, me_method=rs.ME_HEX, n_refs=3, inter=0x13, intra=0x3, transform8x8=1,
                                                  mixed_refs=1, cabac=1, deblock=1)),
    ("umh", (208, 144), 4, "static", dict(qp=28, subme=5, me_method=rs.ME_UMH, n_refs=2, inter=0x13, intra=0x3, transform8x8=1,
                                            mixed_refs=1, cabac=1, deblock=1)),
    ("sub8x8", (208, 144), 4, "moving", dict(qp=22, subme=5, me_method=rs.ME_HEX, n_refs=2, inter=0x33, intra=0x3, transform8x8=1,
                                               mixed_refs=1, cabac=1, deblock=1)),
    ("sub8x8_umh", (200, 120), 4, "static", dict(qp=24, subme=3, me_method=rs.ME_UMH, inter=0x30, n_refs=2, deblock=1, chroma_me=0)),
    ("esa", (208, 144), 3, "moving", dict(qp=26, subme=5, me_method=rs.ME_ESA, me_range=12, n_refs=2, inter=0x33, intra=0x1, mixed_refs=1, cabac=1, deblock=1)),
    ("esa_fpel", (200, 120), 3, "static", dict(qp=30, subme=1, me_method=rs.ME_ESA, me_range=16, inter=0x10)),
    ("nr", (208, 144), 6, "moving", dict(qp=28, subme=5, me_method=rs.ME_HEX, n_refs=2, inter=0x13, intra=0x3, transform8x8=1, mixed_refs=1,
                                         cabac=1, deblock=1, noise_reduction=300)),
    ("jvt", (208, 144), 4, "moving", dict(qp=24, subme=5, me_method=rs.ME_HEX, n_refs=2, inter=0x33, intra=0x3, transform8x8=1, mixed_refs=1,
                                          cabac=1, deblock=1, cqm_preset=1)),
    ("umh_fpel", (200, 120), 4, "moving", dict(qp=32, subme=1, me_method=rs.ME_UMH, me_range=24, inter=0x10, n_refs=2, deblock=1)),
    # lossless (constant QP 0): predictive lossless intra prediction, zigzag residuals, SAD everywhere
    ("lossless", (208, 144), 4, "moving", dict(qp=0, subme=5, me_method=rs.ME_HEX, n_refs=2, inter=0x33, intra=0x3, transform8x8=1, mixed_refs=1,
                                               cabac=1, deblock=1)),
    ("lossless_cavlc", (176, 112), 3, "static", dict(qp=0, subme=2, me_method=rs.ME_UMH, n_refs=1, inter=0x11, intra=0x1, deblock=1)),
]


# round 2: chains through refslice_encode_chain2 -- the entropy writer in the loop (payload bytes in the fixture), the RD levels
# (subme 6 / 7), trellis, psy-rd, adaptive quantisation.  (name, size, frames, clip kind, parameters, ext parameters)
MED = dict(me_method=rs.ME_HEX, n_refs=3, inter=0x13, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1, deblock=1)
MEDB = dict(MED, inter=0x113, n_refs=2)       # + X264_ANALYSE_BSUB16x16
SLOW = dict(me_method=rs.ME_UMH, n_refs=5, inter=0x13, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1, deblock=1)
CASES2 = [
    ("w_medium_ip", (208, 144), 4, "static", dict(qp=26, subme=5, **MED), dict()),                          # the round-1 medium-like chain, now with its payload
    ("rd6", (208, 144), 4, "moving", dict(qp=28, subme=6, **MED), dict()),
    ("rd7_psy", (208, 144), 4, "static", dict(qp=26, subme=7, **MED), dict(psy_rd=1.0)),
    ("rd7_trellis1", (208, 144), 5, "moving", dict(qp=24, subme=7, **MED), dict(trellis=1, psy_rd=1.0)),      # the medium preset's analysis options at constant QP
    ("rd7_trellis2", (200, 120), 4, "static", dict(qp=30, subme=7, **MED), dict(trellis=2, psy_rd=0.0)),      # psy-rd off: the I_PCM decision is live
    ("rd7_aq", (208, 144), 5, "moving", dict(qp=26, subme=7, **MED), dict(trellis=1, psy_rd=1.0, aq_mode=1, aq_strength=1.0)),   # per-macroblock QP
    ("rd7_umh_1ref", (200, 120), 4, "moving", dict(qp=34, subme=7, me_method=rs.ME_UMH, n_refs=1, inter=0x11, intra=0x1, cabac=1, deblock=1),
     dict(trellis=1, psy_rd=1.0, aq_mode=1, aq_strength=1.5)),
    ("t1_subme4", (200, 120), 4, "moving", dict(qp=30, subme=4, me_method=rs.ME_HEX, n_refs=2, inter=0x13, intra=0x3, transform8x8=1, cabac=1, deblock=1),
     dict(trellis=1)),                                                                                       # trellis without the RD levels: every intra block is coded again
    ("rd7_lowqp", (96, 80), 3, "moving", dict(qp=8, subme=7, **MED), dict(trellis=2, psy_rd=0.0)),            # long levels: the escape codes of the level coding
    # B slices (coding order in the fixture: I P B B B P B B ...): the medium preset's analysis options with 3 disposable B frames
    ("b_medium", (208, 144), 8, "moving", dict(qp=26, subme=7, **MEDB), dict(trellis=1, psy_rd=1.0, aq_mode=1, bframes=3, weightb=1, direct_pred=1)),
    ("b_temporal", (200, 120), 7, "static", dict(qp=28, subme=6, **MEDB), dict(trellis=1, psy_rd=1.0, bframes=2, weightb=0, direct_pred=2)),   # B without the RD levels, P with them
    ("b_subme5", (208, 144), 6, "moving", dict(qp=30, subme=5, **MEDB), dict(bframes=2, weightb=1, direct_pred=1)),
    ("b_rd_2b", (200, 120), 7, "static", dict(qp=31, subme=7, me_method=rs.ME_HEX, n_refs=1, inter=0x113, intra=0x3, transform8x8=1, cabac=1, deblock=1),
     dict(trellis=2, psy_rd=0.0, bframes=2, weightb=0, direct_pred=1)),                                            # one reference, psy off: I_PCM live in B slices
    ("b_temporal_rd", (208, 144), 7, "moving", dict(qp=22, subme=7, keyint=6, **MEDB), dict(trellis=2, psy_rd=0.0, bframes=3, weightb=1, direct_pred=2)),   # co-located references outside list 0
    # the lookahead's vectors as candidates of the 16x16 searches (h->frames.b_have_lowres, R/common/macroblock.c:393-398); the vectors
    # themselves are stand-ins (oracle/refslice.py: lowres_vectors), some frames / lists marked "not searched"
    ("lowres_p", (208, 144), 5, "moving", dict(qp=27, subme=7, **MED), dict(trellis=1, psy_rd=1.0, aq_mode=1, lowres_seed=1)),
    ("lowres_b", (208, 144), 8, "moving", dict(qp=25, subme=7, **MEDB), dict(trellis=1, psy_rd=1.0, bframes=3, weightb=1, direct_pred=1, lowres_seed=2)),
    ("lowres_bt", (200, 120), 7, "static", dict(qp=29, subme=6, **MEDB), dict(trellis=1, bframes=2, weightb=0, direct_pred=2, lowres_seed=3)),
    # round 3: the RD refinement of subme 8-9 (x264_me_refine_qpel_rd, x264_intra_rd_refine).  SLOW = BASELINE's "slow" flag set at
    # constant QP: --ref 5 --me umh --subme 8 --8x8dct --partitions p8x8,b8x8,i8x8,i4x4 --trellis 1 --weightb --mixed-refs --direct spatial
    ("rd8_slow_ip", (208, 144), 6, "moving", dict(qp=26, subme=8, **SLOW), dict(trellis=1, psy_rd=1.0, aq_mode=1)),
    ("rd8_t2_psy0", (200, 120), 4, "static", dict(qp=30, subme=8, **MED), dict(trellis=2, psy_rd=0.0)),
    ("rd9_ip", (208, 144), 4, "moving", dict(qp=22, subme=9, **MED), dict(trellis=1, psy_rd=1.0)),
    ("rd8_b_slow", (208, 144), 8, "moving", dict(qp=26, subme=8, **dict(SLOW, inter=0x113)), dict(trellis=1, psy_rd=1.0, aq_mode=1, bframes=3, weightb=1, direct_pred=1)),
]
# chains that pin the TWIN only (the kernel refuses them and says so): B slices at subme 9 (x264_me_refine_bidir_rd and the list-1 forms of
# x264_me_refine_qpel_rd), sub-8x8 partitions under the RD levels (x264_rd_cost_subpart, x264_macroblock_encode_p4x4)
CASES2_TWIN = [
    ("rd9_b", (208, 144), 7, "moving", dict(qp=27, subme=9, **MEDB), dict(trellis=1, psy_rd=1.0, bframes=2, weightb=1, direct_pred=1)),
    ("rd9_b_temporal", (200, 120), 7, "static", dict(qp=30, subme=9, **MEDB), dict(trellis=2, psy_rd=0.0, bframes=3, weightb=0, direct_pred=2)),
    ("rd8_sub8x8", (208, 144), 4, "moving", dict(qp=24, subme=8, **dict(MED, inter=0x33)), dict(trellis=1, psy_rd=1.0)),
    ("rd9_sub8x8_nodec", (96, 120), 4, "static", dict(qp=21, subme=9, me_method=rs.ME_HEX, me_range=8, n_refs=2, inter=0x33, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1,
                                                       deblock=1, dct_decimate=0, chroma_me=0, keyint=4, chroma_qp_offset=-3), dict(trellis=1, psy_rd=1.0)),
]


def static_clip(w, h, n):
    """Synthetic clip whose background is frozen at frame 0 outside a moving window (P_SKIP territory)."""
    y, u, v = rs.clip(w, h, n)
    for t in range(1, n):
        x0, y0 = ((16 + 24 * t) % max(w - 96, 1)) & ~1, ((8 + 16 * t) % max(h - 64, 1)) & ~1
        by, bu, bv = y[0].copy(), u[0].copy(), v[0].copy()
        by[y0:y0 + 64, x0:x0 + 96] = y[t][y0:y0 + 64, x0:x0 + 96]
        bu[y0 // 2:y0 // 2 + 32, x0 // 2:x0 // 2 + 48] = u[t][y0 // 2:y0 // 2 + 32, x0 // 2:x0 // 2 + 48]
        bv[y0 // 2:y0 // 2 + 32, x0 // 2:x0 // 2 + 48] = v[t][y0 // 2:y0 // 2 + 32, x0 // 2:x0 // 2 + 48]
        y[t], u[t], v[t] = by, bu, bv
    return y, u, v


def case_inputs(size, frames, kind):
    return (static_clip if kind == "static" else rs.clip)(size[0], size[1], frames)


def masked(a):
    """mvr of skipped macroblocks / unused references is never read by the reference (undefined): zero it."""
    a = dict(a)
    skip = (a["mb_type"] == rs.P_SKIP) | (a["mb_type"] == rs.B_SKIP)
    m = np.broadcast_to(skip[:, None, :, None], a["mvr"].shape) | \
        (np.arange(a["mvr"].shape[1])[None, :, None, None] >= a["frame_info"][:, 2][:, None, None, None])
    a["mvr"] = np.where(m, 0, a["mvr"]).astype(np.int16)
    return a


def masked2(a):
    a = masked(a)
    a.pop("mb_bits", None)                   # a debugging aid of the harness, not part of the fixture
    return a


def main():
    only = sys.argv[1:]                      # optional: regenerate just the named chains
    for name, size, frames, kind, kw, ekw in CASES2 + CASES2_TWIN:
        if only and name not in only:
            continue
        p = rs.make_params(size[0], size[1], frames, **kw)
        y, u, v = case_inputs(size, frames, kind)
        a = masked2(rs.run_reference2(p, rs.make_ext(**ekw), y, u, v))
        path = os.path.join(GOLDEN, "slice2_%s.npz" % name)
        np.savez_compressed(path, **a)
        types = [np.bincount(a["mb_type"][f], minlength=7).tolist() for f in range(frames)]
        print("%s: %d bytes, payload %s, types per frame (I4 I8 I16 PCM P P8 skip) %s" % (path, os.path.getsize(path), a["payload_len"].tolist(), types))
    for name, size, frames, kind, kw in CASES:
        if only and name not in only:
            continue
        p = rs.make_params(size[0], size[1], frames, **kw)
        y, u, v = case_inputs(size, frames, kind)
        a = masked(rs.run_reference(p, y, u, v))
        path = os.path.join(GOLDEN, "slice_%s.npz" % name)
        np.savez_compressed(path, **a)
        types = [np.bincount(a["mb_type"][f], minlength=7)[[0, 1, 2, 4, 6]].tolist() for f in range(frames)]
        print("%s: %d bytes, types per frame (I4 I8 I16 P skip) %s" % (path, os.path.getsize(path), types))


if __name__ == "__main__":
    main()
