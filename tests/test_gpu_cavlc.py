"""The CAVLC writer (x264hip_cavlc_write_frame: x264_macroblock_write_cavlc + the skip runs of x264_slice_write as a pass over the state the
wavefront variant leaves) against the REFERENCE's own writer run inside its per-macroblock loop (oracle/ref_slice.c refslice_encode_chain2
with cabac = 0): the slice_data() bytes of every frame of I / P chains -- BASELINE config 0's shape (352x288, dia, subme 0, no partitions)
and richer ones (all P partitions incl. sub-8x8, several references, 8x8 transform with its coefficient interleave, intra-heavy low QP).

  * golden: tests/golden/cavlc_*.npz made by oracle/gen_golden_cavlc.py from the reference;
  * live: the same configurations on other clips where oracle/_ref/libx264ref.so is built."""
import os

import numpy as np
import pytest

from oracle import refslice as rs
from x264_vs2008_amd import slice as sl
from x264_vs2008_amd.frame import cqm_init

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libx264ref.so")

CONFIGS = {
    "uf_cif": dict(w=352, h=288, n=6, kw=dict(qp=26, me_method=rs.ME_DIA, subme=0, n_refs=1, inter=0, intra=0, cabac=0, deblock=0, chroma_me=1)),
    "p_all_partitions": dict(w=208, h=144, n=6, kw=dict(qp=27, me_method=rs.ME_HEX, subme=5, n_refs=3, inter=0x33, intra=0x3, transform8x8=1, mixed_refs=1,
                                                       cabac=0, deblock=1)),
    "p_no_sub8x8_umh": dict(w=176, h=144, n=5, kw=dict(qp=31, me_method=rs.ME_UMH, subme=4, n_refs=2, inter=0x13, intra=0x3, transform8x8=1, cabac=0, deblock=1)),
    "intra_low_qp": dict(w=144, h=112, n=4, kw=dict(qp=12, me_method=rs.ME_HEX, subme=2, n_refs=1, inter=0x11, intra=0x3, transform8x8=1, cabac=0, deblock=1, keyint=2)),
    "high_qp_skips": dict(w=192, h=128, n=6, kw=dict(qp=40, me_method=rs.ME_HEX, subme=3, n_refs=2, inter=0x11, intra=0x1, cabac=0, deblock=1)),
    # adaptive quantisation: the raster variant without its writer leaves each macroblock's final QP (x264_macroblock_cache_save's rules and
    # cavlc_qp_delta's empty-I_16x16 one) in the state, the pass codes mb_qp_delta from it
    "aq_p_partitions": dict(w=208, h=144, n=6, ext=dict(aq_mode=1, aq_strength=1.0),
                            kw=dict(qp=28, me_method=rs.ME_HEX, subme=5, n_refs=2, inter=0x33, intra=0x3, transform8x8=1, mixed_refs=1, cabac=0, deblock=1)),
    "aq_strong_high_qp": dict(w=192, h=128, n=6, ext=dict(aq_mode=1, aq_strength=1.8),
                              kw=dict(qp=38, me_method=rs.ME_UMH, subme=3, n_refs=1, inter=0x11, intra=0x3, cabac=0, deblock=1, keyint=4)),
    "aq_intra_qp_clip": dict(w=144, h=112, n=4, ext=dict(aq_mode=1, aq_strength=2.5),
                             kw=dict(qp=47, me_method=rs.ME_HEX, subme=1, n_refs=1, inter=0x11, intra=0x3, transform8x8=1, cabac=0, deblock=1, keyint=2)),
}


def reference(c, t0=0):
    y, u, v = rs.clip(c["w"], c["h"], c["n"], t0)
    a = rs.run_reference2(rs.make_params(c["w"], c["h"], c["n"], **c["kw"]), rs.make_ext(write=1, **c.get("ext", {})), y, u, v)
    return (y, u, v), [bytes(a["payload"][f, :a["payload_len"][f]]) for f in range(c["n"])], a


def encode(hip_lib, c, clip):
    y, u, v = clip
    enc = sl.ChainEncoder(hip_lib, c["w"], c["h"], cqm_init(hip_lib), write=1, **c["kw"], **c.get("ext", {}))
    assert enc.cavlc and enc.raster == ("ext" in c)
    out = []
    try:
        for f in range(c["n"]):
            enc.upload(y[f], u[f], v[f])
            enc.encode_frame()
            enc.status()
            out.append(enc.payloads()[0])
            enc.finish_frame()
    finally:
        enc.close()
    return out


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_cavlc_payload_equals_reference_fixture(hip_lib, name):
    c = CONFIGS[name]
    gold = np.load(os.path.join(ROOT, "tests", "golden", "cavlc_%s.npz" % name))
    got = encode(hip_lib, c, rs.clip(c["w"], c["h"], c["n"], 0))
    for f in range(c["n"]):
        want = bytes(gold["payload"][f, :gold["payload_len"][f]])
        assert got[f] == want, "%s frame %d: CAVLC payload differs (%d vs %d bytes)" % (name, f, len(got[f]), len(want))


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libx264ref.so not built (needs /root/reference)")
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_cavlc_payload_equals_reference_live(hip_lib, name):
    c = CONFIGS[name]
    clip, want, _ = reference(c, t0=37)
    got = encode(hip_lib, c, clip)
    for f in range(c["n"]):
        assert got[f] == want[f], "%s frame %d: CAVLC payload differs (%d vs %d bytes)" % (name, f, len(got[f]), len(want[f]))


def test_cavlc_batch_of_chains(hip_lib):
    """Several chains per launch: every chain's payload is the single-chain one."""
    c = CONFIGS["p_no_sub8x8_umh"]
    clips = [rs.clip(c["w"], c["h"], c["n"], t0) for t0 in (0, 11, 23)]
    single = [encode(hip_lib, c, cl) for cl in clips]
    enc = sl.ChainEncoder(hip_lib, c["w"], c["h"], cqm_init(hip_lib), batch=3, write=1, **c["kw"])
    try:
        for f in range(c["n"]):
            for b, (y, u, v) in enumerate(clips):
                enc.upload(y[f], u[f], v[f], b=b)
            enc.encode_frame()
            enc.status()
            pay = enc.payloads()
            for b in range(3):
                assert pay[b] == single[b][f], "chain %d frame %d" % (b, f)
            enc.finish_frame()
    finally:
        enc.close()
