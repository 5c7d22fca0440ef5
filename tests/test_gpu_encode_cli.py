"""The command line front end end to end on the GPU (x264_vs2008_amd/encode.py: reader -> validated parameters -> encoder -> Annex B file):
BASELINE config 1 from a raw file and from a YUV4MPEG2 file = the reference command line's md5; two inputs coded side by side with the MED
flag set = what the reference's whole encoder (the harness) + this library's headers give for each."""
import hashlib
import os

import pytest

import mux_cases as M
from x264_vs2008_amd import encode as E
from x264_vs2008_amd import mux, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libx264ref.so")
UF = "--qp 26 --no-cabac --me dia --subme 0 --partitions none --no-deblock --aq-mode 0 --scenecut -1 --ref 1 --bframes 0 --b-adapt 0 --no-asm --threads 1"
MED = "--crf 23 --ref 3 --bframes 3 --b-adapt 1 --me hex --subme 7 --8x8dct --partitions p8x8,b8x8,i8x8,i4x4 --trellis 1 --weightb --mixed-refs --direct spatial"


def write_clip(path, w, h, n, t0=0, y4m=False):
    with open(path, "wb") as f:
        if y4m:
            f.write(b"YUV4MPEG2 W%d H%d F25:1 Ip A0:0 C420jpeg\n" % (w, h))
        for t in range(n):
            if y4m:
                f.write(b"FRAME\n")
            for pl in synth.frame(w, h, t0 + t):
                f.write(pl.tobytes())


@pytest.mark.parametrize("y4m", [False, True])
def test_cli_config1_file_has_the_reference_md5(hip_lib, tmp_path, y4m):
    src = str(tmp_path / ("cif30.y4m" if y4m else "cif30.yuv"))
    out = str(tmp_path / "out.264")
    write_clip(src, 352, 288, 30, y4m=y4m)
    assert E.main(UF.split() + ["-o", out, src] + ([] if y4m else ["352x288"])) == 0
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == M.STREAM_MD5["C1_UF_cif30"]


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libx264ref.so not built (needs /root/reference)")
def test_cli_two_inputs_side_by_side_equal_the_reference_encoder(hip_lib, tmp_path):
    w, h, n = 176, 96, 14
    srcs = [str(tmp_path / ("in%d.y4m" % i)) for i in range(2)]
    for i, s in enumerate(srcs):
        write_clip(s, w, h, n, t0=40 * i, y4m=True)
    assert E.main(MED.split() + ["-o", str(tmp_path / "o%d.264")] + srcs) == 0
    o = E.build_parser().parse_args(MED.split() + ["-o", "x"] + srcs)
    p = mux.encoder_params(hip_lib, width=w, height=h, fps_num=25, fps_den=1, **E.param_fields(o))
    for i in range(2):
        a = M.reference_med(p, w, h, n, t0=40 * i)
        want = M.mux_reference_stream(hip_lib, p, a, n)
        got = open(str(tmp_path / ("o%d.264" % i)), "rb").read()
        assert got == want, "stream %d: %d vs %d bytes" % (i, len(got), len(want))


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libx264ref.so not built (needs /root/reference)")
def test_cli_slow_flag_set_with_direct_auto(hip_lib, tmp_path):
    """BASELINE's SLOW flag set through the command line on a small clip: b-adapt 2, UMH, subme 8, five references, --direct auto (the B slice
    headers' direct_spatial_mv_pred bit follows the running scores), the post-encode scene cut -- the file equals the reference encoder's + headers."""
    w, h, n = 176, 96, 16
    src, out = str(tmp_path / "in.y4m"), str(tmp_path / "o.264")
    write_clip(src, w, h, n, t0=17, y4m=True)
    args = "--crf 23 --ref 5 --bframes 3 --b-adapt 2 --me umh --subme 8 --8x8dct --partitions p8x8,b8x8,i8x8,i4x4 --trellis 1 --weightb --mixed-refs --direct auto"
    assert E.main(args.split() + ["-o", out, src]) == 0
    o = E.build_parser().parse_args(args.split() + ["-o", "x", src])
    p = mux.encoder_params(hip_lib, width=w, height=h, fps_num=25, fps_den=1, **E.param_fields(o))
    a = M.reference_med(p, w, h, n, t0=17)
    want = M.mux_reference_stream(hip_lib, p, a, n)
    got = open(out, "rb").read()
    assert got == want, "%d vs %d bytes" % (len(got), len(want))


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libx264ref.so not built (needs /root/reference)")
def test_cli_constant_qp_cabac_without_a_frame_queue(hip_lib, tmp_path):
    """--qp N --scenecut -1 --bframes 0: nothing for x264_slicetype_decide to decide -- the lock-step encoder (raster variant, CABAC in the loop,
    an IDR every --keyint frames) against the reference's loop + this library's headers; trellis, deblock offsets and a second IDR included."""
    from oracle import refslice as rs
    w, h, n = 160, 112, 9
    src, out = str(tmp_path / "in_160x112.yuv"), str(tmp_path / "o.264")
    write_clip(src, w, h, n, t0=5)
    args = "--qp 29 --scenecut -1 --bframes 0 --ref 2 --subme 5 --8x8dct --trellis 1 --deblock=-1:1 --keyint 6 --min-keyint 2 --mixed-refs --aq-mode 0"
    assert E.main(args.split() + ["-o", out, src]) == 0
    o = E.build_parser().parse_args(args.split() + ["-o", "x", src])
    p = mux.encoder_params(hip_lib, width=w, height=h, **E.param_fields(o))
    y, u, v = rs.clip(w, h, n, 5)
    a = rs.run_reference2(rs.make_params(w, h, n, qp=29, me_method=rs.ME_HEX, subme=5, n_refs=2, inter=p.inter, intra=p.intra, transform8x8=1, cabac=1, deblock=1,
                                         alpha_c0=-1, beta=1, keyint=6, mixed_refs=1, chroma_me=1, mv_range=p.mv_range),
                          rs.make_ext(write=1, trellis=1), y, u, v)
    m, want = mux.AnnexB(hip_lib, p), b""
    for t in range(n):
        want += m.frame(frame=t, ftype=mux.TYPE_IDR if t % 6 == 0 else mux.TYPE_P, qp=int(a["frame_info"][t][1]), payload=bytes(a["payload"][t, :a["payload_len"][t]]))
    got = open(out, "rb").read()
    assert got == want, "%d vs %d bytes" % (len(got), len(want))
