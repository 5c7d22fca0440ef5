"""The reference's command line for what this library builds: raw I420 / YUV4MPEG2 in, Annex B `.264` out, on the GPU.

    python -m x264_vs2008_amd.encode --crf 23 --ref 3 --bframes 3 --b-adapt 1 --subme 7 --8x8dct --trellis 1 --weightb --mixed-refs -o out.264 in.y4m
    python -m x264_vs2008_amd.encode --qp 26 --no-cabac --me dia --subme 0 --partitions none --no-deblock --scenecut -1 -o out.264 in.yuv 352x288
    python -m x264_vs2008_amd.encode [options] -o out_%d.264 a.y4m b.y4m c.y4m        # one stream per input, coded side by side (same size and options)

Option names and meanings are `x264 --longhelp`'s (R/x264.c:386-487, x264_param_parse R/common/common.c:206-587; tests/test_cpu_encode_cli.py
parses the same argument lists with the reference's own x264_param_parse and compares x264_param2string).  The parameters go through
x264hip_validate_parameters (what x264_encoder_open does to them), the encoder runs with the validated values, and the output is what
x264_encoder_encode emits: version SEI, SPS, PPS, one slice NAL per frame in coding order (x264_vs2008_amd/mux.py).  For BASELINE's flag
sets the file is the reference command line's, byte for byte (tests/test_gpu_encode_cli.py).

A single stream is one serial chain of macroblocks -- ONE wavefront: a lone 1080p input codes at well under a frame per second.  The
product's throughput comes from thousands of streams in flight (bench.py); this front end is the functional drop-in, and several inputs
given at once are coded as chains of one batch.

Readers: raw I420 (R/muxers.c:63-122: frame i at i * w * h * 3 / 2) and YUV4MPEG2 (:124-316: W, H, F from the stream header, C420* only, every
FRAME header skipped to its newline).  Refused, never approximated: what the library refuses (x264hip_validate_parameters, the encoders' own
checks) -- ABR / VBV / 2-pass, --direct none, B frames or adaptive decisions with --no-cabac, interlaced, threads > 1, b-pyramid.
Scene cuts work as in the reference: by default after the fact (a P picture that should have been intra is coded again, encoder.c:1603-1699), with
--pre-scenecut in the lookahead."""
import argparse
import os
import re
import sys

import numpy as np

from . import mux

ME = {"dia": 0, "hex": 1, "umh": 2, "esa": 3, "tesa": 4}
DIRECT = {"none": 0, "spatial": 1, "temporal": 2, "auto": 3}
CQM = {"flat": 0, "jvt": 1}


def build_parser():
    ap = argparse.ArgumentParser(prog="x264_vs2008_amd.encode", description="x264 core 66's command line on MI355X (the built subset)", allow_abbrev=False)
    a = ap.add_argument
    a("inputs", nargs="+", help="raw I420 (.yuv, with WxH as the last argument or in the name) or YUV4MPEG2 (.y4m) files; several inputs = several streams side by side")
    a("-o", "--output", required=True, help="output .264; with several inputs a pattern with %%d")
    a("--frames", type=int, default=0, help="maximum number of frames to encode")
    a("--fps", default=None, help="float or rational (raw input; y4m carries its own)")
    a("-q", "--qp", type=int, default=None, help="constant quantiser")
    a("--crf", type=float, default=None, help="quality-based VBR (nominal QP)")
    a("-r", "--ref", type=int, default=1)
    a("-b", "--bframes", type=int, default=0)
    a("--b-adapt", type=int, default=1)
    a("--b-bias", type=int, default=0)
    a("-I", "--keyint", type=int, default=250)
    a("-i", "--min-keyint", type=int, default=25)
    a("--scenecut", type=int, default=40)
    a("--pre-scenecut", action="store_true")
    a("--no-cabac", action="store_true")
    a("--no-deblock", "--nf", action="store_true")
    a("-f", "--deblock", default=None, help="alpha:beta (negative alpha: write --deblock=-1:2)")
    a("-A", "--partitions", "--analyse", default=None, help="p8x8,p4x4,b8x8,i8x8,i4x4 / none / all")
    a("--direct", default="spatial", choices=sorted(DIRECT))
    a("-w", "--weightb", action="store_true")
    a("--me", default="hex", choices=sorted(ME))
    a("--merange", type=int, default=16)
    a("--mvrange", type=int, default=-1)
    a("-m", "--subme", type=int, default=6)
    a("--psy-rd", default="1.0:0.0", help="rd:trellis strengths")
    a("--mixed-refs", action="store_true")
    a("--no-chroma-me", action="store_true")
    a("--8x8dct", dest="dct8", action="store_true")       # (no -8: a digit option would make argparse read "--scenecut -1" as two options)
    a("-t", "--trellis", type=int, default=0)
    a("--no-fast-pskip", action="store_true")
    a("--no-dct-decimate", action="store_true")
    a("--nr", type=int, default=0)
    a("--deadzone-inter", type=int, default=21)
    a("--deadzone-intra", type=int, default=11)
    a("--cqm", default="flat", choices=sorted(CQM))
    a("--chroma-qp-offset", type=int, default=0)
    a("--ipratio", type=float, default=1.4)
    a("--pbratio", type=float, default=1.3)
    a("--qcomp", type=float, default=0.6)
    a("--qpmin", type=int, default=10)
    a("--qpmax", type=int, default=51)
    a("--qpstep", type=int, default=4)
    a("--aq-mode", type=int, default=1)
    a("--aq-strength", type=float, default=1.0)
    a("--level", default=None, help="4.1 / 41 ...")
    a("--threads", type=int, default=1)
    a("--no-asm", action="store_true", help="accepted and ignored (there is no asm here)")
    a("--no-psnr", action="store_true", help="accepted and ignored")
    a("--no-ssim", action="store_true", help="accepted and ignored")
    a("--device", type=int, default=0, help="GPU to run on")
    return ap


def param_fields(o):
    """The command line as x264_param_t fields (x264hip_encoder_params' names): what x264_param_parse does with each option."""
    k = dict(frame_reference=o.ref, bframe=o.bframes, bframe_adaptive=o.b_adapt, bframe_bias=o.b_bias, keyint_max=o.keyint, keyint_min=o.min_keyint,
             scenecut_threshold=o.scenecut, pre_scenecut=int(o.pre_scenecut), cabac=int(not o.no_cabac), deblocking_filter=int(not o.no_deblock),
             direct_mv_pred=DIRECT[o.direct], weighted_bipred=int(o.weightb), me_method=ME[o.me], me_range=o.merange, mv_range=o.mvrange, subpel_refine=o.subme,
             mixed_references=int(o.mixed_refs), chroma_me=int(not o.no_chroma_me), transform_8x8=int(o.dct8), trellis=o.trellis, fast_pskip=int(not o.no_fast_pskip),
             dct_decimate=int(not o.no_dct_decimate), noise_reduction=o.nr, luma_deadzone=(o.deadzone_inter, o.deadzone_intra), cqm_preset=CQM[o.cqm],
             chroma_qp_offset=o.chroma_qp_offset, ip_factor=o.ipratio, pb_factor=o.pbratio, qcompress=o.qcomp, qp_min=o.qpmin, qp_max=o.qpmax, qp_step=o.qpstep,
             aq_mode=o.aq_mode, aq_strength=o.aq_strength, threads=o.threads)
    if o.deblock is not None:                      # "deblock" / "filter": alpha[:beta], a lone number gives both (common.c:336-347)
        parts = re.split("[:,]", o.deblock)
        k["deblocking_filter_alphac0"] = int(parts[0])
        k["deblocking_filter_beta"] = int(parts[1]) if len(parts) > 1 else int(parts[0])
        k["deblocking_filter"] = 1
    pr = re.split("[:,]", o.psy_rd)
    k["psy_rd"] = float(pr[0])
    k["psy_trellis"] = float(pr[1]) if len(pr) > 1 else 0.0
    if o.partitions is not None:                   # common.c:398-418: the option rewrites analyse.inter only
        v, inter = o.partitions, 0
        if "all" in v:
            inter = 0xffffffff                     # ~0; x264_validate_parameters masks it
        for name, bit in (("i4x4", 0x1), ("i8x8", 0x2), ("p8x8", 0x10), ("p4x4", 0x20), ("b8x8", 0x100)):
            if name in v:
                inter |= bit
        k["inter"] = inter
    if o.qp is not None:                           # common.c:446-451 / 440-445: the last of --qp / --crf given wins; both given: CRF here as in the usual order
        k.update(rc_method=mux.RC_CQP, qp_constant=o.qp)
    if o.crf is not None:
        k.update(rc_method=mux.RC_CRF, rf_constant=o.crf)
    if o.level is not None:                        # common.c:262-268: "4.1" -> 41, small integers are tenths
        lv = o.level
        k["level_idc"] = int(10 * float(lv) + .5) if "." in lv or int(lv) < 6 else int(lv)
    return k


# ---- readers (R/muxers.c) ----------------------------------------------------------------------------------------------------------------
class RawYuv:
    def __init__(self, path, w, h):
        self.f, self.w, self.h = open(path, "rb"), w, h
        self.size = w * h * 3 // 2
        self.n = os.path.getsize(path) // self.size
        self.fps = None

    def read(self, i):
        self.f.seek(i * self.size)
        b = np.frombuffer(self.f.read(self.size), np.uint8)
        if b.size != self.size:
            raise IOError("short read at frame %d" % i)
        w, h = self.w, self.h
        return b[:w * h].reshape(h, w), b[w * h:w * h * 5 // 4].reshape(h // 2, w // 2), b[w * h * 5 // 4:].reshape(h // 2, w // 2)


class Y4m:
    def __init__(self, path):
        self.f = open(path, "rb")
        head = self.f.readline(96)
        if not head.startswith(b"YUV4MPEG2") or not head.endswith(b"\n"):
            raise ValueError("%s: not a YUV4MPEG2 stream" % path)
        self.w = self.h = 0
        self.fps = None
        for tok in head[10:].split():
            t, v = chr(tok[0]), tok[1:].decode()
            if t == "W":
                self.w = int(v)
            elif t == "H":
                self.h = int(v)
            elif t == "C" and not v.startswith("420"):
                raise ValueError("%s: colorspace %s unhandled (4:2:0 only)" % (path, v))
            elif t == "F":
                n, d = (int(x) for x in v.split(":"))
                if n and d:
                    self.fps = (n, d)
        self.seq_len = len(head)
        self.size = self.w * self.h * 3 // 2
        self.offsets, pos, total = [], self.seq_len, os.path.getsize(path)
        while pos < total:                         # every FRAME header may carry parameters: skip to its newline
            self.f.seek(pos)
            fh = self.f.readline(96)
            if not fh.startswith(b"FRAME") or not fh.endswith(b"\n") or pos + len(fh) + self.size > total:
                break
            self.offsets.append(pos + len(fh))
            pos += len(fh) + self.size
        self.n = len(self.offsets)

    def read(self, i):
        self.f.seek(self.offsets[i])
        b = np.frombuffer(self.f.read(self.size), np.uint8)
        w, h = self.w, self.h
        return b[:w * h].reshape(h, w), b[w * h:w * h * 5 // 4].reshape(h // 2, w // 2), b[w * h * 5 // 4:].reshape(h // 2, w // 2)


def open_inputs(names):
    """The command line's file arguments: inputs, with `WxH` as an optional last argument for raw files (or WxH in a raw file's name)."""
    names = list(names)
    res = None
    if len(names) > 1 and re.fullmatch(r"\d+x\d+", names[-1]):
        res = tuple(int(v) for v in names.pop().split("x"))
    out = []
    for n in names:
        if n.lower().endswith(".y4m"):
            out.append(Y4m(n))
        else:
            r = res
            if r is None:
                m = re.search(r"(\d+)x(\d+)", os.path.basename(n))
                if not m:
                    raise ValueError("%s: raw input needs its resolution (WxH as the last argument or in the file name)" % n)
                r = (int(m.group(1)), int(m.group(2)))
            out.append(RawYuv(n, *r))
    return out


# ---- the encoders -------------------------------------------------------------------------------------------------------------------------
def needs_lookahead(p):
    """x264_encoder_open: h->frames.b_have_lowres (encoder.c:711-716), plus anything that needs the frame queue at all."""
    return bool(p.bframe or p.rc_method == mux.RC_CRF or (p.scenecut_threshold >= 0 and p.keyint_max > 1))


def encode_streams(lib, p, sources, n_frames, sinks):
    """p: validated parameters (mux.encoder_params); sources: readers of equal picture size; sinks: binary files, one per source."""
    from .frame import JVT_LISTS, cqm_init
    from . import slice as sl
    B, w, h = len(sources), p.width, p.height
    dz = (p.luma_deadzone[0], p.luma_deadzone[1])
    cq = cqm_init(lib, JVT_LISTS if p.cqm_preset == 1 else None, luma_deadzone=dz, qp_min=p.qp_min)
    muxers = [mux.AnnexB(lib, p) for _ in range(B)]
    common = dict(qp=p.qp_constant, me_method=p.me_method, me_range=p.me_range, subme=p.subpel_refine, n_refs=p.frame_reference, inter=p.inter, intra=p.intra,
                  transform8x8=p.transform_8x8, fast_pskip=p.fast_pskip, dct_decimate=p.dct_decimate, chroma_me=p.chroma_me, cabac=p.cabac, deblock=p.deblocking_filter,
                  alpha_c0=p.deblocking_filter_alphac0, beta=p.deblocking_filter_beta, keyint=p.keyint_max, mixed_refs=p.mixed_references, noise_reduction=p.noise_reduction,
                  mv_range=p.mv_range, trellis=p.trellis, psy_rd=p.psy_rd, aq_mode=p.aq_mode, aq_strength=p.aq_strength, qp_min=p.qp_min, qp_max=p.qp_max)
    # chroma_qp_offset: the encoders apply the psy shift themselves (as x264_validate_parameters did to p): hand them the value before it
    shift = (1 if p.psy_rd < 0.25 else 2) if p.d_psy_rd_fix8 else 0
    common["chroma_qp_offset"] = p.chroma_qp_offset + shift
    coded = 0
    if not needs_lookahead(p):
        # x264_slicetype_decide has nothing to decide: an IDR every keyint frames, P frames between, constant QP
        enc = sl.ChainEncoder(lib, w, h, cq, batch=B, write=1, **common)
        try:
            for t in range(n_frames):
                for b, s in enumerate(sources):
                    enc.upload(*s.read(t), b=b)
                stype, qp, _ = enc.encode_frame()
                enc.status()
                pays = enc.payloads()
                enc.finish_frame()
                for b in range(B):
                    sinks[b].write(muxers[b].frame(frame=t, ftype=mux.TYPE_IDR if stype == sl.SLICE_I else mux.TYPE_P, qp=qp, payload=pays[b]))
                coded += 1
        finally:
            enc.close()
        return coded
    if not p.cabac:
        raise ValueError("--no-cabac with B frames, CRF or a scene cut: the frame queue's encoder (StreamEncoder) codes CABAC slices; CAVLC streams are the "
                         "constant-QP I / P ones (--qp N --bframes 0 --scenecut -1)")
    from .stream import StreamEncoder
    enc = StreamEncoder(lib, w, h, cq, batch=B, n_frames=n_frames, crf=p.rf_constant if p.rc_method == mux.RC_CRF else None, b_adapt=p.bframe_adaptive,
                        bframe_bias=p.bframe_bias, keyint_min=p.keyint_min, scenecut_threshold=p.scenecut_threshold, pre_scenecut=p.pre_scenecut,
                        ip_factor=p.ip_factor, pb_factor=p.pb_factor, qcompress=p.qcompress, qp_step=p.qp_step, bframes=p.bframe, weightb=p.weighted_bipred,
                        direct_pred=p.direct_mv_pred, **common)

    def fill(pic, f):
        for b, s in enumerate(sources):
            enc.src_ctx.upload(pic, *s.read(f), b=b)

    try:
        idle = 0
        while idle < 2 and coded < n_frames * B:
            out = enc.step(fill)
            idle = 0 if out else idle + bool(enc.flushing)
            if out:
                enc.sync()
                enc.status()
                pays = enc.payloads()
                for cd in out:
                    sinks[cd.chain].write(muxers[cd.chain].frame(frame=cd.frame, ftype=cd.type, qp=cd.qp, payload=pays[cd.chain], n_ref0=cd.n_ref0, n_ref1=cd.n_ref1,
                                                                 direct_spatial=cd.direct_spatial, frame_num_reset=cd.frame_num_reset))
                    coded += 1
    finally:
        enc.close()
    return coded // B


def main(argv=None):
    o = build_parser().parse_args(argv)
    from . import lib as L
    sources = open_inputs(o.inputs)
    w, h = sources[0].w, sources[0].h
    if any((s.w, s.h) != (w, h) for s in sources):
        raise SystemExit("encode: all inputs must have the same picture size")
    n = min(s.n for s in sources)
    if any(s.n != n for s in sources):             # the chains of a batch take their pictures in lock step: every stream ends with the shortest input
        print("encode: inputs of different lengths (%s frames): every stream is coded up to the shortest, %d frames" % (", ".join(str(s.n) for s in sources), n), file=sys.stderr)
    if o.frames > 0:
        n = min(n, o.frames)
    if n < 1:
        raise SystemExit("encode: no frames")
    k = param_fields(o)
    fps = sources[0].fps
    if o.fps is not None:                          # common.c:280-296: "25", "30000/1001", "23.976"
        if "/" in o.fps:
            fps = tuple(int(v) for v in o.fps.split("/"))
        else:
            f = float(o.fps)
            fps = (int(f * 1000 + .5), 1000)
    if fps:
        k.update(fps_num=fps[0], fps_den=fps[1])
    lib = L.load(o.device)
    try:
        p = mux.encoder_params(lib, width=w, height=h, **k)
    except ValueError as ex:
        raise SystemExit("encode: " + str(ex))
    if len(sources) > 1 and "%d" not in o.output:
        raise SystemExit("encode: several inputs need an output pattern with %d")
    names = [o.output % i if "%d" in o.output else o.output for i in range(len(sources))]
    sinks = [open(nm, "wb") for nm in names]
    try:
        done = encode_streams(lib, p, sources, n, sinks)
    except (ValueError, RuntimeError) as ex:
        raise SystemExit("encode: " + str(ex))
    finally:
        for s in sinks:
            s.close()
    print("encoded %d frames of %d stream%s (%dx%d): %s" % (done, len(sources), "" if len(sources) == 1 else "s", w, h, ", ".join(names)), file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
