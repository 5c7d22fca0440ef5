"""Seeded configurations of the lookahead / rate-control tests: clip (with an optional scene change), the encoder options that reach
x264_slicetype_decide and x264_ratecontrol_start, and the drivers that produce comparable records from the reference's queue
(oracle/ref_slice.c refslice_encode_stream) and from the library's host state machine."""
import numpy as np

from oracle import refslice as rs
from x264_vs2008_amd import lookahead as LA
from x264_vs2008_amd import synth


def clip(w, h, frames, cut, t0, slow=1):
    """frames of the synthetic clip from t0; from `cut` on a different scene (the generator far away in time, turned upside down)."""
    fr = []
    for t in range(frames):
        if cut and t >= cut:
            y, u, v = synth.frame(w, h, t0 + 700 + 3 * t)
            fr.append((np.ascontiguousarray(y[::-1, ::-1]), np.ascontiguousarray(u[::-1, ::-1]), np.ascontiguousarray(v[::-1, ::-1])))
        else:
            fr.append(synth.frame(w, h, t0 + t // slow))            # slow > 1: every picture repeated (what b-adapt answers with B frames)
    return tuple(np.ascontiguousarray(np.stack([f[i] for f in fr])) for i in range(3))


def config(seed):
    r = np.random.default_rng(77000 + seed)
    # both above 64 for synth.frame.  Odd numbers of macroblock columns included: there x264_frame_expand_border_lowres pads from column
    # stride - 64, not from the lowres width, and the reference's searches near the right edge read columns nobody wrote (found here as a
    # run-to-run difference of the REFERENCE); oracle/ref_slice.c pins them to 0, the fresh-page value, as the twin and the kernels do
    w, h = 16 * int(r.integers(5, 11)), 16 * int(r.integers(5, 9))
    frames = int(r.integers(9, 19))
    bframes = int(r.choice([0, 1, 2, 3, 3, 4]))
    b_adapt = int(r.choice([0, 1, 1, 2])) if bframes else 0
    pre = int(r.random() < 0.7)
    cut = int(r.integers(3, frames - 2)) if r.random() < 0.5 else 0
    keyint = int(r.choice([250, 250, 6, 9]))
    crf = float(r.choice([18.0, 23.0, 27.5, 33.0])) if r.random() < 0.75 else None
    c = dict(w=w, h=h, frames=frames, bframes=bframes, b_adapt=b_adapt, pre_scenecut=pre, scenecut_threshold=int(r.choice([40, 40, 60])) if pre else -1,
             cut=cut, keyint=keyint, keyint_min=int(r.choice([0, 0, 2])), crf=crf, qp=int(r.integers(20, 36)), bframe_bias=int(r.choice([0, 0, 20, -30])),
             weightb=int(r.random() < 0.5), me=int(r.choice([rs.ME_DIA, rs.ME_HEX, rs.ME_HEX, rs.ME_UMH])), subme=int(r.choice([2, 4, 5, 6])),
             aq=int(r.random() < 0.4), t0=int(r.integers(0, 400)), slow=int(r.choice([1, 1, 2, 3])))
    return c


def reference_records(c):
    """What the reference's encoder does with the clip: per coded frame (input number, type, QP, lowres vectors l0 / l1 or None, i_satd),
    and the whole harness output (payloads ...) for the tests that go on to encode."""
    p = rs.make_params(c["w"], c["h"], c["frames"], qp=c["qp"], me_method=c["me"], subme=c["subme"], n_refs=c.get("n_refs", 2), inter=c.get("inter", 0x33),
                       intra=0x3, transform8x8=1, cabac=1, deblock=1, keyint=c["keyint"], mixed_refs=c.get("mixed_refs", 0), chroma_me=c.get("chroma_me", 1))
    e = rs.make_ext(bframes=c["bframes"], b_adapt=c["b_adapt"], pre_scenecut=c["pre_scenecut"], scenecut_threshold=c["scenecut_threshold"],
                    keyint_min=c["keyint_min"], crf=-1.0 if c["crf"] is None else c["crf"], bframe_bias=c["bframe_bias"], weightb=c["weightb"],
                    aq_mode=c["aq"], aq_strength=1.0, trellis=c.get("trellis", 0), psy_rd=c.get("psy_rd", 0.0), direct_pred=c.get("direct_pred", 1))
    y, u, v = clip(c["w"], c["h"], c["frames"], c["cut"], c["t0"], c["slow"])
    a = rs.run_reference_stream(p, e, y, u, v)
    return a


# X264_TYPE_* of a coded frame from the harness's (slice type, kept_as_ref ...) is not recorded; the slice type and the IDR-ness (POC 0) are
def records_of_reference(a, frames):
    out = []
    for f in range(frames):
        st, qp, _, poc = (int(x) for x in a["frame_info"][f])
        lm = a["look_mv"][f]
        out.append(dict(frame=int(a["frame_info2"][f][0]), slice=st, poc=poc, qp=qp, f_qp_avg=float(a["rc_info"][f][1]),
                        mv0=None if lm[0, 0, 0] == 0x7fff else lm[0].copy(), mv1=None if lm[1, 0, 0] == 0x7fff else lm[1].copy(),
                        satd=int(a["rc_info"][f][2]) if st != rs.SLICE_B else 0))
    return out


def lookahead_params(c):
    return LA.make_params((c["w"] + 15) // 16, (c["h"] + 15) // 16, bframes=c["bframes"], b_adapt=c["b_adapt"], bframe_bias=c["bframe_bias"],
                          keyint_max=c["keyint"], keyint_min=c["keyint_min"], scenecut_threshold=c["scenecut_threshold"],
                          pre_scenecut=c["pre_scenecut"], crf=c["crf"], qp=c["qp"], qp_min=0)


SLICE_OF_TYPE = {LA.TYPE_IDR: rs.SLICE_I, LA.TYPE_I: rs.SLICE_I, LA.TYPE_P: rs.SLICE_P, LA.TYPE_B: rs.SLICE_B}


def compare(mine, ref):
    """mine: look_util.run_chain's list; ref: records_of_reference's.  Returns a list of differences (empty = equal)."""
    bad = []
    if len(mine) != len(ref):
        return ["%d frames coded, the reference codes %d" % (len(mine), len(ref))]
    for i, (m, r) in enumerate(zip(mine, ref)):
        frame, typ, qp, f_qpm, ref0, ref1, mv0, mv1, satd = m
        if frame != r["frame"] or SLICE_OF_TYPE[typ] != r["slice"] or (typ == LA.TYPE_IDR) != (r["slice"] == rs.SLICE_I and r["poc"] == 0):
            bad.append("coded frame %d: input %d type %d, the reference: input %d slice %d poc %d" % (i, frame, typ, r["frame"], r["slice"], r["poc"]))
            continue
        if qp != r["qp"]:
            bad.append("coded frame %d (input %d): QP %d (f_qpm %r), the reference %d" % (i, frame, qp, f_qpm, r["qp"]))
        if abs(f_qpm - r["f_qp_avg"]) > 1e-3 and r["slice"] != -1:
            bad.append("coded frame %d: f_qpm %r, the reference's average %r" % (i, f_qpm, r["f_qp_avg"]))
        if r["slice"] != rs.SLICE_B and satd != r["satd"]:
            bad.append("coded frame %d: i_satd %d, the reference %d" % (i, satd, r["satd"]))
        for l, (a, b) in enumerate(((mv0, r["mv0"]), (mv1, r["mv1"]))):
            if r["slice"] == rs.SLICE_I or (l == 1 and r["slice"] != rs.SLICE_B):
                continue
            if (a is None) != (b is None):
                bad.append("coded frame %d list %d: lowres vectors %s, the reference %s" % (i, l, "absent" if a is None else "present", "absent" if b is None else "present"))
            elif a is not None and not np.array_equal(a, b):
                bad.append("coded frame %d list %d: %d lowres vectors differ" % (i, l, int((a != b).any(1).sum())))
    return bad
