"""End to end: a batch of chains through the StreamEncoder -- the lookahead's cost kernel, the library's slice-type decision and rate
control, the chain-table sweep with every chain coding its own kind of frame at its own QP -- against the REFERENCE's encoder run on the
same clips (oracle/ref_slice.c refslice_encode_stream: x264_encoder_encode's queue around x264_slicetype_decide, x264_ratecontrol_start
and the slice loop).  Compared: the order frames are coded in, their types, QPs and the slice_data() bytes of every one.

  * golden: tests/golden/stream_*.npz made by oracle/gen_golden_stream.py from the reference;
  * live: the same configurations with other clips where oracle/_ref/libx264ref.so is built."""
import os

import numpy as np
import pytest

import look_cases as K
from oracle import refslice as rs
from x264_vs2008_amd import lookahead as LA
from x264_vs2008_amd.frame import cqm_init
from x264_vs2008_amd.stream import StreamEncoder

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libx264ref.so")


def encoder_for(hip_lib, c, batch, pipeline=False):
    return StreamEncoder(hip_lib, c["w"], c["h"], cqm_init(hip_lib), batch=batch, n_frames=c["frames"] if pipeline else None, crf=c["crf"], b_adapt=c["b_adapt"], bframe_bias=c["bframe_bias"],
                         keyint_min=c["keyint_min"], scenecut_threshold=c["scenecut_threshold"], pre_scenecut=c["pre_scenecut"],
                         qp=c["qp"], me_method=c["me"], me_range=16, subme=c["subme"], n_refs=c.get("n_refs", 2), inter=c.get("inter", 0x33), intra=0x3,
                         transform8x8=1, cabac=1, deblock=1, keyint=c["keyint"], mixed_refs=c.get("mixed_refs", 0), chroma_me=c.get("chroma_me", 1),
                         trellis=c.get("trellis", 0), psy_rd=c.get("psy_rd", 0.0), aq_mode=c["aq"], aq_strength=1.0, bframes=c["bframes"],
                         weightb=c["weightb"], direct_pred=c.get("direct_pred", 1), qp_min=0)


def run_stream(hip_lib, cs, pipeline=False):
    """cs: the chains' configurations (one encoder configuration, different clips).  Returns per chain [(frame, slice type, qp, payload)].
    pipeline: the encoder is told how many pictures there are and prepares every next call's lookahead beside the sweep in flight."""
    c0, frames = cs[0], cs[0]["frames"]
    clips = [K.clip(c["w"], c["h"], frames, c["cut"], c["t0"], c["slow"]) for c in cs]
    enc = encoder_for(hip_lib, c0, len(cs), pipeline)
    got = [[] for _ in cs]
    run_stream.resets = [[] for _ in cs]               # per coded frame: the encoder says "scene-cut IDR: frame_num restarts" (Coded.frame_num_reset)
    run_stream.direct_spatial = [[] for _ in cs]       # ... and the direct mode a B slice's header carries

    def fill(pic, f):
        for b, (y, u, v) in enumerate(clips):
            enc.src_ctx.upload(pic, y[f], u[f], v[f], b=b)

    fed, idle = 0, 0
    for _ in range(4 * frames + 40):
        coded = enc.step(fill if fed < frames else None)
        fed += fed < frames
        idle = 0 if coded else idle + (fed >= frames and enc.flushing)
        if idle >= 2:                                    # (a clip shorter than the lookahead's delay is coded by the flush alone)
            break
        if coded:
            enc.sync()
            enc.status()
            pl = enc.payloads()
            for cd in coded:
                got[cd.chain].append((cd.frame, cd.slice_type, cd.qp, pl[cd.chain]))
                run_stream.resets[cd.chain].append(int(cd.frame_num_reset))
                run_stream.direct_spatial[cd.chain].append(int(cd.direct_spatial))
    enc.close()
    return got


def check(got, a, c, what):
    frames = c["frames"]
    assert len(got) == frames, "%s: %d frames coded, the reference codes %d" % (what, len(got), frames)
    for f, (frame, st, qp, payload) in enumerate(got):
        ref = (int(a["frame_info2"][f][0]), int(a["frame_info"][f][0]), int(a["frame_info"][f][1]))
        assert (frame, st, qp) == ref, "%s coded frame %d: (input, slice, qp) %s, the reference %s" % (what, f, (frame, st, qp), ref)
        want = bytes(a["payload"][f, :a["payload_len"][f]])
        assert payload == want, "%s coded frame %d (input %d, slice %d, qp %d): payload differs (%d vs %d bytes)" % (what, f, frame, st, qp, len(payload), len(want))


def run_async(hip_lib, cs, launches=3, drift=2):
    """The same through the AsyncStreamEncoder: every chain's frames as its own kernels finish.  Returns per chain [(frame, slice type, qp, payload)]."""
    import ctypes as C
    from x264_vs2008_amd.stream import AsyncStreamEncoder
    c0, frames = cs[0], cs[0]["frames"]
    clips = [K.clip(c["w"], c["h"], frames, c["cut"], c["t0"], c["slow"]) for c in cs]
    c = c0
    enc = AsyncStreamEncoder(hip_lib, c["w"], c["h"], cqm_init(hip_lib), batch=len(cs), n_frames=frames, launches=launches, drift=drift, crf=c["crf"],
                             b_adapt=c["b_adapt"], bframe_bias=c["bframe_bias"], keyint_min=c["keyint_min"], scenecut_threshold=c["scenecut_threshold"],
                             pre_scenecut=c["pre_scenecut"], qp=c["qp"], me_method=c["me"], me_range=16, subme=c["subme"], n_refs=c.get("n_refs", 2),
                             inter=c.get("inter", 0x33), intra=0x3, transform8x8=1, cabac=1, deblock=1, keyint=c["keyint"], mixed_refs=c.get("mixed_refs", 0),
                             chroma_me=c.get("chroma_me", 1), trellis=c.get("trellis", 0), psy_rd=c.get("psy_rd", 0.0), aq_mode=c["aq"], aq_strength=1.0,
                             bframes=c["bframes"], weightb=c["weightb"], direct_pred=c.get("direct_pred", 1), qp_min=0)
    cap = min(1 << 16, enc.payload_cap - 64)          # (PAYLOAD_LEAD bytes of every chain's slot precede the payload)
    hip_lib.x264hip_host_alloc.restype = C.c_void_p
    pin = hip_lib.x264hip_host_alloc(C.c_size_t(len(cs) * frames * (cap + 64)))
    recs = [[] for _ in cs]

    def fill(pic, f):
        for b, (y, u, v) in enumerate(clips):
            enc.src_ctx.upload(pic, y[f], u[f], v[f], b=b)

    def on_launch(coded, ctx, ev_b):
        for cd in coded:
            k = len(recs[cd.chain])
            base = pin + (cd.chain * frames + k) * (cap + 64)
            enc.payload_async_of(cd, k, ctx, ev_b, base, base + 64, cap)
            recs[cd.chain].append((cd.frame, cd.slice_type, cd.qp, base))

    enc.run(fill, on_launch)
    enc.status()
    got = [[(f, st, qp, C.string_at(base + 64, C.c_int32.from_address(base).value)) for f, st, qp, base in r] for r in recs]
    sizes = list(enc.launch_sizes)
    enc.close()
    hip_lib.x264hip_host_free(C.c_void_p(pin))
    return got, sizes


CONFIGS = {
    "badapt1_crf_aq": dict(w=128, h=96, frames=14, bframes=3, b_adapt=1, crf=23.0, subme=5, me=1, weightb=1, aq=1, n_refs=2),
    "badapt2_crf_rd": dict(w=112, h=96, frames=13, bframes=2, b_adapt=2, crf=28.0, subme=7, me=2, weightb=0, aq=0, n_refs=3, mixed_refs=1, trellis=1, inter=0x13),
    "scenecut_cqp": dict(w=96, h=80, frames=12, bframes=0, b_adapt=0, crf=None, subme=6, me=1, weightb=0, aq=1, keyint=8, inter=0x13),
    "temporal_crf": dict(w=128, h=80, frames=12, bframes=1, b_adapt=1, crf=20.0, subme=4, me=0, weightb=1, aq=0, direct_pred=2),
    # the reference's default scene cut (after the encode: given-up P pictures coded again, queues rearranged) and --direct auto: fixtures, so that
    # both are held to the reference where oracle/_ref is not built too
    "postsc_crf": dict(w=112, h=96, frames=13, bframes=2, b_adapt=1, crf=24.0, subme=5, me=1, weightb=1, aq=1, n_refs=2, inter=0x13, pre_scenecut=0),
    "direct_auto_crf": dict(w=128, h=96, frames=13, bframes=3, b_adapt=1, crf=22.0, subme=6, me=1, weightb=1, aq=0, n_refs=2, inter=0x113, direct_pred=3),
}
SEEDS = {"badapt1_crf_aq": [0, 3, 9], "badapt2_crf_rd": [4, 7], "scenecut_cqp": [5, 11, 12], "temporal_crf": [1, 6], "postsc_crf": [4, 13, 20], "direct_auto_crf": [3, 21]}
STEP_ONLY = {"postsc_crf", "direct_auto_crf"}          # (the step-less scheduler keeps neither the verdict loop nor the running scores)


def chains(name, seeds):
    cs = []
    for s in seeds:
        c = K.config(s)
        c.update(pre_scenecut=1, scenecut_threshold=40, keyint=250, keyint_min=0, bframe_bias=0, qp=26)
        c.update(CONFIGS[name])
        cs.append(c)
    return cs


@pytest.mark.parametrize("pipeline", [False, True])
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_stream_equals_reference_fixture(hip_lib, name, pipeline):
    gold = np.load(os.path.join(ROOT, "tests", "golden", "stream_%s.npz" % name))
    cs = chains(name, SEEDS[name])
    got = run_stream(hip_lib, cs, pipeline)
    for i, c in enumerate(cs):
        a = {k: gold["c%d_%s" % (i, k)] for k in ("frame_info", "frame_info2", "payload", "payload_len")}
        check(got[i], a, c, "%s chain %d" % (name, i))
        for f in range(c["frames"]):                     # a B slice header's direct_spatial_mv_pred (--direct auto: it follows the running scores)
            if int(a["frame_info"][f][0]) == rs.SLICE_B:
                assert run_stream.direct_spatial[i][f] == int(a["frame_info2"][f][3]), "%s chain %d coded frame %d: direct mode" % (name, i, f)


@pytest.mark.parametrize("name", sorted(set(CONFIGS) - STEP_ONLY))
def test_async_stream_equals_reference_fixture(hip_lib, name):
    """Chains stepping on their own (AsyncStreamEncoder): each chain's frames, in its coding order, are the lock-step encoder's and the reference's."""
    gold = np.load(os.path.join(ROOT, "tests", "golden", "stream_%s.npz" % name))
    cs = chains(name, SEEDS[name])
    got, sizes = run_async(hip_lib, cs)
    for i, c in enumerate(cs):
        a = {k: gold["c%d_%s" % (i, k)] for k in ("frame_info", "frame_info2", "payload", "payload_len")}
        check(got[i], a, c, "%s chain %d (async)" % (name, i))
    assert sum(sizes) == len(cs) * cs[0]["frames"]


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libx264ref.so not built (needs /root/reference)")
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_stream_equals_reference_live(hip_lib, name):
    cs = chains(name, [s + 40 for s in SEEDS[name]])
    got = run_stream(hip_lib, cs)
    for i, c in enumerate(cs):
        check(got[i], K.reference_records(c), c, "%s chain %d" % (name, i))


def random_config(seed):
    """A seeded encoder configuration over what the stream path accepts, on top of look_cases.config's clip and lookahead options."""
    r = np.random.default_rng(91000 + seed)
    c = K.config(seed)
    subme = int(r.choice([2, 4, 5, 6, 7, 7, 8]))
    c.update(w=16 * int(r.integers(5, 10)), h=16 * int(r.integers(5, 8)), frames=int(r.integers(8, 13)), subme=subme,
             n_refs=int(r.integers(1, 4)), mixed_refs=int(r.random() < 0.5), inter=int(r.choice([0x13, 0x11, 0x10, 0x33])) if subme < 6 else int(r.choice([0x13, 0x11, 0x10])),
             trellis=int(r.choice([0, 1, 2])), psy_rd=float(r.choice([0.0, 1.0])), direct_pred=int(r.choice([1, 1, 2])), chroma_me=int(r.random() < 0.7),
             pre_scenecut=1, scenecut_threshold=int(r.choice([40, -1])), qp=int(r.integers(18, 36)))
    return c


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libx264ref.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed", list(range(14)))
def test_stream_random_configuration_equals_reference(hip_lib, seed):
    """Seeded option sets (references, partitions, subme 2..8, trellis, psy-rd, AQ, weightb, spatial / temporal direct, b-adapt 0 / 1 / 2, CQP / CRF,
    scene cuts, keyint) through the StreamEncoder, two chains with different clips, against the reference's whole encoder."""
    c = random_config(seed)
    cs = []
    for k in range(2):
        ck = dict(c)
        ck.update(t0=c["t0"] + 61 * k, slow=[c["slow"], 1 + (c["slow"] % 3)][k])
        cs.append(ck)
    got = run_stream(hip_lib, cs, pipeline=bool(seed & 1))
    for i, ck in enumerate(cs):
        check(got[i], K.reference_records(ck), ck, "seed %d chain %d %s" % (seed, i, {k: ck[k] for k in ("subme", "n_refs", "bframes", "b_adapt", "crf", "trellis", "direct_pred", "aq", "inter")}))


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libx264ref.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed", [11, 14, 19, 36, 37, 209, 226])
def test_stream_post_encode_scenecut_equals_reference(hip_lib, seed):
    """No --pre-scenecut (the reference's default): x264_encoder_encode looks at every coded P picture and, where an intra picture would have been as
    good, gives the attempt up and codes again -- the picture as I or IDR, or the B picture before it as the P, with the queues rearranged
    (R/encoder/encoder.c:1603-1699).  Clips with scene changes and repeated pictures, two chains per encoder so that one chain's second attempt
    runs while the other has nothing to do: frame order, types, QPs, payloads, and the frame_num each slice header carries (restarted only by a
    scene-cut IDR) against the reference's encoder."""
    from x264_vs2008_amd import mux
    c = dict(K.config(seed), pre_scenecut=0, subme=5, n_refs=2, inter=0x13)
    if c["scenecut_threshold"] < 0:
        c["scenecut_threshold"] = 40
    cs = [dict(c), dict(c, t0=c["t0"] + 61, cut=max(c["cut"] - 2, 0))]
    # odd seeds: the encoder knows the clip's length and runs the next call's lookahead AHEAD of the verdicts, beside the sweep, with a copy of the judged
    # chains' queues to come back to (x264hip_lookahead_save / _restore)
    got = run_stream(hip_lib, cs, pipeline=bool(seed & 1))
    resets, gave_up = run_stream.resets, 0
    for i, ck in enumerate(cs):
        a = K.reference_records(ck)
        check(got[i], a, ck, "seed %d chain %d" % (seed, i))
        gave_up += int(a["stat"][:ck["frames"], 3].sum())
        # the muxer's own frame_num bookkeeping against the reference's h->i_frame_num
        p = mux.encoder_params(hip_lib, width=ck["w"], height=ck["h"], rc_method=mux.RC_CQP, qp_constant=ck["qp"], bframe=ck["bframes"], keyint_max=ck["keyint"])
        m = mux.AnnexB(hip_lib, p)
        for f, (frame, st, qp, payload) in enumerate(got[i]):
            poc = int(a["frame_info"][f][3])
            ftype = (mux.TYPE_IDR if poc == 0 else mux.TYPE_I) if st == rs.SLICE_I else mux.TYPE_P if st == rs.SLICE_P else mux.TYPE_B
            m.frame(frame=frame, ftype=ftype, qp=qp, payload=payload, frame_num_reset=resets[i][f])
            used = m.frame_num - (0 if ftype == mux.TYPE_B else 1)
            assert used == int(a["look_cost"][f][7]), "seed %d chain %d coded frame %d: frame_num %d, the reference %d" % (seed, i, f, used, int(a["look_cost"][f][7]))
    assert gave_up > 0, "seed %d: no attempt was given up -- the clip does not test the scene cut" % seed



@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libx264ref.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed", [3, 13, 21, 26, 35, 203])
def test_stream_direct_auto_equals_reference(hip_lib, seed):
    """--direct auto: every B macroblock predicts BOTH direct modes (the other one first, then the frame's; R/encoder/analyse.c:2476-2496) and credits each
    with the skip it would give; the running scores pick every B frame's mode and decay as x264_encoder_frame_end lets them (encoder.c:113-118,
    1777-1790).  Clips with B frames (fixed pattern and b-adapt 1 / 2), below and with the RD levels, two chains with different content: order, types,
    QPs, payloads -- and the direct_spatial_mv_pred bit of every B slice header -- against the reference's encoder."""
    base = K.config(seed)
    c = dict(base, direct_pred=3, bframes=max(base["bframes"], 2), subme=[4, 6, 7, 5, 8, 7][seed % 6], n_refs=2, inter=0x113 if seed % 2 else 0x13, trellis=seed % 3 == 0)
    cs = [dict(c), dict(c, t0=c["t0"] + 61, slow=1 + (c["slow"] % 3))]
    got = run_stream(hip_lib, cs)
    n_b, modes = 0, set()
    for i, ck in enumerate(cs):
        a = K.reference_records(ck)
        check(got[i], a, ck, "seed %d chain %d" % (seed, i))
        for f in range(ck["frames"]):
            if int(a["frame_info"][f][0]) == rs.SLICE_B:
                n_b += 1
                modes.add(int(a["frame_info2"][f][3]))
                assert run_stream.direct_spatial[i][f] == int(a["frame_info2"][f][3]), "seed %d chain %d coded frame %d: direct mode" % (seed, i, f)
    assert n_b > 0


def mixed_config(seed):
    """Two chains of a seeded configuration over round 3's additions on top of random_config: --direct auto / temporal / spatial, the post- or pre-encode scene
    cut (or none), B patterns fixed / b-adapt 1 / 2, clips with and without scene changes; and whether the encoder runs its lookahead ahead (pipeline)."""
    r = np.random.default_rng(77000 + seed)
    c = random_config(seed)
    c.update(pre_scenecut=int(r.random() < 0.4), scenecut_threshold=int(r.choice([40, 40, 60, -1])), direct_pred=int(r.choice([1, 2, 3, 3])),
             bframes=int(r.choice([0, 1, 2, 3])), b_adapt=int(r.choice([0, 1, 2])), cut=int(r.choice([0, 4, 7])))
    if c["bframes"] == 0:
        c["b_adapt"] = 0
    if c["subme"] == 8 and c["bframes"] and c["inter"] & 0x20:
        c["inter"] &= ~0x20
    cs = [dict(c), dict(c, t0=c["t0"] + 61, slow=1 + (c["slow"] % 3), cut=max(c["cut"] - 2, 0))]
    return cs, bool(r.random() < 0.5)


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libx264ref.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed", [67, 68, 69, 91, 93, 99, 122, 139, 160, 203, 227, 238])
def test_stream_mixed_round3_options_equal_reference(hip_lib, seed):
    """Seeds of scratch/fuzz_stream_new.py (760 configurations equal) that combine --direct auto with given-up P pictures, with and without the lookahead
    running ahead of the verdicts: order, types, QPs, payloads and every B slice's direct mode against the reference's encoder."""
    cs, pipe = mixed_config(seed)
    got = run_stream(hip_lib, cs, pipeline=pipe)
    for i, ck in enumerate(cs):
        a = K.reference_records(ck)
        check(got[i], a, ck, "seed %d chain %d" % (seed, i))
        for f in range(ck["frames"]):
            if int(a["frame_info"][f][0]) == rs.SLICE_B:
                assert run_stream.direct_spatial[i][f] == int(a["frame_info2"][f][3]), "seed %d chain %d coded frame %d: direct mode" % (seed, i, f)
