"""GPU: the macroblock sweep (x264hip_slice_sweep_frame: cache_load -> x264_macroblock_analyse ->
x264_macroblock_encode -> cache_save for every macroblock, one wavefront per macroblock row) against
golden arrays produced by the REFERENCE's own functions for the same chains (oracle/ref_slice.c,
oracle/gen_golden_slice.py): every decision, every level, every reconstructed pixel before and after
the loop filter, frame after frame (each frame is predicted from the GPU's own previous output)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle.gen_golden_slice import CASES, case_inputs
from x264_vs2008_amd import slice as sl

pytestmark = pytest.mark.gpu

SUPPORTED = list(CASES)
STATE = ["mb_type", "partition", "sub_partition", "ref", "i4mode", "i16mode", "chroma_mode", "qp", "t8", "mv", "cbp", "nnz", "luma", "luma_dc",
         "chroma_dc", "chroma_ac"]


def run_chain(hip_lib, cqm, size, frames, y, u, v, kw, batch=1, shift=0):
    """Encode the clip; returns per frame a dict of downloaded arrays (batch element 0 unless stated)."""
    kw = dict(kw)
    if kw.pop("cqm_preset", 0):                      # --cqm jvt: the quantiser tables x264_cqm_init builds for the JVT matrices
        with np.load(os.path.join(GOLDEN, "cqm_jvt.npz")) as z:
            cqm = {k: z[k] for k in z.files}
    enc = sl.ChainEncoder(hip_lib, size[0], size[1], cqm, batch=batch, **kw)
    out = []
    try:
        for f in range(frames):
            for b in range(batch):
                t = (f + b * shift) % frames if shift else f
                enc.upload(y[t], u[t], v[t], b=b)
            stype, qp, state = enc.encode_frame()
            enc.status()
            recon = enc.last[0]
            d = {k: state.get(k) for k in STATE + ["mvr", "cost_intra", "cost_inter"]}
            d["info"] = (stype, qp)
            for nm in ("y", "u", "v"):
                d["rec_" + nm] = np.stack([enc.ctx.download(recon, nm, padded=False, b=b) for b in range(batch)])
            enc.finish_frame()
            enc.ctx.sync()
            for nm in ("y", "u", "v"):
                d["fin_" + nm] = np.stack([enc.ctx.download(recon, nm, padded=False, b=b) for b in range(batch)])
            out.append(d)
    finally:
        enc.close()
    return out


def check_frame(d, gold, f, n_refs, b=0):
    for k in STATE:
        got, want = d[k][b], gold[k][f]
        assert np.array_equal(got.reshape(want.shape), want), "frame %d: %s differs first at %s" % (f, k, np.argwhere(got.reshape(want.shape) != want)[:3].tolist())
    for nm in ("y", "u", "v"):
        for kind in ("rec_", "fin_"):
            got, want = d[kind + nm][b], gold[kind + nm][f]
            assert np.array_equal(got, want), "frame %d: %s%s differs at %s" % (f, kind, nm, np.argwhere(got != want)[:3].tolist())
    skip = gold["mb_type"][f] == sl.P_SKIP
    nr = int(gold["frame_info"][f][2])
    for r in range(nr):
        got, want = d["mvr"][b][r], gold["mvr"][f][r]
        assert np.array_equal(got[~skip], want[~skip]), "frame %d: mvr[%d]" % (f, r)
    assert d["info"] == (int(gold["frame_info"][f][0]), int(gold["frame_info"][f][1]))
    assert int(d["cost_intra"][b].sum()) == int(gold["stat"][f][0]) and int(d["cost_inter"][b].sum()) == int(gold["stat"][f][1])


@pytest.mark.parametrize("name,size,frames,kind,kw", SUPPORTED, ids=[c[0] for c in SUPPORTED])
def test_sweep_matches_reference_loop(hip_lib, cqm, name, size, frames, kind, kw):
    with np.load(os.path.join(GOLDEN, "slice_%s.npz" % name)) as z:
        gold = {k: z[k] for k in z.files}
    y, u, v = case_inputs(size, frames, kind)
    out = run_chain(hip_lib, cqm, size, frames, y, u, v, kw)
    for f in range(frames):
        check_frame(out[f], gold, f, kw.get("n_refs", 1))


def test_sweep_batched_chains_are_independent(hip_lib, cqm):
    """Three chains in one launch; every element gets the same clip and must equal the golden chain."""
    name, size, frames, kind, kw = next(c for c in SUPPORTED if c[0] == "refs3")
    with np.load(os.path.join(GOLDEN, "slice_%s.npz" % name)) as z:
        gold = {k: z[k] for k in z.files}
    y, u, v = case_inputs(size, frames, kind)
    out = run_chain(hip_lib, cqm, size, frames, y, u, v, kw, batch=3)
    for f in range(frames):
        for b in range(3):
            check_frame(out[f], gold, f, kw.get("n_refs", 1), b=b)


def test_sweep_full_hd_matches_twin(hip_lib, oracle_lib, cqm):
    """BASELINE's frame size: a 3-frame 1920x1080 chain (120x68 macroblocks, the medium-like option set) against the CPU
    twin of the same loop (oracle/slice_oracle.c, itself pinned against the reference on the small chains)."""
    from oracle import refslice as rs
    w, h, n = 1920, 1080, 3
    kw = dict(qp=28, subme=5, me_method=rs.ME_HEX, n_refs=2, inter=0x13, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1, deblock=1)
    y, u, v = rs.clip(w, h, n)
    want = rs.run(oracle_lib, "x264o_encode_chain", rs.make_params(w, h, n, **kw), y, u, v)
    out = run_chain(hip_lib, cqm, (w, h), n, y, u, v, kw)
    for f in range(n):
        for k in STATE:
            got, ref = out[f][k][0], want[k][f]
            assert np.array_equal(got.reshape(ref.shape), ref), "frame %d: %s differs first at %s" % (f, k, np.argwhere(got.reshape(ref.shape) != ref)[:3].tolist())
        for nm in ("y", "u", "v"):
            assert np.array_equal(out[f]["fin_" + nm][0], want["fin_" + nm][f]), "frame %d: filtered %s" % (f, nm)
    t = want["mb_type"][1:]
    assert (t == sl.P_8x8).any() and (t == sl.P_L0).any() and (want["mb_type"][0] == sl.I_4x4).any()


def _against_twin(hip_lib, oracle_lib, cqm, w, h, n, kw, need_types=()):
    from oracle import refslice as rs
    y, u, v = rs.clip(w, h, n)
    want = rs.run(oracle_lib, "x264o_encode_chain", rs.make_params(w, h, n, **kw), y, u, v)
    out = run_chain(hip_lib, cqm, (w, h), n, y, u, v, kw)
    for f in range(n):
        for k in STATE:
            got, ref = out[f][k][0], want[k][f]
            assert np.array_equal(got.reshape(ref.shape), ref), "frame %d: %s differs first at %s" % (f, k, np.argwhere(got.reshape(ref.shape) != ref)[:3].tolist())
        for nm in ("y", "u", "v"):
            kind = "fin_" if kw.get("deblock") else "rec_"
            assert np.array_equal(out[f][kind + nm][0], want[kind + nm][f]), "frame %d: reconstructed %s" % (f, nm)
    for t in need_types:
        assert (want["mb_type"] == t).any()


def test_sweep_cif_ultrafast_matches_twin(hip_lib, oracle_lib, cqm):
    """BASELINE configs[0]'s shape: 352x288, the ultrafast-like option set (dia, subme 0, one reference, 16x16 only, no
    loop filter, CAVLC-side cbp), 10 frames of the synthetic clip against the CPU twin."""
    from oracle import refslice as rs
    _against_twin(hip_lib, oracle_lib, cqm, 352, 288, 10, dict(qp=26, subme=0, me_method=rs.ME_DIA, n_refs=1, cabac=0, deblock=0),
                  need_types=(sl.P_L0, sl.I_16x16))


def test_sweep_2160p_umh_matches_twin(hip_lib, oracle_lib, cqm):
    """BASELINE configs[2]'s shape: 3840x2160 (240x135 macroblocks) with --me umh, the medium-like option set, 2 frames."""
    from oracle import refslice as rs
    _against_twin(hip_lib, oracle_lib, cqm, 3840, 2160, 2,
                  dict(qp=28, subme=5, me_method=rs.ME_UMH, n_refs=1, inter=0x13, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1, deblock=1),
                  need_types=(sl.P_8x8, sl.P_L0))


def _random_case(seed):
    """A random small chain: size (also ragged: not a multiple of 16), clip kind and every option the sweep accepts."""
    r = np.random.default_rng(1000 + seed)
    w, h = int(r.integers(5, 16)) * 16 - int(r.integers(0, 2)) * 8, int(r.integers(5, 11)) * 16 - int(r.integers(0, 2)) * 8
    frames = int(r.integers(3, 6))
    kw = dict(qp=int(r.integers(18, 42)), subme=int(r.integers(0, 6)), me_method=int(r.choice([0, 1, 1, 2, 2, 3])), me_range=int(r.choice([8, 16, 24])),
              n_refs=int(r.integers(1, 5)), inter=int(r.choice([0, 0x1, 0x3, 0x10, 0x13, 0x30, 0x33])), intra=int(r.choice([0, 0x1, 0x2, 0x3])),
              transform8x8=int(r.integers(0, 2)), mixed_refs=int(r.integers(0, 2)), cabac=int(r.integers(0, 2)), deblock=int(r.integers(0, 2)),
              fast_pskip=int(r.integers(0, 2)), dct_decimate=int(r.integers(0, 2)), chroma_me=int(r.integers(0, 2)), keyint=int(r.choice([3, 250])),
              noise_reduction=int(r.choice([0, 0, 0, 60, 400])), chroma_qp_offset=int(r.choice([0, 0, -4, 3, 6])), alpha_c0=int(r.choice([0, 0, -2, 3])),
              beta=int(r.choice([0, 0, 2, -3])), mv_range=int(r.choice([0, 0, 8, 16, 64])), cqm_preset=int(r.choice([0, 0, 1])))
    if kw["me_method"] == 3:
        kw["subme"] = max(kw["subme"], 1); kw["me_range"] = min(kw["me_range"], 16)       # ESA: undefined at subme 0 in the reference; keep the scan small
    if kw["cqm_preset"]:
        kw["qp"] = max(kw["qp"], 10)                 # jvt matrices: qp < 6 overflows the 16-bit multipliers (x264_cqm_init refuses)
    if r.random() < 0.12:                            # lossless: x264_validate_parameters' consequences
        kw["qp"] = 0; kw["cqm_preset"] = 0
        if not kw["cabac"]:
            kw["transform8x8"] = 0
    if not kw["transform8x8"]:
        kw["inter"] &= ~0x2; kw["intra"] &= ~0x2                  # I8x8 needs the 8x8 transform (x264_validate_parameters)
    return (w, h), frames, ("moving" if r.integers(0, 2) else "static"), kw


@pytest.mark.parametrize("seed", range(24))
def test_sweep_random_options_match_twin(hip_lib, oracle_lib, cqm, seed):
    """Random option combinations (dia / hex / umh, subme 0..5, 1..4 references, partitions, mixed refs, intra sizes, 8x8 transform,
    CAVLC / CABAC cbp rules, skip and decimation switches, chroma ME, loop filter, short GOPs) on random, also ragged, frame sizes.
    (scratch/fuzz_gpu.py runs the same generator for as many seeds as wanted: 300 of them were clean in round 1.)"""
    from oracle import refslice as rs
    size, frames, kind, kw = _random_case(seed)
    y, u, v = case_inputs(size, frames, kind)
    want = rs.run(oracle_lib, "x264o_encode_chain", rs.make_params(size[0], size[1], frames, **kw), y, u, v)
    out = run_chain(hip_lib, cqm, size, frames, y, u, v, kw)
    for f in range(frames):
        for k in STATE:
            got, ref = out[f][k][0], want[k][f]
            assert np.array_equal(got.reshape(ref.shape), ref), "frame %d: %s (%s %s)" % (f, k, size, kw)
        for nm in ("y", "u", "v"):
            assert np.array_equal(out[f]["fin_" + nm][0], want[("fin_" if kw["deblock"] else "rec_") + nm][f]), "frame %d: %s (%s %s)" % (f, nm, size, kw)


@pytest.mark.parametrize("size", [(16, 16), (32, 16), (16, 48), (48, 32), (24, 40)])
def test_sweep_tiny_frames_match_twin(hip_lib, oracle_lib, cqm, size):
    """The smallest pictures: one macroblock, one row, one column, ragged 24x40 (mv limits, neighbour availability and the row
    hand-off degenerate here); smooth random content, the medium-like option set with every partition size."""
    from oracle import refslice as rs
    w, h = size
    frames = 4
    rng = np.random.default_rng(w * 131 + h)
    base = rng.integers(0, 256, (h + 16, w + 16)).astype(np.float32)
    for _ in range(3):                                   # blur: textures that motion search can follow
        base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(np.roll(base, 1, 0), 1, 1)) / 4
    y = np.stack([np.clip(base[2 * t:2 * t + h, t:t + w] + rng.integers(-2, 3, (h, w)), 0, 255).astype(np.uint8) for t in range(frames)])
    u = np.stack([np.clip(base[t:t + h // 2, 2 * t:2 * t + w // 2] * 0.5 + 64, 0, 255).astype(np.uint8) for t in range(frames)])
    v = np.stack([np.clip(255 - base[t:t + h // 2, t:t + w // 2] * 0.5, 0, 255).astype(np.uint8) for t in range(frames)])
    kw = dict(qp=24, subme=5, me_method=rs.ME_UMH, n_refs=2, inter=0x33, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1, deblock=1)
    want = rs.run(oracle_lib, "x264o_encode_chain", rs.make_params(w, h, frames, **kw), y, u, v)
    out = run_chain(hip_lib, cqm, size, frames, y, u, v, kw)
    for f in range(frames):
        for k in STATE:
            got, ref = out[f][k][0], want[k][f]
            assert np.array_equal(got.reshape(ref.shape), ref), "frame %d: %s" % (f, k)
        for nm in ("y", "u", "v"):
            assert np.array_equal(out[f]["fin_" + nm][0], want["fin_" + nm][f]), "frame %d: %s" % (f, nm)


def test_sweep_batched_chains_with_different_content(hip_lib, oracle_lib, cqm):
    """Four chains in one launch, each fed a different rotation of the clip: every batch element must equal the twin's result for
    ITS input (no state leaks between the chains of a launch: progress words, LDS, lane-indexed registers, --nr totals)."""
    from oracle import refslice as rs
    size, frames, batch = (208, 144), 5, 4
    kw = dict(qp=27, subme=5, me_method=rs.ME_HEX, n_refs=2, inter=0x33, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1, deblock=1, noise_reduction=150)
    y, u, v = case_inputs(size, frames, "moving")
    out = run_chain(hip_lib, cqm, size, frames, y, u, v, kw, batch=batch, shift=1)
    for b in range(batch):
        order = [(f + b) % frames for f in range(frames)]
        want = rs.run(oracle_lib, "x264o_encode_chain", rs.make_params(size[0], size[1], frames, **kw), y[order], u[order], v[order])
        for f in range(frames):
            for k in STATE:
                got, ref = out[f][k][b], want[k][f]
                assert np.array_equal(got.reshape(ref.shape), ref), "chain %d frame %d: %s" % (b, f, k)
            for nm in ("y", "u", "v"):
                assert np.array_equal(out[f]["fin_" + nm][b], want["fin_" + nm][f]), "chain %d frame %d: %s" % (b, f, nm)


@pytest.mark.gpu
@pytest.mark.parametrize("kw,tweak,needle", [
    (dict(subme=10, cabac=1), None, "subme"),                              # beyond the reference's range (x264_validate_parameters clips to 9)
    (dict(subme=8, cabac=1, inter=0x33), None, "sub-8x8"),                 # sub-8x8 partitions under the RD refinement: their partial bit counts read the previous macroblock's cache
    (dict(subme=6, cabac=0), None, "CABAC"),                               # the RD levels price against the live CABAC contexts: CAVLC RD is not built
    (dict(subme=7, cabac=1, inter=0x30), None, "sub-8x8"),                 # x264_rd_cost_part
    (dict(subme=5, me_method=4), None, "me method"),                       # TESA: not built
    (dict(subme=0, me_method=3), None, "subme"),                           # ESA at subme 0: undefined in the reference
    (dict(subme=2), "lossless_qp", "lossless"),                            # lossless without x264_validate_parameters' consequences
    (dict(subme=2), "nr_no_state", "noise_reduction"),                     # --nr without the per-chain sums
], ids=["subme10", "rd8_sub8x8", "rd_cavlc", "rd_sub8x8", "tesa", "esa_subme0", "lossless_qp26", "nr_without_state"])
def test_sweep_refuses_what_it_does_not_build(hip_lib, cqm, kw, tweak, needle):
    """No fallback: an option the sweep does not implement is an error string, not an approximation."""
    y, u, v = case_inputs((96, 80), 1, "moving")
    enc = sl.ChainEncoder(hip_lib, 96, 80, cqm, qp=26, **kw)
    try:
        if tweak == "lossless_qp":
            enc.lossless = 1
        if tweak == "nr_no_state":
            enc.opt["noise_reduction"] = 100
        enc.upload(y[0], u[0], v[0])
        with pytest.raises(RuntimeError) as e:
            enc.encode_frame()
        assert needle in str(e.value), str(e.value)
    finally:
        enc.close()


def test_abort_path_is_sticky(hip_lib, cqm, monkeypatch):
    """A wave that exhausts its spin budget aborts the launch, and the context remembers: the status call keeps failing after
    later frames have been enqueued (their states reuse -- and clear -- the aborted frame's own flag).  Forced here with a spin
    budget of one poll (X264HIP_SPIN_LIMIT), on a frame tall enough that rows really wait for each other."""
    monkeypatch.setenv("X264HIP_SPIN_LIMIT", "1")
    y, u, v = case_inputs((208, 144), 3, "moving")
    enc = sl.ChainEncoder(hip_lib, 208, 144, cqm, qp=26, subme=2, me_method=1, n_refs=1, cabac=1)
    try:
        enc.upload(y[0], u[0], v[0])
        enc.encode_frame()
        with pytest.raises(RuntimeError) as e:
            enc.status()
        assert "gave up" in str(e.value)
        monkeypatch.delenv("X264HIP_SPIN_LIMIT")
        for f in (1, 2):                     # the ring of states wraps: the aborted frame's own flag is cleared
            enc.finish_frame()
            enc.upload(y[f], u[f], v[f])
            enc.encode_frame()
        with pytest.raises(RuntimeError) as e:
            enc.status()
        assert "EARLIER frame" in str(e.value)
    finally:
        enc.close()
