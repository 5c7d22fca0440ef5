"""Seeded random I P B chains for the raster sweep: the configuration generator and the GPU-vs-twin comparison used by
tests/test_gpu_fuzz_cases.py and scratch/fuzz_gpu_b.py.  The twin (oracle/) is the checker here, as in every parity test: it is pinned
to the reference by the fixtures of tests/golden and by oracle/gen_golden_slice.py's live comparisons.  A noisy bottom-right corner
makes a P frame's last macroblocks end intra (their analysis leftovers are what x264hip_slice_rd.stale carries into the next frame)."""
import numpy as np

from oracle import refslice as rs
from oracle.gen_golden_slice import case_inputs
from x264_vs2008_amd import slice as sl


def config_wide(i):
    """Seeds from 1000: a wider option space, I / P chains included (UMH / ESA, search and vector ranges, noise reduction, chroma QP
    offsets, sub-8x8 partitions below the RD levels, more references)."""
    r = np.random.default_rng(9000 + i)
    w, h = int(r.integers(5, 14)) * 16 - int(r.integers(0, 2)) * 8, int(r.integers(5, 10)) * 16 - int(r.integers(0, 2)) * 8
    if i >= 5000:                                  # seeds from 5000: pictures large enough for long vectors and the vector-range clipping
        w, h = int(r.integers(20, 44)) * 16 - int(r.integers(0, 2)) * 8, int(r.integers(12, 26)) * 16 - int(r.integers(0, 2)) * 8
    frames = int(r.integers(4, 9)) if i < 5000 else int(r.integers(4, 7))
    bframes = int(r.choice([0, 0, 1, 2, 3]))
    subme = int(r.choice([1, 2, 3, 4, 5, 6, 7, 7])) if not bframes else int(r.choice([2, 3, 4, 5, 6, 7, 7]))
    kw = dict(qp=int(r.integers(12, 44)), subme=subme, me_method=int(r.choice([0, 1, 1, 2, 2, 3])), me_range=int(r.choice([8, 16, 24])),
              n_refs=int(r.integers(1, 5)), inter=int(r.choice([0, 0x10, 0x13, 0x13, 0x33])), intra=int(r.choice([0x1, 0x3])),
              transform8x8=int(r.integers(0, 2)), mixed_refs=int(r.integers(0, 2)), cabac=1, deblock=int(r.integers(0, 2)), fast_pskip=int(r.integers(0, 2)),
              dct_decimate=int(r.integers(0, 2)), chroma_me=int(r.integers(0, 2)), keyint=int(r.choice([0, 0, 4, 7])),
              noise_reduction=int(r.choice([0, 0, 0, 80])) if not bframes else 0, chroma_qp_offset=int(r.choice([0, 0, -4, 3])),
              alpha_c0=int(r.choice([0, 0, -2, 3])), beta=int(r.choice([0, 0, 2])), mv_range=int(r.choice([0, 0, 16, 64])))
    if kw["me_method"] == 3:
        kw["me_range"] = min(kw["me_range"], 16); kw["subme"] = max(kw["subme"], 1)
    if subme >= 6:
        kw["inter"] &= ~0x20                       # sub-8x8 partitions with the RD levels: refused
    if bframes:
        kw["inter"] |= int(r.choice([0, 0x100]))
    if not kw["transform8x8"]:
        kw["inter"] &= ~0x2; kw["intra"] &= ~0x2
    ekw = dict(trellis=int(r.choice([0, 1, 2])), psy_rd=float(r.choice([0.0, 0.4, 1.0])), aq_mode=int(r.integers(0, 2)), aq_strength=float(r.choice([0.6, 1.0, 1.4])),
               bframes=bframes, weightb=int(r.integers(0, 2)), direct_pred=int(r.choice([1, 2])))
    if i >= 1500 and r.random() < 0.5:
        ekw["lowres_seed"] = int(i)                # the lookahead's candidates (stand-in vectors, oracle/refslice.py: lowres_vectors)
    kind = "moving" if r.integers(0, 2) else "static"
    y, u, v = case_inputs((w, h), frames, kind)
    y = y.copy()
    y[:, -24:, -40:] = np.random.default_rng(i).integers(0, 256, (frames, 24, 40), dtype=np.uint8)
    return w, h, frames, kind, kw, ekw, y, u, v


def config_refine(i):
    """Seeds from 30000 (config(30000 + i)): subme 6..9 -- the RD refinement of vectors and intra modes (x264_me_refine_qpel_rd, _bidir_rd,
    x264_intra_rd_refine) in I / P / B chains, sub-8x8 partitions under the RD levels, 1..4 references, trellis, psy, AQ."""
    r = np.random.default_rng(31000 + i)
    w, h = int(r.integers(5, 12)) * 16 - int(r.integers(0, 2)) * 8, int(r.integers(5, 9)) * 16 - int(r.integers(0, 2)) * 8
    frames = int(r.integers(3, 7))
    bframes = int(r.choice([0, 0, 1, 2, 3]))
    subme = int(r.choice([6, 7, 8, 8, 9, 9]))
    kw = dict(qp=int(r.integers(14, 42)), subme=subme, me_method=int(r.choice([0, 1, 1, 2])), me_range=int(r.choice([8, 16])),
              n_refs=int(r.integers(1, 5)), inter=int(r.choice([0, 0x10, 0x13, 0x13, 0x33, 0x33])), intra=int(r.choice([0x1, 0x3])),
              transform8x8=int(r.integers(0, 2)), mixed_refs=int(r.integers(0, 2)), cabac=1, deblock=int(r.integers(0, 2)), fast_pskip=int(r.integers(0, 2)),
              dct_decimate=int(r.integers(0, 2)), chroma_me=int(r.integers(0, 2)), keyint=int(r.choice([0, 0, 4])),
              chroma_qp_offset=int(r.choice([0, 0, -3, 2])), mv_range=int(r.choice([0, 0, 16, 64])))
    if bframes:
        kw["inter"] |= int(r.choice([0, 0x100]))
    if not kw["transform8x8"]:
        kw["inter"] &= ~0x2; kw["intra"] &= ~0x2
    ekw = dict(trellis=int(r.choice([0, 1, 2])), psy_rd=float(r.choice([0.0, 0.5, 1.0])), aq_mode=int(r.integers(0, 2)), aq_strength=float(r.choice([0.7, 1.0, 1.3])),
               bframes=bframes, weightb=int(r.integers(0, 2)), direct_pred=int(r.choice([1, 1, 2])))
    kind = "moving" if r.integers(0, 2) else "static"
    y, u, v = case_inputs((w, h), frames, kind)
    y = y.copy()
    y[:, -24:, -40:] = np.random.default_rng(i).integers(0, 256, (frames, 24, 40), dtype=np.uint8)
    return w, h, frames, kind, kw, ekw, y, u, v


def config(i):
    if i >= 30000:
        return config_refine(i - 30000)
    if i >= 1000:
        return config_wide(i)
    r = np.random.default_rng(7000 + i)
    w, h = int(r.integers(5, 12)) * 16, int(r.integers(5, 9)) * 16
    frames = int(r.integers(5, 10))
    kw = dict(qp=int(r.integers(16, 40)), subme=int(r.choice([2, 4, 5, 6, 7, 7])), me_method=int(r.choice([0, 1, 1, 2])), me_range=16,
              n_refs=int(r.integers(1, 4)), inter=int(r.choice([0x100, 0x110, 0x113, 0x113])), intra=int(r.choice([0x1, 0x3])),
              transform8x8=int(r.integers(0, 2)), mixed_refs=int(r.integers(0, 2)), cabac=1, deblock=1, fast_pskip=int(r.integers(0, 2)),
              dct_decimate=1, chroma_me=int(r.integers(0, 2)), keyint=int(r.choice([0, 0, 6])))
    if not kw["transform8x8"]:
        kw["inter"] &= ~0x2; kw["intra"] &= ~0x2
    ekw = dict(trellis=int(r.choice([0, 1, 2])), psy_rd=float(r.choice([0.0, 1.0])), aq_mode=int(r.integers(0, 2)), bframes=int(r.integers(1, 4)),
               weightb=int(r.integers(0, 2)), direct_pred=int(r.choice([1, 2, 2])))
    kind = "moving" if r.integers(0, 2) else "static"
    y, u, v = case_inputs((w, h), frames, kind)
    y = y.copy()
    y[:, -24:, -40:] = np.random.default_rng(i).integers(0, 256, (frames, 24, 40), dtype=np.uint8)      # the last macroblocks: unpredictable
    return w, h, frames, kind, kw, ekw, y, u, v


def compare(hip, twin, cqm, i):
    """-> (description, [frames whose payload differs], the twin's type of every frame's last macroblock)"""
    w, h, frames, kind, kw, ekw, y, u, v = config(i)
    want = rs.run2(twin, "x264o_encode_chain2", rs.make_params(w, h, frames, **kw), rs.make_ext(**ekw), y, u, v)
    what = "%dx%d x%d %s %s %s" % (w, h, frames, kind, kw, ekw)
    ekw = dict(ekw)
    seed = ekw.pop("lowres_seed", None)
    lowres = None
    if seed is not None:
        from x264_vs2008_amd.frame import DeviceArray
        n = ((w + 15) // 16) * ((h + 15) // 16)
        lm = rs.lowres_vectors(seed, frames, n)
        lowres = [tuple(DeviceArray(hip, (1, n, 2), np.int16, np.ascontiguousarray(lm[f, l][None])) for l in range(2)) for f in range(frames)]
    try:
        enc = sl.ChainEncoder(hip, w, h, cqm, batch=1, write=1, **kw, **ekw)
    except (RuntimeError, ValueError) as e:
        return what + " REFUSED " + str(e)[:80], [], []
    order = sl.coding_order(frames, kw["keyint"], ekw["bframes"]) if ekw["bframes"] else \
        [(t, sl.SLICE_I if (t % kw["keyint"] == 0 if kw["keyint"] else t == 0) else sl.SLICE_P) for t in range(frames)]
    diffs = []
    try:
        for f, (disp, stype) in enumerate(order):
            enc.upload(y[disp], u[disp], v[disp])
            try:
                enc.encode_frame(stype=stype, disp=disp, **(dict(lowres_mv=lowres[f][0], lowres_mv1=lowres[f][1]) if lowres else {}))
            except RuntimeError as e:                  # an option combination the sweep refuses (it says so): not a difference
                if "slice_sweep:" in str(e) and ("not built" in str(e) or f == (1 if stype != sl.SLICE_I else 0)):
                    return what + " REFUSED " + str(e)[-90:], [], []
                raise
            enc.status()
            if enc.payloads()[0] != bytes(want["payload"][f, :want["payload_len"][f]]):
                diffs.append("f%d(%s d%d)" % (f, "PBI"[stype], disp))
            enc.finish_frame()
    finally:
        enc.close()
    return what, diffs, [int(want["mb_type"][f, -1]) for f in range(frames)]
