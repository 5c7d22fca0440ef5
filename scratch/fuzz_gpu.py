"""Developer tool (GPU box): random small chains through the GPU sweep vs the CPU twin, every option combination the sweep accepts."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import refslice as rs
from oracle.gen_golden_slice import case_inputs
from x264_vs2008_amd import lib as L, slice as sl

STATE = ["mb_type", "partition", "sub_partition", "ref", "i4mode", "i16mode", "chroma_mode", "qp", "t8", "mv", "cbp", "nnz", "luma", "luma_dc", "chroma_dc", "chroma_ac"]


def run_gpu(hip, cqm, size, frames, y, u, v, kw):
    kw = dict(kw)
    if kw.pop("cqm_preset", 0):
        with np.load(os.path.join(ROOT, "tests", "golden", "cqm_jvt.npz")) as z:
            cqm = {k: z[k] for k in z.files}
    enc = sl.ChainEncoder(hip, size[0], size[1], cqm, batch=1, **kw)
    out = []
    try:
        for f in range(frames):
            enc.upload(y[f], u[f], v[f], b=0)
            stype, qp, state = enc.encode_frame()
            enc.status()
            d = {k: state.get(k) for k in STATE}
            enc.finish_frame()
            enc.ctx.sync()
            for nm in ("y", "u", "v"):
                d["fin_" + nm] = enc.ctx.download(enc.last[0], nm, padded=False, b=0)
            out.append(d)
    finally:
        enc.close()
    return out


def main():
    n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    hip = L.load()
    tw = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    with np.load(os.path.join(ROOT, "tests", "golden", "cqm_flat.npz")) as z:
        cqm = {k: z[k] for k in z.files}
    bad = 0
    for i in range(n_cfg):
        r = np.random.default_rng(1000 + seed0 + i)
        w, h = int(r.integers(5, 16)) * 16 - int(r.integers(0, 2)) * 8, int(r.integers(5, 11)) * 16 - int(r.integers(0, 2)) * 8
        if os.environ.get("FUZZ_BIG"):              # larger pictures: long vectors, many rows in flight
            w, h = int(r.integers(20, 46)) * 16 - int(r.integers(0, 2)) * 8, int(r.integers(12, 31)) * 16 - int(r.integers(0, 2)) * 8
        frames = int(r.integers(3, 6))
        kw = dict(qp=int(r.integers(18, 42)), subme=int(r.integers(0, 6)), me_method=int(r.choice([0, 1, 1, 2, 2, 3])), me_range=int(r.choice([8, 16, 24])),
                  n_refs=int(r.integers(1, 5)), inter=int(r.choice([0, 0x1, 0x3, 0x10, 0x13, 0x30, 0x33])), intra=int(r.choice([0, 0x1, 0x2, 0x3])),
                  transform8x8=int(r.integers(0, 2)), mixed_refs=int(r.integers(0, 2)), cabac=int(r.integers(0, 2)), deblock=int(r.integers(0, 2)),
                  fast_pskip=int(r.integers(0, 2)), dct_decimate=int(r.integers(0, 2)), chroma_me=int(r.integers(0, 2)), keyint=int(r.choice([3, 250])),
                  noise_reduction=int(r.choice([0, 0, 0, 60, 400])), chroma_qp_offset=int(r.choice([0, 0, -4, 3, 6])), alpha_c0=int(r.choice([0, 0, -2, 3])),
                  beta=int(r.choice([0, 0, 2, -3])), mv_range=int(r.choice([0, 0, 8, 16, 64])), cqm_preset=int(r.choice([0, 0, 1])))
        if kw["me_method"] == 3:
            kw["subme"] = max(kw["subme"], 1); kw["me_range"] = min(kw["me_range"], 16)   # ESA: undefined at subme 0 in the reference; keep the scan small
        if kw["cqm_preset"]:
            kw["qp"] = max(kw["qp"], 10)             # jvt matrices: qp < 6 overflows the 16-bit multipliers
        if r.random() < float(os.environ.get("FUZZ_LOSSLESS", "0.12")):     # lossless: x264_validate_parameters' consequences
            kw["qp"] = 0; kw["cqm_preset"] = 0
            if not kw["cabac"]:
                kw["transform8x8"] = 0
        if not kw["transform8x8"]:
            kw["inter"] &= ~0x2; kw["intra"] &= ~0x2          # I8x8 needs the 8x8 transform (x264_validate_parameters)
        kind = "moving" if r.integers(0, 2) else "static"
        y, u, v = case_inputs((w, h), frames, kind)
        want = rs.run(tw, "x264o_encode_chain", rs.make_params(w, h, frames, **kw), y, u, v)
        got = run_gpu(hip, cqm, (w, h), frames, y, u, v, kw)
        diffs = []
        for f in range(frames):
            for k in STATE:
                a, b = got[f][k][0], want[k][f]
                if not np.array_equal(a.reshape(b.shape), b):
                    diffs.append("f%d:%s" % (f, k))
            for nm in ("y", "u", "v"):
                kk = "fin_" if kw["deblock"] else "rec_"
                if not np.array_equal(got[f]["fin_" + nm], want[kk + nm][f]):
                    diffs.append("f%d:%s%s" % (f, kk, nm))
        types = np.bincount(want["mb_type"].ravel(), minlength=7).tolist()
        print("cfg %d %dx%d x%d %s %s -> %s  types %s" % (i, w, h, frames, kind, kw, "OK" if not diffs else "DIFF " + ",".join(diffs[:6]), types), flush=True)
        bad += bool(diffs)
    print("configs with differences: %d of %d" % (bad, n_cfg))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
