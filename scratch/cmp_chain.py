"""Developer tool: run the reference's own per-macroblock loop (oracle/ref_slice.c) and the twin
(oracle/slice_oracle.c) on the same clip and report the first difference per array."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import refslice as rs


def static_clip(w, h, n, seed=0):
    """synthetic clip whose background is frozen (frame 0) outside a moving window: P_SKIP territory."""
    y, u, v = rs.clip(w, h, n)
    for t in range(1, n):
        x0, y0 = (16 + 24 * t) % max(w - 96, 1), (8 + 16 * t) % max(h - 64, 1)
        x0 &= ~1; y0 &= ~1
        by, bu, bv = y[0].copy(), u[0].copy(), v[0].copy()
        by[y0:y0 + 64, x0:x0 + 96] = y[t][y0:y0 + 64, x0:x0 + 96]
        bu[y0 // 2:y0 // 2 + 32, x0 // 2:x0 // 2 + 48] = u[t][y0 // 2:y0 // 2 + 32, x0 // 2:x0 // 2 + 48]
        bv[y0 // 2:y0 // 2 + 32, x0 // 2:x0 // 2 + 48] = v[t][y0 // 2:y0 // 2 + 32, x0 // 2:x0 // 2 + 48]
        y[t], u[t], v[t] = by, bu, bv
    return y, u, v


def compare(a, b, p):
    bad = []
    skip = (a["mb_type"] == rs.P_SKIP) | (a["mb_type"] == rs.B_SKIP)
    for k in a:
        x, z = a[k], b[k]
        if k == "mvr":
            m = np.broadcast_to(skip[:, None, :, None], x.shape) | (np.arange(x.shape[0])[:, None, None, None] == 0)
            m = m | (np.arange(x.shape[1])[None, :, None, None] >= a["frame_info"][:, 2][:, None, None, None])
            x, z = np.where(m, 0, x), np.where(m, 0, z)
        if not np.array_equal(x, z):
            idx = np.argwhere(x != z)
            bad.append((k, len(idx), idx[0].tolist(), x[tuple(idx[0])], z[tuple(idx[0])]))
    return bad


if __name__ == "__main__":
    ora = C.CDLL(os.path.join(os.path.dirname(rs.HERE), "oracle", "liboracle.so"))
    cfgs = [dict(subme=2, me_method=1, inter=0x10, n_refs=1), dict(subme=5, me_method=1, inter=0x13, intra=0x3, transform8x8=1, n_refs=3, cabac=1, deblock=1),
            dict(subme=5, me_method=1, inter=0x13, intra=0x3, transform8x8=1, n_refs=3, cabac=1, deblock=1, mixed_refs=1),
            dict(subme=1, inter=0x10, n_refs=2, mixed_refs=1, fast_pskip=0), dict(subme=4, me_method=1, inter=0x11, intra=0x1, n_refs=2, mixed_refs=1, dct_decimate=0, deblock=1),
            dict(subme=3, inter=0x10, n_refs=2, cabac=1),
            dict(subme=0), dict(subme=1), dict(subme=2, me_method=1), dict(subme=5, me_method=1, n_refs=3, cabac=1, deblock=1),
            dict(subme=4, me_method=1, n_refs=2, inter=0x3, intra=0x3, transform8x8=1, cabac=1, deblock=1),
            dict(subme=3, intra=0x1, inter=0x1, n_refs=2, deblock=1, dct_decimate=0, fast_pskip=0)]
    for size in ((208, 144), (200, 120)):
        for qp in (22, 30, 38):
            for cfg in cfgs:
                for clipf in (rs.clip, static_clip):
                    p = rs.make_params(size[0], size[1], 5, qp=qp, **cfg)
                    y, u, v = clipf(size[0], size[1], 5)
                    a = rs.run_reference(p, y, u, v)
                    b = rs.run(ora, "x264o_encode_chain", p, y, u, v)
                    bad = compare(a, b, p)
                    tc = [np.bincount(a["mb_type"][f], minlength=7)[[0, 1, 2, 4, 6]].tolist() for f in range(5)]
                    print(size, qp, cfg, clipf.__name__, "OK" if not bad else "DIFF", tc[0], tc[2], tc[4])
                    for r in bad[:8]:
                        print("    ", r)
