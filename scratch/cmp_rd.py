"""Developer tool: the reference's loop with the entropy writer in it (refslice_encode_chain2) against the twin
(x264o_encode_chain2) on the same clips; reports the first difference per array, payload bytes included."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import refslice as rs
from scratch.cmp_chain import static_clip, compare


def first_bad(a, b, p):
    bad = compare({k: a[k] for k in a if k != "payload"}, b, p)
    n = a["payload_len"]
    for f in range(len(n)):
        if n[f] != b["payload_len"][f] or not np.array_equal(a["payload"][f, :n[f]], b["payload"][f, :n[f]]):
            mb = np.argwhere(a["mb_bits"][f] != b["mb_bits"][f])
            bad.append(("payload", f, int(n[f]), int(b["payload_len"][f]), "first mb_bits diff at mb %s" % (mb[0].tolist() if len(mb) else None)))
            break
    return bad


def run_case(ora, size, n, clipf, kw, ekw, verbose=True):
    p = rs.make_params(size[0], size[1], n, **kw)
    y, u, v = clipf(size[0], size[1], n)
    a = rs.run_reference2(p, rs.make_ext(**ekw), y, u, v)
    b = rs.run2(ora, "x264o_encode_chain2", p, rs.make_ext(**ekw), y, u, v)
    bad = first_bad(a, b, p)
    if verbose:
        tc = [np.bincount(a["mb_type"][f], minlength=7)[[0, 1, 2, 3, 4, 5, 6]].tolist() for f in range(n)]
        print(size, kw, ekw, clipf.__name__, "OK" if not bad else "DIFF", tc if not bad else "", flush=True)
        for x in bad[:12]:
            print("    ", x)
    return bad


if __name__ == "__main__":
    ora = C.CDLL(os.path.join(os.path.dirname(rs.HERE), "oracle", "liboracle.so"))
    base = dict(me_method=1, n_refs=3, inter=0x13, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1, deblock=1)
    which = sys.argv[1] if len(sys.argv) > 1 else "quick"
    if which == "quick":
        cases = [((208, 144), 4, rs.clip, dict(qp=26, subme=5, **base), dict()),
                 ((208, 144), 4, rs.clip, dict(qp=26, subme=6, **base), dict()),
                 ((208, 144), 4, rs.clip, dict(qp=26, subme=7, **base), dict(psy_rd=1.0)),
                 ((208, 144), 4, rs.clip, dict(qp=26, subme=7, **base), dict(trellis=1, psy_rd=1.0)),
                 ((208, 144), 4, rs.clip, dict(qp=26, subme=7, **base), dict(trellis=2, psy_rd=1.0)),
                 ((208, 144), 4, rs.clip, dict(qp=26, subme=7, **base), dict(trellis=1, psy_rd=1.0, aq_mode=1))]
        for c in cases:
            run_case(ora, *c)
    else:
        nbad = 0
        for size in ((208, 144), (200, 120), (96, 80)):
            for qp in (10, 18, 26, 34, 44):
                for subme in (5, 6, 7):
                    for ekw in (dict(), dict(psy_rd=1.0), dict(trellis=1, psy_rd=1.0), dict(trellis=2, psy_rd=0.0), dict(trellis=1, psy_rd=1.0, aq_mode=1),
                                dict(trellis=2, psy_rd=0.3, aq_mode=1, aq_strength=1.6)):
                        for clipf in (rs.clip, static_clip):
                            for var in (dict(), dict(n_refs=1, mixed_refs=0), dict(transform8x8=0, inter=0x11, intra=0x1), dict(me_method=2, dct_decimate=0, fast_pskip=0)):
                                kw = dict(base); kw.update(var); kw.update(qp=qp, subme=subme)
                                bad = run_case(ora, size, 4, clipf, kw, ekw, verbose=False)
                                if bad:
                                    nbad += 1
                                    print("DIFF", size, kw, ekw, clipf.__name__, bad[:3], flush=True)
            print("done", size, qp, "bad so far", nbad, flush=True)
        print("configurations with differences:", nbad)
