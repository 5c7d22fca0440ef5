"""Developer tool: B chains, reference harness vs twin."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import refslice as rs
from scratch.cmp_chain import static_clip
from scratch.cmp_rd import first_bad

def run_case(ora, size, n, clipf, kw, ekw, verbose=True):
    p = rs.make_params(size[0], size[1], n, **kw)
    y, u, v = clipf(size[0], size[1], n)
    a = rs.run_reference2(p, rs.make_ext(**ekw), y, u, v)
    b = rs.run2(ora, "x264o_encode_chain2", p, rs.make_ext(**ekw), y, u, v)
    bad = first_bad(a, b, p)
    if verbose:
        print(size, kw, ekw, clipf.__name__, "OK" if not bad else "DIFF", flush=True)
        if not bad:
            for f in range(n):
                print("   ", f, a["frame_info"][f].tolist(), np.bincount(a["mb_type"][f].astype(np.int64), minlength=19).tolist())
        for x in bad[:14]:
            print("    ", x)
    return bad

if __name__ == "__main__":
    ora = C.CDLL(os.path.join(os.path.dirname(rs.HERE), "oracle", "liboracle.so"))
    base = dict(me_method=1, n_refs=2, inter=0x113, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1, deblock=1)
    which = sys.argv[1] if len(sys.argv) > 1 else "quick"
    if which == "quick":
        sub = int(sys.argv[2]) if len(sys.argv) > 2 else 5
        dp = int(sys.argv[3]) if len(sys.argv) > 3 else 1
        run_case(ora, (160, 128), 5, rs.clip, dict(qp=28, subme=sub, **base), dict(bframes=2, weightb=0, direct_pred=dp))
    else:
        nbad = 0
        sizes = ((208, 144), (200, 120), (96, 80))
        for size in (sizes if len(sys.argv) < 3 else (sizes[int(sys.argv[2])],)):
            for qp in (12, 22, 30, 40):
                for subme in (2, 5, 6, 7):
                    for ekw in (dict(bframes=1), dict(bframes=2, weightb=1), dict(bframes=3, weightb=1, direct_pred=2), dict(bframes=2, trellis=1, psy_rd=1.0, aq_mode=1, weightb=1),
                                dict(bframes=3, trellis=2, psy_rd=0.0, direct_pred=2), dict(bframes=16, weightb=1, trellis=1, psy_rd=1.0)):
                        for clipf in (rs.clip, static_clip):
                            for var in (dict(), dict(n_refs=1, mixed_refs=0), dict(transform8x8=0, inter=0x11, intra=0x1), dict(me_method=2, n_refs=3, keyint=5), dict(me_method=0, inter=0x113 & ~0x100)):
                                kw = dict(base); kw.update(var); kw.update(qp=qp, subme=subme)
                                bad = run_case(ora, size, 7, clipf, kw, ekw, verbose=False)
                                if bad:
                                    nbad += 1
                                    print("DIFF", size, kw, ekw, clipf.__name__, bad[:3], flush=True)
            print("done", size, "bad so far", nbad, flush=True)
        print("configurations with differences:", nbad)
