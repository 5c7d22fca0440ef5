"""Developer tool (CPU, needs oracle/_ref): the twin against the reference's own loop at subme 8-9 (RD refinement) and with sub-8x8
partitions under the RD levels -- payload bytes and decisions of every frame.  usage: cmp_rd89.py [configurations] [first seed]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import refslice as rs


from fuzz_b import config_refine as config      # seeds 30000 + i of tests/fuzz_b.py


if __name__ == "__main__":
    n, s0 = (int(sys.argv[1]) if len(sys.argv) > 1 else 40), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    tw = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    bad = 0
    for i in range(s0, s0 + n):
        w, h, frames, kind, kw, ekw, y, u, v = config(i)
        a = rs.run2(tw, "x264o_encode_chain2", rs.make_params(w, h, frames, **kw), rs.make_ext(**ekw), y, u, v)
        b = rs.run_reference2(rs.make_params(w, h, frames, **kw), rs.make_ext(**ekw), y, u, v)
        same = [bytes(a["payload"][f, :a["payload_len"][f]]) == bytes(b["payload"][f, :b["payload_len"][f]]) for f in range(frames)]
        if not all(same):
            bad += 1
            f = same.index(False)
            diff = {k: int(np.argwhere(a[k][f] != b[k][f])[0][0]) for k in ("mb_type", "partition", "mv", "ref", "i4mode", "i16mode", "chroma_mode", "cbp", "qp", "t8") if (a[k][f] != b[k][f]).any()}
            print("cfg %d %dx%d x%d %s %s %s: frames %s differ; frame %d type %d first differing mb per array %s" % (i, w, h, frames, kind, kw, ekw, [g for g, ok in enumerate(same) if not ok], f, a["frame_info"][f][0], diff), flush=True)
    print("done: %d of %d configurations differ between the twin and the reference" % (bad, n))
