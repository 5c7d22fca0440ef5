"""Developer tool (CPU, needs /root/reference built into oracle/_ref): the random chains of tests/fuzz_b.py through the twin and through the
reference's own loop -- payload bytes of every frame.  usage: fuzz_twin_ref.py [configurations] [first seed]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_b
from oracle import refslice as rs

n, s0 = (int(sys.argv[1]) if len(sys.argv) > 1 else 50), (int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
tw = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
bad = 0
for i in range(s0, s0 + n):
    w, h, frames, kind, kw, ekw, y, u, v = fuzz_b.config(i)
    a = rs.run2(tw, "x264o_encode_chain2", rs.make_params(w, h, frames, **kw), rs.make_ext(**ekw), y, u, v)
    b = rs.run_reference2(rs.make_params(w, h, frames, **kw), rs.make_ext(**ekw), y, u, v)
    same = [bytes(a["payload"][f, :a["payload_len"][f]]) == bytes(b["payload"][f, :b["payload_len"][f]]) for f in range(frames)]
    if not all(same):
        bad += 1
        print("cfg %d %s %s: frames %s differ" % (i, kw, ekw, [f for f, ok in enumerate(same) if not ok]), flush=True)
print("done: %d of %d configurations differ between the twin and the reference" % (bad, n))
