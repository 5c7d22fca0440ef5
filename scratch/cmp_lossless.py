"""Developer tool: lossless (qp 0) configurations, reference loop vs twin."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import refslice as rs
from cmp_chain import compare, static_clip
ora = C.CDLL(os.path.join(rs.HERE, "liboracle.so"))
cfgs = [dict(subme=2, me_method=1, inter=0x10, intra=1), dict(subme=5, me_method=1, n_refs=2, cabac=1, deblock=1, inter=0x33, intra=0x3, transform8x8=1, mixed_refs=1),
        dict(subme=0, me_method=0), dict(subme=4, me_method=2, inter=0x13, intra=3, transform8x8=1, cabac=0), dict(subme=1, me_method=3, me_range=8, inter=0x30, n_refs=2, deblock=1, cabac=1),
        dict(subme=5, me_method=1, inter=0x3, intra=0x3, transform8x8=1, cabac=1, noise_reduction=100, chroma_qp_offset=3, fast_pskip=1)]
nbad = 0
for size in ((208, 144), (200, 120), (352, 288)):
    for cfg in cfgs:
        for clipf in (rs.clip, static_clip):
            p = rs.make_params(size[0], size[1], 4, qp=0, **cfg)
            y, u, v = clipf(size[0], size[1], 4)
            a = rs.run_reference(p, y, u, v); b = rs.run(ora, "x264o_encode_chain", p, y, u, v)
            bad = compare(a, b, p)
            nbad += bool(bad)
            if bad: print(size, cfg, clipf.__name__, bad[:3])
print("configs with differences:", nbad)
