"""Developer tool: sub-8x8 partition configurations (X264_ANALYSE_PSUB8x8), reference loop vs twin."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import refslice as rs
from cmp_chain import compare, static_clip
ora = C.CDLL(os.path.join(rs.HERE, "liboracle.so"))
cfgs = [dict(subme=2, me_method=1, inter=0x30), dict(subme=5, me_method=1, n_refs=3, cabac=1, deblock=1, inter=0x33, intra=0x3, transform8x8=1, mixed_refs=1),
        dict(subme=0, me_method=0, n_refs=2, inter=0x30), dict(subme=3, me_method=2, inter=0x30, me_range=24), dict(subme=5, me_method=2, n_refs=2, inter=0x31, intra=1, chroma_me=1),
        dict(subme=1, me_method=1, inter=0x30, mixed_refs=1, n_refs=2), dict(subme=4, me_method=1, inter=0x33, intra=3, transform8x8=1, chroma_me=0, n_refs=2),
        dict(subme=5, me_method=0, inter=0x30, chroma_me=0, dct_decimate=0, fast_pskip=0)]
nbad = 0
nsub = 0
for size in ((208, 144), (200, 120), (352, 288)):
    for qp in (20, 30):
        for cfg in cfgs:
            for clipf in (rs.clip, static_clip):
                p = rs.make_params(size[0], size[1], 4, qp=qp, **cfg)
                y, u, v = clipf(size[0], size[1], 4)
                a = rs.run_reference(p, y, u, v); b = rs.run(ora, "x264o_encode_chain", p, y, u, v)
                bad = compare(a, b, p)
                sp = a["sub_partition"][a["mb_type"] == 5]
                nsub += int((sp != 3).sum())
                if not np.array_equal(a["sub_partition"], b["sub_partition"]): bad = list(bad) + ["sub_partition"]
                nbad += bool(bad)
                if bad: print(size, qp, cfg, clipf.__name__, bad[:3])
print("configs with differences:", nbad, " sub-8x8 partitions seen:", nsub)
