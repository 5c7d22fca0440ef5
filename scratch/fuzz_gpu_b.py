"""Developer tool (GPU box): random small I P B chains through the raster sweep vs the CPU twin -- payload bytes of every frame (they
cover every decision).  usage: fuzz_gpu_b.py [configurations] [first seed]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_b import compare
from x264_vs2008_amd import lib as L


def main():
    n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    hip = L.load()
    tw = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    with np.load(os.path.join(ROOT, "tests", "golden", "cqm_flat.npz")) as z:
        cqm = {k: z[k] for k in z.files}
    bad = 0
    for i in range(seed0, seed0 + n_cfg):
        what, diffs, last = compare(hip, tw, cqm, i)
        print("cfg %d %s last-mb types %s: %s" % (i, what, last, "OK" if not diffs else "DIFF " + " ".join(diffs)), flush=True)
        bad += bool(diffs)
    print("done: %d of %d configurations differ" % (bad, n_cfg))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
