"""TEST INFRASTRUCTURE -- regenerates tests/golden/cavlc_*.npz: the slice_data() bytes the REFERENCE's CAVLC writer
(x264_macroblock_write_cavlc inside oracle/ref_slice.c's loop, refslice_encode_chain2 with cabac = 0) produces for the chains of
tests/test_gpu_cavlc.py.  Needs oracle/_ref/libx264ref.so.

    python -m oracle.gen_golden_cavlc
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_cavlc as T
    for name in sorted(T.CONFIGS):
        _, pays, a = T.reference(T.CONFIGS[name])
        types = [int(np.bincount(a["mb_type"][f].astype(np.int64) & 31, minlength=8)[6]) for f in range(len(pays))]
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cavlc_%s.npz" % name), payload=a["payload"], payload_len=a["payload_len"])
        print(name, [len(p) for p in pays], "skips per frame", types)


if __name__ == "__main__":
    main()
