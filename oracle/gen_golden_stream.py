"""TEST INFRASTRUCTURE -- regenerates tests/golden/stream_*.npz: the REFERENCE's whole encoder (frame queue, x264_slicetype_decide,
x264_ratecontrol_start, the slice loop with the entropy coder; oracle/ref_slice.c refslice_encode_stream) on the clips and options of
tests/test_gpu_stream.py -- per coded frame the input number, slice type, QP and the slice_data() bytes.
Needs oracle/_ref/libx264ref.so (`make -C oracle ref`, i.e. /root/reference).

    python -m oracle.gen_golden_stream
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import look_cases as K
    import test_gpu_stream as T
    for name in sorted(T.CONFIGS):
        out = {}
        for i, c in enumerate(T.chains(name, T.SEEDS[name])):
            a = K.reference_records(c)
            for k in ("frame_info", "frame_info2", "payload", "payload_len"):
                out["c%d_%s" % (i, k)] = a[k]
            print(name, i, "".join("PBI"[int(t)] for t in a["frame_info"][:, 0]), [int(q) for q in a["frame_info"][:, 1]], int(a["payload_len"].sum()), "bytes")
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "stream_%s.npz" % name), **out)


if __name__ == "__main__":
    main()
