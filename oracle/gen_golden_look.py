"""TEST INFRASTRUCTURE -- regenerates tests/golden/look_host.npz: what the REFERENCE's encoder (its frame queue, x264_slicetype_decide,
x264_rc_analyse_slice and the CRF / CQP rate control, run through oracle/ref_slice.c refslice_encode_stream) decides for the seeded clips
of tests/look_cases.py -- per coded frame the input number, slice type, POC, QP, the rate control's average QP, i_satd and the lookahead's
vectors offered to the 16x16 search.  Needs oracle/_ref/libx264ref.so (`make -C oracle ref`, i.e. /root/reference).

    python -m oracle.gen_golden_look
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import look_cases as K          # noqa: E402

SEEDS = list(range(0, 28))


def main():
    out = {}
    for seed in SEEDS:
        c = K.config(seed)
        a = K.reference_records(c)
        recs = K.records_of_reference(a, c["frames"])
        n = len(recs[0]["mv0"]) if recs[0]["mv0"] is not None else ((c["w"] + 15) // 16) * ((c["h"] + 15) // 16)
        head = np.array([[r["frame"], r["slice"], r["poc"], r["qp"], r["satd"], r["mv0"] is not None, r["mv1"] is not None] for r in recs], np.int32)
        mv = np.zeros((len(recs), 2, n, 2), np.int16)
        for i, r in enumerate(recs):
            for l in (0, 1):
                if r["mv%d" % l] is not None:
                    mv[i, l] = r["mv%d" % l]
        out["s%d_head" % seed] = head
        out["s%d_qavg" % seed] = np.array([r["f_qp_avg"] for r in recs], np.float32)
        out["s%d_mv" % seed] = mv
        print(seed, "".join("PBI"[r["slice"]] for r in recs), c)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "look_host.npz"), seeds=np.array(SEEDS), **out)


if __name__ == "__main__":
    main()
