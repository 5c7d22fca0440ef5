"""TEST INFRASTRUCTURE -- regenerates tests/golden/mux_uf_cif30.npz: the slice_data() bytes of the 30 frames of BASELINE config 1 (the UF flag
set on the cif30 clip) from the REFERENCE's per-macroblock loop (oracle/ref_slice.c), so that tests/test_cpu_mux.py can assemble the whole
stream around them and compare its md5 with the reference CLI's where oracle/_ref is not built.  Needs oracle/_ref/libx264ref.so.

    python -m oracle.gen_golden_mux
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import mux_cases as M
    a = M.reference_uf(352, 288, 30)
    n = int(a["payload_len"].max())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "mux_uf_cif30.npz"), payload=a["payload"][:, :n], payload_len=a["payload_len"])
    print("cif30 UF:", [int(x) for x in a["payload_len"]])


if __name__ == "__main__":
    main()
