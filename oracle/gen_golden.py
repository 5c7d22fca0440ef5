#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generate tests/golden/*.npz from the reference.

Runs the reference's own C implementation of the six DSP tables
(oracle/_ref/libx264ref.so, built by `make -C oracle ref` from the reference
sources where they lie) through oracle/harness.py and stores inputs and
outputs as golden vectors.  Only runnable where /root/reference exists; the
committed fixtures are what travels.

    python3 oracle/gen_golden.py            # regenerate tests/golden
    python3 oracle/gen_golden.py --check    # also diff the CPU oracle against the reference
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import harness  # noqa: E402
from x264_vs2008_amd.tables import TableSet  # noqa: E402

SEEDS = (1234, 20090216)


def load_ref():
    from oracle import hostpic
    lib = hostpic.load_lazy(os.path.join(HERE, "_ref", "libx264ref.so"))
    for n in ("refshim_quant4_mf", "refshim_quant4_bias", "refshim_quant8_mf", "refshim_quant8_bias"):
        getattr(lib, n).restype = C.POINTER(C.c_uint16)
    for n in ("refshim_dequant4_mf", "refshim_dequant8_mf"):
        getattr(lib, n).restype = C.POINTER(C.c_int)
    return lib


def ref_cqm(lib):
    """Quantiser tables for the flat matrices, straight from x264_cqm_init."""
    assert lib.refshim_cqm_flat_init() == 0
    g = lambda p, n: np.ctypeslib.as_array(p, shape=(n,)).copy()
    cqm = {
        "quant4_mf": np.array([[g(lib.refshim_quant4_mf(c, q), 16) for q in range(52)] for c in range(4)]),
        "quant4_bias": np.array([[g(lib.refshim_quant4_bias(c, q), 16) for q in range(52)] for c in range(4)]),
        "quant8_mf": np.array([[g(lib.refshim_quant8_mf(c, q), 64) for q in range(52)] for c in range(2)]),
        "quant8_bias": np.array([[g(lib.refshim_quant8_bias(c, q), 64) for q in range(52)] for c in range(2)]),
        "dequant4_mf": np.array([g(lib.refshim_dequant4_mf(c), 96).reshape(6, 16) for c in range(4)]),
        "dequant8_mf": np.array([g(lib.refshim_dequant8_mf(c), 384).reshape(6, 64) for c in range(2)]),
    }
    return cqm


def main():
    check = "--check" in sys.argv
    ref = load_ref()
    cqm = ref_cqm(ref)
    gold_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold_dir, exist_ok=True)
    np.savez_compressed(os.path.join(gold_dir, "cqm_flat.npz"), **cqm)
    ora = C.CDLL(os.path.join(HERE, "liboracle.so")) if check else None
    bad_total = 0
    for seed in SEEDS:
        inp = harness.make_inputs(seed, cqm)
        for interlaced in (0, 1):
            fams = None if interlaced == 0 else ("dct",)
            got = harness.run_all(TableSet(ref, "ref", interlaced), inp, fams)
            if interlaced:
                got = {k: v for k, v in got.items() if k.startswith("zigzag.")}
            name = "l1_seed%d_i%d.npz" % (seed, interlaced)
            payload = {"out." + k: v for k, v in got.items()}
            if not interlaced:
                payload.update({"in." + k: v for k, v in inp.items() if not k.startswith("cqm.")})
            np.savez_compressed(os.path.join(gold_dir, name), **payload)
            print("wrote", name, len(got), "cases")
            if check:
                mine = harness.run_all(TableSet(ora, "oracle", interlaced), inp, fams)
                if interlaced:
                    mine = {k: v for k, v in mine.items() if k.startswith("zigzag.")}
                bad = harness.compare(got, mine)
                bad_total += len(bad)
                print("  oracle vs reference: %d mismatching cases %s" % (len(bad), bad[:12]))
    if check and bad_total:
        sys.exit(1)


if __name__ == "__main__":
    main()
