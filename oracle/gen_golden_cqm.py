"""Generate tests/golden/cqm_jvt.npz: the quantiser tables of --cqm jvt, straight from the reference's x264_cqm_init
(oracle/ref_shim.c: refshim_cqm_init_preset).  Test infrastructure."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def ref_tables(lib, preset):
    for n in ("quant4_mf", "quant4_bias", "quant8_mf", "quant8_bias"):
        f = getattr(lib, "refshim_cq_" + n); f.restype = C.POINTER(C.c_uint16); f.argtypes = [C.c_int, C.c_int]
    for n in ("dequant4_mf", "dequant8_mf"):
        f = getattr(lib, "refshim_cq_" + n); f.restype = C.POINTER(C.c_int); f.argtypes = [C.c_int]
    assert lib.refshim_cqm_init_preset(preset) == 0
    g = lambda p, n: np.ctypeslib.as_array(p, shape=(n,)).copy()
    return {
        "quant4_mf": np.array([[g(lib.refshim_cq_quant4_mf(c, q), 16) for q in range(52)] for c in range(4)]),
        "quant4_bias": np.array([[g(lib.refshim_cq_quant4_bias(c, q), 16) for q in range(52)] for c in range(4)]),
        "quant8_mf": np.array([[g(lib.refshim_cq_quant8_mf(c, q), 64) for q in range(52)] for c in range(2)]),
        "quant8_bias": np.array([[g(lib.refshim_cq_quant8_bias(c, q), 64) for q in range(52)] for c in range(2)]),
        "dequant4_mf": np.array([g(lib.refshim_cq_dequant4_mf(c), 96).reshape(6, 16) for c in range(4)]),
        "dequant8_mf": np.array([g(lib.refshim_cq_dequant8_mf(c), 384).reshape(6, 64) for c in range(2)]),
    }


def main():
    from oracle import refslice as rs
    lib = rs.reference_lib()                    # lazy binding: the library leaves encoder.c's symbols undefined
    flat = ref_tables(lib, 0)
    with np.load(os.path.join(ROOT, "tests", "golden", "cqm_flat.npz")) as z:
        assert all(np.array_equal(flat[k], z[k]) for k in z.files), "preset 0 must reproduce cqm_flat.npz"
    jvt = ref_tables(lib, 1)
    path = os.path.join(ROOT, "tests", "golden", "cqm_jvt.npz")
    np.savez_compressed(path, **jvt)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
