"""Developer tool (GPU box): run one CASES2 chain through the raster sweep and the CPU twin; report the first macroblock whose
bit position differs, and state differences."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import refslice as rs
from oracle.gen_golden_slice import CASES2, case_inputs
from x264_vs2008_amd import lib as L
from test_gpu_slice_rd import run_chain2
from test_gpu_slice import STATE
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
ora = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
hip = L.load()
with np.load(os.path.join(ROOT, "tests", "golden", "cqm_flat.npz")) as z:
    cqm = {k: z[k] for k in z.files}
names = sys.argv[1:] or [c[0] for c in CASES2]
for name, size, frames, kind, kw, ekw in CASES2:
    if name not in names:
        continue
    y, u, v = case_inputs(size, frames, kind)
    p = rs.make_params(size[0], size[1], frames, **kw)
    tw = rs.run2(ora, "x264o_encode_chain2", p, rs.make_ext(**ekw), y, u, v)
    out = run_chain2(hip, cqm, size, frames, y, u, v, kw, ekw)
    ok = True
    for f in range(frames):
        n = int(tw["payload_len"][f])
        want = bytes(tw["payload"][f, :n])
        bad = [k for k in STATE if not np.array_equal(out[f][k][0].reshape(tw[k][f].shape), tw[k][f])]
        bits_g, bits_t = out[f]["mb_bits"][0], tw["mb_bits"][f]
        d = np.argwhere(bits_g != bits_t)
        same = out[f]["payload"][0] == want
        if bad or len(d) or not same:
            ok = False
            mb = int(d[0][0]) if len(d) else -1
            print(name, "frame", f, "state diffs:", bad, "first mb_bits diff at mb", mb, "payload", "same" if same else "DIFF", len(out[f]["payload"][0]), n)
            for k in bad[:6]:
                a, b = out[f][k][0].reshape(tw[k][f].shape), tw[k][f]
                i = np.argwhere(a != b)[0]
                print("    ", k, "first at", i.tolist(), "gpu", a[tuple(i)], "twin", b[tuple(i)])
            if mb >= 0:
                print("     mb", mb, "type", tw["mb_type"][f][mb], "part", tw["partition"][f][mb], "cbp", hex(tw["cbp"][f][mb]), "t8", tw["t8"][f][mb], "qp", tw["qp"][f][mb],
                      "bits gpu", bits_g[max(mb - 1, 0):mb + 2].tolist(), "twin", bits_t[max(mb - 1, 0):mb + 2].tolist())
                print("     nnz", tw["nnz"][f][mb].tolist(), "i4mode", tw["i4mode"][f][mb].tolist(), "chroma_mode", tw["chroma_mode"][f][mb], "i16", tw["i16mode"][f][mb])
                if mb > 0:
                    print("     left type", tw["mb_type"][f][mb - 1], "cbp", hex(tw["cbp"][f][mb - 1]))
            break
    print(name, "OK" if ok else "FAILED", flush=True)
