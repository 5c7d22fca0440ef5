"""Developer tool: the B kernel against a golden B chain; prints the first differing macroblock of the first differing frame."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle.gen_golden_slice import CASES2, case_inputs
from x264_vs2008_amd import slice as sl, lib as L
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STATE = ["mb_type", "partition", "sub_partition", "ref", "mv", "i4mode", "i16mode", "chroma_mode", "qp", "cbp", "t8", "nnz", "luma", "luma_dc", "chroma_dc", "chroma_ac"]
import ctypes as C
def get1(enc, state, name, shape, dt):
    out = np.zeros(shape, dt)
    assert enc.ctx.lib.x264hip_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(getattr(state.st, name)), C.c_size_t(out.nbytes)) == 0
    return out
name = sys.argv[1] if len(sys.argv) > 1 else "b_medium"
_, size, frames, kind, kw, ekw = next(c for c in CASES2 if c[0] == name)
with np.load(os.path.join(ROOT, "tests", "golden", "slice2_%s.npz" % name)) as z:
    gold = {k: z[k] for k in z.files}
with np.load(os.path.join(ROOT, "tests", "golden", "cqm_flat.npz")) as z:
    cqm = {k: z[k] for k in z.files}
hip = L.load(0)
y, u, v = case_inputs(size, frames, kind)
kw = dict(kw); kw.pop("cqm_preset", 0)
enc = sl.ChainEncoder(hip, size[0], size[1], cqm, batch=1, write=1, **kw, **{k: v_ for k, v_ in ekw.items() if k != "write"})
order = sl.coding_order(frames, kw.get("keyint", 0), ekw["bframes"])
n = gold["mb_type"].shape[1]
for f, (disp, stype) in enumerate(order):
    enc.upload(y[disp], u[disp], v[disp])
    st, qp, state = enc.encode_frame(stype=stype, disp=disp)
    enc.status()
    got = {k: state.get(k)[0] for k in STATE}
    got["mv1"] = get1(enc, state, "mv1", (n, 16, 2), np.int16); got["ref1"] = get1(enc, state, "ref1", (n, 4), np.int8)
    bad = {}
    for k in STATE + (["mv1", "ref1"] if stype == sl.SLICE_B else []):
        if stype == sl.SLICE_I and k in ("mv", "ref"): continue
        w = gold[k][f]
        if not np.array_equal(got[k], w): bad[k] = int(np.argwhere(got[k].reshape(n, -1) != w.reshape(n, -1))[0][0])
    pay = enc.payloads()[0]; wantp = bytes(gold["payload"][f, :gold["payload_len"][f]])
    print("frame", f, "disp", disp, "type", stype, "bad" if bad else "ok", bad, "payload", "ok" if pay == wantp else "DIFF %d vs %d" % (len(pay), len(wantp)), flush=True)
    if bad:
        mb = min(bad.values())
        for k in ("mb_type", "partition", "sub_partition", "ref", "ref1", "cbp", "t8", "qp"):
            print("  ", k, "got", got[k][mb].tolist(), "want", gold[k][f][mb].tolist())
        for k in ("mv", "mv1"):
            print("  ", k, "got", got[k][mb][[0, 3, 12, 15]].tolist(), "want", gold[k][f][mb][[0, 3, 12, 15]].tolist())
        print("   first bad mb", mb, "of", n, "mb_w", (size[0] + 15) // 16)
        break
    enc.finish_frame(); enc.ctx.sync()
enc.close()
