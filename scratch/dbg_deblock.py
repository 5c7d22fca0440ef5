"""Developer tool: first frame of the refs3 golden chain -- compare the loop filter run on the sweep's
state arrays (layout 1) with the compact layout (0) and with the golden planes."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle.gen_golden_slice import CASES, case_inputs
from x264_vs2008_amd import lib as Lm, slice as sl
from x264_vs2008_amd.frame import DeblockParams, DeviceArray

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
lib = Lm.load()
with np.load(os.path.join(GOLDEN, "cqm_flat.npz")) as z:
    cqm = {k: z[k] for k in z.files}
name, size, frames, kind, kw = next(c for c in CASES if c[0] == "refs3")
with np.load(os.path.join(GOLDEN, "slice_%s.npz" % name)) as z:
    gold = {k: z[k] for k in z.files}
y, u, v = case_inputs(size, frames, kind)
enc = sl.ChainEncoder(lib, size[0], size[1], cqm, **kw)
enc.upload(y[0], u[0], v[0])
stype, qp, state = enc.encode_frame()
enc.status()
recon = enc.last[0]
rec = {nm: enc.ctx.download(recon, nm, padded=False) for nm in ("y", "u", "v")}
print("rec ok", all(np.array_equal(rec[nm], gold["rec_" + nm][0]) for nm in rec))
s = state.st
c = enc.ctx
for layout in (1, 0):
    c.upload(recon, rec["y"][:size[1], :size[0]], rec["u"][:size[1] // 2, :size[0] // 2], rec["v"][:size[1] // 2, :size[0] // 2])
    if layout == 1:
        dp = DeblockParams(mb_type=s.mb_type, qp=s.qp, nnz=s.nnz, transform8x8=s.t8, mv=s.mv, ref=s.ref, alpha_c0_offset=0, beta_offset=0,
                           chroma_qp_offset=0, state_layout=1)
    else:
        t = state.get("mb_type")[0]
        tt = np.where(t <= 3, 1, np.where(t == 6, 2, 0)).astype(np.uint8)
        nz = state.get("nnz")[0]
        nz26 = np.concatenate([nz[:, :24], nz[:, 25:27]], axis=1).astype(np.uint8)
        bufs = [DeviceArray(lib, tt.shape, np.uint8, tt), DeviceArray(lib, nz26.shape, np.uint8, nz26)]
        dp = DeblockParams(mb_type=bufs[0].ptr, qp=s.qp, nnz=bufs[1].ptr, transform8x8=s.t8, mv=s.mv, ref=s.ref, alpha_c0_offset=0,
                           beta_offset=0, chroma_qp_offset=0, state_layout=0)
    c.check(lib.x264hip_deblock_frame(c.h, C.byref(recon), C.byref(dp)), "deblock")
    c.sync()
    for nm in ("y", "u", "v"):
        got = c.download(recon, nm, padded=False)
        want = gold["fin_" + nm][0]
        d = np.argwhere(got != want)
        print("layout", layout, nm, "diffs", len(d), d[:6].tolist())
