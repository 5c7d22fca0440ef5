import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.gen_golden_slice import CASES, case_inputs
from x264_vs2008_amd import lib as L, slice as sl
name = sys.argv[1] if len(sys.argv) > 1 else "sub8x8"
nm, size, frames, kind, kw = next(c for c in CASES if c[0] == name)
gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "slice_%s.npz" % name)))
with np.load(os.path.join(ROOT, "tests", "golden", "cqm_flat.npz")) as z:
    cqm = {k: z[k] for k in z.files}
hip = L.load()
y, u, v = case_inputs(size, frames, kind)
enc = sl.ChainEncoder(hip, size[0], size[1], cqm, batch=1, **kw)
for f in range(2):
    enc.upload(y[f], u[f], v[f], b=0)
    stype, qp, state = enc.encode_frame(); enc.status()
    g = {k: state.get(k)[0] for k in ("mb_type", "partition", "sub_partition", "mv", "ref", "cost_inter", "cost_intra", "cbp", "t8")}
    enc.finish_frame(); enc.ctx.sync()
    if f == 0: continue
    bad = np.argwhere(g["mb_type"] != gold["mb_type"][f]).ravel()
    print("frame", f, "type diffs", len(bad), bad[:10].tolist())
    for mb in list(bad[:4]) + [int(m) for m in np.argwhere((g["mb_type"] == gold["mb_type"][f]) & (gold["mb_type"][f] == 5)).ravel()[:2]]:
        print("mb", mb, "gpu type/part/sub", g["mb_type"][mb], g["partition"][mb], g["sub_partition"][mb].tolist(), "cost inter/intra", g["cost_inter"][mb], g["cost_intra"][mb],
              "| gold", gold["mb_type"][f][mb], gold["partition"][f][mb], gold["sub_partition"][f][mb].tolist())
        print("   gpu mv", g["mv"][mb].reshape(16, 2).tolist()); print("   gold mv", gold["mv"][f][mb].reshape(16, 2).tolist())
enc.close()
