import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import refslice as R
from x264_vs2008_amd import lib as L, lookahead as LA
import look_util as U
w, h, F = 128, 96, 12
kw = dict(bframes=3, b_adapt=1, crf=23.0)
if len(sys.argv) > 1: w, h, F = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p = R.make_params(w, h, F, qp=26, me_method=R.ME_HEX, subme=5, n_refs=2, inter=0x33, intra=0x3, transform8x8=1, cabac=1, deblock=1, keyint=250)
e = R.make_ext(bframes=3, b_adapt=1, pre_scenecut=1, scenecut_threshold=40, crf=23.0, weightb=1)
y, u, v = R.clip(w, h, F)
a = R.run_reference_stream(p, e, y, u, v)
lib = L.open_library()
prm = LA.make_params((w+15)//16, (h+15)//16, bframes=3, b_adapt=1, keyint_max=250, crf=23.0, qp_min=0)
look = U.CpuLook(lib, w, h, R.ME_HEX, 16, 1, 0, 3)
log = []
out = U.run_chain(lib, prm, look, y, u, v, F, log)
for i, o in enumerate(out):
    ref = (int(a['frame_info2'][i][0]), int(a['frame_info'][i][0]), int(a['frame_info'][i][1]))
    print(o[0], o[1], o[2], o[3], o[8], '| ref', ref, a['rc_info'][i], a['look_cost'][i][:4])
    lm = a['look_mv'][i]
    for l in (0, 1):
        mine = o[6 + l]
        refv = None if lm[l, 0, 0] == 0x7fff else lm[l]
        if (mine is None) != (refv is None): print('  lowres presence differs', l, mine is None, refv is None)
        elif mine is not None and not np.array_equal(mine, refv): print('  lowres mv differ list', l, int((mine != refv).any(1).sum()))
print(len(log), 'tasks', log[:12])
