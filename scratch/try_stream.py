import sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import refslice as R
w, h, F = 128, 96, 12
p = R.make_params(w, h, F, qp=26, me_method=R.ME_HEX, subme=5, n_refs=2, inter=0x33, intra=0x3, transform8x8=1, cabac=1, deblock=1, keyint=250)
e = R.make_ext(bframes=3, b_adapt=1, pre_scenecut=1, scenecut_threshold=40, crf=23.0, weightb=1)
y, u, v = R.clip(w, h, F)
a = R.run_reference_stream(p, e, y, u, v)
print(a['frame_info']); print(a['frame_info2']); print(a['rc_info']); print(a['look_cost']); print(a['payload_len'])
