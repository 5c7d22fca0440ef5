import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from x264_vs2008_amd import lib as L
import look_util as U, look_cases as K
lib = L.open_library()
seed = int(sys.argv[1])
c = K.config(seed); print(c)
a = K.reference_records(c); ref = K.records_of_reference(a, c['frames'])
y, u, v = K.clip(c['w'], c['h'], c['frames'], c['cut'], c['t0'], c['slow'])
look = U.CpuLook(lib, c['w'], c['h'], c['me'], 16, c['weightb'], c['bframe_bias'], c['bframes'])
log = []
mine = U.run_chain(lib, K.lookahead_params(c), look, y, u, v, c['frames'], log)
print(log)
for i, (m, r) in enumerate(zip(mine, ref)):
    for l in (0, 1):
        a_, b_ = m[6 + l], r['mv%d' % l]
        if a_ is not None and b_ is not None and not np.array_equal(a_, b_):
            idx = np.nonzero((a_ != b_).any(1))[0]
            print('coded', i, 'input', m[0], 'list', l, 'mb', idx, 'mine', a_[idx], 'ref', b_[idx], 'mb_w', (c['w'] + 15) // 16)
