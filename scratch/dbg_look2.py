import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from x264_vs2008_amd import lib as L
import look_util as U, look_cases as K
lib = L.open_library()
keep = {}
for rep in range(2):
  for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    c = K.config(seed)
    a = K.reference_records(c); ref = K.records_of_reference(a, c['frames'])
    y, u, v = K.clip(c['w'], c['h'], c['frames'], c['cut'], c['t0'], c['slow'])
    look = U.CpuLook(lib, c['w'], c['h'], c['me'], 16, c['weightb'], c['bframe_bias'], c['bframes'])
    mine = U.run_chain(lib, K.lookahead_params(c), look, y, u, v, c['frames'])
    for i, (m, r) in enumerate(zip(mine, ref)):
        for l in (0, 1):
            a_, b_ = m[6 + l], r['mv%d' % l]
            key = (seed, i, l)
            if a_ is None or b_ is None: continue
            if key in keep:
                pm, pr = keep[key]
                if not np.array_equal(pm, a_): print('MINE varies', key, np.nonzero((pm != a_).any(1))[0])
                if not np.array_equal(pr, b_): print('REF varies', key, np.nonzero((pr != b_).any(1))[0], pr[(pr != b_).any(1)], b_[(pr != b_).any(1)])
            else: keep[key] = (a_.copy(), b_.copy())
            if not np.array_equal(a_, b_):
                idx = np.nonzero((a_ != b_).any(1))[0]
                print('rep', rep, 'seed', seed, 'coded', i, 'list', l, 'mb', idx, 'mine', a_[idx].tolist(), 'ref', b_[idx].tolist(), 'mb_w', (c['w'] + 15) // 16, 'mb_h', (c['h']+15)//16)
