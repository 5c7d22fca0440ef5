import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from x264_vs2008_amd import lib as L
import look_util as U, look_cases as K
lib = L.open_library()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
for seed in range(lo, hi):
    c = K.config(seed)
    t = time.time()
    try:
        a = K.reference_records(c)
    except RuntimeError as ex:
        print(seed, 'reference refused', ex, c); continue
    ref = K.records_of_reference(a, c['frames'])
    y, u, v = K.clip(c['w'], c['h'], c['frames'], c['cut'], c['t0'], c['slow'])
    look = U.CpuLook(lib, c['w'], c['h'], c['me'], 16, c['weightb'], c['bframe_bias'], c['bframes'])
    log = []
    mine = U.run_chain(lib, K.lookahead_params(c), look, y, u, v, c['frames'], log)
    bad = K.compare(mine, ref)
    types = ''.join('IPB'[{2:0,0:1,1:2}[r['slice']]] for r in ref)
    print(seed, 'OK' if not bad else 'BAD', types, len(log), 'tasks', '%.1fs' % (time.time() - t), {k: c[k] for k in ('bframes', 'b_adapt', 'pre_scenecut', 'cut', 'keyint', 'crf')})
    for b in bad[:6]: print('   ', b)
