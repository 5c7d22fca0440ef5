import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from x264_vs2008_amd import lib as L, lookahead as LA
import look_util as U, look_cases as K
lib = L.open_library()
worst = {}
for seed in range(0, 120):
    c = K.config(seed)
    y, u, v = K.clip(c['w'], c['h'], c['frames'], c['cut'], c['t0'], c['slow'])
    look = U.CpuLook(lib, c['w'], c['h'], c['me'], 16, c['weightb'], c['bframe_bias'], c['bframes'])
    la = LA.Lookahead(lib, K.lookahead_params(c))
    fed, span = 0, 0
    while True:
        flushing = fed >= c['frames']
        if not flushing:
            num = la.put(); look.add(num, y[num], u[num], v[num]); fed += 1
        while True:
            kind, fr, needs = la.get(flushing)
            if kind != LA.NEED: break
            for (b, p0, p1, ds0, ds1, spec) in needs:
                la.set_cost(b, p0, p1, *look.cost(b, p0, p1, ds0, ds1), speculative=spec)
        span = max(span, fed - la.oldest_live())
        if kind == LA.END: break
        if kind == LA.NONE: continue
        la.end()
    key = (c['bframes'], c['b_adapt'])
    worst[key] = max(worst.get(key, 0), span)
print(worst)
