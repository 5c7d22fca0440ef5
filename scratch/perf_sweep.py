"""Developer tool: time the macroblock sweep at 1080p for a few configurations / batch sizes."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x264_vs2008_amd import lib as Lm, slice as sl, synth

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
lib = Lm.load()
with np.load(os.path.join(GOLDEN, "cqm_flat.npz")) as z:
    cqm = {k: z[k] for k in z.files}
W, H = 1920, 1080
frames = [synth.frame(W, H, t) for t in range(5)]
cfgs = {"uf": dict(qp=26, subme=0), "hex5r3": dict(qp=26, subme=5, me_method=1, n_refs=3, cabac=1, deblock=1),
        "hex2r1": dict(qp=26, subme=2, me_method=1, n_refs=1),
        "bench": dict(qp=26, subme=5, me_method=1, n_refs=3, cabac=1, deblock=1, inter=0x13, intra=0x3, transform8x8=1, mixed_refs=1),
        "p16": dict(qp=26, subme=5, me_method=1, n_refs=3, cabac=1, deblock=1, inter=0x3, intra=0x3, transform8x8=1),
        "med": dict(qp=26, subme=5, me_method=1, n_refs=3, cabac=1, deblock=1, inter=0x3, intra=0x3, transform8x8=1)}
which = sys.argv[1].split(",") if len(sys.argv) > 1 else list(cfgs)
batches = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 8]
for name in which:
    for B in batches:
        enc = sl.ChainEncoder(lib, W, H, cqm, batch=B, **cfgs[name])
        from x264_vs2008_amd.frame import DeviceArray
        enc.profile = DeviceArray(lib, (B, enc.ctx.dims.mb_h, 8), np.int64) if os.environ.get('SW_PROF', '1') == '1' else None
        ev = [lib.x264hip_event_create() for _ in range(3)]
        lib.x264hip_event_create.restype = C.c_void_p
        lib.x264hip_event_elapsed_ms.restype = C.c_float
        ev = [C.c_void_p(lib.x264hip_event_create()) for _ in range(3)]
        for t in range(5):
            for b in range(B):
                enc.upload(*frames[(t + b) % 5] if t else frames[0], b=b)
            enc.ctx.sync()
            lib.x264hip_event_record(ev[0], C.c_void_p(enc.ctx.stream))
            stype, qp, state = enc.encode_frame()
            lib.x264hip_event_record(ev[1], C.c_void_p(enc.ctx.stream))
            enc.finish_frame()
            lib.x264hip_event_record(ev[2], C.c_void_p(enc.ctx.stream))
            enc.status()
            enc.ctx.sync()
            ms_s = lib.x264hip_event_elapsed_ms(ev[0], ev[1]); ms_f = lib.x264hip_event_elapsed_ms(ev[1], ev[2])
            ty = state.get("mb_type")[0]
            pr = (enc.profile.get()[0].astype(np.float64) if enc.profile else np.zeros((enc.ctx.dims.mb_h, 8))) / 100.0 / enc.ctx.dims.mb_w   # us per macroblock
            print("   us/MB (mean over rows): wait %.1f load %.1f inter %.1f intra %.1f encode %.1f store %.1f publish %.1f | row0 wait %.1f" % (
                pr[:, 0].mean(), pr[:, 1].mean(), pr[:, 2].mean(), pr[:, 6].mean(), pr[:, 3].mean(), pr[:, 4].mean(), pr[:, 5].mean(), pr[0, 0]))
            print("%s B=%d frame %d type %d: sweep %.2f ms, filter %.2f ms -> %.1f fps  (types %s)" % (
                name, B, t, stype, ms_s, ms_f, 1000.0 * B / (ms_s + ms_f), np.bincount(ty, minlength=7).tolist()), flush=True)
        enc.close()
