import sys, ctypes as C, traceback
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from x264_vs2008_amd import lib as L
from x264_vs2008_amd.frame import cqm_init
import look_cases as K, test_gpu_stream as T
from x264_vs2008_amd.stream import AsyncStreamEncoder
hip = L.load(0)
cs = T.chains("badapt1_crf_aq", T.SEEDS["badapt1_crf_aq"])
c = cs[0]
clips = [K.clip(c["w"], c["h"], 14, cc["cut"], cc["t0"], cc["slow"]) for cc in cs]
enc = AsyncStreamEncoder(hip, c["w"], c["h"], cqm_init(hip), batch=3, n_frames=14, launches=3, crf=23.0, b_adapt=1, qp=26, me_method=1, subme=5, n_refs=2, inter=0x33, intra=3,
                         transform8x8=1, cabac=1, deblock=1, keyint=250, aq_mode=1, bframes=3, weightb=1, qp_min=0)
def chk(tag):
    rc = hip.x264hip_expand_border(enc.src_ctx.h, C.byref(enc.pool[0]), 0)
    print(tag, 'rc', rc, hip.x264hip_last_error().decode() if rc else '', flush=True)
for f in range(6):
    pic = enc.look.begin_frame(f)
    for b, (y, u, v) in enumerate(clips):
        try:
            enc.src_ctx.upload(pic, y[f], u[f], v[f], b=b)
        except RuntimeError as e:
            print('f', f, 'b', b, 'upload', e); 
    chk('f%d after uploads' % f)
    en, off = enc.aq_slots[enc.look.slot(f)]
    hip.x264hip_adaptive_quant_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    rc = hip.x264hip_adaptive_quant_frame(enc.src_ctx.h, C.byref(pic), C.c_float(1.0), en.p, off.p); print('aq rc', rc)
    chk('f%d after aq' % f)
    enc.look.prepare(f)
    chk('f%d after prepare' % f)
    print('put', [la.put() for la in enc.las]); chk('f%d after puts' % f)
