import sys, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from x264_vs2008_amd import lib as L
from x264_vs2008_amd.frame import cqm_init, FrameCtx, DeviceArray
import look_cases as K, test_gpu_stream as T
hip = L.load(0)
def probe(tag):
    ctx = FrameCtx(hip, 128, 96, batch=1)
    pic = ctx.new_picture(source_only=True)
    y = np.zeros((96,128),np.uint8); u = np.zeros((48,64),np.uint8)
    try:
        ctx.upload(pic, y, u, u, b=0); print(tag, 'ok')
    except RuntimeError as e:
        print(tag, 'STALE ERROR', e)
    ctx.close()
probe('start')
cs = T.chains("badapt1_crf_aq", T.SEEDS["badapt1_crf_aq"])
c = cs[0]
from x264_vs2008_amd.stream import AsyncStreamEncoder, StreamEncoder
hip.x264hip_event_create.restype = C.c_void_p
e = hip.x264hip_event_create(); probe('event_create')
print('query', hip.x264hip_event_query(C.c_void_p(e))); probe('event_query fresh')
hip.x264hip_host_alloc.restype = C.c_void_p
p = hip.x264hip_host_alloc(C.c_size_t(12)); probe('host_alloc 12')
enc = AsyncStreamEncoder(hip, c["w"], c["h"], cqm_init(hip), batch=3, n_frames=14, launches=3, crf=23.0, b_adapt=1, qp=26, me_method=1, subme=5, n_refs=2, inter=0x33, intra=3,
                         transform8x8=1, cabac=1, deblock=1, keyint=250, aq_mode=1, bframes=3, weightb=1, qp_min=0)
probe('after init')
enc.close()
probe('after close')
frames = 14
clips = [K.clip(c["w"], c["h"], frames, cc["cut"], cc["t0"], cc["slow"]) for cc in cs]
enc = AsyncStreamEncoder(hip, c["w"], c["h"], cqm_init(hip), batch=3, n_frames=14, launches=3, crf=23.0, b_adapt=1, qp=26, me_method=1, subme=5, n_refs=2, inter=0x33, intra=3,
                         transform8x8=1, cabac=1, deblock=1, keyint=250, aq_mode=1, bframes=3, weightb=1, qp_min=0)
probe('after init 2')
print('oldest', enc._oldest_needed()); probe('after oldest')
pic = enc.look.begin_frame(0); probe('after begin_frame')
y, u, v = clips[0]
enc.src_ctx.upload(pic, y[0], u[0], v[0], b=0); print('upload b0 ok')
enc.src_ctx.upload(pic, y[0], u[0], v[0], b=1); print('upload b1 ok')
