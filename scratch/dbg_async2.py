import sys, ctypes as C, traceback
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from x264_vs2008_amd import lib as L
import look_cases as K, test_gpu_stream as T
from x264_vs2008_amd import frame as F
hip = L.load(0)
orig = F.FrameCtx.upload
def upload(self, pic, y, u, v, b=None):
    print('upload b', b, 'ctx', hex(self.h.value), 'pic plane0', hex(pic.plane[0] or 0), y.shape, y.flags['C_CONTIGUOUS'], flush=True)
    return orig(self, pic, y, u, v, b)
F.FrameCtx.upload = upload
cs = T.chains("badapt1_crf_aq", T.SEEDS["badapt1_crf_aq"])
try:
    got, sizes = T.run_async(hip, cs)
    print('sizes', sizes)
except Exception:
    traceback.print_exc()
