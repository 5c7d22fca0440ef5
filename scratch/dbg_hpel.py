import ctypes as C, sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from x264_vs2008_amd import lib as L
from x264_vs2008_amd.frame import FrameCtx
from frame_util import make_clip_frame
hip = L.load(0)
ora = C.CDLL('/root/repo/oracle/liboracle.so')
ctx = FrameCtx(hip, 352, 288)
pic = ctx.new_picture()
hp = make_clip_frame(ctx, pic, 5, ora)
hip.x264hip_expand_border(ctx.h, C.byref(pic), 0)
hip.x264hip_hpel_filter_frame(ctx.h, C.byref(pic))
_, stride, w, h, padh, padv = hp.full["y"]
ora.x264o_plane_expand_border(hp.ptr("y"), stride, w, h, padh, padv)
ora.x264o_frame_hpel(hp.ptr("y"), hp.ptr("h"), hp.ptr("vv"), hp.ptr("c"), stride, w, h, ctx.dims.mb_h)
got = ctx.download(pic, "c").astype(int); want = hp.arr("c").astype(int)
bad = np.argwhere(got != want)
print(len(bad))
ys = bad[:,0]-32; xs = bad[:,1]-32
print("y range", ys.min(), ys.max(), "x range", xs.min(), xs.max())
print("x mod 64 hist", np.bincount((xs+4) % 64, minlength=64))
print("y mod 16 hist", np.bincount((ys+8) % 16, minlength=16))
d = (got-want)[got!=want]
print("diff hist", np.unique(d, return_counts=True))
for b in bad[:10]: print(b-32, got[tuple(b)], want[tuple(b)])
