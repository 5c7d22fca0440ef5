"""Developer tool (GPU box): frames/s of the raster sweep at 1080p for a few batch sizes."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x264_vs2008_amd import lib as L, synth
from x264_vs2008_amd import slice as sl
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hip = L.load()
with np.load(os.path.join(ROOT, "tests", "golden", "cqm_flat.npz")) as z:
    cqm = {k: z[k] for k in z.files}
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
opts = dict(qp=26, me_method=1, me_range=16, subme=int(os.environ.get("SUBME", 7)), n_refs=3, fast_pskip=1, dct_decimate=1, chroma_me=1, cabac=1, deblock=1,
            keyint=0, inter=0x13, intra=0x3, transform8x8=1, mixed_refs=1, trellis=int(os.environ.get("TRELLIS", 1)), psy_rd=float(os.environ.get("PSY", 1.0)),
            aq_mode=int(os.environ.get("AQ", 1)), write=1)
BFR = int(os.environ.get("BFR", 0))                 # B frames between anchors (coding order: I P B.. P B..)
if BFR:
    opts.update(bframes=BFR, weightb=1, direct_pred=1, inter=0x113)
frames = [synth.frame(W, H, i) for i in range(4)]
for B in [int(x) for x in os.environ.get("BATCHES", "64,256").split(",")]:
    enc = sl.ChainEncoder(hip, W, H, cqm, batch=B, **opts)
    ctx = enc.ctx
    srcs = []
    for i in range(4):
        pic = ctx.new_picture()
        for b in range(B):
            ctx.upload(pic, *frames[(i + b) % 4], b=b)
        srcs.append(pic)
    from x264_vs2008_amd.frame import DeviceArray
    if os.environ.get('PROF'):
        enc.profile = DeviceArray(hip, (B, ctx.dims.mb_h, 8), np.int64)
    order = sl.coding_order(2 + 2 * (BFR + 1), 0, BFR) if BFR else None
    if BFR:
        while len(srcs) < len(order):
            pic = ctx.new_picture(source_only=True)
            for b in range(B):
                ctx.upload(pic, *synth.frame(W, H, 11 * len(srcs) + (b % 4)), b=b)
            srcs.append(pic)
    for k in range(len(order) if BFR else 4):
        hip.x264hip_device_synchronize()
        t0 = time.perf_counter()
        if BFR:
            enc.encode_frame(srcs[order[k][0]], stype=order[k][1], disp=order[k][0])
        else:
            enc.encode_frame(srcs[k])
        hip.x264hip_device_synchronize()
        t1 = time.perf_counter()
        enc.finish_frame()
        hip.x264hip_device_synchronize()
        t2 = time.perf_counter()
        n = (getattr(enc, "last_bufs", None) or enc.rd_bufs)["payload_len"].get()
        if enc.profile:
            pr = enc.profile.get()[:, -1, :].astype(np.float64).mean(axis=0) / 100e6 / (ctx.dims.mb_w * ctx.dims.mb_h) * 1e6
            tot = enc.profile.get()[:, -1, :].astype(np.float64).sum(axis=1) / 100e6      # seconds per chain (accumulated over the rows)
            print('   chain seconds: min %.3f mean %.3f p90 %.3f max %.3f' % (tot.min(), tot.mean(), np.percentile(tot, 90), tot.max()))
            print('   us/MB: trial-encode %.1f load %.1f inter-ME %.1f final-encode %.1f stores %.1f entropy-write %.1f analysis(rest) %.1f trial-ssd+bits %.1f | total %.1f' % (pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7], pr.sum()))
            enc.profile.set(np.zeros((B, ctx.dims.mb_h, 8), np.int64))
        print("B %d frame %d (%s): sweep %.3f s, filters %.3f s -> %.1f frames/s; payload bytes mean %.0f" % (B, k, ("PBI"[order[k][1]] if BFR else "I" if k == 0 else "P"), t1 - t0, t2 - t1, B / (t2 - t0), n.mean()), flush=True)
    enc.close()
