"""NOT THE HOT PATH: the frame-level entry points of round 1 (exhaustive ME surfaces, residual for given vectors, filters) strung
together for tests/test_gpu_pipeline.py and as a usage example of INTEGRATION.md section 3.  The macroblock loop itself -- analysis,
mode decision, encode, entropy coding -- is slice.py (ChainEncoder) over x264hip_slice_sweep_frame; bench.py times that.

The per-frame pass on the GPU, in the order the reference's frame
loop reaches these operations (R/encoder/encoder.c:1406-1421 lowres + AQ,
R/encoder/analyse.c:2228 motion search per reference, R/encoder/macroblock.c:
596-768 residual, R/encoder/encoder.c:983-1057 deblock + border + half-pel
planes + SSD).  Everything is enqueued on the context's stream; nothing here
synchronises with the host.

A context created with batch = B carries B independent frames (one per GOP
chain) through every launch, so one call sequence advances B chains by one
frame each.

The pass decides nothing the reference decides serially (mode decision,
entropy coding): it produces, for every macroblock at once, the arithmetic
those decisions consume -- costs and vectors per reference, residual levels,
reconstruction.  See DESIGN.md "What the frame pass is and is not".
"""
import ctypes as C

import numpy as np

from .frame import (CqmDevice, DeblockParams, DeviceArray, FrameCtx, MeParams, chroma_qp, cost_mv_table)

LAMBDA_TAB = (1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6,
              6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25, 29, 32, 36, 40, 45, 51, 57, 64, 72, 81, 91)
COST_SPAN = 4 * 2048          # the reference's p_cost_mv span: +-2*4*2048 qpel (analyse.c:191-198); half is ample here


class PFramePass:
    def __init__(self, lib, ctx, cqm, qp=26, me_range=16, transform8x8=1, n_refs=3):
        self.lib, self.ctx, self.qp, self.t8, self.n_refs = lib, ctx, qp, transform8x8, n_refs
        d = ctx.dims
        n = self.n = d.mb_w * d.mb_h
        B = self.B = ctx.batch
        self.me_range = me_range
        self.cost_tab = cost_mv_table(LAMBDA_TAB[qp], COST_SPAN)
        self.cost_dev = DeviceArray(lib, self.cost_tab.shape, np.uint16, self.cost_tab)
        self.cqm = CqmDevice(lib, cqm)
        self.aq = DeviceArray(lib, (B, n), np.int32)
        self.mv9 = [DeviceArray(lib, (B, n, 9, 2), np.int16) for _ in range(n_refs)]
        self.cost9 = [DeviceArray(lib, (B, n, 9), np.int32) for _ in range(n_refs)]
        self.mvq = [DeviceArray(lib, (B, n, 2), np.int16) for _ in range(n_refs)]
        self.costq = [DeviceArray(lib, (B, n), np.int32) for _ in range(n_refs)]
        self.levels_y = DeviceArray(lib, (B, n, 256), np.int16)
        self.levels_c = DeviceArray(lib, (B, n, 128), np.int16)
        self.dc_c = DeviceArray(lib, (B, n, 8), np.int16)
        self.cbp = DeviceArray(lib, (B, n), np.int32)
        self.nnz = DeviceArray(lib, (B, n, 26), np.uint8)
        self.mv16 = DeviceArray(lib, (B, n, 16, 2), np.int16)
        self.refi = DeviceArray(lib, (B, n, 4), np.int8)
        self.mb_type = DeviceArray(lib, (B, n), np.uint8)
        self.qp_arr = DeviceArray(lib, (B, n), np.uint8, np.full((B, n), qp, np.uint8))
        self.t8_arr = DeviceArray(lib, (B, n), np.uint8, np.full((B, n), transform8x8, np.uint8))
        self.ssd = DeviceArray(lib, (B, 3), np.uint64)
        self.me_p = MeParams(range=me_range, cost_mv=self.cost_dev.ptr, cost_mv_range=COST_SPAN, centers=None, mvp=None,
                             sad_surface=None, mv_range=512)
        self.res_p = self.cqm.params(qp, transform8x8, 0)
        self.res_p.mv4x4_out = self.mv16.ptr
        self.res_p.ref_out = self.refi.ptr
        self.db_p = DeblockParams(mb_type=self.mb_type.ptr, qp=self.qp_arr.ptr, nnz=self.nnz.ptr, transform8x8=self.t8_arr.ptr,
                                  mv=self.mv16.ptr, ref=self.refi.ptr, alpha_c0_offset=0, beta_offset=0, chroma_qp_offset=0)
        self.me_events = None      # optional list collecting (start, stop) HIP events around the full-pel kernel

    def make_reference(self, pic):
        """Borders + half-pel planes: what x264_fdec_filter_row leaves for a kept reference."""
        L, c = self.lib, self.ctx
        c.check(L.x264hip_expand_border(c.h, C.byref(pic), 0), "expand_border")
        c.check(L.x264hip_hpel_filter_frame(c.h, C.byref(pic)), "hpel_filter_frame")

    def step(self, cur, refs, recon):
        """Enqueue one P-frame pass for every batch element: `cur` source picture, `refs` list of
        reference pictures (nearest first), `recon` picture that receives the reconstruction."""
        L, c = self.lib, self.ctx
        h = c.h
        c.check(L.x264hip_lowres_init_frame(h, C.byref(cur)), "lowres_init_frame")
        c.check(L.x264hip_aq_var_frame(h, C.byref(cur), self.aq.p), "aq_var_frame")
        for i, ref in enumerate(refs):
            if self.me_events is not None:
                e0, e1 = L.x264hip_event_create(), L.x264hip_event_create()
                L.x264hip_event_record(C.c_void_p(e0), C.c_void_p(c.stream))
            c.check(L.x264hip_me_fullpel_frame(h, C.byref(cur), C.byref(ref), C.byref(self.me_p), self.mv9[i].p, self.cost9[i].p),
                    "me_fullpel_frame")
            if self.me_events is not None:
                L.x264hip_event_record(C.c_void_p(e1), C.c_void_p(c.stream))
                self.me_events.append((e0, e1))
            c.check(L.x264hip_me_subpel_frame(h, C.byref(cur), C.byref(ref), C.byref(self.me_p), self.mv9[i].p, self.mvq[i].p,
                                              self.costq[i].p), "me_subpel_frame")
        c.check(L.x264hip_inter_residual_frame(h, C.byref(cur), C.byref(refs[0]), C.byref(recon), C.byref(self.res_p),
                                               self.mvq[0].p, self.levels_y.p, self.levels_c.p, self.dc_c.p, self.cbp.p, self.nnz.p),
                "inter_residual_frame")
        c.check(L.x264hip_deblock_frame(h, C.byref(recon), C.byref(self.db_p)), "deblock_frame")
        self.make_reference(recon)
        c.check(L.x264hip_ssd_frame_async(h, C.byref(cur), C.byref(recon), self.ssd.p), "ssd_frame_async")

    def results(self, b=0):
        """Copy the last pass's arrays for batch element b to the host (synchronises)."""
        self.ctx.sync()
        g = lambda a: a.get()[b]
        return {"aq": g(self.aq), "mv9": [g(a) for a in self.mv9], "cost9": [g(a) for a in self.cost9],
                "mvq": [g(a) for a in self.mvq], "costq": [g(a) for a in self.costq],
                "levels_y": g(self.levels_y), "levels_c": g(self.levels_c), "dc_c": g(self.dc_c),
                "cbp": g(self.cbp), "nnz": g(self.nnz), "ssd": g(self.ssd).astype(np.int64)}


def setup_event_api(lib):
    lib.x264hip_event_create.restype = C.c_void_p
    lib.x264hip_event_elapsed_ms.restype = C.c_float
