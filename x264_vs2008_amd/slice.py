"""Host-side mirror of the macroblock sweep (include/x264hip.h: x264hip_slice_sweep_frame) and a
chain encoder around it: I frame, then P frames, every frame kept as reference -- the sequencing
x264_encoder_encode / x264_slice_write / x264_fdec_filter_row do on the host
(R/encoder/encoder.c:1316-1560, 1141-1291, 983-1056), restricted to CQP without B-frames.

The arithmetic lives in the HIP library; this file only orders launches and owns device buffers.
"""
import ctypes as C
import math

import numpy as np

from .frame import CqmDevice, DeblockParams, DeviceArray, FrameCtx

SLICE_P, SLICE_B, SLICE_I = 0, 1, 2
I_4x4, I_8x8, I_16x16, I_PCM, P_L0, P_8x8, P_SKIP = range(7)
LAMBDA_TAB = (1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6,
              6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25, 29, 32, 36, 40, 45, 51, 57, 64, 72, 81, 91)   # R/encoder/analyse.c:140-149
COST_SPAN = 2 * 4 * 2048      # p_cost_mv reaches +-2*4*2048 quarter-pels (R/encoder/analyse.c:191-198)

STATE_FIELDS = [("mb_type", np.int8, ()), ("partition", np.int8, ()), ("sub_partition", np.int8, (4,)), ("ref", np.int8, (4,)), ("i4mode", np.int8, (16,)),
                ("i16mode", np.int8, ()), ("chroma_mode", np.int8, ()), ("qp", np.int8, ()), ("t8", np.int8, ()),
                ("mv", np.int16, (16, 2)), ("mvr", np.int16, None), ("cbp", np.int16, ()), ("nnz", np.uint8, (27,)),
                ("luma", np.int16, (256,)), ("luma_dc", np.int16, (16,)), ("chroma_dc", np.int16, (8,)), ("chroma_ac", np.int16, (128,)),
                ("cost_intra", np.int32, ()), ("cost_inter", np.int32, ()), ("cost_intra_alt", np.int32, ())]


class MbState(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _, _ in STATE_FIELDS] + \
               [("progress", C.c_void_p), ("poc", C.c_int), ("n_ref0", C.c_int), ("inv_ref_poc", C.c_int * 8), ("mvd", C.c_void_p),
                ("mv1", C.c_void_p), ("ref1", C.c_void_p), ("mvr1", C.c_void_p), ("mvd1", C.c_void_p), ("skipbp", C.c_void_p),
                ("ref_poc", C.c_int * 8)]


class SliceB(C.Structure):
    """x264hip_slice_b: list 1 of a B slice and what direct prediction reads."""
    _fields_ = [("fref1", C.c_void_p), ("l1_state", C.c_void_p), ("ref1_poc", C.c_int), ("weightb", C.c_int), ("lowres_mv1", C.c_void_p),
                ("direct_spatial", C.c_int), ("direct_score", C.c_void_p)]


class SliceRd(C.Structure):
    """x264hip_slice_rd: the raster-order variant of the sweep (RD levels, trellis, adaptive quantisation, the entropy coder in the loop)."""
    _fields_ = [("trellis", C.c_int), ("psy_rd", C.c_int), ("write", C.c_int), ("cabac_init_idc", C.c_int), ("i_frame", C.c_int),
                ("qp_min", C.c_int), ("qp_max", C.c_int), ("f_qpm", C.c_float), ("aq_offset", C.c_void_p), ("cost_mv_all", C.c_void_p),
                ("unquant4_mf", C.c_void_p), ("unquant8_mf", C.c_void_p), ("payload", C.c_void_p), ("payload_cap", C.c_int),
                ("payload_len", C.c_void_p), ("mb_bits", C.c_void_p), ("stale", C.c_void_p), ("i_frame_stride", C.c_int)]


class CavlcParams(C.Structure):
    """x264hip_cavlc_params (include/x264hip_lookahead.h)"""
    _fields_ = [("slice_type", C.c_int), ("n_ref0", C.c_int), ("analyse_inter", C.c_int), ("transform8x8", C.c_int), ("cqm_custom", C.c_int),
                ("payload", C.c_void_p), ("payload_cap", C.c_int), ("payload_len", C.c_void_p), ("mb_bits", C.c_void_p), ("slice_qp", C.c_int)]


PAYLOAD_LEAD = 64
MB_BYTES_MAX = 8192          # SW_MB_BYTES_MAX (csrc/slice_kernel.h): the sweep stops before a macroblock whose worst case might not fit


class SliceParams(C.Structure):
    _fields_ = [("slice_type", C.c_int), ("qp", C.c_int), ("chroma_qp_offset", C.c_int),
                ("me_method", C.c_int), ("me_range", C.c_int), ("subme", C.c_int), ("chroma_me", C.c_int), ("mv_range", C.c_int),
                ("fast_pskip", C.c_int), ("dct_decimate", C.c_int), ("cabac", C.c_int), ("transform8x8", C.c_int),
                ("analyse_inter", C.c_int), ("analyse_intra", C.c_int),
                ("quant4_mf", C.c_void_p), ("quant4_bias", C.c_void_p), ("quant8_mf", C.c_void_p), ("quant8_bias", C.c_void_p),
                ("dequant4_mf", C.c_void_p), ("dequant8_mf", C.c_void_p),
                ("cost_mv", C.c_void_p), ("cost_mv_range", C.c_int), ("poc", C.c_int), ("ref_poc", C.c_int * 8),
                ("mixed_refs", C.c_int), ("profile", C.c_void_p), ("noise_reduction", C.c_int), ("nr", C.c_void_p), ("lossless", C.c_int),
                ("rd", C.c_void_p), ("lowres_mv", C.c_void_p), ("b", C.c_void_p)]


class NrState(C.Structure):
    """x264hip_nr_state: h->nr_residual_sum / nr_count / nr_offset of every chain (device)."""
    _fields_ = [("sum", C.c_void_p), ("count", C.c_void_p), ("offset", C.c_void_p)]


def bframe_qp(qp, pb_factor=1.3):
    """rc->qp_constant[SLICE_TYPE_B] (R/encoder/ratecontrol.c:369-372)."""
    return min(max(int(qp + 6.0 * math.log(float(np.float32(pb_factor))) / math.log(2.0) + 0.5), 0), 51)


def coding_order(n_frames, keyint, bframes):
    """[(display index, slice type)] in coding order for a fixed pattern of `bframes` disposable B frames (x264_slicetype_decide
    without b-adapt, then x264_encoder_encode's reordering): an anchor every bframes + 1 frames after an IDR, the last frame before
    the next IDR / the end of the clip is an anchor too, every anchor is coded before the B frames it closes."""
    out, t = [], 0
    while t < n_frames:
        if (t % keyint == 0) if keyint > 0 else t == 0:
            out.append((t, SLICE_I))
            t += 1
            continue
        lim = min((t // keyint + 1) * keyint if keyint > 0 else n_frames, n_frames)
        anchor = min(t + bframes, lim - 1)
        out.append((anchor, SLICE_P))
        out += [(b, SLICE_B) for b in range(t, anchor)]
        t = anchor + 1
    return out


def iframe_qp(qp, ip_factor=1.4):
    """rc->qp_constant[SLICE_TYPE_I] (R/encoder/ratecontrol.c:370-372): the float ip_factor's log, truncated."""
    return min(max(int(qp - 6.0 * math.log(float(np.float32(ip_factor))) / math.log(2.0) + 0.5), 0), 51)


class DeviceState:
    """One x264hip_mb_state: allocated by the library, read back as numpy arrays [batch][n][...]."""

    def __init__(self, ctx, levels=True):
        self.ctx, self.st = ctx, MbState()
        # levels=False: X264HIP_STATE_NO_LEVELS -- no coefficient-level arrays (2/3 of a state's bytes), for sweeps that write the payload themselves
        ctx.check(ctx.lib.x264hip_mb_state_alloc_ex(ctx.h, C.byref(self.st), C.c_int(0 if levels else 1)), "mb_state_alloc")

    def get(self, name):
        d, B = self.ctx.dims, self.ctx.batch
        n = d.mb_w * d.mb_h
        _, dt, tail = next(f for f in STATE_FIELDS if f[0] == name)
        shape = (B, 8, n, 2) if name == "mvr" else (B, n) + tail
        out = np.zeros(shape, dt)
        rc = self.ctx.lib.x264hip_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(getattr(self.st, name)), C.c_size_t(out.nbytes))
        assert rc == 0
        return out

    def free(self):
        self.ctx.lib.x264hip_mb_state_free(self.ctx.h, C.byref(self.st))


class ChainEncoder:
    """Encodes `batch` independent chains in lock step, frame t of every chain per sweep launch."""

    def __init__(self, lib, width, height, cqm, batch=1, qp=26, me_method=0, me_range=16, subme=0, n_refs=1, inter=0, intra=0,
                 transform8x8=0, fast_pskip=1, dct_decimate=1, chroma_me=1, cabac=0, deblock=0, alpha_c0=0, beta=0,
                 chroma_qp_offset=0, keyint=0, mixed_refs=0, noise_reduction=0, mv_range=0,
                 trellis=0, psy_rd=0.0, aq_mode=0, aq_strength=1.0, write=0, cabac_init_idc=0, qp_min=0, qp_max=51, payload_cap=0, raster=None,
                 bframes=0, weightb=0, direct_pred=1, lanes=0, levels=True):
        self.lib = lib
        # x264_validate_parameters (R/encoder/encoder.c:493-522): what the RD-side options do to each other
        trellis = min(max(trellis, 0), 2) if cabac else 0
        psy_rd = 0.0 if subme < 6 else min(max(float(psy_rd), 0.0), 10.0)
        self.psy_rd_fix = int(np.float32(psy_rd) * 256 + 0.5)             # FIX8
        if self.psy_rd_fix:
            chroma_qp_offset = min(max(chroma_qp_offset - (1 if psy_rd < 0.25 else 2), -12), 12)
        aq_strength = min(max(float(aq_strength), 0.0), 3.0)
        aq_mode = 0 if aq_strength == 0 else min(max(aq_mode, 0), 1)
        # the raster-order variant of the sweep: needed by the RD levels, trellis, adaptive quantisation, or simply to get the payload
        # a CAVLC slice's payload is written by a pass over the state the sweep leaves (x264hip_cavlc_write_frame): CAVLC has no adaptive
        # state, so it needs no place in the macroblock loop and the wavefront variant stays the one that codes such slices (with adaptive
        # quantisation: the raster variant without its writer, whose QP rules leave every macroblock's final QP in the state)
        self.cavlc = bool(write and not cabac and subme < 6 and not trellis and not bframes)
        self.raster = bool(subme >= 6 or trellis or aq_mode or (write and not self.cavlc)) if raster is None else bool(raster)
        self.rd_opt = dict(trellis=trellis, aq_mode=aq_mode, aq_strength=aq_strength, write=int(bool((write and not self.cavlc) or subme >= 6 or trellis)),
                           cabac_init_idc=cabac_init_idc, qp_min=qp_min, qp_max=qp_max)
        self.lossless = int(qp == 0)
        if self.lossless:              # x264_validate_parameters, R/encoder/encoder.c:401-421
            fast_pskip, noise_reduction, chroma_qp_offset = 0, 0, 0
            transform8x8 = int(bool(transform8x8 and cabac))
            if not transform8x8:
                inter, intra = inter & ~2, intra & ~2
        self.ctx = FrameCtx(lib, width, height, batch=batch)
        self.opt = dict(qp=qp, me_method=me_method, me_range=me_range, subme=subme, n_refs=n_refs, inter=inter, intra=intra,
                        transform8x8=transform8x8, fast_pskip=fast_pskip, dct_decimate=dct_decimate, chroma_me=chroma_me, cabac=cabac,
                        deblock=deblock, alpha_c0=alpha_c0, beta=beta, chroma_qp_offset=chroma_qp_offset, keyint=keyint, mixed_refs=mixed_refs,
                        noise_reduction=noise_reduction, mv_range=mv_range)
        self.cqm = CqmDevice(lib, cqm)
        self.cost = {}
        self.rd_bufs = None
        if self.raster:
            d, B = self.ctx.dims, batch
            n = d.mb_w * d.mb_h
            cap = payload_cap or (n * 800 + MB_BYTES_MAX + 128 + PAYLOAD_LEAD)
            rb = self._frame_bufs(B, n, cap, aq_mode)
            # p_cost_mv of every QP and the unquant tables, built by the library's host C (x264hip_cost_mv_table / _unquant_table)
            tabs = np.zeros((52, 2 * COST_SPAN + 1), np.int16)
            for q in range(52):
                lib.x264hip_cost_mv_table(C.c_int(LAMBDA_TAB[q]), C.c_int(COST_SPAN), tabs[q].ctypes.data_as(C.c_void_p))
            rb["cost_mv_all"] = DeviceArray(lib, tabs.shape, np.int16, tabs)
            if "unquant4_mf" in cqm:                   # tables from frame.cqm_init (x264hip_cqm_init): complete
                u4, u8 = np.ascontiguousarray(cqm["unquant4_mf"], np.int32), np.ascontiguousarray(cqm["unquant8_mf"], np.int32)
            else:                                      # a table set without them (the reference's x264_cqm_init output as the tests hold it)
                q4 = np.ascontiguousarray(cqm["quant4_mf"][:, 6:12, :].astype(np.int32))       # the shift is zero at qp 6..11 (4x4) / 0..5 (8x8)
                q8 = np.ascontiguousarray(cqm["quant8_mf"][:, 0:6, :].astype(np.int32))
                u4, u8 = np.zeros((4, 52, 16), np.int32), np.zeros((2, 52, 64), np.int32)
                lib.x264hip_unquant_table(q4.ctypes.data_as(C.c_void_p), C.c_int(4), C.c_int(16), u4.ctypes.data_as(C.c_void_p))
                lib.x264hip_unquant_table(q8.ctypes.data_as(C.c_void_p), C.c_int(2), C.c_int(64), u8.ctypes.data_as(C.c_void_p))
            rb["unquant4_mf"] = DeviceArray(lib, u4.shape, np.int32, u4)
            rb["unquant8_mf"] = DeviceArray(lib, u8.shape, np.int32, u8)
            # x264hip_slice_rd.stale: the motion-cache entry that survives macroblocks and frames (read when temporal direct prediction
            # fails); one record per chain, zero like the reference's freshly allocated x264_t
            rb["stale"] = DeviceArray(lib, (B, 8), np.int16)
            self.rd_bufs = rb
            self.payload_cap = cap
        if self.cavlc and not levels:
            raise ValueError("CAVLC payloads are written from the coefficient levels: levels=False does not go with cabac=0, write=1")
        if self.cavlc and not self.raster:
            if not levels:
                raise ValueError("CAVLC payloads are written from the coefficient levels: levels=False does not go with cabac=0, write=1")
            d = self.ctx.dims
            n = d.mb_w * d.mb_h
            self.payload_cap = payload_cap or (n * 800 + MB_BYTES_MAX + 128 + PAYLOAD_LEAD)
            self.rd_bufs = self._frame_bufs(batch, n, self.payload_cap, 0)
        self.i_frame, self.i_frame_stride = 0, 0      # shard.py sets both when the chains are the GOPs of one stream
        self._fenc = None              # the picture upload() fills: allocated on first use (a caller with its own source pictures never needs it)
        # B frames (disposable, one list-1 picture): encode_frame(src, stype, disp) in coding_order(); the DPB then holds
        # max(n_refs, 2) pictures (sps->vui.i_max_dec_frame_buffering, R/encoder/set.c:196-200)
        if bframes and direct_pred not in (1, 2) and not (direct_pred == 3 and getattr(self, "_direct_auto_ok", False)):
            # --direct auto picks a B frame's direct mode from running skip scores and evaluates BOTH modes in every macroblock (R/encoder/analyse.c:2476-2496,
            # encoder.c:113-118,1777-1790): the running scores are the stream encoder's (stream.py: x264hip_slice_b.direct_score); --direct none is not built
            raise ValueError("direct_pred %d: spatial (1) and temporal (2) direct prediction are built, --direct auto (3) in StreamEncoder (it needs the running "
                             "skip scores of the stream's B frames); --direct none is refused" % direct_pred)
        self.bopt = dict(bframes=bframes, weightb=int(bool(weightb)), direct_spatial=int(direct_pred != 2))
        self.dpb = max(n_refs, 2 if bframes else 1)
        self.pool = [self.ctx.new_picture() for _ in range(self.dpb + 1)]
        if not levels and not self.rd_opt["write"]:
            raise ValueError("levels=False: the coefficient levels are the only product unless the sweep writes the payload (write=1)")
        self.states = [DeviceState(self.ctx, levels) for _ in range(self.dpb + 1)]
        self.refs = []                 # [(picture, state, poc)], newest first
        # lanes: the B frames between two anchors predict from the anchors only, never from each other, so they are independent
        # of one another and of the NEXT anchor.  With lanes = K each B frame is enqueued on one of K extra streams (own
        # reconstruction, state and payload buffers), ordered behind the anchor it needs by an event, and runs beside the
        # following anchor: a launch's slow chains no longer hold the whole device.  Same results, bit for bit.
        self.lanes, self.lane_i, self.anchor_ev, self.b_readers = [], 0, None, []
        if lanes and bframes:
            if not self.raster:
                raise ValueError("lanes: B frames run in the raster variant")
            if direct_pred == 2:
                raise ValueError("lanes: temporal direct prediction chains every frame to the one coded before it (x264hip_slice_rd.stale)")
            d = self.ctx.dims
            for _ in range(lanes):
                lc = FrameCtx(lib, width, height, batch=batch)
                self.lanes.append(dict(ctx=lc, recon=lc.new_picture(source_only=True), state=DeviceState(lc, levels),
                                       bufs=self._frame_bufs(batch, d.mb_w * d.mb_h, self.payload_cap, aq_mode)))
        self.t = 0
        self.last_idr = 0
        self.profile = None            # DeviceArray [batch][mb_h][8] int64 when phase timing is wanted
        self.events = None             # set to [] to collect (start, stop, slice_type, n_refs) HIP events per sweep launch
        self.nr = None
        if noise_reduction:
            self.nr = NrState()
            self.ctx.check(lib.x264hip_nr_state_alloc(self.ctx.h, C.byref(self.nr)), "nr_state_alloc")
        lib.x264hip_event_create.restype = C.c_void_p
        lib.x264hip_event_elapsed_ms.restype = C.c_float

    def _frame_bufs(self, B, n, cap, aq_mode):
        """What one frame in flight writes: the payload, its length, the bit position after every macroblock, the AQ arrays."""
        lib = self.lib
        rb = dict(payload=DeviceArray(lib, (B, cap), np.uint8), payload_len=DeviceArray(lib, (B,), np.int32), mb_bits=DeviceArray(lib, (B, n), np.int32))
        if aq_mode:
            rb["aq_energy"] = DeviceArray(lib, (B, n), np.int32)
            rb["aq_offset"] = DeviceArray(lib, (B, n), np.float32)
        return rb

    def sync(self):
        """Everything enqueued so far, on the main stream and on the lanes, has finished."""
        self.ctx.sync()
        for ln in self.lanes:
            ln["ctx"].sync()

    def cost_table(self, qp):
        if qp not in self.cost:                        # p_cost_mv of this QP from the library's host C (x264hip_cost_mv_table), as cost_mv_all
            tab = np.zeros(2 * COST_SPAN + 1, np.int16)
            self.lib.x264hip_cost_mv_table(C.c_int(LAMBDA_TAB[qp]), C.c_int(COST_SPAN), tab.ctypes.data_as(C.c_void_p))
            self.cost[qp] = DeviceArray(self.lib, tab.shape, np.int16, tab)
        return self.cost[qp]

    @property
    def fenc(self):
        if self._fenc is None:
            self._fenc = self.ctx.new_picture(source_only=True)
        return self._fenc

    def upload(self, y, u, v, b=0):
        for ln in self.lanes:          # a B frame still in flight on a lane may be reading this picture
            ln["ctx"].sync()
        self.ctx.upload(self.fenc, y, u, v, b=b)

    def encode_frame(self, src=None, stype=None, disp=None, lowres_mv=None, lowres_mv1=None):
        """The macroblock sweep for the frame held by `src` (default: the picture upload() fills) in every
        batch element.  Returns (slice_type, qp, state) -- the state's arrays are valid after ctx.sync().
        Without stype: I / P chains in display order (an IDR every keyint frames).  With stype / disp (see coding_order): the
        frame's slice type and display index, frames arriving in coding order -- the way B frames are coded."""
        L, c, o = self.lib, self.ctx, self.opt
        fenc = self.fenc if src is None else src
        if stype is None:
            idr = (self.t % o["keyint"] == 0) if o["keyint"] > 0 else self.t == 0
            stype, disp = (SLICE_I if idr else SLICE_P), self.t
        idr, is_b = stype == SLICE_I, stype == SLICE_B
        if idr:
            self.refs, self.last_idr = [], disp
        poc = 2 * (disp - self.last_idr)
        used = [r[0] for r in self.refs]
        lane = None
        if is_b and self.lanes:        # a lane's stream: behind the last anchor's filters, beside whatever else is in flight
            lane = self.lanes[self.lane_i % len(self.lanes)]
            self.lane_i += 1
            c, recon, state = lane["ctx"], lane["recon"], lane["state"]
            if self.anchor_ev:
                L.x264hip_stream_wait_event(C.c_void_p(c.stream), C.c_void_p(self.anchor_ev))
        else:
            pic_i = next(i for i, p in enumerate(self.pool) if not any(p is q for q in used))
            recon, state = self.pool[pic_i], self.states[pic_i]
            keep = []
            for ev, reads in self.b_readers:       # B frames on the lanes that still read the picture / state about to be overwritten
                if pic_i in reads:
                    L.x264hip_stream_wait_event(C.c_void_p(c.stream), C.c_void_p(ev))
                    reads.discard(pic_i)
                if reads:
                    keep.append((ev, reads))
                else:
                    L.x264hip_event_destroy(C.c_void_p(ev))
            self.b_readers = keep
        # x264_reference_build_list (R/encoder/encoder.c:911-981): list 0 = earlier pictures, nearest first; list 1 = later ones
        refs = sorted([r for r in self.refs if r[2] < poc], key=lambda r: -r[2])[:o["n_refs"]]
        refs1 = sorted([r for r in self.refs if r[2] > poc], key=lambda r: r[2])[:1] if is_b else []
        qp = iframe_qp(o["qp"]) if idr else bframe_qp(o["qp"]) if is_b else o["qp"]
        self.last_is_b, self.last_poc = is_b, poc
        b = self.cqm.bufs
        p = SliceParams(slice_type=stype, qp=qp, chroma_qp_offset=o["chroma_qp_offset"], me_method=o["me_method"], me_range=o["me_range"],
                        subme=o["subme"], chroma_me=o["chroma_me"], mv_range=o["mv_range"] or 512, fast_pskip=o["fast_pskip"], dct_decimate=o["dct_decimate"],
                        cabac=o["cabac"], transform8x8=o["transform8x8"], analyse_inter=o["inter"], analyse_intra=o["intra"],
                        quant4_mf=b["quant4_mf"].ptr, quant4_bias=b["quant4_bias"].ptr, quant8_mf=b["quant8_mf"].ptr,
                        quant8_bias=b["quant8_bias"].ptr, dequant4_mf=b["dequant4_mf"].ptr, dequant8_mf=b["dequant8_mf"].ptr,
                        cost_mv=self.cost_table(qp).ptr, cost_mv_range=COST_SPAN, poc=poc, mixed_refs=o["mixed_refs"],
                        profile=self.profile.ptr if self.profile else None,
                        noise_reduction=o["noise_reduction"], nr=C.addressof(self.nr) if self.nr else None, lossless=self.lossless,
                        lowres_mv=lowres_mv.ptr if lowres_mv is not None else None)        # DeviceArray [batch][n_mb][2] int16: the lookahead's vectors
        if self.raster:
            rb, ro = dict(self.rd_bufs, **lane["bufs"]) if lane else self.rd_bufs, self.rd_opt
            self.last_bufs = rb
            if ro["aq_mode"]:                  # x264_adaptive_quant_frame on the source (R/encoder/encoder.c:1421)
                L.x264hip_adaptive_quant_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
                c.check(L.x264hip_adaptive_quant_frame(c.h, C.byref(fenc), C.c_float(ro["aq_strength"]), rb["aq_energy"].p, rb["aq_offset"].p), "adaptive_quant_frame")
            self.rd = SliceRd(trellis=ro["trellis"], psy_rd=self.psy_rd_fix, write=ro["write"], cabac_init_idc=ro["cabac_init_idc"], i_frame=self.i_frame,
                              qp_min=ro["qp_min"], qp_max=ro["qp_max"], f_qpm=float(qp), aq_offset=rb["aq_offset"].ptr if ro["aq_mode"] else None,
                              cost_mv_all=rb["cost_mv_all"].ptr, unquant4_mf=rb["unquant4_mf"].ptr, unquant8_mf=rb["unquant8_mf"].ptr,
                              payload=rb["payload"].ptr, payload_cap=self.payload_cap, payload_len=rb["payload_len"].ptr, mb_bits=rb["mb_bits"].ptr,
                              stale=rb["stale"].ptr,
                              i_frame_stride=self.i_frame_stride)
            p.rd = C.addressof(self.rd)
        if is_b:
            self.sb = SliceB(fref1=C.addressof(refs1[0][0]), l1_state=C.addressof(refs1[0][1].st), ref1_poc=refs1[0][2],
                             weightb=self.bopt["weightb"], direct_spatial=self.bopt["direct_spatial"],
                             lowres_mv1=lowres_mv1.ptr if lowres_mv1 is not None else None)
            p.b = C.addressof(self.sb)
        for i, r in enumerate(refs):
            p.ref_poc[i] = r[2]
        arr = (C.c_void_p * max(len(refs), 1))(*[C.addressof(r[0]) for r in refs]) if refs else None
        l0 = C.byref(refs[0][1].st) if refs else None
        ev = None
        if self.events is not None:            # HIP events on the launch stream around the sweep kernel (bench.py)
            ev = (L.x264hip_event_create(), L.x264hip_event_create())
            L.x264hip_event_record(C.c_void_p(ev[0]), C.c_void_p(c.stream))
        c.check(L.x264hip_slice_sweep_frame(c.h, C.byref(fenc), arr, len(refs), C.byref(recon), C.byref(p), l0, C.byref(state.st)),
                "slice_sweep_frame")
        if ev:
            L.x264hip_event_record(C.c_void_p(ev[1]), C.c_void_p(c.stream))
            self.events.append((ev[0], ev[1], stype, len(refs) + len(refs1)))
        if self.cavlc:                                 # x264_macroblock_write_cavlc for every macroblock of every chain, from the state just written
            rb = self.rd_bufs
            cp = CavlcParams(slice_type=stype, n_ref0=len(refs), analyse_inter=o["inter"], transform8x8=o["transform8x8"], cqm_custom=0,
                             payload=rb["payload"].ptr, payload_cap=self.payload_cap, payload_len=rb["payload_len"].ptr, mb_bits=rb["mb_bits"].ptr,
                             slice_qp=qp)
            c.check(L.x264hip_cavlc_write_frame(c.h, C.byref(state.st), C.byref(cp)), "cavlc_write_frame")
            self.last_bufs = rb
        if self.nr:                            # x264_noise_reduction_update at the end of every frame (R/encoder/encoder.c:1755)
            c.check(L.x264hip_noise_reduction_update(c.h, C.byref(self.nr), o["noise_reduction"]), "noise_reduction_update")
        if lane:                               # whoever overwrites one of the pictures this frame reads waits for it
            ev = L.x264hip_event_create()
            L.x264hip_event_record(C.c_void_p(ev), C.c_void_p(c.stream))
            reads = {i for i, p in enumerate(self.pool) if any(p is r[0] for r in refs + refs1)}
            self.b_readers.append((ev, reads))
        self.last = (recon, state)
        self.last_ctx = c
        return stype, qp, state

    def finish_frame(self):
        """x264_fdec_filter_row for the whole frame: loop filter, borders, half-pel planes; then the frame joins the reference list."""
        L, c, o = self.lib, self.ctx, self.opt
        recon, state = self.last
        if getattr(self, "last_is_b", False):          # a disposable B frame: neither filtered nor kept (R/encoder/encoder.c:986-1024,1060-1068)
            self.t += 1
            self.i_frame += 1
            return
        if o["deblock"]:
            s = state.st
            dp = DeblockParams(mb_type=s.mb_type, qp=s.qp, nnz=s.nnz, transform8x8=s.t8, mv=s.mv, ref=s.ref,
                               alpha_c0_offset=o["alpha_c0"], beta_offset=o["beta"], chroma_qp_offset=o["chroma_qp_offset"], state_layout=1,
                               sub8x8=1 if o["inter"] & 0x20 else 0)
            c.check(L.x264hip_deblock_frame(c.h, C.byref(recon), C.byref(dp)), "deblock_frame")
        c.check(L.x264hip_expand_border(c.h, C.byref(recon), 0), "expand_border")
        c.check(L.x264hip_hpel_filter_frame(c.h, C.byref(recon)), "hpel_filter_frame")
        self.refs.insert(0, (recon, state, getattr(self, "last_poc", 2 * (self.t - self.last_idr))))
        del self.refs[self.dpb:]
        if self.lanes:                         # the point the B frames that predict from this anchor wait for
            if self.anchor_ev:
                L.x264hip_event_destroy(C.c_void_p(self.anchor_ev))
            self.anchor_ev = L.x264hip_event_create()
            L.x264hip_event_record(C.c_void_p(self.anchor_ev), C.c_void_p(c.stream))
        self.t += 1
        self.i_frame += 1

    def payloads(self):
        """slice_data() of the last frame of every chain (valid after ctx.sync()): list of bytes objects."""
        rb = getattr(self, "last_bufs", None) or self.rd_bufs
        n = rb["payload_len"].get()
        raw = rb["payload"].get()
        return [bytes(raw[b, PAYLOAD_LEAD:PAYLOAD_LEAD + n[b]]) for b in range(len(n))]

    def payload_async(self, b, host_len, host_buf, nbytes):
        """Enqueue, behind the sweep just launched, the copy of chain b's payload length and of the first `nbytes` bytes of its payload
        into pinned host memory (x264hip_host_alloc): no synchronisation, the bytes are there once the stream has passed this point."""
        rb = getattr(self, "last_bufs", None) or self.rd_bufs
        st = C.c_void_p(self.last_ctx.stream)
        self.lib.x264hip_memcpy_d2h_async(C.c_void_p(host_len), C.c_void_p(rb["payload_len"].ptr + 4 * b), C.c_size_t(4), st)
        self.lib.x264hip_memcpy_d2h_async(C.c_void_p(host_buf), C.c_void_p(rb["payload"].ptr + self.payload_cap * b + PAYLOAD_LEAD), C.c_size_t(nbytes), st)

    def status(self):
        c = self.ctx
        if getattr(self, "last", None) is None:          # nothing launched yet
            return
        c.check(self.lib.x264hip_slice_sweep_status(c.h, C.byref(self.last[1].st if not self.lanes or not self.last_is_b else self.states[0].st)), "slice_sweep_status")
        for ln in self.lanes:                  # each lane's context keeps its own sticky abort count
            ln["ctx"].check(self.lib.x264hip_slice_sweep_status(ln["ctx"].h, C.byref(ln["state"].st)), "slice_sweep_status")

    def close(self):
        for ev, _ in self.b_readers:
            self.lib.x264hip_event_destroy(C.c_void_p(ev))
        if self.anchor_ev:
            self.lib.x264hip_event_destroy(C.c_void_p(self.anchor_ev))
        self.b_readers, self.anchor_ev = [], None
        if self.nr:
            self.lib.x264hip_nr_state_free(self.ctx.h, C.byref(self.nr))
            self.nr = None
        for ln in self.lanes:                  # each lane's payload / AQ buffers, state, reconstruction and stream
            for d in ln["bufs"].values():
                d.free()
            ln["state"].free()
            ln["ctx"].close()
        self.lanes = []
        for s in self.states:
            s.free()
        for d in self.cost.values():
            d.free()
        for d in (self.rd_bufs or {}).values():
            d.free()
        self.cqm.free()
        self.ctx.close()
