"""x264_encoder_encode for a batch of GOP chains with the real lookahead and rate control in front of the macroblock sweep
(R/encoder/encoder.c:1340-1600): x264_slicetype_decide places every chain's B frames (b-adapt 1 / 2, the pre-encode scene cut),
x264_ratecontrol_start prices every frame (CQP or CRF), and the lookahead's half-resolution vectors are the first candidates of the
16x16 searches -- so the chains stop moving in lock step: at one step chain A codes a P frame from references {9, 8} at QP 24 while
chain B codes a B frame between 8 and 12 at QP 27.

    enc = StreamEncoder(lib, w, h, cqm, batch=B, crf=23.0, bframes=3, b_adapt=1, ...)      # ChainEncoder's options + the lookahead's
    frames = enc.step(fill)                # one picture per chain in (fill(picture, input number)), at most one coded frame per chain out
    frames = enc.step(None)                # flushing: no more input; [] when every chain is done
    enc.payloads()                         # slice_data() of the frames just coded, per chain (None where the chain coded nothing)

Host side: x264_vs2008_amd.lookahead (the library's host C state machine per chain + the batched cost kernel).  Device side: one
chain-table launch per kernel kind (x264hip_slice_sweep_chains), each chain's entry naming its own source picture (the lookahead's slot),
references, reconstruction and QP; the end-of-frame filters run on the elements that coded a kept frame (x264hip_frame_ctx_elements).
The arithmetic lives in the HIP library; this file orders launches and owns device buffers."""
import ctypes as C
import os
import sys

import numpy as np

from . import lookahead as LA
from .frame import DeblockParams, DeviceArray
from .slice import (COST_SPAN, ChainEncoder, MbState, SLICE_B, SLICE_I, SLICE_P, SliceB, SliceParams, SliceRd)


class ChainSweep(C.Structure):
    """x264hip_chain_sweep"""
    _fields_ = [("chain", C.c_int), ("fenc", C.c_void_p), ("refs", C.c_void_p), ("n_refs", C.c_int), ("recon", C.c_void_p),
                ("params", C.c_void_p), ("l0", C.c_void_p), ("out", C.c_void_p)]


class Coded:
    """What one chain coded in a step."""
    __slots__ = ("chain", "frame", "type", "slice_type", "qp", "f_qpm", "poc", "n_ref0", "n_ref1", "i_satd", "frame_num_reset", "direct_spatial")

    def __repr__(self):
        return "Coded(chain=%d frame=%d type=%d qp=%d poc=%d)" % (self.chain, self.frame, self.type, self.qp, self.poc)


class StreamEncoder(ChainEncoder):
    def __init__(self, lib, width, height, cqm, batch=1, crf=None, b_adapt=1, bframe_bias=0, keyint_min=0, scenecut_threshold=40, pre_scenecut=1,
                 ip_factor=1.4, pb_factor=1.3, qcompress=0.6, qp_step=4, n_slots=0, speculative=True, limits=None, n_frames=None, lookahead_priority=False, b_cus=0, **kw):
        kw.setdefault("write", 1)
        kw.setdefault("levels", False)
        if kw.get("lanes"):
            raise ValueError("StreamEncoder: the chains of a step already run side by side; lanes belong to the lock-step encoder")
        self._direct_auto_ok = True                     # --direct auto: the running scores live here (c_dscore)
        super().__init__(lib, width, height, cqm, batch=batch, **kw)
        if not self.raster or not self.rd_opt["write"]:
            raise ValueError("StreamEncoder: the chain-table sweep is the raster variant with the entropy coder in the loop")
        o, d = self.opt, self.ctx.dims
        keyint = o["keyint"] if o["keyint"] > 0 else 1 << 30
        self.la_params = LA.make_params(d.mb_w, d.mb_h, bframes=self.bopt["bframes"], b_adapt=b_adapt if self.bopt["bframes"] else 0, bframe_bias=bframe_bias,
                                        keyint_max=keyint, keyint_min=keyint_min, scenecut_threshold=scenecut_threshold, pre_scenecut=pre_scenecut,
                                        crf=crf, qp=o["qp"], ip_factor=ip_factor, pb_factor=pb_factor, qcompress=qcompress,
                                        qp_min=self.rd_opt["qp_min"], qp_max=self.rd_opt["qp_max"], qp_step=qp_step)
        bf = self.bopt["bframes"]
        delay = (max(bf, 3) * 4 if b_adapt == 2 and bf else bf)
        self.n_slots = n_slots or (delay + bf + 3)        # oldest live frame .. newest input spans at most delay + bframes + 2 (asserted when a slot is reused)
        # The lookahead works on a stream of its own (same geometry, so its pictures are the sweep's sources): with n_frames given (how many
        # pictures every chain will take) step() prepares the NEXT call's decisions right after launching this one's sweep, and the
        # lookahead's kernels fill the wave slots the step's B chains leave when they finish ahead of its P chains.
        from .frame import FrameCtx
        # b_cus: the step's B kernel on compute units [0, b_cus), its I / P kernel (and everything else of this context) on the rest
        self._cu_streams = []
        if b_cus:
            lib.x264hip_stream_create_cu_range.restype = C.c_void_p
            sb_, sp_ = lib.x264hip_stream_create_cu_range(0, b_cus), lib.x264hip_stream_create_cu_range(b_cus, 256 - b_cus)
            if not sb_ or not sp_:
                raise RuntimeError("x264hip_stream_create_cu_range failed")
            self._cu_streams = [sb_, sp_]
            old = self.ctx
            self.ctx = FrameCtx(lib, width, height, stream=sp_, batch=batch)         # same geometry: the pool's pictures and states stay where they are
            self.ctx.pictures, old.pictures = old.pictures, []
            for st_ in self.states:
                st_.ctx = self.ctx
            old.close()
            self.ctx.check(lib.x264hip_frame_ctx_set_b_stream(self.ctx.h, C.c_void_p(sb_)), "frame_ctx_set_b_stream")
        self._hp_stream = None
        if lookahead_priority:
            lib.x264hip_stream_create_high_priority.restype = C.c_void_p
            self._hp_stream = lib.x264hip_stream_create_high_priority()
        self.src_ctx = FrameCtx(lib, width, height, stream=self._hp_stream, batch=batch)
        self.n_frames = n_frames
        self._prep = None
        self._coding = set()
        self.look = LA.LookaheadDevice(self.src_ctx, self.n_slots, bf, me_method=o["me_method"], me_range=o["me_range"], weightb=self.bopt["weightb"],
                                       bframe_bias=bframe_bias, subme=o["subme"], lossless=self.lossless)
        self.lb = LA.LookaheadBatch(self.src_ctx, self.la_params, self.look, speculative=speculative, limits=limits)
        B, n = batch, d.mb_w * d.mb_h
        # fenc->f_qp_offset of every slot (x264_adaptive_quant_frame runs when the picture comes in, encoder.c:1420-1421)
        self.aq_slots = None
        if self.rd_opt["aq_mode"]:
            self.aq_slots = [(DeviceArray(lib, (B, n), np.int32), DeviceArray(lib, (B, n), np.float32)) for _ in range(self.n_slots)]
        # per chain: [(pool index, poc, MbState copy with this chain's frame-level scalars)], newest first; the IDR's input number; frames coded
        self.crefs = [[] for _ in range(B)]
        self.c_last_idr = [0] * B
        self.c_coded = [0] * B
        lib.x264hip_chain_sweep_bytes.restype = C.c_size_t
        lib.x264hip_host_alloc.restype = C.c_void_p
        tb = lib.x264hip_chain_sweep_bytes()
        self.tab_host = lib.x264hip_host_alloc(C.c_size_t(tb * B))
        self.tab_dev = DeviceArray(lib, (tb * B,), np.uint8)
        self.elems_dev = [DeviceArray(lib, (B,), np.int32) for _ in range(len(self.pool))]
        self.flushing = False
        self.coded_now = [None] * B
        self.n_sweeps = 0
        # --direct auto (h->mb.b_direct_auto_write): every B macroblock predicts both direct modes and credits each with the skip it would give
        # (the sweep leaves the frame's two sums in dscore_dev); h->stat.i_direct_score, the running scores that pick the next B frame's mode
        # (encoder.c:113-118) and decay as x264_encoder_frame_end lets them (:1777-1790), are kept per chain here
        self.direct_auto = bool(self.bopt["bframes"] and kw.get("direct_pred", 1) == 3)
        self.dscore_dev = DeviceArray(lib, (B, 2), np.int32) if self.direct_auto else None
        self.c_dscore = [[0, 0] for _ in range(B)]
        self._dscore_pending = []
        # Without --pre-scenecut the reference looks at every P frame AFTER coding it (x264_encoder_encode, encoder.c:1603-1699) and, if it finds a
        # scene cut, codes again: the picture as I / IDR, or the B picture before it as the P.  The decision is made from the sweep's own
        # statistics (x264hip_frame_stats + x264hip_scenecut_post), the queue surgery by the library (x264hip_lookahead_scenecut), the second
        # attempt inside the same step (step()).
        self.post_scenecut = bool(scenecut_threshold >= 0 and not pre_scenecut)
        self._post, self.stats_dev, self._undo, self.n_given_up = [], None, {}, 0
        if self.post_scenecut:
            if o["keyint"] <= 0:
                raise ValueError("StreamEncoder: the post-encode scene cut needs a finite keyint")
            if o["noise_reduction"]:
                raise ValueError("StreamEncoder: --nr with the post-encode scene cut: a given-up attempt has already added to the noise-reduction sums "
                                 "(the reference's has too, but per frame, not per batch); run with pre_scenecut=1")
            self.stats_dev = [DeviceArray(lib, (B, 32), np.uint8) for _ in self.pool]
        self.sweep_events = None       # set to [] to collect (start, stop, chains, algorithmic bytes) HIP events around every step's sweep launches

    # ---- one call of x264_encoder_encode for every chain --------------------------------------------------------------------------
    def _prepare(self, fill):
        """The lookahead's part of one x264_encoder_encode call for every chain: a picture comes in (or the flush begins), the queues decide.
        Runs on the lookahead's stream; everything it launched has finished when it returns."""
        self._put(fill)
        return self._decide()

    def _put(self, fill):
        """The picture comes in (x264_encoder_encode up to the frame queue: copy, lowres planes, intra costs, adaptive quantisation)."""
        more = fill is not None and (self.n_frames is None or self.lb.fed < self.n_frames)
        if not more:
            self.flushing = True
        else:
            # the slot the new picture takes must not be one a sweep still in flight reads (possible only when the ring is at its bound)
            if self.look.frame_of_slot[self.look.slot(self.lb.fed)] in self._coding:
                self.ctx.sync()
            self.lb.put(lambda pic, f: self._fill(fill, pic, f))

    def _decide(self):
        """The queues decide (x264_slicetype_decide, x264_rc_analyse_slice, x264_ratecontrol_start) with the cost kernel behind them."""
        frames = self.lb.get(self.flushing)
        self.src_ctx.sync()
        return frames

    def step(self, fill):
        """fill(picture, frame): write input picture `frame` of every chain into `picture` (enc.src_ctx.upload / .synth); None: flush.
        Returns the list of Coded for the chains that coded a frame (empty while the B buffer fills; empty for good once flushed).
        With n_frames given to the constructor the next call's lookahead is prepared before this one returns (fill is then also asked
        for the following picture) and the flush starts by itself after n_frames pictures."""
        # with n_frames known the next call's lookahead (picture in, costs, decisions) runs beside this call's sweep
        frames = self._prep if self._prep is not None else self._prepare(fill)
        self._prep = None
        self.coded_now = [None] * self.ctx.batch
        pipelined = self.n_frames is not None
        if not any(fr is not None for fr in frames):
            self._coding = set()
            if pipelined and not self.flushing:
                self._prep = self._prepare(fill)
            return []
        out = self._sweep(frames)
        if not self.post_scenecut:
            if pipelined and not self.flushing:
                self._prep = self._prepare(fill)        # beside the sweep just launched
            return out
        # x264_encoder_encode looks at the P picture it just coded (encoder.c:1603-1699) and, if an intra picture would have been as good, gives the
        # attempt up and codes again -- the same picture as I / IDR or the B picture before it as the P -- inside the same call.  Here: wait for the
        # sweep, ask (x264hip_frame_stats + x264hip_scenecut_post), and run the given-up chains again until none is left.  Each round costs a whole
        # frame time for a few chains: fine for a handful of streams, a reason to decide scene cuts in the lookahead (pre_scenecut = 1) for thousands.
        # The next call's lookahead does not wait for the verdicts: it runs ahead beside the sweep, on the assumption that nothing is given up,
        # with a copy of the judged chains' queues to come back to (x264hip_lookahead_save / _restore).
        done = [cd.chain for cd in out]
        judged = sorted({ci for _, ci, _, _ in self._post})
        saved = None
        if pipelined and not self.flushing:
            saved = {ci: self.lb.chains[ci].save() for ci in judged}
            if any(v is None for v in saved.values()):
                saved = None
        if saved is not None:
            self.lb.end(done)
            self._prep = self._prepare(fill)
        redone = []
        while True:
            self.ctx.sync()
            hits = self._scenecut_hits()
            if not hits:
                break
            for ci in hits:
                if saved is not None and ci not in redone:
                    self.lb.chains[ci].restore(saved[ci])
                self._give_up(ci)
            redone += [ci for ci in hits if ci not in redone]
            again = self.lb.get(self.flushing, only=hits)
            self.src_ctx.sync()
            redo = {cd.chain: cd for cd in self._sweep(again)}
            out = [redo.get(cd.chain, cd) for cd in out]
            self.n_given_up += len(hits)
        if saved is None:
            self.lb.end(done)
        elif redone:                                    # their decisions for the next call were made on the wrong assumption: made again
            self.lb.end(redone)
            nxt = self.lb.get(self.flushing, only=redone)
            self.src_ctx.sync()
            for ci in redone:
                self._prep[ci] = nxt[ci]
        return out

    def _sweep(self, frames):
        """The chain-table launch for the chains that have a frame in `frames` (per chain an x264hip_look_frame or None), then the filters of the kept ones."""
        L, c, o, ro = self.lib, self.ctx, self.opt, self.rd_opt
        B = c.batch
        todo = [(ci, fr) for ci, fr in enumerate(frames) if fr is not None]
        if self._dscore_pending:                        # the B frames of the last sweep: their skip sums join the running scores before the next mode is picked
            c.sync()
            fs, n_mb = self.dscore_dev.get(), c.dims.mb_w * c.dims.mb_h
            for ci in self._dscore_pending:
                sc = self.c_dscore[ci]
                if sc[0] + sc[1] > n_mb:
                    sc[0], sc[1] = sc[0] * 9 // 10, sc[1] * 9 // 10
                sc[0] += int(fs[ci][0]); sc[1] += int(fs[ci][1])
            self._dscore_pending = []
        keep = []                                       # everything the C call reads must outlive it
        entries = (ChainSweep * len(todo))()
        b = self.cqm.bufs
        rb = self.rd_bufs
        written, filt = set(), {}
        out = []
        for k, (ci, fr) in enumerate(todo):
            entries[k], cd, pic_i = self._entry(ci, fr, keep)
            written.add(pic_i)
            out.append(cd)
            if cd.slice_type != SLICE_B:                   # kept: filtered below
                filt.setdefault(pic_i, []).append(ci)
        c.sync()                                        # the previous step's sweep and filters are done: their tables, element lists and pictures are free
        for pic_i in written:
            c.check(L.x264hip_mb_state_clear_progress(c.h, C.byref(self.states[pic_i].st)), "mb_state_clear_progress")
        ev = None
        if self.sweep_events is not None:
            ev = (L.x264hip_event_create(), L.x264hip_event_create())
            L.x264hip_event_record(C.c_void_p(ev[0]), C.c_void_p(c.stream))
        c.check(L.x264hip_slice_sweep_chains(c.h, entries, len(todo), C.c_void_p(self.tab_host), self.tab_dev.p), "slice_sweep_chains")
        if ev:
            L.x264hip_event_record(C.c_void_p(ev[1]), C.c_void_p(c.stream))
            px = self.ctx.dims.mb_w * 16 * self.ctx.dims.lines_y
            # per frame the source (1.5 B/px), each reference's four luma planes + chroma (4.5 B/px) and the reconstruction (1.5 B/px)
            self.sweep_events.append((ev[0], ev[1], len(todo), sum(px * (3.0 + 4.5 * (cd.n_ref0 + cd.n_ref1)) for cd in out),
                                      "".join("PBI"[cd.slice_type] for cd in out[:1]) + ":%d" % len(todo)))
        self.n_sweeps += 1
        if self.nr:                                        # x264_noise_reduction_update at the end of every frame (encoder.c:1755)
            c.check(L.x264hip_noise_reduction_update(c.h, C.byref(self.nr), o["noise_reduction"]), "noise_reduction_update")
        # x264_fdec_filter_row for the kept frames: loop filter, borders, half-pel planes, on the elements that were just written
        for pic_i, chains in filt.items():
            recon, s = self.pool[pic_i], self.states[pic_i].st
            el = self.elems_dev[pic_i]
            lst = np.zeros(c.batch, np.int32)
            lst[:len(chains)] = chains
            el.set(lst)
            c.check(L.x264hip_frame_ctx_elements(c.h, el.p, len(chains)), "frame_ctx_elements")
            if o["deblock"]:
                dp = DeblockParams(mb_type=s.mb_type, qp=s.qp, nnz=s.nnz, transform8x8=s.t8, mv=s.mv, ref=s.ref,
                                   alpha_c0_offset=o["alpha_c0"], beta_offset=o["beta"], chroma_qp_offset=o["chroma_qp_offset"], state_layout=1,
                                   sub8x8=1 if o["inter"] & 0x20 else 0)
                c.check(L.x264hip_deblock_frame(c.h, C.byref(recon), C.byref(dp)), "deblock_frame")
            c.check(L.x264hip_expand_border(c.h, C.byref(recon), 0), "expand_border")
            c.check(L.x264hip_hpel_filter_frame(c.h, C.byref(recon)), "hpel_filter_frame")
        c.check(L.x264hip_frame_ctx_elements(c.h, None, 0), "frame_ctx_elements")
        if self.post_scenecut:
            self._post = []
            for pic_i, chains in filt.items():
                ps = [ci for ci in chains if self.coded_now[ci].slice_type == SLICE_P]
                if ps:
                    c.check(L.x264hip_frame_stats(c.h, C.byref(self.states[pic_i].st), self.stats_dev[pic_i].p), "frame_stats")
                    self._post += [(pic_i, ci, self.coded_now[ci].frame, self.coded_now[ci].frame - self.c_last_idr[ci]) for ci in ps]
        self._keep = keep
        if not self.post_scenecut:
            self.lb.end([ci for ci, _ in todo])
        self.last_bufs, self.last_ctx = rb, c
        self._coding = {cd.frame for cd in out}
        return out

    def _scenecut_hits(self):
        """After sync(): the chains whose P picture of the last sweep the reference would give up (encoder.c:1603-1644)."""
        if not self._post:
            return []
        L, d, lp = self.lib, self.ctx.dims, self.la_params
        L.x264hip_scenecut_post.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        got, hits = {}, []
        for pic_i, ci, frame, gop in self._post:
            if pic_i not in got:
                got[pic_i] = self.stats_dev[pic_i].get()
            rec = np.ascontiguousarray(got[pic_i][ci])
            if L.x264hip_scenecut_post(rec.ctypes.data_as(C.c_void_p), d.mb_w * d.mb_h, gop, lp.scenecut_threshold, lp.keyint_min, lp.keyint_max):
                hits.append(ci)
        self._post = []
        return hits

    def _give_up(self, ci):
        """The attempt of chain ci is dropped: its reconstruction never joined the DPB, its payload is never read, its frame was never counted
        (x264_reference_update and x264_encoder_frame_end are not reached for it); the chain's queue rearranges itself (x264hip_lookahead_scenecut)."""
        self.crefs[ci], self.c_coded[ci] = self._undo[ci]
        self.coded_now[ci] = None
        self.lb.chains[ci].scenecut()
        self.lb.pending[ci] = None

    def _entry(self, ci, fr, keep):
        """One chain's sweep for the frame its queue handed it: the x264hip_chain_sweep record (whatever it points at goes into `keep`), the
        chain's DPB bookkeeping (x264_reference_build_list before, x264_reference_update after), the Coded record, the pool picture written."""
        o, ro, b, rb = self.opt, self.rd_opt, self.cqm.bufs, self._bufs_of(ci)
        idr = fr.type == LA.TYPE_IDR
        stype = SLICE_I if fr.type in (LA.TYPE_IDR, LA.TYPE_I) else SLICE_B if fr.type == LA.TYPE_B else SLICE_P
        if idr:
            self.crefs[ci], self.c_last_idr[ci] = [], fr.frame
        refs_all = self.crefs[ci]
        poc = fr.poc
        used = {r[0] for r in refs_all}
        pic_i = next(i for i in range(len(self.pool)) if i not in used)
        recon, state = self.pool[pic_i], self.states[pic_i]
        refs = sorted([r for r in refs_all if r[1] < poc], key=lambda r: -r[1])[:o["n_refs"]] if stype != SLICE_I else []
        refs1 = sorted([r for r in refs_all if r[1] > poc], key=lambda r: r[1])[:1] if stype == SLICE_B else []
        qp = fr.qp
        slot = self.look.slot(fr.frame)
        assert self.look.frame_of_slot[slot] == fr.frame, "input frame %d left its lookahead slot before it was coded (n_slots too small)" % fr.frame
        lw0 = lw1 = None
        n = self.look.n
        if stype != SLICE_I and fr.lowres_l0:     # the kernel adds the chain's offset in an [batch][n_mb][2] array to the pointer it is given
            lw0 = self.look.mv_ptr(ci, fr.frame, 0, fr.frame - fr.ref0_frame) - 4 * n * ci
        if stype == SLICE_B and fr.lowres_l1:
            lw1 = self.look.mv_ptr(ci, fr.frame, 1, fr.ref1_frame - fr.frame) - 4 * n * ci
        p = SliceParams(slice_type=stype, qp=qp, chroma_qp_offset=o["chroma_qp_offset"], me_method=o["me_method"], me_range=o["me_range"],
                        subme=o["subme"], chroma_me=o["chroma_me"], mv_range=o["mv_range"] or 512, fast_pskip=o["fast_pskip"], dct_decimate=o["dct_decimate"],
                        cabac=o["cabac"], transform8x8=o["transform8x8"], analyse_inter=o["inter"], analyse_intra=o["intra"],
                        quant4_mf=b["quant4_mf"].ptr, quant4_bias=b["quant4_bias"].ptr, quant8_mf=b["quant8_mf"].ptr,
                        quant8_bias=b["quant8_bias"].ptr, dequant4_mf=b["dequant4_mf"].ptr, dequant8_mf=b["dequant8_mf"].ptr,
                        cost_mv=rb["cost_mv_all"].ptr + qp * (2 * COST_SPAN + 1) * 2, cost_mv_range=COST_SPAN, poc=poc, mixed_refs=o["mixed_refs"],      # (a row of the table of all QPs: no allocation while kernels run -- hipMalloc waits for the device)
                        noise_reduction=o["noise_reduction"], nr=C.addressof(self.nr) if self.nr else None, lossless=self.lossless,
                        lowres_mv=lw0)
        rd = SliceRd(trellis=ro["trellis"], psy_rd=self.psy_rd_fix, write=1, cabac_init_idc=ro["cabac_init_idc"], i_frame=self.c_coded[ci],
                     qp_min=ro["qp_min"], qp_max=ro["qp_max"], f_qpm=fr.f_qpm, aq_offset=self.aq_slots[slot][1].ptr if self.aq_slots else None,
                     cost_mv_all=rb["cost_mv_all"].ptr, unquant4_mf=rb["unquant4_mf"].ptr, unquant8_mf=rb["unquant8_mf"].ptr,
                     payload=rb["payload"].ptr, payload_cap=self.payload_cap, payload_len=rb["payload_len"].ptr, mb_bits=rb["mb_bits"].ptr,
                     stale=rb["stale"].ptr, i_frame_stride=0)
        p.rd = C.addressof(rd)
        keep += [p, rd]
        dsp = self.bopt["direct_spatial"]
        if stype == SLICE_B:
            if self.direct_auto:                       # x264_slice_header_init, encoder.c:113-118
                dsp = int(self.c_dscore[ci][1] > self.c_dscore[ci][0])
                self._dscore_pending.append(ci)
            sb = SliceB(fref1=C.addressof(self.pool[refs1[0][0]]), l1_state=C.addressof(refs1[0][2]), ref1_poc=refs1[0][1],
                        weightb=self.bopt["weightb"], direct_spatial=dsp, lowres_mv1=lw1,
                        direct_score=self.dscore_dev.ptr if self.direct_auto else None)
            p.b = C.addressof(sb)
            keep.append(sb)
        for i, r in enumerate(refs):
            p.ref_poc[i] = r[1]
        arr = (C.c_void_p * max(len(refs), 1))(*[C.addressof(self.pool[r[0]]) for r in refs]) if refs else None
        mine = MbState.from_buffer_copy(state.st)        # this chain's view of the state: the device arrays + its own frame-level scalars
        keep += [arr, mine]
        entry = ChainSweep(ci, C.addressof(self.look.pics[slot]), C.cast(arr, C.c_void_p) if arr else None, len(refs), C.addressof(recon),
                                C.addressof(p), C.addressof(refs[0][2]) if refs else None, C.addressof(mine))
        cd = Coded()
        cd.chain, cd.frame, cd.type, cd.slice_type, cd.qp, cd.f_qpm, cd.poc = ci, fr.frame, fr.type, stype, qp, fr.f_qpm, poc
        cd.n_ref0, cd.n_ref1, cd.i_satd = len(refs), len(refs1), fr.i_satd
        cd.frame_num_reset = int(getattr(fr, "frame_num_reset", 0))
        cd.direct_spatial = dsp
        self._undo[ci] = (list(refs_all), self.c_coded[ci])
        self.coded_now[ci] = cd
        if stype != SLICE_B:                           # kept: filtered below, then this chain's newest reference
            self.crefs[ci] = ([(pic_i, poc, mine)] + refs_all)[:self.dpb]
        self.c_coded[ci] += 1
        return entry, cd, pic_i

    def _bufs_of(self, ci):
        """The buffers the chain's next frame writes (AsyncStreamEncoder alternates two sets per chain)."""
        return self.rd_bufs

    def _fill(self, fill, pic, frame):
        fill(pic, frame)
        if self.aq_slots:
            c, L, ro = self.src_ctx, self.lib, self.rd_opt
            en, off = self.aq_slots[self.look.slot(frame)]
            L.x264hip_adaptive_quant_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
            c.check(L.x264hip_adaptive_quant_frame(c.h, C.byref(pic), C.c_float(ro["aq_strength"]), en.p, off.p), "adaptive_quant_frame")

    def payloads(self):
        """slice_data() of the frame each chain coded in the last step (None where it coded nothing); valid after sync()."""
        all_ = super().payloads()
        return [p if self.coded_now[i] is not None else None for i, p in enumerate(all_)]

    def status(self):
        c = self.ctx
        for s in self.states:
            c.check(self.lib.x264hip_slice_sweep_status(c.h, C.byref(s.st)), "slice_sweep_status")

    def close(self):
        self.sync()
        self.src_ctx.sync()
        self.lb.close()
        self.look.close()
        for pair in self.aq_slots or []:
            for a in pair:
                a.free()
        for a in [self.tab_dev] + self.elems_dev + (self.stats_dev or []) + ([self.dscore_dev] if self.dscore_dev else []):
            a.free()
        if self.tab_host:
            self.lib.x264hip_host_free(C.c_void_p(self.tab_host))
            self.tab_host = None
        super().close()
        self.src_ctx.close()
        if self._hp_stream:
            self.lib.x264hip_stream_destroy(C.c_void_p(self._hp_stream))
            self._hp_stream = None


class AsyncStreamEncoder(StreamEncoder):
    """The same streams without the step: every chain gets its next frame as soon as ITS OWN kernel is done.

    In StreamEncoder a step ends when its slowest chain does; a P chain's frame takes about twice as long as a B chain's, so the wave slots
    of the B chains idle for the rest of every step.  Here a host scheduler polls the completion of the I / P kernel and of the B kernel of
    every launch in flight (x264hip_slice_sweep_chains_events, x264hip_event_query), asks the freed chains' queues for their next frames
    (batching the lookahead's cost requests over them) and launches those at once, on one of a small ring of contexts (streams) -- so
    launches of different "ages" overlap and the device stays full.  Chains therefore drift apart: a picture is prepared for ALL chains
    (lowres planes, intra costs, AQ) when the first chain needs it, and the slot ring is `drift` pictures longer; a chain that would run
    further ahead than the ring allows waits.  The results per chain are exactly StreamEncoder's (tests/test_gpu_stream.py).

        enc = AsyncStreamEncoder(lib, w, h, cqm, batch=B, n_frames=N, ...)
        enc.run(fill, on_launch)      # codes N frames of every chain; on_launch(coded, stream) after every launch (e.g. to enqueue a
                                      # payload copy behind it: a chain's payload buffer is reused by its next frame)
    """

    def __init__(self, lib, width, height, cqm, batch=1, n_frames=None, drift=2, launches=8, **kw):
        if n_frames is None:
            raise ValueError("AsyncStreamEncoder: n_frames (pictures per chain) is needed")
        if kw.get("noise_reduction"):
            raise ValueError("AsyncStreamEncoder: --nr updates its tables once per frame for all chains together: lock-step only")
        if kw.get("direct_pred", 1) == 2:
            pass                                         # temporal direct: x264hip_slice_rd.stale is per chain, frames of a chain stay in order
        bf = kw.get("bframes", 0)
        b_adapt = kw.get("b_adapt", 1)
        delay = (max(bf, 3) * 4 if b_adapt == 2 and bf else bf)
        kw.setdefault("n_slots", delay + bf + 3 + drift)
        super().__init__(lib, width, height, cqm, batch=batch, n_frames=None, lookahead_priority=True, **kw)
        from .frame import FrameCtx
        self.total, self.delay, self.drift = n_frames, delay, drift
        B = batch
        lib.x264hip_host_alloc.restype = C.c_void_p
        lib.x264hip_event_create.restype = C.c_void_p
        tb = lib.x264hip_chain_sweep_bytes()
        self.lctx = []
        for _ in range(launches):
            lc = FrameCtx(lib, width, height, batch=B)
            # (element lists in pinned host memory, read by the filter kernels in place: with the device kept full, even a tiny upload's copy
            # kernel would wait for a wave slot)
            self.lctx.append(dict(ctx=lc, tab_host=lib.x264hip_host_alloc(C.c_size_t(tb * B)), tab_dev=DeviceArray(lib, (tb * B,), np.uint8),
                                  elems=[lib.x264hip_host_alloc(C.c_size_t(4 * B)) for _ in range(len(self.pool))],
                                  ev_ip=lib.x264hip_event_create(), ev_b=lib.x264hip_event_create(), busy=False, ip=[], b=[], keep=None))
        # a chain's next frame may be launched while the copy of its previous payload is still queued: two sets of what a frame writes
        d = self.ctx.dims
        alt = self._frame_bufs(B, d.mb_w * d.mb_h, self.payload_cap, 0)
        self.rd_bufs_alt = dict(self.rd_bufs, **alt)
        self._alt_only = alt
        self.las = self.lb.chains                        # one host state machine per chain (the batch object's lock-step driver is not used)
        self.fed = [0] * B
        self.ncoded = [0] * B
        self.inflight_frame = [-1] * B
        self.prepared = 0                                # pictures prepared on the device (for all chains)
        self.n_launches = 0
        self.launch_sizes = []
        if self.post_scenecut:
            raise ValueError("AsyncStreamEncoder: the post-encode scene cut's check is made per step (StreamEncoder); run with pre_scenecut=1")
        if self.direct_auto:
            raise ValueError("AsyncStreamEncoder: --direct auto's running scores are kept per step (StreamEncoder)")

    def _bufs_of(self, ci):
        return self.rd_bufs_alt if self.c_coded[ci] & 1 else self.rd_bufs

    def payload_bufs(self, cd_index, ci):
        """The buffer set chain ci's frame number cd_index (0-based, coding order) was written to."""
        return self.rd_bufs_alt if cd_index & 1 else self.rd_bufs

    def payload_async_of(self, cd, index, c, ev_b, host_len, host_buf, nbytes):
        """From on_launch: enqueue, behind the frame `cd` (its chain's frame number `index`) of the launch on context c, the copy of its payload
        length and first nbytes payload bytes into pinned host memory; valid once c's stream has passed this point (run() returns after that)."""
        from .slice import PAYLOAD_LEAD
        rb = self.payload_bufs(index, cd.chain)
        st = C.c_void_p(c.stream)
        if cd.slice_type == SLICE_B:                     # the B kernel runs on the library's second stream of this context
            self.lib.x264hip_stream_wait_event(st, C.c_void_p(ev_b))
        self.lib.x264hip_memcpy_d2h_async(C.c_void_p(host_len), C.c_void_p(rb["payload_len"].ptr + 4 * cd.chain), C.c_size_t(4), st)
        self.lib.x264hip_memcpy_d2h_async(C.c_void_p(host_buf), C.c_void_p(rb["payload"].ptr + self.payload_cap * cd.chain + PAYLOAD_LEAD), C.c_size_t(nbytes), st)

    # -- pictures --------------------------------------------------------------------------------------------------------------------
    def _oldest_needed(self):
        m = self.prepared
        for ci, la in enumerate(self.las):
            if self.inflight_frame[ci] >= 0:
                m = min(m, self.inflight_frame[ci])
            if self.ncoded[ci] + (1 if self.inflight_frame[ci] >= 0 else 0) < self.total:
                m = min(m, la.oldest_live() if self.fed[ci] else 0)
        return m

    def _prepare_picture(self, fill):
        """Picture number self.prepared for every chain: source, lowres planes, intra costs, AQ offsets.  False: its slot is still in use."""
        f = self.prepared
        if f >= self.total or f - self._oldest >= self.n_slots:
            return False
        pic = self.look.begin_frame(f)
        self._fill(fill, pic, f)
        self.look.prepare(f)
        # what says "picture f is in place" to the sweeps that will read it (they run on other streams)
        if not hasattr(self, "pic_events"):
            self.pic_events = {}
        ev = self.pic_events.pop(f - self.n_slots, None) or self.lib.x264hip_event_create()
        self.lib.x264hip_event_record(C.c_void_p(ev), C.c_void_p(self.src_ctx.stream))
        self.pic_events[f] = ev
        self.prepared += 1
        return True

    # -- the scheduler ---------------------------------------------------------------------------------------------------------------
    def _decide(self, chains, fill):
        """The lookahead's part for `chains` (idle, or with a frame in flight -- the decision does not read its result): pictures in, queues
        asked.  A chain ends up with its next frame in self.next_frame, or pending on a batch of cost tasks launched here (not waited for), or
        back in self.undecided (no slot free for a picture / no launch buffer free)."""
        tasks, specs, owners = [], [], []
        for ci in chains:
            if self.ncoded[ci] + (1 if self.inflight_frame[ci] >= 0 else 0) >= self.total or ci in self.next_frame:
                continue
            want = min(self.total, self.ncoded[ci] + (1 if self.inflight_frame[ci] >= 0 else 0) + self.delay + 1)
            ok = True
            while self.fed[ci] < want:
                if self.fed[ci] >= self.prepared and not self._prepare_picture(fill):
                    ok = False                           # the ring is full: this chain is too far ahead of the slowest one, it waits
                    break
                assert self.las[ci].put() == self.fed[ci]
                self.fed[ci] += 1
            if not ok:
                self.undecided.add(ci)
                continue
            kind, fr, needs = self.las[ci].get(self.fed[ci] >= self.total, self.lb.speculative)
            if kind == LA.FRAME:
                self.next_frame[ci] = fr
            elif kind == LA.NEED:
                if len(tasks) + len(needs) > self.look.max_tasks:
                    self.undecided.add(ci)
                    continue
                for (b, p0, p1, ds0, ds1, spec) in needs:
                    tasks.append((ci, b, p0, p1, ds0, ds1)); specs.append(spec)
                owners.append(ci)
            elif kind != LA.END:
                raise RuntimeError("AsyncStreamEncoder: chain %d has no frame although %d pictures are in" % (ci, self.fed[ci]))
        if tasks:
            h = self.look.run_async(tasks)
            if h is None:                                # every launch buffer is in flight: ask again later
                self.undecided.update(owners)
            else:
                self.lb.rounds += 1
                self.cost_batches.append((h, tasks, specs, owners))

    def run(self, fill, on_launch=None, until=None, poll_s=0.001):
        """Code frames until every chain has coded `until` of them (default: all n_frames), then wait for the device.  May be called again with a
        larger `until` (a benchmark's warm-up, then its timed part)."""
        import time
        L, B = self.lib, self.ctx.batch
        target = self.total if until is None else min(until, self.total)
        if not hasattr(self, "coded_all"):
            self.coded_all = [[] for _ in range(B)]
            self.next_frame, self.cost_batches, self.undecided = {}, [], set()
            self.pic_events = {}
        idle = {ci for ci in range(B) if self.inflight_frame[ci] < 0}           # chains whose previous frame (if any) is done
        done = sum(1 for ci in range(B) if self.ncoded[ci] >= target)
        self._oldest = self._oldest_needed()
        self._decide([ci for ci in sorted(idle) if self.ncoded[ci] < target], fill)
        stats = os.environ.get("X264HIP_ASYNC_STATS")
        t_last, acc = time.perf_counter(), [0.0, 0.0, 0.0, 0.0, 0.0]
        while done < B:
            progressed = False
            if stats:                                    # time-weighted: idle chains, idle chains without a decision, launches in flight, cost batches in flight
                now = time.perf_counter(); dt_ = now - t_last; t_last = now
                n_idle = sum(1 for ci in idle if self.ncoded[ci] < target)
                acc[0] += dt_; acc[1] += dt_ * n_idle; acc[2] += dt_ * sum(1 for ci in idle if self.ncoded[ci] < target and ci not in self.next_frame)
                acc[3] += dt_ * sum(1 for lc in self.lctx if lc["busy"]); acc[4] += dt_ * len(self.cost_batches)
            # 1. sweeps: the B kernel and the I / P kernel (+ its filters) of every launch in flight, separately
            for lc in self.lctx:
                if not lc["busy"]:
                    continue
                for kind in ("b", "ip"):
                    if lc[kind] and L.x264hip_event_query(C.c_void_p(lc["ev_" + kind])) == 1:
                        for ci in lc[kind]:
                            self.inflight_frame[ci] = -1
                            self.ncoded[ci] += 1
                            if self.ncoded[ci] == target:
                                done += 1
                        idle.update(lc[kind]); lc[kind] = []; progressed = True
                if not lc["b"] and not lc["ip"]:
                    lc["busy"], lc["keep"] = False, None
            # 2. cost batches: results in, the chains that asked go on deciding
            if self.cost_batches:
                keep_b, again = [], []
                for (h, tasks, specs, owners) in self.cost_batches:
                    res = self.look.poll(h)
                    if res is None:
                        keep_b.append((h, tasks, specs, owners))
                        continue
                    for (ci, b, p0, p1, ds0, ds1), spec, r in zip(tasks, specs, res):
                        self.las[ci].set_cost(b, p0, p1, int(r[0]), int(r[1]), int(r[2]), spec)
                    again += owners
                    progressed = True
                self.cost_batches = keep_b
                if again:
                    self._oldest = self._oldest_needed()
                    self._decide(again, fill)
            if self.undecided and progressed:
                again, self.undecided = sorted(self.undecided), set()
                self._oldest = self._oldest_needed()
                self._decide(again, fill)
            # 3. launch: idle chains whose next frame is decided
            free = [lc for lc in self.lctx if not lc["busy"]]
            go = [ci for ci in idle if ci in self.next_frame and self.ncoded[ci] < target] if free else []
            if not go:
                if not progressed:
                    time.sleep(poll_s)
                continue
            lc = free[0]
            c = lc["ctx"]
            keep, written, filt, out = [], set(), {}, []
            entries = (ChainSweep * len(go))()
            frames_read = set()
            for k, ci in enumerate(sorted(go)):
                fr = self.next_frame.pop(ci)
                entries[k], cd, pic_i = self._entry(ci, fr, keep)
                written.add(pic_i)
                out.append(cd)
                if cd.slice_type != SLICE_B:
                    filt.setdefault(pic_i, []).append(ci)
                self.inflight_frame[ci] = fr.frame
                self.coded_all[ci].append(cd)
                idle.discard(ci)
                frames_read.add(fr.frame)
            # the pictures these sweeps read were prepared on the lookahead's stream: wait (on the device) for the ones that may not be finished
            for f in frames_read:
                ev = self.pic_events.get(f)
                if ev is not None and L.x264hip_event_query(C.c_void_p(ev)) != 1:
                    L.x264hip_stream_wait_event(C.c_void_p(c.stream), C.c_void_p(ev))
            for pic_i in written:
                c.check(L.x264hip_mb_state_clear_progress(c.h, C.byref(self.states[pic_i].st)), "mb_state_clear_progress")
            for pic_i, chains in filt.items():
                np.ctypeslib.as_array(C.cast(lc["elems"][pic_i], C.POINTER(C.c_int32)), (B,))[:len(chains)] = chains
            ev = None
            if self.sweep_events is not None:
                ev = (L.x264hip_event_create(), L.x264hip_event_create())
                L.x264hip_event_record(C.c_void_p(ev[0]), C.c_void_p(c.stream))
            c.check(L.x264hip_slice_sweep_chains_events(c.h, entries, len(go), C.c_void_p(lc["tab_host"]), lc["tab_dev"].p, C.c_void_p(lc["ev_ip"]),
                                                        C.c_void_p(lc["ev_b"])), "slice_sweep_chains_events")
            if ev:
                L.x264hip_event_record(C.c_void_p(ev[1]), C.c_void_p(c.stream))
                px = self.ctx.dims.mb_w * 16 * self.ctx.dims.lines_y
                self.sweep_events.append((ev[0], ev[1], sum(1 for cd in out if cd.slice_type != SLICE_B), sum(px * (3.0 + 4.5 * (cd.n_ref0 + cd.n_ref1)) for cd in out if cd.slice_type != SLICE_B), "IP"))
            o = self.opt
            for pic_i, chains in filt.items():
                recon, s_ = self.pool[pic_i], self.states[pic_i].st
                c.check(L.x264hip_frame_ctx_elements(c.h, C.c_void_p(lc["elems"][pic_i]), len(chains)), "frame_ctx_elements")
                if o["deblock"]:
                    dp = DeblockParams(mb_type=s_.mb_type, qp=s_.qp, nnz=s_.nnz, transform8x8=s_.t8, mv=s_.mv, ref=s_.ref,
                                       alpha_c0_offset=o["alpha_c0"], beta_offset=o["beta"], chroma_qp_offset=o["chroma_qp_offset"], state_layout=1,
                                       sub8x8=1 if o["inter"] & 0x20 else 0)
                    c.check(L.x264hip_deblock_frame(c.h, C.byref(recon), C.byref(dp)), "deblock_frame")
                c.check(L.x264hip_expand_border(c.h, C.byref(recon), 0), "expand_border")
                c.check(L.x264hip_hpel_filter_frame(c.h, C.byref(recon)), "hpel_filter_frame")
            c.check(L.x264hip_frame_ctx_elements(c.h, None, 0), "frame_ctx_elements")
            self.last_ctx = c
            if on_launch is not None:
                on_launch(out, c, lc["ev_b"])            # e.g. copies of the payloads just produced (a B chain's: behind ev_b)
            # the kept frames' chains are free when the filters are done: the event that says so is recorded behind them
            L.x264hip_event_record(C.c_void_p(lc["ev_ip"]), C.c_void_p(c.stream))
            lc["busy"], lc["keep"] = True, keep
            lc["ip"] = [cd.chain for cd in out if cd.slice_type != SLICE_B]
            lc["b"] = [cd.chain for cd in out if cd.slice_type == SLICE_B]
            launched = [cd.chain for cd in out]
            for ci in launched:
                self.las[ci].end()
            self.n_launches += 1
            self.launch_sizes.append(len(out))
            # their next decisions now, beside the sweeps (x264_ratecontrol_start does not read what the frame in flight produces)
            self._oldest = self._oldest_needed()
            self._decide(launched, fill)
        if stats and acc[0] > 0:
            print("async stats: %.1f s, mean idle chains %.0f (undecided %.0f) of %d, launches in flight %.1f, cost batches in flight %.1f, launches %d"
                  % (acc[0], acc[1] / acc[0], acc[2] / acc[0], B, acc[3] / acc[0], acc[4] / acc[0], self.n_launches), file=sys.stderr)
        for lc in self.lctx:
            lc["ctx"].sync()
        assert self.lib.x264hip_device_synchronize() == 0

    def status(self):
        super().status()
        for lc in self.lctx:
            for s in self.states:
                lc["ctx"].check(self.lib.x264hip_slice_sweep_status(lc["ctx"].h, C.byref(s.st)), "slice_sweep_status")

    def close(self):
        assert self.lib.x264hip_device_synchronize() == 0
        for a in self._alt_only.values():
            a.free()
        for lc in self.lctx:
            lc["tab_dev"].free()
            for hp in [lc["tab_host"]] + lc["elems"]:
                self.lib.x264hip_host_free(C.c_void_p(hp))
            for e in (lc["ev_ip"], lc["ev_b"]):
                self.lib.x264hip_event_destroy(C.c_void_p(e))
            lc["ctx"].close()
        self.lctx = []
        super().close()
