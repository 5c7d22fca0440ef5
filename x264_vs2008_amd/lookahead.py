"""Host mirror of the encoder's frame queue for one GOP chain: ctypes over x264hip_lookahead_* (include/x264hip.h; the logic is the
library's host C, csrc/lookahead_host.hip -- x264_slicetype_decide, x264_rc_analyse_slice, the CQP / CRF rate control).

    la = Lookahead(lib, LookaheadParams(...))
    la.put()                      # a picture enters frames.next (R/encoder/encoder.c:1404-1421)
    kind, frame, needs = la.get(flushing)
        NONE  : the B buffer is filling
        NEED  : `needs` = the x264_slicetype_frame_cost tasks whose results the decision lacks (the first one asked for, the rest speculative);
                compute them, la.set_cost(...) each, call get again
        FRAME : `frame` = what x264_encoder_encode would code now (type, QP, which lowres vectors the 16x16 search is offered)
        END   : flushed
    la.end()                      # x264_ratecontrol_end + x264_reference_update for that frame

Nothing here computes a cost: the GPU does (x264hip_lookahead_cost_frames through x264_vs2008_amd.slice), or, in the CPU tests, the oracle."""
import ctypes as C

NONE, FRAME, NEED, END = 0, 1, 2, 3
TYPE_IDR, TYPE_I, TYPE_P, TYPE_B = 1, 2, 3, 5          # R/x264.h:116-121
RC_CQP, RC_CRF = 0, 1
MAX_NEED = 16


class LookaheadParams(C.Structure):
    """x264hip_lookahead_params"""
    _fields_ = [("mb_w", C.c_int), ("mb_h", C.c_int), ("bframes", C.c_int), ("b_adapt", C.c_int), ("bframe_bias", C.c_int),
                ("keyint_max", C.c_int), ("keyint_min", C.c_int), ("scenecut_threshold", C.c_int), ("pre_scenecut", C.c_int),
                ("rc_method", C.c_int), ("qp_constant", C.c_int), ("rf_constant", C.c_float), ("ip_factor", C.c_float),
                ("pb_factor", C.c_float), ("qcompress", C.c_float), ("qp_min", C.c_int), ("qp_max", C.c_int), ("qp_step", C.c_int)]


def make_params(mb_w, mb_h, bframes=0, b_adapt=1, bframe_bias=0, keyint_max=250, keyint_min=0, scenecut_threshold=40, pre_scenecut=1,
                crf=None, qp=26, ip_factor=1.4, pb_factor=1.3, qcompress=0.6, qp_min=10, qp_max=51, qp_step=4):
    """The reference's defaults (x264_param_default, R/common/common.c:46-150) where not given; keyint_min as x264_validate_parameters
    leaves it (R/encoder/encoder.c:455-457).  crf None: constant QP `qp`."""
    if keyint_max <= 0:
        keyint_max = 1 << 30
    if keyint_min <= 0:
        keyint_min = keyint_max // 10
    keyint_min = max(1, min(keyint_min, keyint_max // 2 + 1))
    return LookaheadParams(mb_w, mb_h, bframes, b_adapt, bframe_bias, keyint_max, keyint_min, scenecut_threshold, pre_scenecut,
                           RC_CQP if crf is None else RC_CRF, qp, 0.0 if crf is None else crf, ip_factor, pb_factor, qcompress, qp_min, qp_max, qp_step)


class Need(C.Structure):
    """x264hip_look_need"""
    _fields_ = [("b", C.c_int), ("p0", C.c_int), ("p1", C.c_int), ("do_search", C.c_int * 2), ("speculative", C.c_int)]


class Frame(C.Structure):
    """x264hip_look_frame"""
    _fields_ = [("frame", C.c_int), ("type", C.c_int), ("poc", C.c_int), ("kept_as_ref", C.c_int), ("qp", C.c_int), ("f_qpm", C.c_float),
                ("ref0_frame", C.c_int), ("ref1_frame", C.c_int), ("lowres_l0", C.c_int), ("lowres_l1", C.c_int), ("i_satd", C.c_int),
                ("frame_num_reset", C.c_int)]


def bind(lib):
    lib.x264hip_lookahead_new.restype = C.c_void_p
    lib.x264hip_lookahead_new.argtypes = [C.POINTER(LookaheadParams)]
    lib.x264hip_lookahead_delete.argtypes = [C.c_void_p]
    lib.x264hip_lookahead_put.argtypes = [C.c_void_p]
    lib.x264hip_lookahead_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(Frame), C.POINTER(Need), C.c_int, C.POINTER(C.c_int)]
    lib.x264hip_lookahead_set_cost.argtypes = [C.c_void_p] + [C.c_int] * 7
    lib.x264hip_lookahead_end.argtypes = [C.c_void_p]
    lib.x264hip_lookahead_scenecut.argtypes = [C.c_void_p]
    lib.x264hip_lookahead_state_bytes.restype = C.c_size_t
    lib.x264hip_lookahead_save.argtypes = [C.c_void_p, C.c_void_p]
    lib.x264hip_lookahead_restore.argtypes = [C.c_void_p, C.c_void_p]
    lib.x264hip_lookahead_oldest_live.argtypes = [C.c_void_p]
    return lib


class Lookahead:
    def __init__(self, lib, params):
        self.lib = bind(lib)
        self.params = params
        self.h = lib.x264hip_lookahead_new(C.byref(params))
        if not self.h:
            raise ValueError("x264hip_lookahead_new refused the parameters (bframes > 16, unknown rate control, ...)")
        self._need = (Need * MAX_NEED)()
        self._n = C.c_int(0)

    def close(self):
        if self.h:
            self.lib.x264hip_lookahead_delete(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def put(self):
        return self.lib.x264hip_lookahead_put(self.h)

    def get(self, flushing=False, speculative=True):
        fr = Frame()
        kind = self.lib.x264hip_lookahead_get(self.h, int(flushing), C.byref(fr), self._need, MAX_NEED, C.byref(self._n))
        if kind < 0:
            raise RuntimeError("x264hip_lookahead_get: end() of the previous frame is missing")
        needs = []
        if kind == NEED:
            for i in range(self._n.value):
                nd = self._need[i]
                if nd.speculative and not speculative:
                    continue
                needs.append((nd.b, nd.p0, nd.p1, nd.do_search[0], nd.do_search[1], nd.speculative))
        return kind, (fr if kind == FRAME else None), needs

    def set_cost(self, b, p0, p1, score, intra_mbs, cost00, speculative=0):
        self.lib.x264hip_lookahead_set_cost(self.h, b, p0, p1, int(score), int(intra_mbs), int(cost00), int(speculative))

    def end(self):
        self.lib.x264hip_lookahead_end(self.h)

    def scenecut(self):
        """Instead of end(): the post-encode scene cut gave up the P picture just coded (x264hip_lookahead_scenecut).  1: get() hands the same
        picture out again as I / IDR; 2: another picture (the last B before it, as the P)."""
        rc = self.lib.x264hip_lookahead_scenecut(self.h)
        if rc < 0:
            raise RuntimeError("x264hip_lookahead_scenecut: no P picture in flight")
        return rc

    def save(self):
        """A copy of the queue's state (x264hip_lookahead_save) to come back to with restore(), or None if it does not fit."""
        if getattr(self, "_state", None) is None:
            self._state = C.create_string_buffer(self.lib.x264hip_lookahead_state_bytes())
        return self._state if self.lib.x264hip_lookahead_save(self.h, self._state) == 0 else None

    def restore(self, state):
        self.lib.x264hip_lookahead_restore(self.h, state)

    def oldest_live(self):
        return self.lib.x264hip_lookahead_oldest_live(self.h)


# ------------------------------------------------------------------------------------------------------------------------------------
# The device half: every chain's lookahead data in HBM and the batched cost kernel (x264hip_lookahead_cost_frames).

class LookSlot(C.Structure):
    """x264hip_look_slot"""
    _fields_ = [("pic", C.c_void_p), ("intra_cost", C.c_void_p), ("mv", C.c_void_p), ("mv_cost", C.c_void_p)]


class LookTask(C.Structure):
    """x264hip_look_task"""
    _fields_ = [("chain", C.c_int), ("slot_b", C.c_int), ("slot_p0", C.c_int), ("slot_p1", C.c_int), ("d0", C.c_int), ("d1", C.c_int),
                ("do_search", C.c_int * 2)]


class LookParams(C.Structure):
    """x264hip_look_params"""
    _fields_ = [("me_method", C.c_int), ("me_range", C.c_int), ("weighted_bipred", C.c_int), ("bframes", C.c_int), ("bframe_bias", C.c_int),
                ("subme_param", C.c_int), ("lossless", C.c_int), ("cost_mv", C.c_void_p), ("cost_mv_range", C.c_int)]


COST_SPAN = 2 * 4 * 2048      # p_cost_mv reaches +-2*4*2048 quarter-pels (R/encoder/analyse.c:191-198)


class LookaheadDevice:
    """The lookahead's per-frame data of every chain of a FrameCtx (batch = chains): a ring of slots, input frame f in slot f % n_slots --
    the picture (source planes + the four half-resolution planes), its intra costs, and fenc->lowres_mvs / lowres_mv_costs for both lists
    and every distance.  Chains take pictures in lock step (one per encoder call), so one ring serves them all; how long a frame stays
    alive differs per chain and is bounded by n_slots (x264hip_lookahead_oldest_live)."""

    def __init__(self, ctx, n_slots, bframes, me_method=1, me_range=16, weightb=0, bframe_bias=0, subme=5, lossless=0, max_tasks=None):
        from .frame import DeviceArray, Picture
        import numpy as np
        self.np, self.ctx, self.lib, self.n_slots, self.bframes = np, ctx, ctx.lib, n_slots, bframes
        lib = self.lib
        d = ctx.dims
        self.n = d.mb_w * d.mb_h
        self.nd = bframes + 1
        self.pics, self.intra, self.mv, self.mv_cost = [], [], [], []
        self.frame_of_slot = [-1] * n_slots
        for _ in range(n_slots):
            pic = Picture()
            ctx.check(lib.x264hip_picture_alloc_lookahead(ctx.h, C.byref(pic)), "picture_alloc_lookahead")
            ctx.pictures.append(pic)
            self.pics.append(pic)
            self.intra.append(DeviceArray(lib, (ctx.batch, self.n), np.int32))
            self.mv.append(DeviceArray(lib, (ctx.batch, 2, self.nd, self.n, 2), np.int16))
            self.mv_cost.append(DeviceArray(lib, (ctx.batch, 2, self.nd, self.n), np.int32))
            # x264hip_malloc zero-fills: edge macroblocks are never searched and read as zero vectors (frame.c:93: memset at allocation)
        self.slots = (LookSlot * n_slots)()
        for i in range(n_slots):
            self.slots[i] = LookSlot(C.addressof(self.pics[i]), self.intra[i].ptr, self.mv[i].ptr, self.mv_cost[i].ptr)
        tab = np.zeros(2 * COST_SPAN + 1, np.int16)
        lib.x264hip_cost_mv_table(C.c_int(1), C.c_int(COST_SPAN), tab.ctypes.data_as(C.c_void_p))      # a->i_lambda = x264_lambda_tab[12] = 1
        self.cost_mv = DeviceArray(lib, tab.shape, np.int16, tab)
        self.params = LookParams(me_method, me_range, weightb, bframes, bframe_bias, subme, lossless, self.cost_mv.ptr, COST_SPAN)
        self.max_tasks = max_tasks or 8 * ctx.batch
        lib.x264hip_lookahead_task_bytes.restype = C.c_size_t
        tb = lib.x264hip_lookahead_task_bytes()
        lib.x264hip_host_alloc.restype = C.c_void_p
        self.staging = lib.x264hip_host_alloc(C.c_size_t(tb * self.max_tasks))
        self.tasks_dev = DeviceArray(lib, (tb * self.max_tasks,), np.uint8)
        self.out_dev = DeviceArray(lib, (self.max_tasks, 4), np.int32)
        self.out_host = lib.x264hip_host_alloc(C.c_size_t(16 * self.max_tasks))
        if not self.staging or not self.out_host:
            raise MemoryError("x264hip_host_alloc")
        self.n_launches = self.n_tasks_run = 0

    def slot(self, frame):
        return frame % self.n_slots

    def picture(self, frame):
        s = self.slot(frame)
        assert self.frame_of_slot[s] == frame, "input frame %d is no longer in its lookahead slot" % frame
        return self.pics[s]

    def begin_frame(self, frame):
        """Claim the slot of input frame `frame`; the caller fills the returned picture's Y, U, V (upload / synth) and calls prepare()."""
        s = self.slot(frame)
        self.frame_of_slot[s] = frame
        return self.pics[s]

    def prepare(self, frame):
        """x264_frame_init_lowres + the intra half of the cost for every chain's copy of `frame` (encoder.c:1415-1418, slicetype.c:186-245)."""
        s, ctx, lib = self.slot(frame), self.ctx, self.lib
        ctx.check(lib.x264hip_lowres_init_frame(ctx.h, C.byref(self.pics[s])), "lowres_init_frame")
        ctx.check(lib.x264hip_lookahead_intra_frame(ctx.h, C.byref(self.pics[s]), self.intra[s].p), "lookahead_intra_frame")

    def run(self, tasks):
        """tasks: [(chain, b, p0, p1, do_search0, do_search1)] with input frame numbers.  Returns int32 [n][3]: score, intra_mbs, cost00."""
        np, ctx, lib = self.np, self.ctx, self.lib
        out = np.zeros((len(tasks), 3), np.int32)
        for base in range(0, len(tasks), self.max_tasks):
            part = tasks[base:base + self.max_tasks]
            arr = (LookTask * len(part))()
            for i, (chain, b, p0, p1, ds0, ds1) in enumerate(part):
                for f in (b, p0, p1):
                    assert self.frame_of_slot[self.slot(f)] == f, "input frame %d is no longer in its lookahead slot" % f
                arr[i] = LookTask(chain, self.slot(b), self.slot(p0), self.slot(p1), b - p0, p1 - b, (C.c_int * 2)(ds0, ds1))
            ctx.check(lib.x264hip_lookahead_cost_frames(ctx.h, self.slots, self.n_slots, arr, len(part), C.byref(self.params), C.c_void_p(self.staging),
                                                        self.tasks_dev.p, self.out_dev.p), "lookahead_cost_frames")
            ctx.check(lib.x264hip_memcpy_d2h_async(C.c_void_p(self.out_host), self.out_dev.p, C.c_size_t(16 * len(part)), C.c_void_p(ctx.stream)), "memcpy_d2h_async")
            ctx.sync()
            res = np.ctypeslib.as_array(C.cast(self.out_host, C.POINTER(C.c_int32)), (len(part), 4))
            out[base:base + len(part)] = res[:, :3]
            self.n_launches += 1
            self.n_tasks_run += len(part)
        return out

    # -- the same without waiting: for a scheduler that keeps the device full (AsyncStreamEncoder) -------------------------------------
    def run_async(self, tasks):
        """Enqueue the tasks' launch; returns a handle for poll().  Task records and results live in pinned host memory that the kernel reads
        and writes in place (no copy kernels, which would queue for a wave slot behind whatever fills the device)."""
        np, ctx, lib = self.np, self.ctx, self.lib
        assert 0 < len(tasks) <= self.max_tasks
        if not hasattr(self, "_ring"):
            lib.x264hip_event_create.restype = C.c_void_p
            tb = lib.x264hip_lookahead_task_bytes()
            self._ring = [dict(staging=lib.x264hip_host_alloc(C.c_size_t(tb * self.max_tasks)), out=lib.x264hip_host_alloc(C.c_size_t(16 * self.max_tasks)),
                               ev=lib.x264hip_event_create(), busy=False) for _ in range(16)]
        slot = next((r for r in self._ring if not r["busy"]), None)
        if slot is None:
            return None
        arr = (LookTask * len(tasks))()
        for i, (chain, b, p0, p1, ds0, ds1) in enumerate(tasks):
            for f in (b, p0, p1):
                assert self.frame_of_slot[self.slot(f)] == f, "input frame %d is no longer in its lookahead slot" % f
            arr[i] = LookTask(chain, self.slot(b), self.slot(p0), self.slot(p1), b - p0, p1 - b, (C.c_int * 2)(ds0, ds1))
        ctx.check(lib.x264hip_lookahead_cost_frames(ctx.h, self.slots, self.n_slots, arr, len(tasks), C.byref(self.params), C.c_void_p(slot["staging"]),
                                                    C.c_void_p(slot["staging"]), C.c_void_p(slot["out"])), "lookahead_cost_frames")
        lib.x264hip_event_record(C.c_void_p(slot["ev"]), C.c_void_p(ctx.stream))
        slot["busy"], slot["n"] = True, len(tasks)
        self.n_launches += 1
        self.n_tasks_run += len(tasks)
        return slot

    def poll(self, handle):
        """None while the launch is running, then its results (int32 [n][3])."""
        if self.lib.x264hip_event_query(C.c_void_p(handle["ev"])) != 1:
            return None
        res = self.np.ctypeslib.as_array(C.cast(handle["out"], C.POINTER(C.c_int32)), (handle["n"], 4))[:, :3].copy()
        handle["busy"] = False
        return res

    def mv_ptr(self, chain, frame, lst, dist):
        """Device address of frames[frame]->lowres_mvs[lst][dist - 1] of one chain ([n_mb][2] int16)."""
        s = self.slot(frame)
        return self.mv[s].ptr + 4 * self.n * ((chain * 2 + lst) * self.nd + dist - 1)

    def mv_host(self, chain, frame, lst, dist):
        a = self.mv[self.slot(frame)].get()
        return a[chain, lst, dist - 1]

    def close(self):
        for a in self.intra + self.mv + self.mv_cost + [self.cost_mv, self.tasks_dev, self.out_dev]:
            a.free()
        for r in getattr(self, "_ring", []):
            self.lib.x264hip_host_free(C.c_void_p(r["staging"])); self.lib.x264hip_host_free(C.c_void_p(r["out"]))
            self.lib.x264hip_event_destroy(C.c_void_p(r["ev"]))
        self._ring = []
        for p in (self.staging, self.out_host):
            if p:
                self.lib.x264hip_host_free(C.c_void_p(p))
        self.staging = self.out_host = None


class LookaheadBatch:
    """The encoder's frame queue for every chain of a FrameCtx: one host state machine per chain, their cost requests computed together.

        lb.put(fill)          # one picture per chain enters; fill(picture, frame) writes its Y, U, V for every chain
        frames = lb.get(flushing)   # per chain: x264hip_look_frame (what to code now) or None (buffer filling / flushed)
        ...encode...
        lb.end(chains)        # those chains' frames are done
    """

    def __init__(self, ctx, params, device, speculative=True, limits=None):
        """limits: pictures per chain (None: unbounded) -- a chain whose stream is shorter stops taking pictures and flushes on its own."""
        self.ctx, self.dev, self.speculative = ctx, device, speculative
        self.chains = [Lookahead(ctx.lib, params) for _ in range(ctx.batch)]
        self.pending = [None] * ctx.batch
        self.rounds = 0
        self.limits = list(limits) if limits is not None else None
        self.fed = 0

    def put(self, fill):
        frame = self.fed
        for ci, la in enumerate(self.chains):
            if self.limits is None or frame < self.limits[ci]:
                f = la.put()
                assert f == frame
        self.fed += 1
        old = self.dev.frame_of_slot[self.dev.slot(frame)]
        assert old < 0 or old < self.oldest_live(), "lookahead ring of %d slots too small: input frame %d is still alive when %d arrives" % (self.dev.n_slots, old, frame)
        fill(self.dev.begin_frame(frame), frame)
        self.dev.prepare(frame)
        return frame

    def get(self, flushing=False, only=None):
        """only: the chains to ask (the others keep None) -- the second attempt after a post-encode scene cut."""
        n = len(self.chains)
        out, waiting = [None] * n, list(range(n) if only is None else only)
        while waiting:
            tasks, owners, still = [], [], []
            for ci in waiting:
                kind, fr, needs = self.chains[ci].get(flushing or (self.limits is not None and self.fed >= self.limits[ci]), self.speculative)
                if kind == NEED:
                    for (b, p0, p1, ds0, ds1, spec) in needs:
                        tasks.append((ci, b, p0, p1, ds0, ds1))
                        owners.append(spec)
                    still.append(ci)
                elif kind == FRAME:
                    out[ci] = fr
                    self.pending[ci] = fr
            if tasks:
                res = self.dev.run(tasks)
                self.rounds += 1
                for (ci, b, p0, p1, ds0, ds1), spec, r in zip(tasks, owners, res):
                    self.chains[ci].set_cost(b, p0, p1, int(r[0]), int(r[1]), int(r[2]), spec)
            waiting = still
        return out

    def end(self, chains=None):
        for ci in (range(len(self.chains)) if chains is None else chains):
            if self.pending[ci] is not None:
                self.chains[ci].end()
                self.pending[ci] = None

    def oldest_live(self):
        return min(la.oldest_live() for la in self.chains)

    def close(self):
        for la in self.chains:
            la.close()
