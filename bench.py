#!/usr/bin/env python3
"""bench.py -- frames/sec of x264's per-macroblock hot loop, with the encoder's own lookahead and rate control around it, on 1..N MI355X.

DEFAULT (--stream 1): every GPU runs B independent STREAMS (chains), each a segment of ONE long synthetic 1080p clip (SURVEY.md 8(d)'s
integer generator; chain g's pictures start g * 4096 frames into it), through what x264_encoder_encode does with BASELINE.md's MED flag
set:

  --crf 23 --ref 3 --bframes 3 --b-adapt 1 --me hex --subme 7 --8x8dct --partitions p8x8,b8x8,i8x8,i4x4 --trellis 1 --weightb
  --mixed-refs --direct spatial, psy-rd 1.0, aq-mode 1, keyint 250, scenecut 40 (looked for after every coded P frame), CABAC, deblock

A "step" is one x264_encoder_encode call for every chain (x264_vs2008_amd/stream.py):

  x264hip_picture_synth            the chain's next picture, synthesised ON THE DEVICE into the lookahead's slot (nothing uploaded)
  x264hip_lowres_init_frame, x264hip_lookahead_intra_frame, x264hip_adaptive_quant_frame     what the encoder does when a picture comes in
  x264hip_lookahead_* (host C) + x264hip_lookahead_cost_frames     x264_slicetype_decide (b-adapt 1, scene cut), x264_rc_analyse_slice and
                                   x264_ratecontrol_start (CRF): every chain places its own B frames and prices its own frames; the
                                   per-frame costs they read (x264_slicetype_frame_cost) are batched GPU launches, one wavefront per task
  x264hip_slice_sweep_chains       the per-macroblock loop in raster order for the frame each chain's queue hands it -- cache_load ->
                                   x264_macroblock_analyse (RD mode decision) -> x264_macroblock_encode (trellis) ->
                                   x264_macroblock_write_cabac -> cache_save for all 8160 macroblocks, one wavefront per chain, the I / P
                                   chains and the B chains of the step side by side; the slice's CABAC payload comes out of the launch
  x264hip_deblock_frame, x264hip_expand_border, x264hip_hpel_filter_frame   a kept frame becomes a reference (the elements that coded one)

config.matches_baseline is true for the default run with config.flag_set saying what that means: MED as BASELINE.md states it -- the
post-encode scene cut is evaluated after every P frame of every chain (a given-up attempt would be coded again inside the step; the
clip has none); --pre-scenecut 1 decides cuts in the lookahead instead.  PARITY IS CHECKED IN THIS RUN: rank 0's
chain 0 also goes through the REFERENCE's whole encoder on the host (frame queue, slice-type decision, rate control, slice loop; the
cpu_baseline leg, before the GPU is touched), and for every frame the GPU side coded for that chain -- warm-up and timed steps alike --
the input number, slice type, QP and payload bytes must equal the reference's, or the run fails (config.parity_checked_frames).

--stream 0: round 2's lock-step chains (closed GOPs of --keyint 12, constant QP, B frames in a fixed pattern, no lookahead);
--preset uhd: BASELINE config 2 (3840x2160, --me umh); --wavefront 1: round 1's configuration (subme 5, no RD / trellis / AQ / entropy
coding; one wavefront per macroblock row); --strong 1: the batch is the TOTAL number of chains, spread over the ranks (strong scaling).

Chains shard across ranks with no data-path collective (independent streams / closed GOPs).
One JSON line on stdout (rank 0).  Launch for N > 1:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from x264_vs2008_amd import lib as L, synth  # noqa: E402
from x264_vs2008_amd import slice as sl  # noqa: E402
from x264_vs2008_amd.frame import cqm_init  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
DIRECT_NAMES = {0: "none", 1: "spatial", 2: "temporal", 3: "auto"}
ME_NAMES = {0: "dia", 1: "hex", 2: "umh", 3: "esa"}
CAPTURE = 1 << 20          # bytes of chain 0's payload copied out per step for the parity check


def analysis_options(args):
    cif = getattr(args, "cif", False)              # BASELINE config 0: --no-cabac --no-deblock (the UF flag set)
    return dict(qp=args.qp, me_method=args.me, me_range=16, subme=args.subme, n_refs=args.refs, fast_pskip=1, dct_decimate=1,
                chroma_me=1, cabac=0 if cif else 1, deblock=0 if cif else 1, keyint=args.keyint, inter=args.inter, intra=args.intra, transform8x8=args.dct8,
                mixed_refs=args.mixed_refs, mv_range=128 if cif else 0)       # the level x264_validate_parameters picks: 1.3 for CIF (mv range 128), 4.0 for 1080p (512)


def rd_options(args):
    """What the raster-order variant adds (x264hip_slice_rd)."""
    o = dict(trellis=args.trellis, psy_rd=args.psy_rd, aq_mode=args.aq_mode, aq_strength=1.0)
    if args.bframes:
        o.update(bframes=args.bframes, weightb=args.weightb, direct_pred=getattr(args, "direct", 1))
    return o


def gpu_options(args):
    """How the GPU side schedules the same work (not an encoder option: the CPU leg does not see it)."""
    o = rd_options(args)
    if args.bframes and args.lanes:
        o.update(lanes=args.lanes)
    return o


def clip_time(d, g, keyint, g_total):
    """Frame number, in the one long clip, of display index d of the chain that codes GOPs g, g + g_total, ..."""
    return ((d // keyint) * g_total + g) * keyint + d % keyint


def _cpu_chain(job):
    """One chain through the reference's loop on one core (own process: the reference keeps process-global tables)."""
    import time as _t
    from oracle import refslice as rs
    width, height, n, kw, ekw, raster, g, keyint, g_total, want_payload = job
    fr = [synth.frame(width, height, clip_time(d, g, keyint, g_total)) for d in range(n)]
    y, u, v = (np.ascontiguousarray(np.stack([f[i] for f in fr])) for i in range(3))
    p = rs.make_params(width, height, n, **kw)
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libx264ref.so")
    t0 = _t.perf_counter()
    if os.path.exists(ref_so):
        out = rs.run_reference2(p, rs.make_ext(**ekw), y, u, v) if raster or not kw["cabac"] else rs.run_reference(p, y, u, v)
        kind = "reference"
    else:
        lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        out = rs.run2(lib, "x264o_encode_chain2", p, rs.make_ext(**ekw), y, u, v) if raster else rs.run(lib, "x264o_encode_chain", p, y, u, v)
        kind = "port"
    spent = _t.perf_counter() - t0
    pays = [bytes(out["payload"][f, :int(out["payload_len"][f])]) for f in range(n)] if want_payload and (raster or not kw["cabac"]) else None
    return spent, kind, pays


SEG = 4096                 # frames of the one long clip between the starts of two chains' segments (stream mode)


def _cpu_stream(job):
    """One chain through the reference's WHOLE encoder on one core: x264_encoder_encode's frame queue, x264_slicetype_decide, the CRF
    rate control and the slice loop (oracle/ref_slice.c refslice_encode_stream).  Own process, like _cpu_chain."""
    import time as _t
    from oracle import refslice as rs
    width, height, n_in, kw, ekw, g, want_payload = job
    fr = [synth.frame(width, height, g * SEG + f) for f in range(n_in)]
    y, u, v = (np.ascontiguousarray(np.stack([f[i] for f in fr])) for i in range(3))
    p = rs.make_params(width, height, n_in, **kw)
    t0 = _t.perf_counter()
    out = rs.run_reference_stream(p, rs.make_ext(**ekw), y, u, v)
    spent = _t.perf_counter() - t0
    recs = None
    if want_payload:
        recs = [(int(out["frame_info2"][f][0]), int(out["frame_info"][f][0]), int(out["frame_info"][f][1]), bytes(out["payload"][f, :int(out["payload_len"][f])]))
                for f in range(n_in)]
    return spent, "reference", recs


def stream_ext(args):
    o = rd_options(args)
    o.update(b_adapt=args.b_adapt, pre_scenecut=1, scenecut_threshold=args.scenecut, crf=args.crf, keyint_min=0)
    return o


def cpu_baseline_stream(args):
    """The stream mode's CPU leg: the reference's whole encoder (lookahead and rate control included) on chain 0's pictures -- its
    coded frames are the parity check's -- then one such encoder per host core on other chains' pictures."""
    import multiprocessing as mp
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libx264ref.so")):
        return None, None
    n_in = args.cpu_frames
    kw, ekw = analysis_options(args), stream_ext(args)
    spent1, kind, recs = _cpu_stream((args.width, args.height, n_in, kw, ekw, 0, True))
    avail = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    cores = max(1, min(avail, args.cpu_procs or avail))
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_stream, [(args.width, args.height, n_in, kw, ekw, 1 + i, False) for i in range(cores)], chunksize=1)
    spent_all = max(r[0] for r in res)
    return {"value": round(cores * n_in / spent_all, 4), "unit": "frames/s", "cores": cores, "kind": kind, "one_core": round(n_in / spent1, 4),
            "sample": "the reference's whole encoder -- frame queue, x264_slicetype_decide (b-adapt %d, pre-scenecut), x264_ratecontrol_start (CRF %.1f), the per-macroblock "
                      "loop with the same options, entropy coding included -- on chains of %d %dx%d frames of the same clip: chain 0 on one core (%.1f s), then %d processes, "
                      "one chain each, on the %d host cores (%.1f s); C compiled -O3, no asm"
                      % (args.b_adapt, args.crf, n_in, args.width, args.height, spent1, cores, cores, spent_all)}, recs


def cpu_baseline(args, g_total):
    """The same loop on the host cores: the REFERENCE's own x264_macroblock_cache_load / _analyse / _encode / _write_cabac /
    _cache_save + x264_frame_deblock_row + x264_frame_filter, compiled from the reference's sources where they lie
    (oracle/_ref/libx264ref.so via oracle/ref_slice.c); our restatement (liboracle.so) when that library is not there.  Measured
    twice on a bounded chain of whole frames: one process on one core -- chain 0 of the GPU run, whose payload bytes are kept for the
    parity check -- and one process per host core (each another chain of the clip).  Runs BEFORE the GPU is initialised (forked
    workers and a live HIP runtime do not mix)."""
    import multiprocessing as mp
    n = args.cpu_frames
    kw, ekw, raster = analysis_options(args), rd_options(args), not args.wavefront
    spent1, kind, pays = _cpu_chain((args.width, args.height, n, kw, ekw, raster, 0, args.keyint, g_total, True))
    avail = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)   # a one-GPU box's CPU share is 16 cores
    cores = max(1, min(avail, args.cpu_procs or avail))
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_chain, [(args.width, args.height, n, kw, ekw, raster, 1 + i, args.keyint, g_total, False) for i in range(cores)], chunksize=1)
    spent_all = max(r[0] for r in res)                # the chains run side by side: the slowest one's encode time (clip synthesis is not counted)
    return {"value": round(cores * n / spent_all, 4), "unit": "frames/s", "cores": cores, "kind": kind,
            "one_core": round(n / spent1, 4),
            "sample": "the same per-macroblock loop with the same options on chains of %d %dx%d frames of the same clip: chain 0 on one core (%.1f s), "
                      "then %d processes, one chain each, on the %d host cores (%.1f s); C compiled -O3, no asm%s"
                      % (n, args.width, args.height, spent1, cores, cores, spent_all, ", entropy coding included" if raster else ", no entropy coding on either side")}, pays


def stream_bytes_per_chain(args, delay):
    """What one stream keeps in HBM (x264_vs2008_amd/stream.py): the lookahead's ring of slots (source + lowres planes, vectors and their costs for both
    lists and every distance, intra costs, AQ), the DPB's pictures with their half-pel planes + one being written, their states, the payload buffer."""
    al = lambda v, a: (v + a - 1) // a * a
    w16, h16 = al(args.width, 16), al(args.height, 16)
    sy = al(w16 + 64, 16)
    y, c = sy * (h16 + 65), al(sy // 2, 16) * (h16 // 2 + 33)
    low = al(w16 // 2 + 64, 16) * (h16 // 2 + 65)
    n = (w16 // 16) * (h16 // 16)
    bf = args.bframes
    slot = y + 2 * c + 4 * low + n * 2 * (bf + 1) * 8 + n * 4 + (n * 8 if args.aq_mode else 0)
    dpb = max(args.refs, 2 if bf else 1)
    return (delay + bf + 3) * slot + (dpb + 1) * (4 * y + 2 * c + n * 440) + args.payload_cap + n * 4


def fit_batch(args, hip, B, delay):
    """The chains that fit: a run that dies of a failed allocation measures nothing.  Leaves 5 % of the free memory alone."""
    free, total = C.c_size_t(0), C.c_size_t(0)
    if hip.x264hip_mem_info(C.byref(free), C.byref(total)) != 0:
        return B
    per = stream_bytes_per_chain(args, delay)
    fit = int(0.95 * free.value // per)
    if fit >= B:
        return B
    fit = max(64, fit // 64 * 64)
    print("bench.py: %d streams need %.0f GB, %.0f GB of HBM are free: running %d" % (B, B * per / 1e9, free.value / 1e9, fit), file=sys.stderr)
    return fit


def run_stream(args, hip, dist, json_fd, rank, world, B, g_first, g_step, cpu, ref_recs, delay):
    """The default: every chain is a stream of its own through the whole encoder.  A step = one x264_encoder_encode call per chain: a
    picture comes in (synthesised on the device into the lookahead's slot; lowres planes, intra costs, AQ offsets follow), the
    lookahead answers what the slice-type decision and the rate control ask (batched x264_slicetype_frame_cost launches), and every
    chain codes the frame its own queue hands it, at its own QP, in one chain-table launch per kernel kind.
    --groups G drives the chains of a GPU as G independently stepping groups (own stream and host thread each).  A step's P chains take
    about twice as long as its B chains and a group waits for its slowest chain, so the idea was that another group's work fills the wave
    slots the finished B chains leave; measured, it does not pay (the kernels are latency-bound per wavefront and every resident chain
    is a wavefront: fewer resident chains run faster each, more run slower), so the default is one group."""
    import threading
    from x264_vs2008_amd.stream import StreamEncoder
    if args.async_:
        return run_stream_async(args, hip, dist, json_fd, rank, world, B, g_first, g_step, cpu, ref_recs, delay)
    o = rd_options(args)
    G = max(1, min(args.groups, B))
    sizes = [len(range(j, B, G)) for j in range(G)]
    offs = [sum(sizes[:j]) for j in range(G)]
    n_coded = args.warmup + args.steps
    check = ref_recs is not None and rank == 0
    cap_n = min(CAPTURE, args.payload_cap - sl.PAYLOAD_LEAD)
    hip.x264hip_host_alloc.restype = C.c_void_p
    pin = hip.x264hip_host_alloc(C.c_size_t(n_coded * (cap_n + 64))) if check else None
    coded0 = []                                          # chain 0's coded frames: (input number, slice type, qp)
    encs = [StreamEncoder(hip, args.width, args.height, cqm_init(hip), batch=sizes[j], crf=args.crf, b_adapt=args.b_adapt, scenecut_threshold=args.scenecut,
                          pre_scenecut=args.pre_scenecut, write=1, levels=False, payload_cap=args.payload_cap, qp_min=0, n_frames=(delay + n_coded) if args.pipeline else None, b_cus=args.b_cus,
                          **analysis_options(args), **o) for j in range(G)]
    d = encs[0].ctx.dims
    px = d.mb_w * 16 * d.lines_y
    kinds = [{"P": 0, "B": 0, "I": 0} for _ in range(G)]
    errors = []
    gate = threading.Barrier(G + 1)
    pcie = None
    if args.pcie:
        if G != 1:
            raise SystemExit("bench.py --pcie: one group")
        from x264_vs2008_amd.frame import FrameCtx
        up = FrameCtx(hip, args.width, args.height, batch=sizes[0])                 # its stream: the transfers' own
        scratch = up.new_picture(source_only=True)
        w, h = args.width, args.height
        host_pic = hip.x264hip_host_alloc(C.c_size_t(w * h * 3 // 2))
        y0, u0, v0 = synth.frame(w, h, 7)
        np.ctypeslib.as_array(C.cast(host_pic, C.POINTER(C.c_uint8)), (w * h * 3 // 2,))[:] = np.concatenate([y0.ravel(), u0.ravel(), v0.ravel()])
        down_bytes = min(args.payload_cap, 256 << 10)                                  # per chain and frame: more than any slice of this clip
        host_pay = hip.x264hip_host_alloc(C.c_size_t(sizes[0] * down_bytes))
        pcie = dict(ctx=up, pic=scratch, host=host_pic, pay=host_pay, down=down_bytes, h2d=0, d2h=0)
        hip.x264hip_picture_upload_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]

    def move_over_pcie(enc):
        """What a host-fed encoder moves per step: every chain's next picture up, every chain's payload down (on the transfers' stream)."""
        up, w, h = pcie["ctx"], args.width, args.height
        st = C.c_void_p(up.stream)
        for b in range(up.batch):
            up.select(b)
            up.check(hip.x264hip_picture_upload_async(up.h, C.byref(pcie["pic"]), C.c_void_p(pcie["host"]), w, C.c_void_p(pcie["host"] + w * h), w // 2,
                                                      C.c_void_p(pcie["host"] + w * h * 5 // 4), w // 2, st), "picture_upload_async")
        pcie["h2d"] += up.batch * w * h * 3 // 2
        rb = enc.rd_bufs
        for b in range(up.batch):          # the payloads of the frames coded one step ago are final: this step's sweep was launched behind a wait for them
            hip.x264hip_memcpy_d2h_async(C.c_void_p(pcie["pay"] + b * pcie["down"]), C.c_void_p(rb["payload"].ptr + enc.payload_cap * b), C.c_size_t(pcie["down"]), st)
        pcie["d2h"] += up.batch * pcie["down"]

    def one_step(j, timed):
        enc = encs[j]
        # group j's chain b is chain g_first + (offs[j] + b) * g_step of the job: its pictures start SEG frames after the previous chain's
        out = enc.step(lambda pic, f: enc.src_ctx.synth(pic, (g_first + offs[j] * g_step) * SEG + f, g_step * SEG))
        if out and check and j == 0:
            c0 = enc.coded_now[0]
            if c0 is not None and len(coded0) < n_coded:
                k = len(coded0)
                enc.payload_async(0, pin + k * (cap_n + 64), pin + k * (cap_n + 64) + 64, cap_n)
                coded0.append((c0.frame, c0.slice_type, c0.qp))
        if timed:
            for cd in out:
                kinds[j]["PBI"[cd.slice_type]] += 1
        if pcie is not None and out:
            move_over_pcie(enc)
        return out

    def worker(j):
        try:
            enc = encs[j]
            for _ in range(delay):                       # the B buffer fills: nothing is coded yet (encoder.c:1423-1430)
                assert not one_step(j, False)
            for _ in range(args.warmup):
                assert len(one_step(j, False)) == sizes[j]
            enc.sync()
            enc.status()
            gate.wait()                                  # everyone warmed up
            gate.wait()                                  # the clock runs
            enc.sweep_events = []
            for _ in range(args.steps):
                assert len(one_step(j, True)) == sizes[j]
            enc.sync()
        except BaseException as ex:                      # noqa: B036 -- reported by the main thread
            errors.append(ex)
            gate.abort()
            return
        gate.wait()

    threads = [threading.Thread(target=worker, args=(j,)) for j in range(G)]
    for t in threads:
        t.start()
    try:
        gate.wait()
        assert hip.x264hip_device_synchronize() == 0
        if dist is not None:
            dist.barrier()
        rounds0, tasks0 = sum(e.lb.rounds for e in encs), sum(e.look.n_tasks_run for e in encs)
        t0 = time.perf_counter()
        gate.wait()
        gate.wait()                                      # every group's timed steps are done and synchronised
        assert hip.x264hip_device_synchronize() == 0
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
    except threading.BrokenBarrierError:
        pass
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    for e in encs:
        e.status()
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
        nb = torch.tensor([B], dtype=torch.int64)
        dist.all_reduce(nb, op=dist.ReduceOp.SUM)
        chains_total = int(nb[0])
    else:
        chains_total = B

    checked = 0
    if check:
        for k, (frame, st, qp) in enumerate(coded0):
            base = pin + k * (cap_n + 64)
            n = C.c_int32.from_address(base).value
            got = C.string_at(base + 64, min(n, cap_n))
            rf, rst, rqp, want = ref_recs[k]
            if (frame, st, qp) != (rf, rst, rqp):
                raise SystemExit("bench.py: PARITY FAILURE -- chain 0, coded frame %d: the GPU side codes input %d as slice type %d at QP %d, the reference input %d as "
                                 "type %d at QP %d" % (k, frame, st, qp, rf, rst, rqp))
            if n != len(want) or got != want[:cap_n]:
                raise SystemExit("bench.py: PARITY FAILURE -- chain 0, coded frame %d (input %d): the GPU's slice payload (%d bytes) differs from the reference's "
                                 "(%d bytes)" % (k, frame, n, len(want)))
            checked += 1
        hip.x264hip_host_free(C.c_void_p(pin))

    # the dominant kernels, timed live with HIP events on their launch stream: per step and group the chain-table launches together
    evs = [ev for e in encs for ev in e.sweep_events]
    ms_all = [hip.x264hip_event_elapsed_ms(C.c_void_p(a), C.c_void_p(b)) for a, b, _, _, _ in evs]
    by_all = [by for _, _, _, by, _ in evs]
    for a, b, _, _, _ in evs:
        hip.x264hip_event_destroy(C.c_void_p(a)); hip.x264hip_event_destroy(C.c_void_p(b))
    sweep_ms, sweep_bytes = float(np.mean(ms_all)), int(np.mean(by_all))
    achieved = sweep_bytes / (sweep_ms * 1e-3) / 1e9
    # HBM-side traffic of a step's sweep launches: the committed rocprofv3 --pmc measurement of this configuration (bytes per coded frame of a
    # chain, by slice type) times this run's mix; null for any other configuration
    traffic, traffic_note = None, "no rocprofv3 PMC measurement committed for this configuration"
    tpath = os.path.join(ROOT, "profiles", "r03_stream_traffic.json")
    defaults = (args.width, args.height, args.me, args.inter & 0x33, args.intra, args.dct8, args.mixed_refs, args.subme, args.refs, args.bframes, args.b_adapt) == \
               (1920, 1080, 1, 0x13, 0x3, 1, 1, 7, 3, 3, 1) and (args.trellis, args.psy_rd, args.aq_mode, args.crf, args.keyint, args.scenecut, args.weightb) == (1, 1.0, 1, 23.0, 250, 40, 1)
    if defaults and os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        per = tj["bytes_per_chain_frame"]
        ks = {k: sum(kk[k] for kk in kinds) for k in "PBI"}
        n_l = max(len(evs), 1)
        if all(k in per for k in "PBI" if ks[k]):
            traffic = int(sum(ks[k] * (per[k]["fetch"] + per[k]["write"]) for k in "PBI" if ks[k]) / n_l)
            traffic_note = ("FETCH_SIZE + WRITE_SIZE per coded frame of a chain by slice type (rocprofv3 --pmc, separate passes, raw request-granular counters, measured at %d chains; "
                            "profiles/r03_stream_traffic.json) times this run's slice types, per step" % tj["batch"])
    if rank == 0:
        fps = chains_total * args.steps / dt
        frame_bytes = px * (1.5 + 4.5 * args.refs + 1.5 + 3.0 + 4.0)
        size = "%dp" % args.height
        ksum = {k: sum(kk[k] for kk in kinds) for k in "PBI"}
        total = sum(ksum.values())
        tasks = sum(e.look.n_tasks_run for e in encs) - tasks0
        rounds = sum(e.lb.rounds for e in encs) - rounds0
        psc = " --pre-scenecut" if args.pre_scenecut else ""
        metric = ("encoded frames/sec, %s, preset=medium's flag set with the encoder's own lookahead and rate control (--crf %.0f --b-adapt %d%s, %s, subme %d RD, "
                  "trellis %d, psy-rd, aq-mode %d, CABAC payload on the GPU); 1/2/4/8 MI355X (bit-exact)"
                  % (size, args.crf, args.b_adapt, psc, ME_NAMES[args.me], args.subme, args.trellis, args.aq_mode))
        what = ("%dx%d streams (one per chain, segments of one synthetic clip) through x264_encoder_encode's path on the GPU: pictures synthesised on the device, "
                "x264_frame_init_lowres + lookahead costs (x264_slicetype_frame_cost, one wavefront per task) feeding the library's x264_slicetype_decide / x264_ratecontrol_start "
                "(host C), then the per-macroblock loop in raster order (one wavefront per chain: cache_load, x264_macroblock_analyse with RD mode decision, x264_macroblock_encode, "
                "x264_macroblock_write_cabac, cache_save), deblock, borders, half-pel planes; --crf %.1f --ref %d --bframes %d --b-adapt %d --weightb --direct %s --me %s "
                "--subme %d --trellis %d --psy-rd %.1f --aq-mode %d --8x8dct %d --mixed-refs %d --partitions 0x%x/0x%x --keyint %d --scenecut %d%s, chroma ME, fast "
                "P-skip, dct-decimate, CABAC; payload bytes stay on the device (the host downloads them and writes the headers around them: x264_vs2008_amd/mux.py)"
                % (args.width, args.height, args.crf, args.refs, args.bframes, args.b_adapt, DIRECT_NAMES[args.direct], ME_NAMES[args.me], args.subme, args.trellis, args.psy_rd, args.aq_mode,
                   args.dct8, args.mixed_refs, args.inter, args.intra, args.keyint, args.scenecut, psc))
        # Default: BASELINE.md's MED as it stands -- no --pre-scenecut: x264_encoder_encode's look at every coded P frame (the post-encode scene cut) is
        # evaluated from the sweep's statistics every step; a hit would be coded again inside the step (StreamEncoder), and this clip has none.
        # The reference leg (the harness) only has the pre-encode scene cut; neither fires here, and the harness run is the reference CLI's stream byte
        # for byte (tests/test_cpu_mux.py: the md5 of BASELINE config 2).
        missing = []
        flagset = ("MED (BASELINE.md) + --pre-scenecut on both sides: the pre-encode scene cut instead of the one that re-encodes a frame -- what the reference does itself with "
                   "--threads > 1") if args.pre_scenecut else \
                  ("MED (BASELINE.md) as it stands: the post-encode scene cut is evaluated after every P frame of every chain (x264hip_frame_stats + x264hip_scenecut_post); "
                   "a given-up attempt is coded again inside the step (x264hip_lookahead_scenecut) -- no chain of this clip has one.  The whole stream of this flag set on the hd24 clip "
                   "(version SEI, parameter sets, slice headers: x264_vs2008_amd/mux.py) has the md5 of the reference CLI's file (tests/test_gpu_mux.py)")
        line = {
            "metric": metric,
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": what, "matches_baseline": bool(args.preset == "hd" and defaults), "flag_set": flagset, "missing": missing,
                       "baseline_metric": "encoded frames/sec, 1080p preset=medium, 1/2/4/8 MI355X (bit-exact)",
                       "frames_per_step": chains_total, "frames_in_flight": chains_total, "keyint": args.keyint, "groups_per_gpu": G,
                       "per_chain_fps": round(fps / chains_total, 4),
                       "latency_note": "throughput exists only with thousands of streams in flight: one chain advances one frame per step",
                       "slice_types_in_timed_steps": {k: round(v / max(total, 1), 4) for k, v in ksum.items()},
                       "lookahead": {"cost_tasks_per_step_and_chain": round(tasks / max(args.steps * B, 1), 3),
                                     "cost_launch_rounds_per_step_and_group": round(rounds / max(args.steps * G, 1), 3), "slots": encs[0].n_slots, "delay": delay},
                       "parallelism": "B streams per GPU in %d independently stepping groups (own stream and host thread each); per group and step one chain-table launch per kernel kind "
                                      "(I / P and B chains side by side), one wavefront per chain walking its frame in raster order; lookahead cost tasks one wavefront each; chains "
                                      "shard across GPUs with no data-path collective" % G,
                       "parity_checked_frames": checked,
                       "pcie": None if pcie is None else {"host_to_device_bytes_per_step": pcie["h2d"] // max(args.warmup + args.steps, 1),
                                                          "device_to_host_bytes_per_step": pcie["d2h"] // max(args.warmup + args.steps, 1),
                                                          "note": "every step moved one I420 picture per chain up from pinned memory and %d KB of every chain's payload buffer down, "
                                                                  "on a stream of their own beside the kernels: value is the PCIe-inclusive rate" % (pcie["down"] >> 10)},
                       "parity": ("chain 0 of rank 0, all %d coded frames of this run (%d of them timed): input order, slice types, QPs and payload bytes equal the reference's whole "
                                  "encoder (frame queue, slicetype decision, CRF, per-macroblock loop) run on the same pictures" % (checked, max(0, checked - args.warmup))) if checked else
                                 "not checked in this run (no CPU leg: --no-cpu, no oracle/_ref, or more than one rank)"},
            "roofline": {"bound": "hbm", "kernel": "k_slice_sweep<raster, chain table>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "traffic_note": traffic_note,
                         "avg_launch_ms": round(sweep_ms, 4), "algorithmic_bytes_per_launch": sweep_bytes,
                         "note": "one 'launch' = one group's chain-table launches of a step together (the I / P kernel and the B kernel, side by side on two streams; HIP events around "
                                 "the pair), %d chains; %d groups overlap, so the launches' durations add up to more than the wall clock; the sweep is bound by the serial macroblock "
                                 "chain of a slice (%d macroblocks one after the other per frame, %d frames in flight), not by bandwidth; whole-frame algorithmic bytes = %d -> %.1f GB/s at "
                                 "this fps" % (sizes[0], G, d.mb_w * d.mb_h, B, frame_bytes, frame_bytes * (fps / world) / 1e9)},
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    for e in encs:
        e.close()
    if dist is not None:
        dist.destroy_process_group()


def run_stream_async(args, hip, dist, json_fd, rank, world, B, g_first, g_step, cpu, ref_recs, delay):
    """Stream mode without steps (x264_vs2008_amd/stream.py: AsyncStreamEncoder): warm-up = every chain codes `warmup` frames, then the
    clock runs while every chain codes `steps` more, each at its own pace; the device is drained at both ends of the timed part."""
    from x264_vs2008_amd.stream import AsyncStreamEncoder
    o = rd_options(args)
    n_coded = args.warmup + args.steps
    enc = AsyncStreamEncoder(hip, args.width, args.height, cqm_init(hip), batch=B, n_frames=delay + n_coded, drift=args.drift, launches=args.launches, crf=args.crf,
                             b_adapt=args.b_adapt, scenecut_threshold=args.scenecut, pre_scenecut=1, write=1, levels=False, payload_cap=args.payload_cap, qp_min=0,
                             **analysis_options(args), **o)
    d = enc.ctx.dims
    px = d.mb_w * 16 * d.lines_y
    check = ref_recs is not None and rank == 0
    cap_n = min(CAPTURE, args.payload_cap - sl.PAYLOAD_LEAD)
    hip.x264hip_host_alloc.restype = C.c_void_p
    pin = hip.x264hip_host_alloc(C.c_size_t(n_coded * (cap_n + 64))) if check else None
    coded0 = []

    def fill(pic, f):
        enc.src_ctx.synth(pic, g_first * SEG + f, g_step * SEG)

    def on_launch(coded, ctx, ev_b):
        if not check:
            return
        for cd in coded:
            if cd.chain == 0 and len(coded0) < n_coded:
                k = len(coded0)
                enc.payload_async_of(cd, k, ctx, ev_b, pin + k * (cap_n + 64), pin + k * (cap_n + 64) + 64, cap_n)
                coded0.append((cd.frame, cd.slice_type, cd.qp))

    enc.run(fill, on_launch, until=args.warmup)
    enc.status()
    if dist is not None:
        dist.barrier()
    enc.sweep_events = []
    rounds0, tasks0, launches0 = enc.lb.rounds, enc.look.n_tasks_run, enc.n_launches
    t0 = time.perf_counter()
    enc.run(fill, on_launch, until=n_coded)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    enc.status()
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
        nb = torch.tensor([B], dtype=torch.int64)
        dist.all_reduce(nb, op=dist.ReduceOp.SUM)
        chains_total = int(nb[0])
    else:
        chains_total = B
    checked = 0
    if check:
        for k, (frame, st, qp) in enumerate(coded0):
            base = pin + k * (cap_n + 64)
            n = C.c_int32.from_address(base).value
            got = C.string_at(base + 64, min(n, cap_n))
            rf, rst, rqp, want = ref_recs[k]
            if (frame, st, qp) != (rf, rst, rqp):
                raise SystemExit("bench.py: PARITY FAILURE -- chain 0, coded frame %d: the GPU side codes input %d as slice type %d at QP %d, the reference input %d as "
                                 "type %d at QP %d" % (k, frame, st, qp, rf, rst, rqp))
            if n != len(want) or got != want[:cap_n]:
                raise SystemExit("bench.py: PARITY FAILURE -- chain 0, coded frame %d (input %d): the GPU's slice payload (%d bytes) differs from the reference's "
                                 "(%d bytes)" % (k, frame, n, len(want)))
            checked += 1
        hip.x264hip_host_free(C.c_void_p(pin))
    evs = enc.sweep_events
    ms_all = [hip.x264hip_event_elapsed_ms(C.c_void_p(a), C.c_void_p(b)) for a, b, _, _, _ in evs]
    n_ip = [n for _, _, n, _, _ in evs]
    by_all = [by for _, _, _, by, _ in evs]
    for a, b, _, _, _ in evs:
        hip.x264hip_event_destroy(C.c_void_p(a)); hip.x264hip_event_destroy(C.c_void_p(b))
    sweep_ms = float(np.mean(ms_all)) if ms_all else 0.0
    sweep_bytes = int(np.mean(by_all)) if by_all else 0
    achieved = sweep_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms else 0.0
    if rank == 0:
        fps = chains_total * args.steps / dt
        frame_bytes = px * (1.5 + 4.5 * args.refs + 1.5 + 3.0 + 4.0)
        size = "%dp" % args.height
        ksum = {"P": 0, "B": 0, "I": 0}
        for per in enc.coded_all:
            for cd in per[args.warmup:n_coded]:
                ksum["PBI"[cd.slice_type]] += 1
        total = sum(ksum.values())
        sizes = enc.launch_sizes[launches0:]
        metric = ("encoded frames/sec, %s, preset=medium's flag set with the encoder's own lookahead and rate control (--crf %.0f --b-adapt %d --pre-scenecut, %s, subme %d RD, "
                  "trellis %d, psy-rd, aq-mode %d, CABAC payload on the GPU); 1/2/4/8 MI355X (bit-exact)"
                  % (size, args.crf, args.b_adapt, ME_NAMES[args.me], args.subme, args.trellis, args.aq_mode))
        what = ("%dx%d streams (one per chain, segments of one synthetic clip) through x264_encoder_encode's path on the GPU: pictures synthesised on the device, "
                "x264_frame_init_lowres + lookahead costs (x264_slicetype_frame_cost, one wavefront per task) feeding the library's x264_slicetype_decide / x264_ratecontrol_start "
                "(host C), then the per-macroblock loop in raster order (one wavefront per chain: cache_load, x264_macroblock_analyse with RD mode decision, x264_macroblock_encode, "
                "x264_macroblock_write_cabac, cache_save), deblock, borders, half-pel planes; --crf %.1f --ref %d --bframes %d --b-adapt %d --weightb --direct %s --me %s "
                "--subme %d --trellis %d --psy-rd %.1f --aq-mode %d --8x8dct %d --mixed-refs %d --partitions 0x%x/0x%x --keyint %d --scenecut %d --pre-scenecut, chroma ME, fast "
                "P-skip, dct-decimate, CABAC; payload bytes stay on the device (slice / NAL headers and the download are the host's).  No global step: a host scheduler launches "
                "every chain's next frame as soon as that chain's own kernel has finished; warm-up = %d frames of every chain, timed = the next %d frames of every chain, the "
                "device drained before and after" % (args.width, args.height, args.crf, args.refs, args.bframes, args.b_adapt, DIRECT_NAMES[args.direct], ME_NAMES[args.me], args.subme, args.trellis,
                                                      args.psy_rd, args.aq_mode, args.dct8, args.mixed_refs, args.inter, args.intra, args.keyint, args.scenecut, args.warmup, args.steps))
        # the flag set is BASELINE.md's MED with --pre-scenecut ON BOTH SIDES (GPU and the reference leg): the reference forces that flag itself with
        # --threads > 1, and BASELINE.md prescribes it for GOP-sharded runs; the scene cut that re-encodes a frame is the only thing it replaces
        missing = []
        flagset = ("MED (BASELINE.md) + --pre-scenecut on both sides: the pre-encode scene cut instead of the one that re-encodes a frame -- what the reference does itself with "
                   "--threads > 1")
        line = {
            "metric": metric,
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": what, "matches_baseline": False, "flag_set": flagset, "missing": missing + ["the step-less scheduler is an experiment: the default run is --async 0"],
                       "baseline_metric": "encoded frames/sec, 1080p preset=medium, 1/2/4/8 MI355X (bit-exact)",
                       "frames_per_step": chains_total, "frames_in_flight": chains_total, "keyint": args.keyint,
                       "per_chain_fps": round(fps / chains_total, 4),
                       "latency_note": "throughput exists only with thousands of streams in flight: one chain advances one frame per 'step' on average",
                       "slice_types_in_timed_steps": {k: round(v / max(total, 1), 4) for k, v in ksum.items()},
                       "scheduler": {"launches_in_timed_part": len(sizes), "mean_chains_per_launch": round(float(np.mean(sizes)), 1) if sizes else 0,
                                     "streams": args.launches, "drift_pictures": args.drift},
                       "lookahead": {"cost_tasks_per_frame": round((enc.look.n_tasks_run - tasks0) / max(args.steps * B, 1), 3),
                                     "cost_launches": enc.lb.rounds - rounds0, "slots": enc.n_slots, "delay": delay},
                       "parallelism": "B streams per GPU, each a wavefront per frame; a host scheduler polls the I / P and the B kernel of every launch in flight and launches the freed "
                                      "chains' next frames at once (chain-table launches on a ring of streams), so chains drift apart by up to %d pictures; lookahead cost tasks one "
                                      "wavefront each; chains shard across GPUs with no data-path collective" % args.drift,
                       "parity_checked_frames": checked,
                       "parity": ("chain 0 of rank 0, all %d coded frames of this run (%d of them timed): input order, slice types, QPs and payload bytes equal the reference's whole "
                                  "encoder (frame queue, slicetype decision, CRF, per-macroblock loop) run on the same pictures" % (checked, max(0, checked - args.warmup))) if checked else
                                 "not checked in this run (no CPU leg: --no-cpu, no oracle/_ref, or more than one rank)"},
            "roofline": {"bound": "hbm", "kernel": "k_slice_sweep<raster, chain table>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                         "traffic_note": "no rocprofv3 PMC measurement committed for this configuration",
                         "avg_launch_ms": round(sweep_ms, 4), "algorithmic_bytes_per_launch": sweep_bytes,
                         "note": "one 'launch' = the I / P chain-table kernel of one scheduler launch (HIP events on its stream; its B chains run in a kernel of their own on a second "
                                 "stream and are not in this figure), mean %d I / P chains; many launches overlap, so their durations add up to far more than the wall clock; the "
                                 "sweep is bound by the serial macroblock chain of a slice (%d macroblocks one after the other per frame, %d frames in flight), not by bandwidth; "
                                 "whole-frame algorithmic bytes = %d -> %.1f GB/s at this fps" % (int(np.mean(n_ip)) if n_ip else 0, d.mb_w * d.mb_h, B, frame_bytes,
                                                                                               frame_bytes * (fps / world) / 1e9)},
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    enc.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="0: 12 (raster variant) / 24 (--wavefront 1)")
    ap.add_argument("--warmup", type=int, default=-1, help="-1: 2 (raster variant) / 3 (--wavefront 1)")
    ap.add_argument("--batch", type=int, default=0, help="independent GOP chains advanced per step on each GPU; 0: 2048 (8 wavefronts on each of the 256 CUs) for the raster "
                    "variant (one wavefront per chain, all resident: what its LDS and registers allow), 512 with --preset uhd, 240 with --wavefront 1")
    ap.add_argument("--strong", type=int, default=0, help="1: --batch is the total number of chains of the job, spread round-robin over the ranks (strong scaling: total work fixed)")
    ap.add_argument("--preset", default="hd", choices=["hd", "uhd", "slow", "cif"], help="cif: BASELINE config 0 (352x288, the UF flag set: --qp 26 --no-cabac --me dia --subme 0 --partitions none "
                    "--no-deblock --ref 1 --bframes 0, lock-step I / P chains in the wavefront variant, payload from the CAVLC writer, parity-checked); "
                    "hd: BASELINE config 1 (1920x1080, hex); uhd: config 2 (3840x2160, --me umh); slow: config 4's flag set on "
                    "one GPU as BASELINE.md states it (1920x1080, --ref 5 --b-adapt 2 --me umh --subme 8 --direct auto, the post-encode scene cut)")
    ap.add_argument("--direct", type=int, default=1, help="param.analyse.i_direct_mv_pred: 1 spatial, 2 temporal, 3 auto (stream mode; --preset slow sets it)")
    ap.add_argument("--wavefront", type=int, default=0, help="1: round 1's configuration (wavefront schedule, subme 5, no RD / trellis / AQ / entropy coding)")
    ap.add_argument("--trellis", type=int, default=1)
    ap.add_argument("--bframes", type=int, default=-1, help="disposable B frames between anchors, fixed pattern (-1: 3 for the raster variant = the medium "
                    "preset's --bframes 3 without b-adapt; 0 with --wavefront 1)")
    ap.add_argument("--weightb", type=int, default=1)
    ap.add_argument("--payload-cap", type=int, default=0, help="bytes of payload buffer per chain and frame in flight (0: 1 MiB at 1080p, 4 MiB at 2160p; the library's default, "
                    "800 B per macroblock, is x264's worst case)")
    ap.add_argument("--lanes", type=int, default=0, help="extra streams for the B frames of a mini-GOP, which then run beside the next anchor (0: one stream, frames in lock step -- "
                    "with every wave slot taken by a launch's chains the lanes gain nothing, DESIGN.md 3.1c; -1: one per B frame of the pattern)")
    ap.add_argument("--psy-rd", type=float, default=1.0)
    ap.add_argument("--aq-mode", type=int, default=1)
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the all-core CPU leg (0: one per host core)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--qp", type=int, default=26)
    ap.add_argument("--refs", type=int, default=3)
    ap.add_argument("--subme", type=int, default=0, help="0: 7 (raster variant) / 5 (--wavefront 1)")
    ap.add_argument("--me", type=int, default=-1, help="param.analyse.i_me_method: 0 dia, 1 hex (the medium preset), 2 umh (-1: by --preset)")
    ap.add_argument("--keyint", type=int, default=0, help="0: 12 (raster variant) / 24 (--wavefront 1)")
    ap.add_argument("--inter", type=lambda v: int(v, 0), default=0x13, help="param.analyse.inter: X264_ANALYSE_I4x4 0x1 | I8x8 0x2 | PSUB16x16 0x10 | "
                    "PSUB8x8 0x20 (the medium preset's p8x8 = 0x10; 0x33 adds p4x4 / p8x4 / p4x8)")
    ap.add_argument("--mixed-refs", type=int, default=1, help="param.analyse.b_mixed_references")
    ap.add_argument("--intra", type=lambda v: int(v, 0), default=0x3, help="param.analyse.intra")
    ap.add_argument("--dct8", type=int, default=1, help="param.analyse.b_transform_8x8")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU leg's chains (0: warmup + steps, every GPU frame of chain 0 is then checked)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU leg (and with it the parity check of this run)")
    ap.add_argument("--stream", type=int, default=-1, help="1: the real lookahead and rate control in front of the sweep (x264_vs2008_amd/stream.py): --crf, --b-adapt, pre-scenecut, "
                    "lowres motion candidates, every chain placing its own B frames and pricing its own frames (-1: 1 for the raster variant, 0 with --wavefront 1); "
                    "0: round 2's lock-step chains at constant QP with a fixed B pattern")
    ap.add_argument("--crf", type=float, default=23.0)
    ap.add_argument("--async", dest="async_", type=int, default=0, help="stream mode: 1: no steps -- a host scheduler hands every chain its next frame as soon as that chain's own kernel is done "
                    "(AsyncStreamEncoder; the timed part is then steps frames of every chain, not steps launches; set GPU_MAX_HW_QUEUES=24: the launches need hardware queues of their "
                    "own); 0 (default): one launch set per step, the step as long as its slowest chain.  Measured on the default workload: 404 frames/s against 455 -- the chains of the "
                    "synthetic clip decide alike for their first frames, so they finish together anyway and the scheduler only adds launches (DESIGN.md 3.3)")
    ap.add_argument("--drift", type=int, default=2, help="--async 1: pictures a chain may be ahead of the slowest one (each costs a lookahead slot per chain)")
    ap.add_argument("--launches", type=int, default=12, help="--async 1: launches in flight (streams)")
    ap.add_argument("--pcie", type=int, default=0, help="stream mode (steps): 1: every step also moves what a host-fed encoder would move over PCIe, beside the kernels on a stream of its "
                    "own -- one I420 picture per chain up (x264hip_picture_upload_async from pinned memory into a scratch picture; the content that is coded still comes from the device "
                    "generator) and every chain's payload buffer down -- so that value becomes the PCIe-inclusive rate (DESIGN.md 6; never the default)")
    ap.add_argument("--b-cus", type=int, default=0, help="stream mode (steps): compute units [0, N) for the step's B kernel, the rest for its I / P kernel (0: both everywhere)")
    ap.add_argument("--pipeline", type=int, default=1, help="stream mode: 1: every step prepares the next step's lookahead (picture in, costs, decisions) beside its own sweep, on a stream "
                    "of its own -- the lookahead's kernels and the host's work fill the time the step's P chains run on after its B chains; 0: one after the other")
    ap.add_argument("--groups", type=int, default=1, help="stream mode: independently stepping groups of chains per GPU (own stream and host thread each); measured: 1 is best -- "
                    "the kernels are bound by each wavefront's own latency, so groups out of phase only slow one another (2048 chains: 428 frames/s in one group, 256 / 185 / 130 "
                    "with 1536 chains in 2 / 3 / 4)")
    ap.add_argument("--b-adapt", type=int, default=1)
    ap.add_argument("--scenecut", type=int, default=40, help="param.i_scenecut_threshold")
    ap.add_argument("--pre-scenecut", dest="pre_scenecut", type=int, default=0, help="stream mode: 1: the scene cut decided in the lookahead (what the reference forces with --threads > 1); "
                    "0 (the reference's default): x264_encoder_encode's look at every coded P frame, a given-up attempt coded again inside the step")
    args = ap.parse_args()
    args.cif = args.preset == "cif"
    if args.cif:
        args.wavefront, args.stream, args.subme_zero = 1, 0, True
        args.width, args.height = args.width or 352, args.height or 288
        args.me = 0 if args.me < 0 else args.me
        args.refs, args.inter, args.intra, args.dct8, args.mixed_refs, args.bframes = 1, 0, 1, 0, 0, 0      # `--partitions none` clears analyse.inter only: I slices keep I4x4
        args.keyint = args.keyint or 250
        args.batch = args.batch or 2048
        args.steps = args.steps or 27                    # with the 3 warm-up steps: chain 0 codes the 30 frames of BASELINE config 1's clip
    wf = bool(args.wavefront)
    uhd, slow = args.preset == "uhd", args.preset == "slow"
    if slow:
        args.direct = 3 if args.direct == 1 else args.direct       # SLOW as BASELINE.md states it: --direct auto (the running skip scores of every chain's B frames)
        args.refs = 5 if args.refs == 3 else args.refs
        args.b_adapt = 2 if args.b_adapt == 1 else args.b_adapt
        args.subme = args.subme or 8
    args.width = args.width or (3840 if uhd else 1920)
    args.height = args.height or (2160 if uhd else 1080)
    args.me = args.me if args.me >= 0 else (2 if uhd or slow else 1)
    args.steps = args.steps or (24 if wf else 12)
    args.warmup = args.warmup if args.warmup >= 0 else (3 if wf else 2)
    args.stream = (0 if wf else 1) if args.stream < 0 else args.stream
    if args.stream and wf:
        raise SystemExit("bench.py: --stream needs the raster variant")
    args.batch = args.batch or (240 if wf else 512 if uhd else 1024 if slow else 2048)
    args.subme = 0 if args.cif else (args.subme or (5 if wf else 7))
    args.keyint = args.keyint or (24 if wf else 250 if args.stream else 12)
    args.payload_cap = args.payload_cap or ((4 << 20) if uhd else (1 << 20))
    if wf:
        args.trellis, args.psy_rd, args.aq_mode = 0, 0.0, 0
    args.bframes = (0 if wf else 3) if args.bframes < 0 else args.bframes
    args.lanes = args.bframes if args.lanes < 0 else args.lanes
    if args.bframes:
        args.inter |= 0x100                          # X264_ANALYSE_BSUB16x16: the medium preset's b8x8
    n_coded = args.warmup + args.steps
    delay = (max(args.bframes, 3) * 4 if args.b_adapt == 2 else args.bframes) if args.stream and args.bframes else 0     # h->frames.i_delay, encoder.c:703-706
    args.cpu_frames = args.cpu_frames or (n_coded + delay)

    # stdout carries ONE line, the JSON: everything else any library prints there (gloo announces its connections on stdout) goes
    # to stderr -- file descriptor 1 points at stderr until the result is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    # chains of the job: weak scaling gives every rank `batch` chains; strong scaling spreads `batch` chains over the ranks
    if args.strong:
        g_total, g_first, g_step = args.batch, rank, world
        B = len(range(rank, args.batch, world))
        if B < 1:
            raise SystemExit("bench.py --strong: fewer chains than ranks")
    else:
        g_total, g_first, g_step, B = world * args.batch, rank * args.batch, 1, args.batch

    # ---- the CPU leg first: the reference on chain 0's frames (rank 0 of a single-GPU run only), before any HIP call ----
    cpu, ref_pays = None, None
    if rank == 0 and world == 1 and not args.no_cpu and args.cpu_frames > 0:
        cpu, ref_pays = cpu_baseline_stream(args) if args.stream else cpu_baseline(args, g_total)

    dist = None
    if world > 1:
        # gloo on CPU tensors only: barrier + max-reduce of the wall time.  torch.cuda is never
        # initialised in this process (its bundled HIP runtime cannot share the GPU with the
        # system runtime libx264hip.so links against); the data path has no collective.
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    ndev = L.open_library().x264hip_device_count()
    if ndev <= 0:
        raise SystemExit("bench.py: no MI355X visible to libx264hip.so (there is no CPU fallback)")
    hip = L.load(local % ndev)                       # one rank per GPU; wraps only when rehearsing on fewer GPUs
    # the raster variant's product is the payload: no coefficient-level arrays in the states, and a payload buffer sized for the
    # content (the sweep stops with an error, never writes past it, if a chain's slice does not fit)
    ropt = (dict(write=1) if args.cif else {}) if wf else dict(write=1, levels=False, payload_cap=args.payload_cap, **gpu_options(args))
    if args.stream:
        if not args.strong:
            B = fit_batch(args, hip, B, delay)
        return run_stream(args, hip, dist, json_fd, rank, world, B, g_first, g_step, cpu, ref_pays, delay)
    enc = sl.ChainEncoder(hip, args.width, args.height, cqm_init(hip), batch=B, **analysis_options(args), **ropt)      # quantiser tables: x264hip_cqm_init (flat matrices)
    ctx = enc.ctx
    d = ctx.dims
    px = d.mb_w * 16 * d.lines_y

    # sources: synthesised on the device, one picture per step (a small ring when B frames run on lanes of their own and may still
    # read theirs while the next step's is being made).  No frame of a chain repeats, so no frame meets itself among its references.
    srcs = [ctx.new_picture(source_only=not wf) for _ in range(1 + (args.lanes if args.bframes else 0) + (1 if args.lanes and args.bframes else 0))]

    # with B frames the chains are coded in coding order: I P B B B P B B B ... (x264_vs2008_amd/slice.py: coding_order)
    order = sl.coding_order(n_coded, args.keyint, args.bframes) if args.bframes else [(t, None) for t in range(n_coded)]

    # chain 0's payload of every step, copied out behind the step's sweep into pinned memory (no synchronisation in the loop)
    hip.x264hip_host_alloc.restype = C.c_void_p
    check = ref_pays is not None and (not wf or args.cif)
    cap_n = min(CAPTURE, (enc.payload_cap if args.cif else args.payload_cap) - sl.PAYLOAD_LEAD)
    pin = hip.x264hip_host_alloc(C.c_size_t(n_coded * (cap_n + 64))) if check else None

    def one_step(k):
        disp, stype = order[k]
        src = srcs[k % len(srcs)]
        ctx.synth(src, clip_time(disp, g_first, args.keyint, g_total), g_step * args.keyint)
        if stype is None:
            enc.encode_frame(src)
        else:
            enc.encode_frame(src, stype=stype, disp=disp)
        if check and k < len(ref_pays):
            enc.payload_async(0, pin + k * (cap_n + 64), pin + k * (cap_n + 64) + 64, cap_n)
        enc.finish_frame()

    def sync_all():
        assert hip.x264hip_device_synchronize() == 0

    for k in range(args.warmup):
        one_step(k)
    sync_all()
    enc.status()
    if dist is not None:
        dist.barrier()
    enc.events = []
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(args.warmup + k)
    sync_all()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    enc.status()                                     # a sweep that gave up waiting would have produced garbage: fail loudly
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
        nb = torch.tensor([B], dtype=torch.int64)
        dist.all_reduce(nb, op=dist.ReduceOp.SUM)
        chains_total = int(nb[0])
    else:
        chains_total = B

    # ---- parity of THIS run: every frame the GPU coded for chain 0 against the reference's bytes ----
    checked, stream_md5 = 0, None
    if check:
        for k in range(min(n_coded, len(ref_pays))):
            base = pin + k * (cap_n + 64)
            n = C.c_int32.from_address(base).value
            got = C.string_at(base + 64, min(n, cap_n))
            want = ref_pays[k]
            if n != len(want) or got != want[:cap_n]:
                raise SystemExit("bench.py: PARITY FAILURE -- chain 0, coded frame %d (display %d): the GPU's slice payload (%d bytes) differs from the "
                                 "reference's (%d bytes)" % (k, order[k][0], n, len(want)))
            checked += 1
        # BASELINE config 1 as a file: chain 0's first 30 frames are the cif30 clip, so their payloads inside the library's version SEI, parameter
        # sets and slice headers (x264_vs2008_amd/mux.py) must be, byte for byte, what the reference's command line wrote (md5 from SURVEY.md 8(c))
        if args.cif and checked >= 30 and g_first == 0 and (args.width, args.height, args.keyint, args.qp, args.me) == (352, 288, 250, 26, 0):
            import hashlib
            from x264_vs2008_amd import mux
            mp = mux.encoder_params(hip, width=352, height=288, rc_method=mux.RC_CQP, qp_constant=26, cabac=0, me_method=0, subpel_refine=0, inter=0, deblocking_filter=0,
                                    aq_mode=0, scenecut_threshold=-1, frame_reference=1, bframe=0, bframe_adaptive=0)
            mx, hsh = mux.AnnexB(hip, mp), hashlib.md5()
            for k in range(30):
                base = pin + k * (cap_n + 64)
                n = C.c_int32.from_address(base).value
                hsh.update(mx.frame(frame=k, ftype=mux.TYPE_IDR if k == 0 else mux.TYPE_P, qp=sl.iframe_qp(26) if k == 0 else 26, payload=C.string_at(base + 64, n)))
            if hsh.hexdigest() != "02b208eecef842e084cbb9c83bc1a757":
                raise SystemExit("bench.py: PARITY FAILURE -- chain 0's Annex B stream (30 frames of cif30) has md5 %s, the reference CLI's file 02b208eecef842e084cbb9c83bc1a757" % hsh.hexdigest())
            stream_md5 = ("chain 0's whole Annex B stream of its first 30 frames (the cif30 clip: version SEI, SPS, PPS, slice headers, CAVLC payloads) has the md5 of the "
                          "reference command line's output for BASELINE config 1, 02b208eecef842e084cbb9c83bc1a757")
        hip.x264hip_host_free(C.c_void_p(pin))

    # the dominant kernel (k_slice_sweep), timed live with HIP events on its launch stream: every launch of the timed region
    # (P launches with 1..R references, B launches, and the I launch at the keyint), so that the mean is the one rocprofv3's kernel
    # trace of the same command shows for the same launches (profiles/)
    ms_all = [hip.x264hip_event_elapsed_ms(C.c_void_p(a), C.c_void_p(b)) for a, b, st, nr in enc.events]
    # algorithmic bytes of each launch (SURVEY 8(d) terms that belong to this kernel): per frame the source (1.5 B/px), each
    # reference's four luma planes + chroma (4.5 B/px) and the reconstruction (1.5 B/px)
    by_all = [B * px * (1.5 + 4.5 * (nr if st != sl.SLICE_I else 0) + 1.5) for a, b, st, nr in enc.events]      # nr: list 0 + list 1
    for a, b, _, _ in enc.events:
        hip.x264hip_event_destroy(C.c_void_p(a)); hip.x264hip_event_destroy(C.c_void_p(b))
    launches = " ".join("%s:%d:%.0f" % ("PBI"[st], nr, ms) for ms, (_, _, st, nr) in zip(ms_all, enc.events))
    if os.environ.get("BENCH_LAUNCHES"):             # developer aid: every timed sweep launch, "slice type:references:ms"
        print(launches, file=sys.stderr)
    sweep_ms = float(np.mean(ms_all))
    sweep_bytes = int(np.mean(by_all))
    achieved = sweep_bytes / (sweep_ms * 1e-3) / 1e9

    # HBM-side traffic of one sweep launch: not measurable from inside this process (PMC counters need rocprofv3), so the
    # figure is the committed rocprofv3 measurement of this very configuration, and null for any other configuration
    traffic, traffic_note = None, "no rocprofv3 PMC measurement committed for this configuration"
    tname = "r01_sweep_traffic.json" if wf else "r03_raster_traffic.json"
    tpath = os.path.join(ROOT, "profiles", tname)
    defaults = (args.width, args.height, args.qp, args.me, args.inter & 0x33, args.intra, args.dct8, args.mixed_refs) == (1920, 1080, 26, 1, 0x13, 0x3, 1, 1) and \
               (args.subme, args.keyint) == ((5, 24) if wf else (7, 12)) and (wf or (args.trellis, args.psy_rd, args.aq_mode) == (1, 1.0, 1))
    if defaults and os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        import collections
        kinds = collections.Counter(("I" if st == sl.SLICE_I else "B" if st == sl.SLICE_B else "P", nr) for _, _, st, nr in enc.events)
        (kname, knr), _ = kinds.most_common(1)[0]
        hit = [l for l in tj["launches"] if l["slice"] == kname and l["refs"] == knr]
        if hit and tj.get("bframes", 0) == args.bframes:
            # (measured with tj["batch"] chains per launch; per macroblock it is the same work, scaled to this run's batch)
            traffic = int((hit[0]["fetch_bytes"] + hit[0]["write_bytes"]) * (B / tj["batch"]))
            traffic_note = ("FETCH_SIZE + WRITE_SIZE of one %s launch with %d reference pictures (the most frequent launch of the timed region), rocprofv3 --pmc, "
                            "separate passes, raw request-granular counters, measured at %d chains per launch and scaled to %d (profiles/%s)" % (kname, knr, tj["batch"], B, tname))

    if rank == 0:
        fps = chains_total * args.steps / dt
        frame_bytes = px * (1.5 + 4.5 * args.refs + 1.5 + 3.0 + 4.0)       # + deblock read/write + hpel planes
        n_i = sum(1 for e in enc.events if e[2] == sl.SLICE_I)
        size = "%dp" % args.height
        if args.cif:
            metric = "encoded frames/sec, 352x288, the ultrafast flag set of BASELINE config 0 (--qp 26 --no-cabac --me dia --subme 0 --partitions none --no-deblock --ref 1), CAVLC payload on the GPU (bit-exact)"
            what = ("%dx%d I / P chains through the reference's per-macroblock loop on the GPU (wavefront schedule) and the CAVLC writer (x264hip_cavlc_write_frame) behind every sweep: "
                    "dia ME range 16, subme 0, 1 ref, no partitions, no deblock, CQP %d, keyint %d" % (args.width, args.height, args.qp, args.keyint))
            missing = []
            par = "B chains per GPU in every launch (one wavefront per macroblock row per chain, then one wavefront per chain for the CAVLC pass); chains shard across GPUs with no data-path collective"
        elif wf:
            metric = "I/P macroblock-loop frames/sec, %s, medium minus {B-frames, RD (subme 7 -> 5), trellis, AQ, entropy coding} (round-1 configuration, bit-exact)" % size
            what = ("%dx%d I/P chains through the reference's per-macroblock loop on the GPU, wavefront schedule (one wavefront per macroblock row): "
                    "%s ME range 16, subme %d, %d refs, chroma ME, fast P-skip, dct-decimate, CQP %d, keyint %d; analyse.inter 0x%x intra 0x%x 8x8dct %d "
                    "mixed-refs %d; no RD, no trellis, no AQ; entropy coding not done" % (args.width, args.height, ME_NAMES[args.me], args.subme, args.refs,
                                                                                       args.qp, args.keyint, args.inter, args.intra, args.dct8, args.mixed_refs))
            missing = ["B slices (--bframes 3 --b-adapt 1 --weightb --direct spatial)", "RD mode decision (subme 7)", "trellis 1", "psy-rd", "aq-mode 1",
                       "CRF rate control", "lookahead / scenecut", "entropy coding"]
            par = "B closed-GOP chains per GPU in every launch (one wavefront per macroblock row per chain); chains shard across GPUs with no data-path collective"
        else:
            metric = ("encoded frames/sec, %s, %s slices with preset=medium's analysis (%s, subme %d RD, trellis %d, psy-rd, aq-mode %d, CABAC payload on the GPU) "
                      "at constant QP%s; CRF and lookahead not built yet; 1/2/4/8 MI355X (bit-exact)"
                      % (size, "I/P/B" if args.bframes else "I/P", ME_NAMES[args.me], args.subme, args.trellis, args.aq_mode,
                         ", %d B frames in a fixed pattern, weightb, spatial direct" % args.bframes if args.bframes else "; no B slices"))
            what = ("%dx%d closed-GOP chains of one synthetic clip through the reference's per-macroblock loop on the GPU, raster order (one wavefront per chain): cache_load, "
                    "x264_macroblock_analyse with RD mode decision, x264_macroblock_encode, x264_macroblock_write_cabac (the slice payload is produced "
                    "by the same launch), cache_save, then deblock, borders, half-pel planes; --ref %d --me %s --subme %d --trellis %d --psy-rd %.1f "
                    "--aq-mode %d --8x8dct %d --mixed-refs %d --partitions 0x%x/0x%x, chroma ME, fast P-skip, dct-decimate, CABAC, CQP %d, keyint %d; sources "
                    "synthesised on the device inside the timed region, payload bytes stay on the device (slice / NAL headers and the download are the host's)"
                    % (args.width, args.height, args.refs, ME_NAMES[args.me], args.subme, args.trellis, args.psy_rd, args.aq_mode, args.dct8, args.mixed_refs,
                       args.inter, args.intra, args.qp, args.keyint))
            missing = (["B slices (--bframes 3 --b-adapt 1 --weightb --direct spatial): about 3/4 of a medium encode's frames"] if not args.bframes else
                       ["adaptive B placement (--b-adapt 1): the B frames are placed in a fixed pattern of %d" % args.bframes]) + [
                       "CRF rate control (--crf 23): constant QP %d + adaptive quantisation here" % args.qp, "lookahead (b-adapt, scenecut, lowres motion candidates)",
                       "keyint %d so that thousands of closed GOPs exist (the preset's default is 250)" % args.keyint]
            par = ("B closed-GOP chains per GPU in every launch, one wavefront per chain walking its frame in raster order (the RD levels, trellis and AQ "
                   "make a slice one serial chain of macroblocks); chains shard across GPUs with no data-path collective")
        line = {
            "metric": metric,
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": what, "matches_baseline": False, "missing": missing,
                       "baseline_metric": "encoded frames/sec, 1080p preset=medium, 1/2/4/8 MI355X (bit-exact)",
                       "frames_per_step": chains_total, "frames_in_flight": chains_total, "keyint": args.keyint,
                       "per_chain_fps": round(fps / chains_total, 4),
                       "latency_note": "throughput exists only with thousands of closed GOPs in flight: one chain advances one frame per step",
                       "i_frames_in_timed_steps": n_i, "parallelism": par,
                       "parity_checked_frames": checked,
                       "parity": ("payload bytes of rank 0's chain 0, all %d coded frames of this run (%d of them timed), equal the reference's own x264_macroblock_analyse / "
                                  "_encode / _write_cabac output for the same frames" % (checked, max(0, checked - args.warmup))) if checked else
                                 "not checked in this run (no CPU leg: --no-cpu, --wavefront 1 or more than one rank)",
                       "stream_md5": stream_md5,
                       "timed_launches": launches},
            "roofline": {"bound": "hbm", "kernel": "k_slice_sweep" + ("" if wf else "<raster>"), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_note": traffic_note,
                         "avg_launch_ms": round(sweep_ms, 4), "algorithmic_bytes_per_launch": sweep_bytes,
                         "note": "mean over the timed launches (P with 1..R references, B, and the I launch at the keyint); the sweep is bound by the serial "
                                 "macroblock chain of a slice (%s), not by bandwidth; whole-frame algorithmic bytes = %d -> %.1f GB/s at this fps"
                                 % ("mb_w + 2*mb_h = %d dependent steps per frame" % (d.mb_w + 2 * d.mb_h - 2) if wf else
                                    "%d macroblocks one after the other per frame, %d frames in flight" % (d.mb_w * d.mb_h, B),
                                    frame_bytes, frame_bytes * (fps / world) / 1e9)},
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    enc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
