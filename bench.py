#!/usr/bin/env python3
"""bench.py -- frames/sec of the per-macroblock hot loop on 1..N MI355X.

A "step" advances B independent closed-GOP chains by one 1920x1080 frame each; the source frames are already resident in HBM.
Per step and chain the GPU does what x264_slice_write + x264_fdec_filter_row do for one frame (R/encoder/encoder.c:1141-1291,
983-1056):

  x264hip_adaptive_quant_frame   x264_adaptive_quant_frame: per-macroblock QP offsets from the source's AC energy
  x264hip_slice_sweep_frame      raster-order variant (x264hip_slice_rd): cache_load -> x264_macroblock_analyse (RD mode decision,
                                 subme 7) -> x264_macroblock_encode (trellis 1) -> x264_macroblock_write_cabac -> cache_save for all
                                 8160 macroblocks, one wavefront per chain; the slice's CABAC payload comes out of the same launch
  x264hip_deblock_frame, x264hip_expand_border, x264hip_hpel_filter_frame   the frame becomes a reference

Default options = BASELINE.md's MED flag set as far as it is built: --ref 3 --me hex --subme 7 --8x8dct --partitions p8x8,i8x8,i4x4
--trellis 1 --mixed-refs, psy-rd 1.0, aq-mode 1, CABAC, deblock -- at CONSTANT QP (CRF needs the lookahead) and with I/P slices
only (B slices are not built yet): config.matches_baseline is false and config.missing lists what is left.  Every decision, level,
pixel and payload byte of this loop is bit-exact against the reference's own functions (tests/test_gpu_slice_rd.py).
--wavefront 1 selects round 1's configuration instead (subme 5, no RD / trellis / AQ / entropy coding; one wavefront per macroblock row).

Chains shard across ranks with no data-path collective (closed GOPs); scaling is weak.
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

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ME_NAMES = {0: "dia", 1: "hex", 2: "umh"}


def load_cqm():
    with np.load(os.path.join(ROOT, "tests", "golden", "cqm_flat.npz")) as z:
        return {k: z[k] for k in z.files}


def analysis_options(args):
    return dict(qp=args.qp, me_method=args.me, me_range=16, subme=args.subme, n_refs=args.refs, fast_pskip=1, dct_decimate=1,
                chroma_me=1, cabac=1, deblock=1, keyint=args.keyint, inter=args.inter, intra=args.intra, transform8x8=args.dct8,
                mixed_refs=args.mixed_refs)


def rd_options(args):
    """What the raster-order variant adds (x264hip_slice_rd)."""
    o = dict(trellis=args.trellis, psy_rd=args.psy_rd, aq_mode=args.aq_mode, aq_strength=1.0)
    if args.bframes:
        o.update(bframes=args.bframes, weightb=args.weightb, direct_pred=1)
    return o


def gpu_options(args):
    """How the GPU side schedules the same work (not an encoder option: the CPU leg does not see it)."""
    o = rd_options(args)
    if args.bframes and args.lanes:
        o.update(lanes=args.lanes)
    return o


def _cpu_chain(job):
    """One chain through the reference's loop on one core (own process: the reference keeps process-global tables)."""
    import time as _t
    from oracle import refslice as rs
    width, height, n, kw, ekw, raster, seed = job
    y, u, v = rs.clip(width, height, n, t0=seed)
    p = rs.make_params(width, height, n, **kw)
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libx264ref.so")
    t0 = _t.perf_counter()
    if os.path.exists(ref_so):
        if raster:
            rs.run_reference2(p, rs.make_ext(**ekw), y, u, v)
        else:
            rs.run_reference(p, y, u, v)
        kind = "reference"
    else:
        lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        if raster:
            rs.run2(lib, "x264o_encode_chain2", p, rs.make_ext(**ekw), y, u, v)
        else:
            rs.run(lib, "x264o_encode_chain", p, y, u, v)
        kind = "port"
    return _t.perf_counter() - t0, kind


def cpu_baseline(args):
    """The same loop on the host cores: the REFERENCE's own x264_macroblock_cache_load / _analyse / _encode / _write_cabac /
    _cache_save + x264_frame_deblock_row + x264_frame_filter, compiled from the reference's sources where they lie
    (oracle/_ref/libx264ref.so via oracle/ref_slice.c); our restatement (liboracle.so) when that library is not there.  Measured
    twice on a bounded chain of whole frames: one process on one core, and one process per host core (each its own chain)."""
    import multiprocessing as mp
    n = args.cpu_frames
    kw, ekw, raster = analysis_options(args), rd_options(args), not args.wavefront
    spent1, kind = _cpu_chain((args.width, args.height, n, kw, ekw, raster, 0))
    avail = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)   # a one-GPU box's CPU share is 16 cores
    cores = max(1, min(avail, args.cpu_procs or avail))
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_chain, [(args.width, args.height, n, kw, ekw, raster, 11 * i) for i in range(cores)], chunksize=1)
    spent_all = max(r[0] for r in res)                # the chains run side by side: the slowest one's encode time (clip synthesis is not counted)
    return {"value": round(cores * n / spent_all, 4), "unit": "frames/s", "cores": cores, "kind": kind,
            "one_core": round(n / spent1, 4),
            "sample": "the same per-macroblock loop with the same options on chains of %d %dx%d frames: one chain on one core (%.1f s), "
                      "then %d processes, one chain each, on the %d host cores (%.1f s); C compiled -O3, no asm%s"
                      % (n, args.width, args.height, spent1, cores, cores, spent_all, ", entropy coding included" if raster else ", no entropy coding on either side")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="0: 12 (raster variant) / 24 (--wavefront 1)")
    ap.add_argument("--warmup", type=int, default=-1, help="-1: 2 (raster variant) / 3 (--wavefront 1)")
    ap.add_argument("--batch", type=int, default=0, help="independent GOP chains advanced per step on each GPU; 0: 2048 (8 wavefronts on each of the 256 CUs) for the raster "
                    "variant (one wavefront per chain, all resident: what its 20 KB of LDS and 234 VGPRs allow), 240 with --wavefront 1")
    ap.add_argument("--wavefront", type=int, default=0, help="1: round 1's configuration (wavefront schedule, subme 5, no RD / trellis / AQ / entropy coding)")
    ap.add_argument("--trellis", type=int, default=1)
    ap.add_argument("--bframes", type=int, default=-1, help="disposable B frames between anchors, fixed pattern (-1: 3 for the raster variant = the medium "
                    "preset's --bframes 3 without b-adapt; 0 with --wavefront 1)")
    ap.add_argument("--weightb", type=int, default=1)
    ap.add_argument("--payload-cap", type=int, default=1 << 20, help="bytes of payload buffer per chain and frame in flight (the library's default, 800 B per macroblock, is x264's worst case)")
    ap.add_argument("--lanes", type=int, default=0, help="extra streams for the B frames of a mini-GOP, which then run beside the next anchor (0: one stream, frames in lock step -- "
                    "with every wave slot taken by a launch's chains the lanes gain nothing, DESIGN.md 3.1c; -1: one per B frame of the pattern -- with half the chains, 1024, "
                    "they reach 90%% of the default's rate in half the memory)")
    ap.add_argument("--psy-rd", type=float, default=1.0)
    ap.add_argument("--aq-mode", type=int, default=1)
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the all-core CPU leg (0: one per host core)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--qp", type=int, default=26)
    ap.add_argument("--refs", type=int, default=3)
    ap.add_argument("--subme", type=int, default=0, help="0: 7 (raster variant) / 5 (--wavefront 1)")
    ap.add_argument("--me", type=int, default=1, help="param.analyse.i_me_method: 0 dia, 1 hex (the medium preset), 2 umh")
    ap.add_argument("--keyint", type=int, default=0, help="0: 12 (raster variant) / 24 (--wavefront 1)")
    ap.add_argument("--inter", type=lambda v: int(v, 0), default=0x13, help="param.analyse.inter: X264_ANALYSE_I4x4 0x1 | I8x8 0x2 | PSUB16x16 0x10 | "
                    "PSUB8x8 0x20 (the medium preset's p8x8 = 0x10; 0x33 adds p4x4 / p8x4 / p4x8)")
    ap.add_argument("--mixed-refs", type=int, default=1, help="param.analyse.b_mixed_references")
    ap.add_argument("--intra", type=lambda v: int(v, 0), default=0x3, help="param.analyse.intra")
    ap.add_argument("--dct8", type=int, default=1, help="param.analyse.b_transform_8x8")
    ap.add_argument("--cpu-frames", type=int, default=0, help="0: 12 (raster variant) / 40 (--wavefront 1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--host-sources", action="store_true", help="fill the source ring with one host upload per chain and picture instead of device-side copies "
                    "(slow set-up; what the rocprofv3 --pmc passes of profiles/ were taken with, at 512 chains: with multi-gigabyte buffers --pmc of ROCm 7.2 crashes)")
    args = ap.parse_args()
    wf = bool(args.wavefront)
    args.steps = args.steps or (24 if wf else 12)
    args.warmup = args.warmup if args.warmup >= 0 else (3 if wf else 2)
    args.batch = args.batch or (240 if wf else 2048)
    args.subme = args.subme or (5 if wf else 7)
    args.keyint = args.keyint or (24 if wf else 12)
    args.cpu_frames = args.cpu_frames or (40 if wf else 12)
    if wf:
        args.trellis, args.psy_rd, args.aq_mode = 0, 0.0, 0
    args.bframes = (0 if wf else 3) if args.bframes < 0 else args.bframes
    args.lanes = args.bframes if args.lanes < 0 else args.lanes
    if args.bframes:
        args.inter |= 0x100                          # X264_ANALYSE_BSUB16x16: the medium preset's b8x8

    # stdout carries ONE line, the JSON: everything else any library prints there (gloo announces its connections on stdout) goes
    # to stderr -- file descriptor 1 points at stderr until the result is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
    B = args.batch
    # the raster variant's product is the payload: no coefficient-level arrays in the states, and a payload buffer sized for the
    # content (the sweep stops with an error, never writes past it, if a chain's slice does not fit)
    ropt = {} if wf else dict(write=1, levels=False, payload_cap=args.payload_cap, **gpu_options(args))
    enc = sl.ChainEncoder(hip, args.width, args.height, load_cqm(), batch=B, **analysis_options(args), **ropt)
    ctx = enc.ctx
    d = ctx.dims
    px = d.mb_w * 16 * d.lines_y

    # resident working set: a ring of source pictures, each holding one frame of every chain.  Chains and
    # steps see different frames of the synthetic clip (rank-dependent offset).
    # The raster variant: eight sources in rotation, so that no frame meets its own picture among its references
    # (with fewer sources than the DPB reaches back, a frame's reference holds the identical content and the search is trivial).
    n_src, pool_n = (8, 16) if wf else (min(max(args.keyint, 4), 8), 16)     # (8 x 7 GB of sources at 2048 chains: what fits beside the DPBs and states)
    pool_n = max(pool_n, n_src + 1)
    pool = [synth.frame(args.width, args.height, rank * 97 + i) for i in range(pool_n)]
    srcs = []
    if wf or B < pool_n or args.host_sources:
        for i in range(n_src):
            pic = ctx.new_picture(source_only=not wf)
            for b in range(B):
                ctx.upload(pic, *pool[(i + 3 * b) % pool_n], b=b)
            srcs.append(pic)
    else:       # the pool goes up once (into the first elements of a staging picture); the sources are filled on the device
        stage = ctx.new_picture(source_only=True)
        for k in range(pool_n):
            ctx.upload(stage, *pool[k], b=k)
        for i in range(n_src):
            pic = ctx.new_picture(source_only=True)
            for b in range(B):
                ctx.copy_element(pic, b, stage, (i + 3 * b) % pool_n)
            srcs.append(pic)
        ctx.sync()

    # with B frames the chains are coded in coding order: I P B B B P B B B ... (x264_vs2008_amd/slice.py: coding_order)
    order = sl.coding_order(args.warmup + args.steps + args.keyint, args.keyint, args.bframes) if args.bframes else None

    def one_step(k):
        if order:
            disp, stype = order[k]
            enc.encode_frame(srcs[disp % n_src], stype=stype, disp=disp)
        else:
            enc.encode_frame(srcs[k % n_src])
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

    # the dominant kernel (k_slice_sweep), timed live with HIP events on its launch stream: every launch of the timed region
    # (P launches with 1..R references and the I launch at the keyint), so that the mean is the one rocprofv3's kernel trace of
    # the same command shows for the same launches (profiles/r01_bench_sweep_launches.json)
    ms_all = [hip.x264hip_event_elapsed_ms(C.c_void_p(a), C.c_void_p(b)) for a, b, st, nr in enc.events]
    # algorithmic bytes of each launch (SURVEY 8(d) terms that belong to this kernel): per frame the source (1.5 B/px), each
    # reference's four luma planes + chroma (4.5 B/px) and the reconstruction (1.5 B/px)
    by_all = [B * px * (1.5 + 4.5 * (nr if st != sl.SLICE_I else 0) + 1.5) for a, b, st, nr in enc.events]      # nr: list 0 + list 1
    for a, b, _, _ in enc.events:
        hip.x264hip_event_destroy(C.c_void_p(a)); hip.x264hip_event_destroy(C.c_void_p(b))
    if os.environ.get("BENCH_LAUNCHES"):             # developer aid: every timed sweep launch, "slice type:references:ms"
        print(" ".join("%s:%d:%.0f" % ("PBI"[st], nr, ms) for ms, (_, _, st, nr) in zip(ms_all, enc.events)), file=sys.stderr)
    sweep_ms = float(np.mean(ms_all))
    sweep_bytes = int(np.mean(by_all))
    achieved = sweep_bytes / (sweep_ms * 1e-3) / 1e9

    # HBM-side traffic of one P sweep launch: not measurable from inside this process (PMC counters need rocprofv3), so the
    # figure is the committed rocprofv3 measurement of this very configuration, and null for any other configuration
    traffic, traffic_note = None, "no rocprofv3 PMC measurement committed for this configuration"
    tname = "r01_sweep_traffic.json" if wf else "r02_raster_traffic.json"
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
        fps = world * B * args.steps / dt
        frame_bytes = px * (1.5 + 4.5 * args.refs + 1.5 + 3.0 + 4.0)       # + deblock read/write + hpel planes
        n_i = sum(1 for e in enc.events if e[2] == sl.SLICE_I)
        if wf:
            metric = "I/P macroblock-loop frames/sec, 1080p, medium minus {B-frames, RD (subme 7 -> 5), trellis, AQ, entropy coding} (round-1 configuration, bit-exact)"
            what = ("%dx%d I/P chains through the reference's per-macroblock loop on the GPU, wavefront schedule (one wavefront per macroblock row): "
                    "%s ME range 16, subme %d, %d refs, chroma ME, fast P-skip, dct-decimate, CQP %d, keyint %d; analyse.inter 0x%x intra 0x%x 8x8dct %d "
                    "mixed-refs %d; no RD, no trellis, no AQ; entropy coding not done" % (args.width, args.height, ME_NAMES[args.me], args.subme, args.refs,
                                                                                       args.qp, args.keyint, args.inter, args.intra, args.dct8, args.mixed_refs))
            missing = ["B slices (--bframes 3 --b-adapt 1 --weightb --direct spatial)", "RD mode decision (subme 7)", "trellis 1", "psy-rd", "aq-mode 1",
                       "CRF rate control", "lookahead / scenecut", "entropy coding"]
            par = "B closed-GOP chains per GPU in every launch (one wavefront per macroblock row per chain); chains shard across GPUs with no data-path collective"
        else:
            metric = ("encoded frames/sec, 1080p, %s slices with preset=medium's analysis (subme 7 RD, trellis 1, psy-rd, aq-mode 1, CABAC payload on the GPU) "
                      "at constant QP%s; CRF and lookahead not built yet; 1/2/4/8 MI355X (bit-exact)"
                      % ("I/P/B" if args.bframes else "I/P", ", %d B frames in a fixed pattern, weightb, spatial direct" % args.bframes if args.bframes else "; no B slices"))
            what = ("%dx%d I/P chains through the reference's per-macroblock loop on the GPU, raster order (one wavefront per chain): cache_load, "
                    "x264_macroblock_analyse with RD mode decision, x264_macroblock_encode, x264_macroblock_write_cabac (the slice payload is produced "
                    "by the same launch), cache_save, then deblock, borders, half-pel planes; --ref %d --me %s --subme %d --trellis %d --psy-rd %.1f "
                    "--aq-mode %d --8x8dct %d --mixed-refs %d --partitions 0x%x/0x%x, chroma ME, fast P-skip, dct-decimate, CABAC, CQP %d, keyint %d"
                    % (args.width, args.height, args.refs, ME_NAMES[args.me], args.subme, args.trellis, args.psy_rd, args.aq_mode, args.dct8, args.mixed_refs,
                       args.inter, args.intra, args.qp, args.keyint))
            missing = (["B slices (--bframes 3 --b-adapt 1 --weightb --direct spatial): about 3/4 of a medium encode's frames"] if not args.bframes else
                       ["adaptive B placement (--b-adapt 1): the B frames are placed in a fixed pattern of %d" % args.bframes]) + [
                       "CRF rate control (--crf 23): constant QP %d + adaptive quantisation here" % args.qp, "lookahead (b-adapt, scenecut, lowres motion candidates)",
                       "slice / NAL headers around the payload"]
            par = ("B closed-GOP chains per GPU in every launch, one wavefront per chain walking its frame in raster order (the RD levels, trellis and AQ "
                   "make a slice one serial chain of macroblocks); chains shard across GPUs with no data-path collective")
        line = {
            "metric": metric,
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": what, "matches_baseline": False, "missing": missing,
                       "baseline_metric": "encoded frames/sec, 1080p preset=medium, 1/2/4/8 MI355X (bit-exact)",
                       "frames_per_step": B, "i_frames_in_timed_steps": n_i, "parallelism": par},
            "roofline": {"bound": "hbm", "kernel": "k_slice_sweep" + ("" if wf else "<raster>"), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_note": traffic_note,
                         "avg_launch_ms": round(sweep_ms, 4), "algorithmic_bytes_per_launch": sweep_bytes,
                         "note": "mean over the timed launches (P with 1..R references and the I launch at the keyint); the sweep is bound by the serial "
                                 "macroblock chain of a slice (%s), not by bandwidth; whole-frame algorithmic bytes = %d -> %.1f GB/s at this fps"
                                 % ("mb_w + 2*mb_h = %d dependent steps per frame" % (d.mb_w + 2 * d.mb_h - 2) if wf else
                                    "%d macroblocks one after the other per frame, %d frames in flight" % (d.mb_w * d.mb_h, B),
                                    frame_bytes, frame_bytes * (fps / world) / 1e9)},
        }
        if world == 1 and not args.no_cpu and args.cpu_frames > 0:
            line["cpu_baseline"] = cpu_baseline(args)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    enc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
