import re
p='/root/repo/bench.py'
s=open(p).read()
# --- doc string
a=s.index('"""bench.py')
b=s.index('"""', a+3)+3
s=s[:a]+'''"""bench.py -- frames/sec of the per-macroblock hot loop on 1..N MI355X.

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

Chains shard across ranks with no data-path collective (closed GOPs, SURVEY 8(e)); scaling is weak.
One JSON line on stdout (rank 0).  Launch for N > 1:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...
"""'''+s[b:]

s=s.replace('''def analysis_options(args):
    return dict(qp=args.qp, me_method=args.me, me_range=16, subme=args.subme, n_refs=args.refs, fast_pskip=1, dct_decimate=1,
                chroma_me=1, cabac=1, deblock=1, keyint=args.keyint, inter=args.inter, intra=args.intra, transform8x8=args.dct8,
                mixed_refs=args.mixed_refs)
''','''def analysis_options(args):
    return dict(qp=args.qp, me_method=args.me, me_range=16, subme=args.subme, n_refs=args.refs, fast_pskip=1, dct_decimate=1,
                chroma_me=1, cabac=1, deblock=1, keyint=args.keyint, inter=args.inter, intra=args.intra, transform8x8=args.dct8,
                mixed_refs=args.mixed_refs)


def rd_options(args):
    """What the raster-order variant adds (x264hip_slice_rd)."""
    return dict(trellis=args.trellis, psy_rd=args.psy_rd, aq_mode=args.aq_mode, aq_strength=1.0)


def _cpu_chain(job):
    """One chain through the reference's loop on one core (own process: the reference keeps process-global tables, SURVEY 0.7)."""
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
''')

a=s.index("def cpu_baseline(args):")
b=s.index("def main():")
s=s[:a]+'''def cpu_baseline(args):
    """The same loop on the host cores: the REFERENCE's own x264_macroblock_cache_load / _analyse / _encode / _write_cabac /
    _cache_save + x264_frame_deblock_row + x264_frame_filter, compiled from the reference's sources where they lie
    (oracle/_ref/libx264ref.so via oracle/ref_slice.c); our restatement (liboracle.so) when that library is not there.  Measured
    twice on a bounded chain of whole frames: one process on one core, and one process per host core (each its own chain)."""
    import multiprocessing as mp
    n = args.cpu_frames
    kw, ekw, raster = analysis_options(args), rd_options(args), not args.wavefront
    spent1, kind = _cpu_chain((args.width, args.height, n, kw, ekw, raster, 0))
    cores = max(1, min(os.cpu_count() or 1, args.cpu_procs or (os.cpu_count() or 1)))
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_cpu_chain, [(args.width, args.height, n, kw, ekw, raster, 11 * i) for i in range(cores)])
    spent_all = time.perf_counter() - t0
    return {"value": round(cores * n / spent_all, 4), "unit": "frames/s", "cores": cores, "kind": kind,
            "one_core": round(n / spent1, 4),
            "sample": "the same per-macroblock loop with the same options on chains of %d %dx%d frames: one chain on one core (%.1f s), "
                      "then %d processes, one chain each, on the %d host cores (%.1f s); C compiled -O3, no asm%s"
                      % (n, args.width, args.height, spent1, cores, cores, spent_all, ", entropy coding included" if raster else ", no entropy coding on either side")}


'''+s[b:]

s=s.replace('''    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=240, help="independent GOP chains advanced per step on each GPU "
                    "(240 x 68 macroblock rows = 16320 row waves for 2048 wave slots at 2 waves/SIMD: 8.5 slots per chain, so the "
                    "68 rows of a frame go through in 8 full generations; later rows take the slots of finished ones)")''','''    ap.add_argument("--steps", type=int, default=0, help="0: 12 (raster variant) / 24 (--wavefront 1)")
    ap.add_argument("--warmup", type=int, default=-1, help="-1: 2 (raster variant) / 3 (--wavefront 1)")
    ap.add_argument("--batch", type=int, default=0, help="independent GOP chains advanced per step on each GPU; 0: 1792 for the raster "
                    "variant (one wavefront per chain: 7 per CU, what its LDS footprint allows), 240 with --wavefront 1")
    ap.add_argument("--wavefront", type=int, default=0, help="1: round 1's configuration (wavefront schedule, subme 5, no RD / trellis / AQ / entropy coding)")
    ap.add_argument("--trellis", type=int, default=1)
    ap.add_argument("--psy-rd", type=float, default=1.0)
    ap.add_argument("--aq-mode", type=int, default=1)
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the all-core CPU leg (0: one per host core)")''')
s=s.replace('''    ap.add_argument("--subme", type=int, default=5)''','''    ap.add_argument("--subme", type=int, default=0, help="0: 7 (raster variant) / 5 (--wavefront 1)")''')
s=s.replace('''    ap.add_argument("--keyint", type=int, default=24)''','''    ap.add_argument("--keyint", type=int, default=0, help="0: 12 (raster variant) / 24 (--wavefront 1)")''')
s=s.replace('''    ap.add_argument("--cpu-frames", type=int, default=40)''','''    ap.add_argument("--cpu-frames", type=int, default=0, help="0: 12 (raster variant) / 40 (--wavefront 1)")''')
s=s.replace('''    args = ap.parse_args()
''','''    args = ap.parse_args()
    wf = bool(args.wavefront)
    args.steps = args.steps or (24 if wf else 12)
    args.warmup = args.warmup if args.warmup >= 0 else (3 if wf else 2)
    args.batch = args.batch or (240 if wf else 1792)
    args.subme = args.subme or (5 if wf else 7)
    args.keyint = args.keyint or (24 if wf else 12)
    args.cpu_frames = args.cpu_frames or (40 if wf else 12)
    if wf:
        args.trellis, args.psy_rd, args.aq_mode = 0, 0.0, 0
''')
s=s.replace('''    enc = sl.ChainEncoder(hip, args.width, args.height, load_cqm(), batch=B, **analysis_options(args))''','''    ropt = {} if wf else dict(write=1, **rd_options(args))
    enc = sl.ChainEncoder(hip, args.width, args.height, load_cqm(), batch=B, **analysis_options(args), **ropt)''')
s=s.replace('''    n_src, pool_n = 8, 16''','''    n_src, pool_n = (8, 16) if wf else (3, 8)''')
open(p,'w').write(s)
