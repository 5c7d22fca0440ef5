"""Frame sharding across GPUs: closed GOPs are the unit.

Frames between two IDR frames reference nothing outside their GOP, so GOPs go
to ranks round-robin with no pixel exchange; the only cross-GOP state in the
reference is scalar rate-control history, which stays on the host.  The first
half of this module is the bookkeeping; encode_clip below is the data path built on it (one rank's GOPs = the batch of the
raster sweep), gather_digests the only exchange between ranks (results, after coding)."""


def gop_bounds(n_frames, keyint):
    """[(first, last+1)] of every closed GOP when an IDR is forced every `keyint` frames."""
    if keyint <= 0:
        raise ValueError("keyint must be positive")
    return [(s, min(s + keyint, n_frames)) for s in range(0, n_frames, keyint)]


def gops_for_rank(n_frames, keyint, rank, world):
    """GOPs owned by `rank`: round-robin over GOP index."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return [g for i, g in enumerate(gop_bounds(n_frames, keyint)) if i % world == rank]


def frames_for_rank(n_frames, keyint, rank, world):
    return [f for a, b in gops_for_rank(n_frames, keyint, rank, world) for f in range(a, b)]


def frame_num_and_poc(frame, keyint):
    """frame_num / POC restart at every IDR, so they follow from the GOP-local index alone
    (R/encoder/encoder.c:1105-1110, 1514-1515): what lets a rank number its frames without
    talking to the others."""
    local = frame % keyint
    return local, 2 * local


# ---------------------------------------------------------------------------------------------------------------------------
# The data path: ONE clip, split into its closed GOPs, the GOPs spread over the ranks (one process per GPU).  A rank advances all
# of its GOPs together -- they are the chains of the raster sweep's batch, one wavefront each -- so that frame t of every GOP it
# owns is coded by the same kernel launch.  Nothing is exchanged while coding (a closed GOP references nothing outside itself);
# what the ranks exchange afterwards is the result: the slice payloads or, as here, their digests.

def encode_clip(hip, cqm, frames, keyint, rank=0, world=1, **options):
    """Code `frames` (a list of (y, u, v) numpy planes, display order, I/P only) as ONE stream with an IDR every `keyint`
    frames; this rank codes the GOPs shard.gops_for_rank gives it.  Returns {gop_index: [payload bytes of its frames]}.
    The payloads are those of a single-process run of the whole clip, bit for bit: frame_num / POC restart at every IDR and the
    only stream-global quantity in a slice's data, the padding bit x264_cabac_encode_flush derives from the number of frames
    coded so far (R/common/cabac.c:918), is given to the kernel per chain (x264hip_slice_rd.i_frame_stride)."""
    from . import slice as sl
    if options.get("lanes"):
        raise ValueError("encode_clip: lanes are a scheduling option of single-stream chains, not of a sharded clip")
    n = len(frames)
    gops = gop_bounds(n, keyint)
    mine = [i for i in range(len(gops)) if i % world == rank]
    out = {g: [] for g in mine}
    if not mine:
        return out
    h, w = frames[0][0].shape
    enc = sl.ChainEncoder(hip, w, h, cqm, batch=len(mine), write=1, keyint=keyint, **options)
    try:
        # chain b is GOP mine[b] = rank + b * world: frame t of it is frame (rank + b * world) * keyint + t of the stream
        enc.i_frame, enc.i_frame_stride = rank * keyint, world * keyint
        for t in range(keyint):
            live = [b for b, g in enumerate(mine) if gops[g][0] + t < gops[g][1]]
            if not live:
                break
            for b, g in enumerate(mine):            # a GOP cut short by the end of the clip repeats its last frame; the result is dropped
                y, u, v = frames[min(gops[g][0] + t, gops[g][1] - 1)]
                enc.upload(y, u, v, b=b)
            enc.encode_frame()
            enc.status()
            pay = enc.payloads()
            enc.finish_frame()
            for b in live:
                out[mine[b]].append(pay[b])
        enc.ctx.sync()
    finally:
        enc.close()
    return out


def encode_clip_stream(hip, cqm, frames, keyint, rank=0, world=1, **options):
    """The same sharding with the whole encoder per GOP: `frames` is cut at every multiple of `keyint` and each piece is a stream of its
    own through x264_vs2008_amd.stream.StreamEncoder -- its own lookahead (b-adapt, pre-encode scene cut), rate control (CRF or constant
    QP) and B frames -- so every piece starts with an IDR and references nothing outside itself; this rank codes the pieces
    gops_for_rank gives it, all of them in every launch.  Returns {gop_index: [(input number within the piece, slice type, qp, payload)]}
    in coding order.  A piece's result is what the reference's encoder produces for that piece ALONE (tests/test_gpu_shard.py): the
    lookahead and the rate control of a single long stream see across the cut, so the concatenation is a valid stream of closed GOPs
    but not the single-process stream of the whole clip -- the price of coding GOPs side by side, the same in the reference when it is
    run once per segment."""
    from .stream import StreamEncoder
    n = len(frames)
    gops = gop_bounds(n, keyint)
    mine = [i for i in range(len(gops)) if i % world == rank]
    out = {g: [] for g in mine}
    if not mine:
        return out
    h, w = frames[0][0].shape
    lens = [gops[g][1] - gops[g][0] for g in mine]
    options = dict(options)
    options.setdefault("keyint", keyint)
    enc = StreamEncoder(hip, w, h, cqm, batch=len(mine), limits=lens, **options)
    try:
        def fill(pic, f):
            for b, g in enumerate(mine):            # a piece cut short by the end of the clip repeats its last picture; nobody reads it
                y, u, v = frames[min(gops[g][0] + f, gops[g][1] - 1)]
                enc.src_ctx.upload(pic, y, u, v, b=b)

        fed, idle = 0, 0
        while idle < 2:
            coded = enc.step(fill if fed < max(lens) else None)
            fed += fed < max(lens)
            if coded:
                idle = 0
                enc.sync()
                enc.status()
                pay = enc.payloads()
                for cd in coded:
                    out[mine[cd.chain]].append((cd.frame, cd.slice_type, cd.qp, pay[cd.chain]))
            elif fed >= max(lens):
                idle += 1
    finally:
        enc.close()
    return out


def payload_digests(per_gop):
    """{gop_index: sha256 over the GOP's slice payloads (each preceded by its length)} -- what the ranks gather."""
    import hashlib
    d = {}
    for g, pays in per_gop.items():
        hsh = hashlib.sha256()
        for p in pays:
            if isinstance(p, tuple):                 # encode_clip_stream's records: the order and the decisions belong to the result
                hsh.update(repr(p[:3]).encode())
                p = p[3]
            hsh.update(len(p).to_bytes(4, "little"))
            hsh.update(p)
        d[g] = hsh.hexdigest()
    return d


def gather_digests(local, dist=None):
    """All ranks' {gop: digest} merged on every rank (torch.distributed all_gather_object over gloo -- the payload bytes stay
    where they are; a muxer would fetch them by GOP index).  Without a process group: the local dictionary."""
    if dist is None:
        return dict(local)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local)
    merged = {}
    for p in parts:
        merged.update(p)
    return merged
