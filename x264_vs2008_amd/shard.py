"""Frame sharding across GPUs: closed GOPs are the unit.

Frames between two IDR frames reference nothing outside their GOP, so GOPs go
to ranks round-robin with no pixel exchange; the only cross-GOP state in the
reference is scalar rate-control history, which stays on the host.  This
module is pure bookkeeping (no GPU, no collective on the data path)."""


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
