"""TEST INFRASTRUCTURE -- hash fixtures at BASELINE's frame sizes, produced by the REFERENCE's own loop (oracle/ref_slice.c in
oracle/_ref/libx264ref.so).  The arrays of a 1920x1080 or 3840x2160 chain are far too big to commit, so the fixture holds a
sha256 per array per frame (tests/golden/hash_*.json); tests/ compare the twin (CPU) and the GPU sweep against them.  Runs only
where /root/reference exists.

    python -m oracle.gen_golden_hash
"""
import hashlib
import json
import os
import sys

import numpy as np

from oracle import refslice as rs
from oracle.gen_golden_slice import MED, MEDB, SLOW, masked2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

ARRAYS = ["mb_type", "partition", "sub_partition", "mv", "ref", "nnz", "i4mode", "i16mode", "chroma_mode", "qp", "cbp", "t8", "luma", "luma_dc", "chroma_dc",
          "chroma_ac", "rec_y", "rec_u", "rec_v", "fin_y", "fin_u", "fin_v"]

# (name, size, frames, parameters, ext parameters or None for the round-1 entry)
HASH_CASES = [
    ("hd_subme5", (1920, 1080), 3, dict(qp=28, subme=5, me_method=rs.ME_HEX, n_refs=2, inter=0x13, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1, deblock=1), None),
    ("hd_medium_rd", (1920, 1080), 3, dict(qp=26, subme=7, **MED), dict(trellis=1, psy_rd=1.0, aq_mode=1, aq_strength=1.0)),
    ("uhd_umh_subme5", (3840, 2160), 2, dict(qp=28, subme=5, me_method=rs.ME_UMH, n_refs=2, inter=0x13, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1, deblock=1), None),
    ("uhd_umh_medium_rd", (3840, 2160), 2, dict(qp=26, subme=7, **dict(MED, me_method=rs.ME_UMH, n_refs=2)), dict(trellis=1, psy_rd=1.0, aq_mode=1, aq_strength=1.0)),
    # the medium preset's analysis options with its GOP shape: I P B B B P in coding order (3 disposable B frames, weightb, spatial direct)
    ("hd_medium_b", (1920, 1080), 6, dict(qp=26, subme=7, **dict(MEDB, n_refs=3)), dict(trellis=1, psy_rd=1.0, aq_mode=1, aq_strength=1.0, bframes=3, weightb=1, direct_pred=1)),
    # round 3.  BASELINE config 3's analysis (SURVEY 8(d) SLOW_SHARD: --ref 5 --bframes 3 --me umh --subme 8 --8x8dct --trellis 1 --weightb --mixed-refs --direct spatial)
    # at constant QP, I P B B B P: the RD refinement in the I / P slices, mode-decision RD in the B slices
    ("hd_slow_b", (1920, 1080), 6, dict(qp=26, subme=8, **dict(SLOW, inter=0x113)), dict(trellis=1, psy_rd=1.0, aq_mode=1, aq_strength=1.0, bframes=3, weightb=1, direct_pred=1)),
    # BASELINE config 2 as stated: 3840x2160, the medium set with --me umh, 3 references, 3 B frames
    ("uhd_umh_medium_b", (3840, 2160), 6, dict(qp=26, subme=7, **dict(MEDB, me_method=rs.ME_UMH, n_refs=3)), dict(trellis=1, psy_rd=1.0, aq_mode=1, aq_strength=1.0, bframes=3, weightb=1, direct_pred=1)),
]


def arrays_for(ekw):
    return ARRAYS + (["mv1", "ref1"] if ekw and ekw.get("bframes") else [])


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def hashes(arrs, frames, names=ARRAYS):
    """{array: [sha256 per frame]} (+ the payload bytes and the frame info when the writer ran)."""
    out = {k: [digest(arrs[k][f]) for f in range(frames)] for k in names}
    out["frame_info"] = arrs["frame_info"].tolist()
    out["stat"] = arrs["stat"].tolist()
    if "payload_len" in arrs:
        n = arrs["payload_len"]
        out["payload_len"] = [int(x) for x in n]
        out["payload"] = [hashlib.sha256(bytes(arrs["payload"][f, :n[f]])).hexdigest() for f in range(frames)]
    return out


def run_case(lib_run, lib_run2, size, frames, kw, ekw):
    y, u, v = rs.clip(size[0], size[1], frames)
    p = rs.make_params(size[0], size[1], frames, **kw)
    if ekw is None:
        from oracle.gen_golden_slice import masked
        return masked(lib_run(p, y, u, v))
    return masked2(lib_run2(p, rs.make_ext(**ekw), y, u, v))


def main():
    only = sys.argv[1:]
    for name, size, frames, kw, ekw in HASH_CASES:
        if only and name not in only:
            continue
        a = run_case(rs.run_reference, rs.run_reference2, size, frames, kw, ekw)
        h = hashes(a, frames, arrays_for(ekw))
        if ekw and ekw.get("bframes"):
            h["frame_info2"] = a["frame_info2"].tolist()
        h["mb_type_counts"] = [np.bincount(a["mb_type"][f], minlength=7).tolist() for f in range(frames)]
        path = os.path.join(GOLDEN, "hash_%s.json" % name)
        with open(path, "w") as f:
            json.dump(h, f, indent=0)
        print(path, h["mb_type_counts"], h.get("payload_len"))


if __name__ == "__main__":
    main()
