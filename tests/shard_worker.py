"""Worker of tests/test_gpu_shard.py::test_two_process_gloo_shard: one process per rank (torch.distributed.run), gloo for the
exchange of results, every rank on the one GPU of the box.  Exit code 0 = the two ranks' GOPs together are the single-rank stream."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist                    # noqa: E402  (CPU tensors / objects only: torch.cuda is never touched)

from x264_vs2008_amd import lib as L, shard, synth  # noqa: E402


def load_cqm_flat():
    with np.load(os.path.join(ROOT, "tests", "golden", "cqm_flat.npz")) as z:
        return {k: z[k] for k in z.files}


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    hip = L.load(0)
    w, h, keyint, n = 208, 144, 4, 15
    frames = [synth.frame(w, h, t) for t in range(n)]
    opts = dict(qp=27, me_method=1, subme=7, n_refs=2, inter=0x13, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1, deblock=1,
                trellis=1, psy_rd=1.0, aq_mode=1)
    mine = shard.payload_digests(shard.encode_clip(hip, load_cqm_flat(), frames, keyint, rank, world, **opts))
    merged = shard.gather_digests(mine, dist)
    ok = 1
    if rank == 0:
        whole = shard.payload_digests(shard.encode_clip(hip, load_cqm_flat(), frames, keyint, 0, 1, **opts))
        ok = int(merged == whole and len(whole) == 4)
        print("gops", sorted(merged), "equal" if ok else "DIFFERENT", flush=True)
    flag = [ok]
    dist.broadcast_object_list(flag, src=0)
    dist.destroy_process_group()
    sys.exit(0 if flag[0] else 1)


if __name__ == "__main__":
    main()
