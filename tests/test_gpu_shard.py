"""The multi-GPU data path: one clip, its closed GOPs spread over the ranks, every rank's GOPs coded together as
the chains of the raster sweep (x264_vs2008_amd/shard.py).  What the ranks produce together must be the single-rank stream, and
that must be the reference's: the oracle codes the whole clip as ONE chain with the same keyint."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import refslice as rs
from x264_vs2008_amd import shard, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPTS = dict(qp=27, me_method=1, subme=7, n_refs=2, inter=0x13, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1, deblock=1)
EXT = dict(trellis=1, psy_rd=1.0, aq_mode=1)


def test_gop_sharded_clip_is_the_single_chain_stream(hip_lib, oracle_lib, cqm):
    w, h, keyint, n = 208, 144, 4, 11                      # GOPs 0..3 frames, 4..7, 8..10 (cut short by the end of the clip)
    frames = [synth.frame(w, h, t) for t in range(n)]
    y, u, v = (np.ascontiguousarray(np.stack([f[i] for f in frames])) for i in range(3))
    want = rs.run2(oracle_lib, "x264o_encode_chain2", rs.make_params(w, h, n, keyint=keyint, **OPTS), rs.make_ext(**EXT), y, u, v)
    stream = [bytes(want["payload"][f, :want["payload_len"][f]]) for f in range(n)]
    whole = shard.encode_clip(hip_lib, cqm, frames, keyint, 0, 1, **OPTS, **EXT)
    assert sorted(whole) == [0, 1, 2] and [len(whole[g]) for g in range(3)] == [4, 4, 3]
    assert [p for g in range(3) for p in whole[g]] == stream
    # two ranks (here one after the other on the one GPU): disjoint GOPs, together the same stream
    parts = [shard.encode_clip(hip_lib, cqm, frames, keyint, r, 2, **OPTS, **EXT) for r in range(2)]
    assert sorted(parts[0]) == [0, 2] and sorted(parts[1]) == [1]
    merged = {**parts[0], **parts[1]}
    assert merged == whole
    assert shard.gather_digests(shard.payload_digests(merged)) == shard.payload_digests(whole)
    # the padding bit of the flush depends on the frame's position in the stream: a GOP coded as if it were the first differs
    alone = shard.encode_clip(hip_lib, cqm, frames[keyint:2 * keyint], keyint, 0, 1, **OPTS, **EXT)
    assert alone[0] != whole[1] and [len(p) for p in alone[0]] == [len(p) for p in whole[1]]


def test_two_process_gloo_shard():
    """One process per rank over torch.distributed (gloo), both on this box's GPU: tests/shard_worker.py."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "tests", "shard_worker.py")], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "equal" in out, out[-2000:]


REF_SO = os.path.join(ROOT, "oracle", "_ref", "libx264ref.so")


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libx264ref.so not built (needs /root/reference)")
def test_gop_sharded_streams_with_b_frames_equal_the_reference_per_piece(hip_lib, cqm):
    """encode_clip_stream: the clip cut at every keyint, each piece through the WHOLE encoder (lookahead, adaptive B frames, CRF) on the
    GPU side, pieces of unequal length in one batch; each piece equals the reference's encoder run on that piece, and two ranks produce
    disjoint pieces that together are the single-rank result."""
    w, h, keyint, n = 128, 96, 9, 24                        # pieces of 9, 9 and 6 pictures
    frames = [synth.frame(w, h, t // 3) for t in range(n)]      # every picture three times: what b-adapt answers with B frames
    kw = dict(qp=26, me_method=1, subme=6, n_refs=2, inter=0x13, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1, deblock=1)
    ekw = dict(trellis=1, psy_rd=1.0, aq_mode=1, bframes=2, weightb=1)
    look = dict(crf=24.0, b_adapt=1, pre_scenecut=1, scenecut_threshold=40)
    whole = shard.encode_clip_stream(hip_lib, cqm, frames, keyint, 0, 1, qp_min=0, **kw, **ekw, **look)
    assert sorted(whole) == [0, 1, 2] and [len(whole[g]) for g in range(3)] == [9, 9, 6]
    for g, (a, b) in enumerate(shard.gop_bounds(n, keyint)):
        y, u, v = (np.ascontiguousarray(np.stack([f[i] for f in frames[a:b]])) for i in range(3))
        ref = rs.run_reference_stream(rs.make_params(w, h, b - a, keyint=keyint, **kw), rs.make_ext(**ekw, **look), y, u, v)
        want = [(int(ref["frame_info2"][f][0]), int(ref["frame_info"][f][0]), int(ref["frame_info"][f][1]), bytes(ref["payload"][f, :ref["payload_len"][f]]))
                for f in range(b - a)]
        assert [r[:3] for r in whole[g]] == [r[:3] for r in want], "piece %d: order / types / QPs" % g
        assert whole[g] == want, "piece %d: payloads" % g
        assert any(r[1] == 1 for r in want), "piece %d has no B frame: the test lost its point" % g
    parts = [shard.encode_clip_stream(hip_lib, cqm, frames, keyint, r, 2, qp_min=0, **kw, **ekw, **look) for r in range(2)]
    assert sorted(parts[0]) == [0, 2] and sorted(parts[1]) == [1]
    merged = {**parts[0], **parts[1]}
    assert merged == whole
    assert shard.gather_digests(shard.payload_digests(merged)) == shard.payload_digests(whole)
