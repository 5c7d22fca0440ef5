"""B slices on the GPU (round 2): the raster sweep's B instantiation against the reference's own loop -- coding order, list 0 /
list 1, spatial direct prediction, bi-prediction with and without weights, b8x8 / 16x8 / 8x16, the B RD decision, bidirectional
refinement, the CABAC B syntax -- frame after frame, decisions, pixels and payload bytes."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle.gen_golden_slice import CASES2, case_inputs
from x264_vs2008_amd import slice as sl

pytestmark = pytest.mark.gpu
B_CASES = [c for c in CASES2 if c[5].get("bframes")]      # spatial and temporal direct prediction
STATE = ["mb_type", "partition", "sub_partition", "ref", "mv", "i4mode", "i16mode", "chroma_mode", "qp", "cbp", "t8", "nnz", "luma", "luma_dc", "chroma_dc", "chroma_ac"]


def get1(enc, state, name, shape, dt):
    out = np.zeros(shape, dt)
    import ctypes as C
    assert enc.ctx.lib.x264hip_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(getattr(state.st, name)), C.c_size_t(out.nbytes)) == 0
    return out


@pytest.mark.parametrize("name,size,frames,kind,kw,ekw", B_CASES, ids=[c[0] for c in B_CASES])
def test_b_slices_match_reference_loop_and_payload(hip_lib, cqm, name, size, frames, kind, kw, ekw):
    with np.load(os.path.join(GOLDEN, "slice2_%s.npz" % name)) as z:
        gold = {k: z[k] for k in z.files}
    y, u, v = case_inputs(size, frames, kind)
    kw, ekw = dict(kw), dict(ekw)
    kw.pop("cqm_preset", 0)
    from test_gpu_slice_rd import lowres_arrays
    lowres = lowres_arrays(hip_lib, ekw.pop("lowres_seed", None), frames, size)
    enc = sl.ChainEncoder(hip_lib, size[0], size[1], cqm, batch=1, write=1, **kw, **{k: v_ for k, v_ in ekw.items() if k != "write"})
    order = sl.coding_order(frames, kw.get("keyint", 0), ekw["bframes"])
    assert [d for d, _ in order] == gold["frame_info2"][:, 0].tolist() and [t for _, t in order] == gold["frame_info"][:, 0].tolist()
    n = gold["mb_type"].shape[1]
    try:
        for f, (disp, stype) in enumerate(order):
            enc.upload(y[disp], u[disp], v[disp])
            lw = dict(lowres_mv=lowres[f][0], lowres_mv1=lowres[f][1]) if lowres else {}
            st, qp, state = enc.encode_frame(stype=stype, disp=disp, **lw)
            enc.status()
            assert (st, qp) == (int(gold["frame_info"][f, 0]), int(gold["frame_info"][f, 1])), "frame %d" % f
            got = {k: state.get(k)[0] for k in STATE}
            if stype == sl.SLICE_B:
                got["mv1"] = get1(enc, state, "mv1", (n, 16, 2), np.int16)
                got["ref1"] = get1(enc, state, "ref1", (n, 4), np.int8)
            for k in STATE + (["mv1", "ref1"] if stype == sl.SLICE_B else []):
                if stype == sl.SLICE_I and k in ("mv", "ref"):
                    continue
                want = gold[k][f]
                assert np.array_equal(got[k], want), "frame %d (%s, display %d): %s differs first at %s" % (
                    f, "IPB"[[2, 0, 1].index(stype)], disp, k, np.argwhere(got[k] != want)[:4].tolist())
            pay = enc.payloads()[0]
            want = bytes(gold["payload"][f, :gold["payload_len"][f]])
            assert pay == want, "frame %d: payload differs (%d vs %d bytes)" % (f, len(pay), len(want))
            recon = enc.last[0]
            for nm in ("y", "u", "v"):
                assert np.array_equal(enc.ctx.download(recon, nm, padded=False, b=0), gold["rec_" + nm][f]), "frame %d rec_%s" % (f, nm)
            enc.finish_frame()
            enc.ctx.sync()
    finally:
        enc.close()
    t = gold["mb_type"]
    assert (t == 16).any() and (t == 17).any() and ((t == 18).any() or gold["frame_info"][:, 1].min() < 26)      # B_BI_BI, B_8x8, B_SKIP (none at the low-QP case)


LANE_CASES = [c for c in B_CASES if c[5].get("direct_pred", 1) == 1 and "lowres_seed" not in c[5]][:2]      # (temporal direct prediction chains the frames: no lanes)


@pytest.mark.parametrize("name,size,frames,kind,kw,ekw", LANE_CASES, ids=[c[0] for c in LANE_CASES])
def test_b_frames_on_lanes_run_beside_the_next_anchor(hip_lib, cqm, name, size, frames, kind, kw, ekw):
    """lanes = 3: every B frame on a stream of its own, ordered by events only (no host synchronisation until the clip is
    enqueued).  The frames still resident at the end -- the last anchor and the B frames after it -- match the reference."""
    with np.load(os.path.join(GOLDEN, "slice2_%s.npz" % name)) as z:
        gold = {k: z[k] for k in z.files}
    y, u, v = case_inputs(size, frames, kind)
    kw = dict(kw)
    kw.pop("cqm_preset", 0)
    enc = sl.ChainEncoder(hip_lib, size[0], size[1], cqm, batch=1, write=1, lanes=3, **kw, **{k: v_ for k, v_ in ekw.items() if k != "write"})
    order = sl.coding_order(frames, kw.get("keyint", 0), ekw["bframes"])
    srcs = []
    for d in range(frames):
        pic = enc.ctx.new_picture(source_only=True)
        enc.ctx.upload(pic, y[d], u[d], v[d], b=0)
        srcs.append(pic)
    held = {}                                  # frame index in coding order -> (recon, state, bufs) of what is still resident at the end
    try:
        for f, (disp, stype) in enumerate(order):
            enc.encode_frame(srcs[disp], stype=stype, disp=disp)
            held[id(enc.last[1])] = (f, stype, enc.last[0], enc.last[1], enc.last_bufs)
            enc.finish_frame()
        enc.sync()
        enc.status()
        n = gold["mb_type"].shape[1]
        checked_b = 0
        for f, stype, recon, state, bufs in held.values():
            if stype != sl.SLICE_B and f != max(g for g, t, *_ in held.values() if t != sl.SLICE_B):
                continue                       # an older anchor's payload buffer has been reused since
            for k in STATE:
                if stype == sl.SLICE_I and k in ("mv", "ref"):
                    continue
                assert np.array_equal(state.get(k)[0], gold[k][f]), "frame %d: %s" % (f, k)
            if stype == sl.SLICE_B:
                assert np.array_equal(get1(enc, state, "mv1", (n, 16, 2), np.int16), gold["mv1"][f])
                checked_b += 1
            ln = int(bufs["payload_len"].get()[0])
            pay = bytes(bufs["payload"].get()[0, sl.PAYLOAD_LEAD:sl.PAYLOAD_LEAD + ln])
            assert pay == bytes(gold["payload"][f, :gold["payload_len"][f]]), "frame %d: payload" % f
            for nm in ("y", "u", "v") if stype == sl.SLICE_B else ():       # (the anchor has been through the loop filter since)
                assert np.array_equal(enc.ctx.download(recon, nm, padded=False, b=0), gold["rec_" + nm][f]), "frame %d rec_%s" % (f, nm)
        assert checked_b >= 2
    finally:
        enc.close()


def test_source_only_picture_is_refused_as_a_reference_and_lanes_are_freed(hip_lib, cqm):
    """A picture without half-pel planes (x264hip_picture_alloc_source) handed to the sweep as a reference is an error string, not a
    device fault; and an encoder with lanes gives all of its device memory back on close() (allocate / close in a loop)."""
    import ctypes as C
    w, h = 208, 144
    enc = sl.ChainEncoder(hip_lib, w, h, cqm, batch=1, write=1, qp=28, subme=5, me_method=1, n_refs=1, cabac=1, deblock=1)
    try:
        y = np.full((h, w), 90, np.uint8); u = np.full((h // 2, w // 2), 128, np.uint8)
        enc.upload(y, u, u)
        enc.encode_frame()
        enc.finish_frame()
        bad = enc.ctx.new_picture(source_only=True)
        enc.refs[0] = (bad,) + tuple(enc.refs[0][1:])
        with pytest.raises(RuntimeError, match="half-pel"):
            enc.encode_frame()
    finally:
        enc.close()
    hip_lib.x264hip_mem_info.restype = C.c_int
    free = []
    for it in range(4):
        enc = sl.ChainEncoder(hip_lib, 352, 288, cqm, batch=4, write=1, lanes=3, bframes=3, qp=28, subme=7, me_method=1, n_refs=2, inter=0x113, intra=0x3,
                              transform8x8=1, cabac=1, deblock=1, trellis=1, aq_mode=1)
        enc.close()
        f, t = C.c_size_t(0), C.c_size_t(0)
        assert hip_lib.x264hip_mem_info(C.byref(f), C.byref(t)) == 0
        free.append(f.value)
    assert free[-1] >= free[0] - (1 << 20), free          # nothing accumulates from one encoder to the next


def test_subme9_with_b_slices_is_refused_not_approximated(hip_lib, cqm):
    """subme 9 refines a B macroblock's vectors by RD (x264_me_refine_bidir_rd and the list-1 form of x264_me_refine_qpel_rd): the twin has
    it (fixtures slice2_rd9_b*.npz), the B kernel does not -- the sweep says so on the first B slice instead of coding it at subme 8."""
    w, h = 208, 144
    enc = sl.ChainEncoder(hip_lib, w, h, cqm, batch=1, write=1, qp=28, subme=9, me_method=1, n_refs=2, inter=0x113, intra=0x3, transform8x8=1, cabac=1, deblock=1,
                          trellis=1, bframes=2, weightb=1)
    try:
        y = np.full((h, w), 90, np.uint8); u = np.full((h // 2, w // 2), 128, np.uint8)
        for disp, stype in sl.coding_order(4, 0, 2)[:2]:      # I and P at subme 9 run (the refinement kernel)
            enc.upload(y, u, u)
            enc.encode_frame(stype=stype, disp=disp)
            enc.status()
            enc.finish_frame()
        enc.upload(y, u, u)
        with pytest.raises(RuntimeError, match="subme 9"):
            enc.encode_frame(stype=sl.SLICE_B, disp=1)
    finally:
        enc.close()
