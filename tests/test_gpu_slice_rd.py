"""GPU: the raster-order variant of the sweep (include/x264hip.h: x264hip_slice_rd) -- the RD levels (subme 6 / 7), trellis,
psy-rd, adaptive quantisation and the CABAC entropy coder inside the macroblock loop -- against chains the REFERENCE's own loop
produced with x264_macroblock_write_cabac in it (oracle/ref_slice.c refslice_encode_chain2, fixtures tests/golden/slice2_*.npz):
every decision, level and pixel as in test_gpu_slice.py, the per-macroblock QP, AND the slice payload bytes."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle.gen_golden_slice import CASES2, case_inputs
from x264_vs2008_amd import slice as sl
from test_gpu_slice import STATE, check_frame

pytestmark = pytest.mark.gpu


def run_chain2(hip_lib, cqm, size, frames, y, u, v, kw, ekw, batch=1):
    kw = dict(kw)
    kw.pop("cqm_preset", 0)
    enc = sl.ChainEncoder(hip_lib, size[0], size[1], cqm, batch=batch, write=1, **kw, **{k: v_ for k, v_ in ekw.items() if k != "write"})
    out = []
    try:
        for f in range(frames):
            for b in range(batch):
                enc.upload(y[f], u[f], v[f], b=b)
            stype, qp, state = enc.encode_frame()
            enc.status()
            recon = enc.last[0]
            d = {k: state.get(k) for k in STATE + ["mvr", "cost_intra", "cost_inter"]}
            d["info"] = (stype, qp)
            d["payload"] = enc.payloads()
            d["mb_bits"] = enc.rd_bufs["mb_bits"].get()
            for nm in ("y", "u", "v"):
                d["rec_" + nm] = np.stack([enc.ctx.download(recon, nm, padded=False, b=b) for b in range(batch)])
            enc.finish_frame()
            enc.ctx.sync()
            for nm in ("y", "u", "v"):
                d["fin_" + nm] = np.stack([enc.ctx.download(recon, nm, padded=False, b=b) for b in range(batch)])
            out.append(d)
    finally:
        enc.close()
    return out


IP_CASES2 = [c for c in CASES2 if not c[5].get("bframes")]        # the kernel codes I and P slices; the B chains pin the twin (round 3: the kernel)


@pytest.mark.parametrize("name,size,frames,kind,kw,ekw", IP_CASES2, ids=[c[0] for c in IP_CASES2])
def test_raster_sweep_matches_reference_loop_and_payload(hip_lib, cqm, name, size, frames, kind, kw, ekw):
    with np.load(os.path.join(GOLDEN, "slice2_%s.npz" % name)) as z:
        gold = {k: z[k] for k in z.files}
    y, u, v = case_inputs(size, frames, kind)
    out = run_chain2(hip_lib, cqm, size, frames, y, u, v, kw, ekw)
    for f in range(frames):
        n = int(gold["payload_len"][f])
        want = bytes(gold["payload"][f, :n])
        # the decisions first (their message says where), then the bytes
        check_frame(out[f], gold, f, kw.get("n_refs", 1))
        assert out[f]["payload"][0] == want, "frame %d: payload differs (%d vs %d bytes)" % (f, len(out[f]["payload"][0]), n)


def test_raster_sweep_batched_chains(hip_lib, cqm):
    """Three chains in one launch (one wavefront each): every element must equal the golden chain, bytes included."""
    name, size, frames, kind, kw, ekw = next(c for c in CASES2 if c[0] == "rd7_aq")
    with np.load(os.path.join(GOLDEN, "slice2_%s.npz" % name)) as z:
        gold = {k: z[k] for k in z.files}
    y, u, v = case_inputs(size, frames, kind)
    out = run_chain2(hip_lib, cqm, size, frames, y, u, v, kw, ekw, batch=3)
    for f in range(frames):
        n = int(gold["payload_len"][f])
        for b in range(3):
            check_frame(out[f], gold, f, kw.get("n_refs", 1), b=b)
            assert out[f]["payload"][b] == bytes(gold["payload"][f, :n])
