"""GPU: the raster-order variant of the sweep (include/x264hip.h: x264hip_slice_rd) -- the RD levels (subme 6 / 7), trellis,
psy-rd, adaptive quantisation and the CABAC entropy coder inside the macroblock loop -- against chains the REFERENCE's own loop
produced with x264_macroblock_write_cabac in it (oracle/ref_slice.c refslice_encode_chain2, fixtures tests/golden/slice2_*.npz):
every decision, level and pixel as in test_gpu_slice.py, the per-macroblock QP, AND the slice payload bytes."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle.gen_golden_slice import CASES2, case_inputs
from x264_vs2008_amd import slice as sl
from test_gpu_slice import STATE, check_frame

pytestmark = pytest.mark.gpu


def lowres_arrays(hip_lib, seed, frames, size, batch=1):
    """The fixture's stand-in lookahead vectors (oracle/refslice.py: lowres_vectors) on the device: [(list 0, list 1)] per frame in coding order."""
    if seed is None:
        return None
    from oracle.refslice import lowres_vectors
    from x264_vs2008_amd.frame import DeviceArray
    n = ((size[0] + 15) // 16) * ((size[1] + 15) // 16)
    lm = lowres_vectors(seed, frames, n)
    return [tuple(DeviceArray(hip_lib, (batch, n, 2), np.int16, np.ascontiguousarray(np.broadcast_to(lm[f, l], (batch, n, 2)))) for l in range(2)) for f in range(frames)]


def run_chain2(hip_lib, cqm, size, frames, y, u, v, kw, ekw, batch=1):
    kw = dict(kw)
    kw.pop("cqm_preset", 0)
    ekw = dict(ekw)
    lowres = lowres_arrays(hip_lib, ekw.pop("lowres_seed", None), frames, size, batch)
    enc = sl.ChainEncoder(hip_lib, size[0], size[1], cqm, batch=batch, write=1, **kw, **{k: v_ for k, v_ in ekw.items() if k != "write"})
    out = []
    # B frames: the chain in coding order (the golden fixtures of such chains are in coding order too)
    order = sl.coding_order(frames, kw.get("keyint", 0), ekw["bframes"]) if ekw.get("bframes") else None
    try:
        for f in range(frames):
            disp = order[f][0] if order else f
            for b in range(batch):
                enc.upload(y[disp], u[disp], v[disp], b=b)
            lw = dict(lowres_mv=lowres[f][0], lowres_mv1=lowres[f][1]) if lowres else {}
            stype, qp, state = enc.encode_frame(stype=order[f][1], disp=disp, **lw) if order else enc.encode_frame(**lw)
            enc.status()
            recon = enc.last[0]
            d = {k: state.get(k) for k in STATE + ["mvr", "cost_intra", "cost_inter"]}
            if order:
                n_mb = state.get("mb_type").shape[1]
                for nm, tail, dt in (("mv1", (16, 2), np.int16), ("ref1", (4,), np.int8)):
                    a = np.zeros((batch, n_mb) + tail, dt)
                    assert enc.ctx.lib.x264hip_memcpy_d2h(a.ctypes.data_as(C.c_void_p), C.c_void_p(getattr(state.st, nm)), C.c_size_t(a.nbytes)) == 0
                    if stype != sl.SLICE_B:               # the reference reports zeros / -1 outside B slices
                        a[...] = 0 if nm == "mv1" else -1
                    d[nm] = a
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


def test_payload_only_states_and_a_full_payload_buffer(hip_lib, cqm):
    """levels=False (X264HIP_STATE_NO_LEVELS: the states carry no coefficient-level arrays, the payload is the product) gives the
    same bytes; and a payload buffer the slice does not fit stops the sweep with an error instead of writing past it."""
    name, size, frames, kind, kw, ekw = next(c for c in CASES2 if c[0] == "rd7_aq")
    with np.load(os.path.join(GOLDEN, "slice2_%s.npz" % name)) as z:
        gold = {k: z[k] for k in z.files}
    y, u, v = case_inputs(size, frames, kind)
    kw = dict(kw)
    kw.pop("cqm_preset", 0)
    ekw = {k: v_ for k, v_ in ekw.items() if k != "write"}
    enc = sl.ChainEncoder(hip_lib, size[0], size[1], cqm, batch=1, write=1, levels=False, **kw, **ekw)
    try:
        assert not enc.states[0].st.luma and not enc.states[0].st.chroma_ac and enc.states[0].st.mv
        for f in range(frames):
            enc.upload(y[f], u[f], v[f])
            enc.encode_frame()
            enc.status()
            assert enc.payloads()[0] == bytes(gold["payload"][f, :int(gold["payload_len"][f])]), "frame %d" % f
            enc.finish_frame()
    finally:
        enc.close()
    # (at QP 6 the first slice of this clip is far beyond the 2 KB such a buffer leaves before the guard's margin: one macroblock's proven worst case)
    enc = sl.ChainEncoder(hip_lib, size[0], size[1], cqm, batch=1, write=1, payload_cap=8192 + 128 + 2048 + sl.PAYLOAD_LEAD, **dict(kw, qp=6), **ekw)
    try:
        enc.upload(y[0], u[0], v[0])
        enc.encode_frame()
        with pytest.raises(RuntimeError, match="aborted"):
            enc.status()
    finally:
        enc.close()
