"""A C-compiled consumer of the C ABI (examples/encode_chain.c: gcc -std=c99, include/x264hip.h only) codes an I P B B B P ... chain
with the medium preset's options; its slice payloads must equal, byte for byte, what the Python host (ChainEncoder, the path the
reference-pinned tests drive) produces from the same pictures."""
import os
import struct
import subprocess

import numpy as np
import pytest

from x264_vs2008_amd import lib as L, synth
from x264_vs2008_amd import slice as sl

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("bframes", [0, 3])
def test_c_consumer_matches_python_host(hip_lib, cqm, tmp_path, bframes):
    w, h, frames = 176, 144, 7
    exe = os.path.join(ROOT, "examples", "encode_chain")
    if not os.path.exists(exe):
        L.build_examples()
    pics = [synth.frame(w, h, 3 * t) for t in range(frames)]
    yuv = tmp_path / "in.yuv"
    with open(yuv, "wb") as f:
        for y, u, v in pics:
            f.write(np.ascontiguousarray(y).tobytes()); f.write(np.ascontiguousarray(u).tobytes()); f.write(np.ascontiguousarray(v).tobytes())
    out = tmp_path / "out.bin"
    r = subprocess.run([exe, str(yuv), str(w), str(h), str(frames), str(out), str(bframes)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got, raw, o = [], out.read_bytes(), 0
    while o < len(raw):
        disp, stype, n = struct.unpack_from("<iii", raw, o)
        got.append((disp, stype, raw[o + 12:o + 12 + n]))
        o += 12 + n
    assert len(got) == frames

    enc = sl.ChainEncoder(hip_lib, w, h, cqm, batch=1, qp=26, me_method=1, me_range=16, subme=7, n_refs=3, inter=0x113, intra=0x3, transform8x8=1,
                          fast_pskip=1, dct_decimate=1, chroma_me=1, cabac=1, deblock=1, mixed_refs=1, trellis=1, psy_rd=1.0, aq_mode=1, aq_strength=1.0,
                          write=1, bframes=bframes, weightb=1, direct_pred=1)
    order = sl.coding_order(frames, 0, bframes) if bframes else [(t, sl.SLICE_I if t == 0 else sl.SLICE_P) for t in range(frames)]
    try:
        for f, (disp, stype) in enumerate(order):
            enc.upload(*pics[disp])
            enc.encode_frame(stype=stype, disp=disp)
            enc.status()
            assert (disp, stype) == got[f][:2], "frame %d: order" % f
            assert enc.payloads()[0] == got[f][2], "frame %d (display %d): payload differs" % (f, disp)
            enc.finish_frame()
    finally:
        enc.close()
    assert sum(len(g[2]) for g in got) > 2000
