"""GPU: the whole per-frame hot-path pass (lowres, AQ energy, motion search on
three references, residual, deblock, borders, half-pel planes, SSD), chained
over several frames so every reconstruction becomes the next frame's
reference, against the same chain run on the CPU twin.  Every array and every
plane -- padding included -- must be identical after each frame."""
import numpy as np
import pytest

from frame_util import HostPic
from oracle import hostpic
from x264_vs2008_amd import synth
from x264_vs2008_amd.frame import FrameCtx, chroma_qp
from x264_vs2008_amd.pipeline import PFramePass, setup_event_api

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size,qp,t8,frames", [((352, 288), 26, 1, 4), ((200, 120), 32, 0, 3)])
def test_chained_pass_matches_cpu_twin(hip_lib, oracle_lib, cqm, size, qp, t8, frames):
    setup_event_api(hip_lib)
    ctx = FrameCtx(hip_lib, *size)
    try:
        d = ctx.dims
        g = hostpic.Geometry(d.width, d.height)
        pas = PFramePass(hip_lib, ctx, cqm, qp=qp, transform8x8=t8, n_refs=3)
        # three initial references = source frames 0..2 (as if coded losslessly)
        dev_refs, host_refs = [], []
        for t in (2, 1, 0):
            pic = ctx.new_picture(); hp = HostPic(ctx, pic)
            y, u, v = synth.frame(d.width, d.height, t)
            ctx.upload(pic, y, u, v); hp.load_yuv(oracle_lib, "x264o_", y, u, v)
            pas.make_reference(pic); hostpic.make_reference(oracle_lib, "x264o_", hp)
            dev_refs.append(pic); host_refs.append(hp)
        for t in range(3, 3 + frames):
            cur = ctx.new_picture(); hcur = HostPic(ctx, cur)
            y, u, v = synth.frame(d.width, d.height, t)
            ctx.upload(cur, y, u, v); hcur.load_yuv(oracle_lib, "x264o_", y, u, v)
            recon = ctx.new_picture(); hrec = HostPic(ctx, recon)
            pas.step(cur, dev_refs, recon)
            got = pas.results()
            want = hostpic.cpu_pframe_pass(oracle_lib, "x264o_", g, hcur, host_refs, hrec, cqm, qp, chroma_qp(qp),
                                           pas.cost_tab, len(pas.cost_tab) // 2, pas.me_range, t8)
            for k in ("aq", "cbp", "nnz", "levels_y", "levels_c", "dc_c", "ssd"):
                assert np.array_equal(got[k], want[k]), "frame %d: %s" % (t, k)
            for i in range(3):
                for k in ("mv9", "cost9", "mvq", "costq"):
                    assert np.array_equal(got[k][i], want[k][i]), "frame %d ref %d: %s" % (t, i, k)
            for name in ("y", "u", "v", "h", "vv", "c"):
                assert np.array_equal(ctx.download(recon, name), hrec.arr(name)), "frame %d recon plane %s" % (t, name)
            for name in ("l0", "lh", "lv", "lc"):
                assert np.array_equal(ctx.download(cur, name), hcur.arr(name)), "frame %d lowres %s" % (t, name)
            dev_refs = [recon] + dev_refs[:2]; host_refs = [hrec] + host_refs[:2]
        assert got["cbp"].any()
    finally:
        ctx.close()
