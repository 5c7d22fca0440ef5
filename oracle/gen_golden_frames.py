#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden vectors for the per-frame DRIVERS.

Runs the reference's own x264_frame_expand_border_mod16, x264_frame_init_lowres,
x264_frame_deblock_row, x264_frame_expand_border, x264_frame_filter and
x264_frame_expand_border_filtered (compiled from R/common/*.c into
oracle/_ref/libx264ref.so; oracle/ref_shim.c only builds the structs they
read) on seeded inputs and stores inputs + resulting padded planes under
tests/golden/.  tests/test_oracle_frame_golden.py then holds the CPU twin
(oracle/frame_oracle.c) to these outputs, which pins the twins' driver logic
(band bounds, border rules, bS derivation, alpha/beta/tc0 tables, array
layouts) to the reference, not just to our reading of it.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import hostpic  # noqa: E402
from x264_vs2008_amd import synth  # noqa: E402

CASES = [((352, 288), 1, 0, 0, 0, 18, 44), ((200, 120), 2, 3, -2, 2, 8, 51)]


class RefFrame(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("stride_y", C.c_int), ("stride_c", C.c_int),
                ("stride_lowres", C.c_int), ("plane", hostpic.u8p * 3), ("filtered", hostpic.u8p * 4),
                ("lowres", hostpic.u8p * 4)]


def ref_frame(pic):
    g = pic.g
    f = RefFrame(g.width, g.height, g.stride_y, g.stride_c, g.stride_lowres)
    for i, n in enumerate(("y", "u", "v")):
        f.plane[i] = pic.ptr(n)
    for i, n in enumerate(("y", "h", "vv", "c")):
        f.filtered[i] = pic.ptr(n)
    for i, n in enumerate(("l0", "lh", "lv", "lc")):
        f.lowres[i] = pic.ptr(n)
    return f


def blocky(img, r, step, amp):
    h, w = img.shape
    off = r.randint(-amp, amp + 1, ((h + step - 1) // step, (w + step - 1) // step))
    big = np.kron(off, np.ones((step, step), np.int64))[:h, :w]
    return np.clip(img.astype(np.int64) + big, 0, 255).astype(np.uint8)


def make_case(size, seed, qlo, qhi):
    """Seeded inputs: a blocky frame and random per-macroblock side information."""
    g = hostpic.Geometry(*size)
    n = g.mb_w * g.mb_h
    r = np.random.RandomState(seed)
    y, u, v = synth.frame(size[0], size[1], seed)
    y, u, v = blocky(blocky(y, r, 4, 3), r, 16, 6), blocky(u, r, 4, 4), blocky(v, r, 4, 4)
    mb_type = r.choice([0, 0, 0, 1, 2, 3], n).astype(np.uint8)
    qp = r.randint(qlo, qhi + 1, n).astype(np.uint8)
    t8 = (r.rand(n) < 0.4).astype(np.uint8)
    nnz = (r.rand(n, 26) < 0.3).astype(np.uint8)
    nnz[mb_type == 2] = 0
    nnz[r.rand(n) < 0.3] = 0
    for mb in np.nonzero(t8)[0]:
        for b in range(4):
            nnz[mb, 4 * b:4 * b + 4] = nnz[mb, 4 * b]
    mv = np.repeat(r.randint(-6, 7, (n, 1, 2)).astype(np.int16), 16, axis=1)
    ref = np.repeat(r.randint(0, 2, (n, 1)).astype(np.int8), 4, axis=1)
    for mb in np.nonzero(mb_type == 3)[0]:
        m8 = r.randint(-6, 7, (2, 2, 2)); r8 = r.randint(0, 2, 4)
        for by in range(4):
            for bx in range(4):
                mv[mb, bx + 4 * by] = m8[by >> 1, bx >> 1]
        ref[mb] = r8
    mv[mb_type == 1] = 0; ref[mb_type == 1] = -1
    return g, dict(y=y, u=u, v=v, mb_type=mb_type, qp=qp, t8=t8, nnz=nnz, mv=np.ascontiguousarray(mv), ref=np.ascontiguousarray(ref))


class RefRef(C.Structure):
    _fields_ = [("y", hostpic.u8p * 4), ("u", hostpic.u8p), ("v", hostpic.u8p)]


ENC_CASES = [((352, 288), 24, 0, 0), ((352, 288), 27, 1, 0), ((200, 120), 33, 1, 1), ((200, 120), 14, 0, 0)]


def encode_inputs(size, qp):
    """Seeded vectors (one per 4x4 block, mixed partition shapes) and reference indices for two references."""
    g = hostpic.Geometry(*size)
    n = g.mb_w * g.mb_h
    r = np.random.RandomState(qp)
    mv16 = np.zeros((n, 16, 2), np.int16)
    ref8 = r.randint(0, 2, (n, 4)).astype(np.int8)
    for mb in range(n):
        shape = r.randint(0, 4)
        for by in range(4):
            for bx in range(4):
                key = {0: 0, 1: by >> 1, 2: bx >> 1, 3: bx + 4 * by}[shape]
                mv16[mb, bx + 4 * by] = np.random.RandomState(1000 * mb + key).randint(-9, 10, 2)
        if shape == 0:
            ref8[mb] = ref8[mb, 0]
        elif shape == 1:
            ref8[mb, 1] = ref8[mb, 0]; ref8[mb, 3] = ref8[mb, 2]
        elif shape == 2:
            ref8[mb, 2] = ref8[mb, 0]; ref8[mb, 3] = ref8[mb, 1]
    mv16[r.rand(n) < 0.3] = 0
    pskip = r.randint(-3, 4, (n, 2)).astype(np.int16)
    pskip[r.rand(n) < 0.4] = 0
    pskip[:3] = (-400, 300)
    return g, mv16, ref8, pskip


def encode_cases(lib):
    """Golden outputs of the reference's x264_macroblock_encode and x264_macroblock_probe_skip.
    Source = synthetic frame 6; references = frames 6 and 4 with the reference's own borders and
    half-pel planes.  Frames are regenerated from x264_vs2008_amd/synth.py by the test, so only the
    side information and the outputs are stored."""
    from x264_vs2008_amd.frame import chroma_qp
    vp = hostpic.vp
    lib.refshim_lambda2.restype = C.c_int
    for size, qp, t8, field in ENC_CASES:
        g, mv16, ref8, pskip = encode_inputs(size, qp)
        n = g.mb_w * g.mb_h
        pics = []
        for t in (6, 6, 4):
            hp = hostpic.HostPic(g)
            y, u, v = synth.frame(size[0], size[1], t)
            for nm, img in (("y", y), ("u", u), ("v", v)):
                hp.set_visible(nm, img)
            f = ref_frame(hp)
            lib.refshim_source_prepare(C.byref(f), 0)
            pics.append(hp)
        cur, refs = pics[0], pics[1:]
        for hp in refs:
            f = ref_frame(hp)
            lib.refshim_fdec_filter(C.byref(f), 0, None, None, None, None, None, None, 0, 0, 0, 1)
        rr = (RefRef * 2)()
        for i, hp in enumerate(refs):
            for k, nm in enumerate(("y", "h", "vv", "c")):
                rr[i].y[k] = hp.ptr(nm)
            rr[i].u = hp.ptr("u"); rr[i].v = hp.ptr("v")
        rec = hostpic.HostPic(g)
        ly = np.zeros((n, 256), np.int16); lc = np.zeros((n, 128), np.int16); dc = np.zeros((n, 8), np.int16)
        cbp = np.zeros(n, np.int32); nnz = np.zeros((n, 26), np.uint8)
        qpc = chroma_qp(qp)
        lib.refshim_inter_encode_frame(cur.ptr("y"), cur.ptr("u"), cur.ptr("v"), rr, 2, rec.ptr("y"), rec.ptr("u"), rec.ptr("v"),
                                       size[0], size[1], g.stride_y, g.stride_c, qp, qpc, t8, field, vp(mv16), vp(ref8),
                                       vp(ly), vp(lc), vp(dc), vp(cbp), vp(nnz))
        skip = np.zeros((2, n), np.uint8)
        for w in range(2):
            lib.refshim_probe_skip_frame(cur.ptr("y"), cur.ptr("u"), cur.ptr("v"), C.byref(rr[w]), size[0], size[1], g.stride_y,
                                         g.stride_c, qp, qpc, field, vp(pskip), vp(skip[w]))
        name = "encode_%dx%d_qp%d_t%d_f%d.npz" % (size[0], size[1], qp, t8, field)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name), mv16=mv16, ref8=ref8, pskip=pskip,
                            lambda2=np.array([lib.refshim_lambda2(qpc)]), levels_y=ly, levels_c=lc, dc_c=dc, cbp=cbp, nnz=nnz,
                            skip=skip, rec_y=rec.visible("y").copy(), rec_u=rec.visible("u").copy(), rec_v=rec.visible("v").copy())
        print("wrote", name, "coded MBs", int((cbp != 0).sum()), "of", n, "skippable", skip.sum(axis=1))


ME_CASES = [((352, 288), 1, 16, 7, 1, 26), ((352, 288), 0, 16, 5, 1, 30), ((200, 120), 1, 16, 2, 0, 22), ((200, 120), 1, 8, 1, 0, 36),
            ((352, 288), 1, 16, 3, 1, 40), ((200, 120), 0, 16, 0, 0, 26)]
ME_LAMBDA = (1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6,
             6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25, 29, 32, 36, 40, 45, 51, 57, 64, 72, 81, 91)


def me_inputs(size, seed, n_refs=3):
    """Seeded predictors: mvp per (mb, ref) and up to 5 extra candidates (zeros, duplicates of mvp,
    far-away vectors that the search has to clip)."""
    g = hostpic.Geometry(*size)
    n = g.mb_w * g.mb_h
    r = np.random.RandomState(seed)
    mvp = r.randint(-14, 15, (n, n_refs, 2)).astype(np.int16)
    mvp[r.rand(n) < 0.25] = 0
    mvp[r.rand(n) < 0.05] = (300, -260)
    mvc = r.randint(-40, 41, (n, n_refs, 8, 2)).astype(np.int16)
    mvc[:, :, 1] = 0
    mvc[:, :, 3] = mvp
    mvc[r.rand(n) < 0.1, :, 2] = (-700, 500)
    n_mvc = r.randint(0, 6, (n, n_refs)).astype(np.uint8)
    return g, mvp, mvc, n_mvc


def me_frames(lib_prepare, size):
    """cur = synthetic frame 7, refs = frames 6, 5, 4 made into references by `lib_prepare`."""
    g = hostpic.Geometry(*size)
    pics = []
    for t in (7, 6, 5, 4):
        hp = hostpic.HostPic(g)
        lib_prepare(hp, synth.frame(size[0], size[1], t), t != 7)
        pics.append(hp)
    return pics[0], pics[1:]


def me_cases(lib):
    """Golden outputs of the reference's x264_me_search_ref (16x16) over whole frames."""
    from x264_vs2008_amd.frame import cost_mv_table
    vp = hostpic.vp

    def prep(hp, yuv, is_ref):
        for nm, img in zip(("y", "u", "v"), yuv):
            hp.set_visible(nm, img)
        f = ref_frame(hp)
        lib.refshim_source_prepare(C.byref(f), 0)
        if is_ref:
            lib.refshim_fdec_filter(C.byref(f), 0, None, None, None, None, None, None, 0, 0, 0, 1)

    for size, method, me_range, subme, chroma_me, qp in ME_CASES:
        g, mvp, mvc, n_mvc = me_inputs(size, 100 * subme + qp)
        n = g.mb_w * g.mb_h
        cur, refs = me_frames(prep, size)
        rr = (RefRef * 3)()
        for i, hp in enumerate(refs):
            for k, nm in enumerate(("y", "h", "vv", "c")):
                rr[i].y[k] = hp.ptr(nm)
            rr[i].u = hp.ptr("u"); rr[i].v = hp.ptr("v")
        lam = ME_LAMBDA[qp]
        span = 4 * 2048
        tab = cost_mv_table(lam, span).view(np.int16)
        ref_cost = (lam * np.array([1, 3, 3])).astype(np.int32)
        out_mv = np.zeros((n, 3, 2), np.int16); out_cost = np.zeros((n, 3), np.int32); best = np.zeros((n, 4), np.int32)
        center = C.cast(tab.ctypes.data + 2 * span, C.c_void_p)
        lib.refshim_me_search16_frame(cur.ptr("y"), cur.ptr("u"), cur.ptr("v"), rr, 3, size[0], size[1], g.stride_y, g.stride_c,
                                      method, me_range, subme, chroma_me, 512, center, vp(mvp), vp(mvc), vp(n_mvc), vp(ref_cost),
                                      vp(out_mv), vp(out_cost), vp(best))
        name = "me16_%dx%d_m%d_r%d_s%d_c%d_qp%d.npz" % (size[0], size[1], method, me_range, subme, chroma_me, qp)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name), mvp=mvp, mvc=mvc, n_mvc=n_mvc, ref_cost=ref_cost,
                            out_mv=out_mv, out_cost=out_cost, best=best)
        print("wrote", name, "best-ref histogram", np.bincount(best[:, 0], minlength=3), "fractional mvs",
              int((out_mv % 4 != 0).any(axis=2).sum()))


def main():
    lib = hostpic.load_lazy(os.path.join(HERE, "_ref", "libx264ref.so"))
    vp = hostpic.vp
    me_cases(lib)
    encode_cases(lib)
    for size, seed, a_off, b_off, c_off, qlo, qhi in CASES:
        g, inp = make_case(size, seed, qlo, qhi)
        out = {}
        # source side: mod-16 edge replication + lowres planes
        src = hostpic.HostPic(g)
        for nm in ("y", "u", "v"):
            src.set_visible(nm, inp[nm])
        f = ref_frame(src)
        lib.refshim_source_prepare(C.byref(f), 1)
        for nm in ("y", "u", "v", "l0", "lh", "lv", "lc"):
            out["src." + nm] = src.arr(nm).copy()
        # reconstruction side: deblock + border + half-pel planes, row by row as the reference does
        rec = hostpic.HostPic(g)
        for nm in ("y", "u", "v"):
            rec.set_visible(nm, inp[nm])
        f = ref_frame(rec)
        lib.refshim_source_prepare(C.byref(f), 0)          # coded area beyond the visible picture
        lib.refshim_fdec_filter(C.byref(f), 1, vp(inp["mb_type"]), vp(inp["qp"]), vp(inp["nnz"]), vp(inp["t8"]), vp(inp["mv"]),
                                vp(inp["ref"]), a_off, b_off, c_off, 1)
        for nm in ("y", "u", "v", "h", "vv", "c"):
            out["rec." + nm] = rec.arr(nm).copy()
        changed = int((rec.visible("y")[:size[1], :size[0]] != inp["y"]).sum())
        name = "frame_drivers_%dx%d.npz" % size
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name), offsets=np.array([a_off, b_off, c_off]),
                            **{"in." + k: v for k, v in inp.items()}, **{"out." + k: v for k, v in out.items()})
        print("wrote", name, "deblock changed", changed, "luma pixels")


if __name__ == "__main__":
    main()
