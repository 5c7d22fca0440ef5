"""CPU: the C-ABI library loads and exports every symbol include/*.h declares
(no compute, no GPU), host-side helpers, and the N > 1 frame-sharding path on
two gloo ranks."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(path):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)
    names = re.findall(r"\b(x264(?:hip)?_\w+)\s*\(", src)
    return sorted(set(n for n in names if not n.endswith("_t")))


def test_library_exports_every_declared_symbol():
    from x264_vs2008_amd import lib as L
    if not os.path.exists(L.SO_PATH):
        L.build()
    lib = L.open_library()
    for header in ("x264hip.h", "x264hip_lookahead.h", "x264hip_stream.h"):
        declared = _declared_functions(os.path.join(ROOT, "include", header))
        assert len(declared) > (30 if header == "x264hip.h" else 8)
        missing = [n for n in declared if not hasattr(lib, n)]
        assert not missing, "declared in include/%s but not exported: %s" % (header, missing)


def test_lookahead_struct_sizes_match_header():
    """The ctypes mirrors of the round-3 structures (x264_vs2008_amd/lookahead.py, stream.py) against the C headers."""
    from x264_vs2008_amd import lookahead as LA
    from x264_vs2008_amd import stream as ST
    from x264_vs2008_amd import slice as SL
    from x264_vs2008_amd import mux as MX
    prog = r'''
#include <stdio.h>
#include "x264hip_lookahead.h"
#include "x264hip_stream.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(x264hip_lookahead_params), sizeof(x264hip_look_need), sizeof(x264hip_look_frame),
 sizeof(x264hip_look_slot), sizeof(x264hip_look_task), sizeof(x264hip_look_params), sizeof(x264hip_chain_sweep), sizeof(x264hip_cavlc_params),
 sizeof(x264hip_encoder_params), sizeof(x264hip_slice_header), sizeof(x264hip_frame_stat)); return 0; }
'''
    exe = os.path.join(ROOT, "tests", "_sizes2.bin")
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=prog.encode(), check=True)
    try:
        got = [int(v) for v in subprocess.check_output([exe]).split()]
    finally:
        os.remove(exe)
    want = [C.sizeof(LA.LookaheadParams), C.sizeof(LA.Need), C.sizeof(LA.Frame), C.sizeof(LA.LookSlot), C.sizeof(LA.LookTask), C.sizeof(LA.LookParams),
            C.sizeof(ST.ChainSweep), C.sizeof(SL.CavlcParams), C.sizeof(MX.EncoderParams), C.sizeof(MX.SliceHeader), 32]
    assert got == want, (got, want)


def test_init_fails_loudly_without_gpu_or_inits_with_one():
    """No CPU fallback: without a device x264hip_init returns < 0 with a message and the
    table fillers refuse to fill; with a device it returns 0."""
    from x264_vs2008_amd import lib as L
    from x264_vs2008_amd.tables import PixelTable
    lib = L.open_library()
    n = lib.x264hip_device_count()
    cfg = L.Cfg(0, 0)
    rc = lib.x264hip_init(C.byref(cfg))
    if n == 0:
        assert rc < 0 and lib.x264hip_last_error()
        t = PixelTable()
        assert lib.x264_pixel_init_hip(C.byref(t)) == -1
        assert not any(bool(f) for f in t.sad)
    else:
        assert rc == 0


def test_table_struct_sizes_match_header():
    """ctypes mirrors vs the C header: compile a tiny program printing sizeof() of each table."""
    from x264_vs2008_amd import tables as T
    prog = r'''
#include <stdio.h>
#include "x264hip.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(x264hip_pixel_function_t), sizeof(x264hip_dct_function_t),
 sizeof(x264hip_zigzag_function_t), sizeof(x264hip_quant_function_t), sizeof(x264hip_mc_functions_t),
 sizeof(x264hip_deblock_function_t), sizeof(x264hip_run_level_t), sizeof(x264hip_picture), sizeof(x264hip_me_params)); return 0; }
'''
    exe = os.path.join(ROOT, "tests", "_sizes.bin")
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=prog.encode(), check=True)
    try:
        got = [int(v) for v in subprocess.check_output([exe]).split()]
    finally:
        os.remove(exe)
    from x264_vs2008_amd.frame import MeParams, Picture
    want = [C.sizeof(T.PixelTable), C.sizeof(T.DctTable), C.sizeof(T.ZigzagTable), C.sizeof(T.QuantTable), C.sizeof(T.McTable),
            C.sizeof(T.DeblockTable), C.sizeof(T.RunLevel), C.sizeof(Picture), C.sizeof(MeParams)]
    assert got == want


def test_cost_mv_table_properties():
    from x264_vs2008_amd.frame import cost_mv_table
    t = cost_mv_table(4, 64).astype(np.int64)
    assert t[64] == int(4 * 0.718 + 0.5) and np.array_equal(t, t[::-1]) and (np.diff(t[64:]) >= 0).all()


def test_gop_sharding_covers_every_frame_once():
    from x264_vs2008_amd import shard
    for n, k, w in ((100, 12, 4), (24, 250, 8), (97, 10, 3), (8, 1, 8)):
        seen = sorted(f for r in range(w) for f in shard.frames_for_rank(n, k, r, w))
        assert seen == list(range(n))
        for r in range(w):
            for a, b in shard.gops_for_rank(n, k, r, w):
                assert a % k == 0 and b - a <= k


WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from x264_vs2008_amd import shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
mine = shard.frames_for_rank(1000, 24, rank, world)
mask = torch.zeros(1000, dtype=torch.int32); mask[mine] = 1
dist.all_reduce(mask)                                   # every frame owned exactly once across ranks
t = torch.tensor([0.5 + rank], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)   # bench.py's timing reduce
dist.barrier()
assert int(mask.min()) == 1 and int(mask.max()) == 1 and float(t[0]) == world - 0.5
print("rank", rank, "ok", len(mine))
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_cost_mv_table_built_in_c_matches_twin_and_numpy(oracle_lib):
    """p_cost_mv feeds every motion-vector decision and is float arithmetic in the reference (R/encoder/analyse.c:182-198).  The
    product builds it in the library's host C (x264hip_cost_mv_table, the reference's expression, -ffp-contract=off); it must equal
    the twin's C table (oracle/slice_oracle.c:s_load_cost_mv -- the twin is pinned to the reference's decisions on thousands of
    chains) and the NumPy restatement in frame.py, for every QP over the full +-2*4*2048 span."""
    import ctypes as C
    import numpy as np
    from x264_vs2008_amd import lib as L
    from x264_vs2008_amd.frame import cost_mv_table
    from x264_vs2008_amd.slice import COST_SPAN, LAMBDA_TAB
    lib = L.open_library()
    oracle_lib.x264o_cost_mv_row.restype = C.c_void_p
    for qp in range(52):
        got = np.zeros(2 * COST_SPAN + 1, np.int16)
        lib.x264hip_cost_mv_table(C.c_int(LAMBDA_TAB[qp]), C.c_int(COST_SPAN), got.ctypes.data_as(C.c_void_p))
        twin = np.ctypeslib.as_array((C.c_int16 * (2 * COST_SPAN + 1)).from_address(oracle_lib.x264o_cost_mv_row(qp)))
        assert np.array_equal(got, twin), qp
        assert np.array_equal(got.view(np.uint16), cost_mv_table(LAMBDA_TAB[qp], COST_SPAN)), qp


def test_nal_encode_matches_reference():
    """x264hip_nal_encode (host C of the library) against the reference's own x264_nal_encode (oracle/_ref, built here from the
    reference's sources) on payloads full of the byte patterns emulation prevention exists for; where the reference library is
    absent, against the rule's statement in ITU-T H.264 7.4.1 (a 0x03 before any byte <= 3 that follows two zeros)."""
    import ctypes as C
    import os
    from x264_vs2008_amd import lib as L

    lib = L.open_library()
    lib.x264hip_nal_encode.restype = C.c_int
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libx264ref.so")
    ref = None
    if os.path.exists(ref_so):
        from oracle import refslice as rs
        ref = rs.reference_lib()

    class Nal(C.Structure):                      # x264_nal_t, R/x264.h:395-403
        _fields_ = [("i_ref_idc", C.c_int), ("i_type", C.c_int), ("i_payload", C.c_int), ("p_payload", C.c_void_p)]

    r = np.random.default_rng(5)
    for trial in range(60):
        n = int(r.integers(0, 400))
        pay = r.choice(np.array([0, 0, 0, 1, 2, 3, 4, 255], np.uint8), n) if trial % 2 else r.integers(0, 256, n).astype(np.uint8)
        pay = np.ascontiguousarray(pay)
        ref_idc, typ, annexb = int(r.integers(0, 4)), int(r.choice([1, 5, 6, 7, 8])), int(r.integers(0, 2))
        got = np.zeros(5 + n * 3 // 2 + 8, np.uint8)
        m = lib.x264hip_nal_encode(got.ctypes.data_as(C.c_void_p), annexb, ref_idc, typ, pay.ctypes.data_as(C.c_void_p), n)
        if ref is not None:
            want = np.zeros_like(got)
            size = C.c_int(0)
            nal = Nal(ref_idc, typ, n, pay.ctypes.data)
            ref.x264_nal_encode(want.ctypes.data_as(C.c_void_p), C.byref(size), annexb, C.byref(nal))
            assert m == size.value and np.array_equal(got[:m], want[:m]), trial
        out, zeros = ([0, 0, 0, 1] if annexb else []) + [(ref_idc << 5) | typ], 0
        for v in pay.tolist():
            if zeros == 2 and v <= 3:
                out.append(3); zeros = 0
            zeros = zeros + 1 if v == 0 else 0
            out.append(v)
        assert got[:m].tolist() == out, trial


from x264_vs2008_amd.frame import JVT4I, JVT4P, JVT8I, JVT8P          # the standard's default scaling lists (--cqm jvt)


def test_cqm_init_in_library_matches_reference_tables(oracle_lib):
    """x264hip_cqm_init (the library's host C restatement of x264_cqm_init, R/common/set.c:68-168) against the tables the REFERENCE's
    x264_cqm_init produced (tests/golden/cqm_flat.npz, cqm_jvt.npz: oracle/gen_golden_cqm.py), flat and JVT matrices, every QP; the
    unquant tables against the library's older x264hip_unquant_table and the twin's x264o_cqm_unquant."""
    import ctypes as C
    from x264_vs2008_amd import lib as L
    from x264_vs2008_amd.frame import cqm_init
    lib = L.open_library()
    for name, lists, qp_min in (("cqm_flat", None, 0), ("cqm_jvt", [JVT4I, JVT4P, JVT4I, JVT4P, JVT8I, JVT8P], 6)):
        t = cqm_init(lib, lists, qp_min=qp_min)
        with np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")) as z:
            for k in z.files:
                assert np.array_equal(t[k], z[k]), (name, k)
        preset = int(lists is not None)
        for cat in range(4):
            for qp in range(52):
                unq = np.zeros(16, np.int32)
                oracle_lib.x264o_cqm_unquant(preset, cat, qp, 0, unq.ctypes.data_as(C.c_void_p))
                assert np.array_equal(unq, t["unquant4_mf"][cat, qp]), (name, cat, qp)
        for cat in range(2):
            for qp in range(52):
                unq = np.zeros(64, np.int32)
                oracle_lib.x264o_cqm_unquant(preset, cat, qp, 1, unq.ctypes.data_as(C.c_void_p))
                assert np.array_equal(unq, t["unquant8_mf"][cat, qp]), (name, cat, qp)
    with pytest.raises(RuntimeError, match="overflow"):            # the JVT matrices below QP 6: "Quantization overflow", set.c:160-166
        cqm_init(lib, [JVT4I, JVT4P, JVT4I, JVT4P, JVT8I, JVT8P], qp_min=0)


def test_generated_cabac_tables_are_one_file():
    """oracle/cabac_tables.h (twin) and csrc/cabac_tables.h (product) are both written by oracle/gen_cabac_tables.py: the data must not drift."""
    import re
    def body(path):
        with open(path) as f:
            return re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S).split()
    a, b = body(os.path.join(ROOT, "oracle", "cabac_tables.h")), body(os.path.join(ROOT, "x264_vs2008_amd", "csrc", "cabac_tables.h"))
    strip = lambda toks: [t for t in toks if t not in ("static", "__device__", "const", "__constant__")]
    ints = lambda toks: re.findall(r"-?\d+", " ".join(toks))
    assert ints(strip(a)) == ints(strip(b))


def test_encode_clip_refuses_what_it_cannot_shard():
    from x264_vs2008_amd import shard
    with pytest.raises(ValueError):
        shard.encode_clip(None, None, [(np.zeros((16, 16), np.uint8),) * 3], 4, lanes=1)
