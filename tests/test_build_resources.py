"""The built library's kernels, as the code objects describe themselves (no GPU needed).

The macroblock sweep must not use private (scratch) memory: at 12 waves per CU it is neither L1- nor L2-resident, and a single
dynamically indexed struct member or a conditional pointer to a local silently moves whole structs there (581 ->
775 frames/s when that was removed).  This test reads .private_segment_fixed_size from the AMDGPU metadata notes of every kernel
in libx264hip.so."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_metadata(tmp_path):
    so = os.path.join(ROOT, "x264_vs2008_amd", "libx264hip.so")
    if not os.path.exists(so):
        import sys
        sys.path.insert(0, ROOT)
        from x264_vs2008_amd import lib as L
        L.build()                                        # hipcc cross-compiles without a GPU
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump")
    shutil.copy(so, tmp_path / "lib.so")
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp_path, check=True, capture_output=True)
    out = {}
    for f in sorted(os.listdir(tmp_path)):
        if not f.endswith("gfx950"):
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if name:
                out[name.group(1)] = {k: int(v) for k, v in re.findall(r"\.(private_segment_fixed_size|vgpr_count|vgpr_spill_count|group_segment_fixed_size):\s+(\d+)", blk)}
    return out

RASTER = re.compile(r"ILi\dELb[01]ELb1ELb[01]ELb[01]ELb[01]ELb[01]EEv")      # k_slice_sweep<WPE, LL, RD = true, BS, TD, RF, CH>
REFINE = re.compile(r"ILi\dELb[01]ELb1ELb[01]ELb[01]ELb1ELb[01]EEv")         # ... with the RD refinement of subme 8-9 (RF = true)
TABLE = re.compile(r"ILi\dELb[01]ELb1ELb[01]ELb[01]ELb[01]ELb1EEv")           # ... launched from a chain table (CH = true)


def test_sweep_kernels_use_no_scratch_memory(tmp_path):
    md = kernel_metadata(tmp_path)
    # the wavefront-schedule variants <1>, <2>, <3> and the lossless one; the raster-order variant (third template argument true:
    # RASTER) and its B-slice instantiation are checked separately below
    sweeps = {k: v for k, v in md.items() if "k_slice_sweep" in k and not RASTER.search(k)}
    assert len(sweeps) >= 4, sorted(md)
    for k, v in sweeps.items():
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0, (k, v)
    default = [v for k, v in sweeps.items() if "ILi3ELb0" in k][0]
    assert default["vgpr_count"] <= 168                  # 3 waves per SIMD
    assert 12 * default["group_segment_fixed_size"] <= 160 * 1024      # 12 waves per CU fit in LDS


def test_raster_sweep_resources_are_bounded(tmp_path):
    """The raster-order variant (RD levels, trellis, the entropy coder in the loop): no private memory either.  (While its encoder
    had two call sites the compiler kept it as a function and 1.8 KB per lane of shared variables in scratch: rocprofv3 counted 64 KB
    of HBM writes per macroblock, profiles/r02_raster_traffic.json; one call site -> inlined -> registers.)"""
    md = kernel_metadata(tmp_path)
    rd = [v for k, v in md.items() if "k_slice_sweep" in k and RASTER.search(k) and not REFINE.search(k) and not TABLE.search(k)]
    assert len(rd) == 3                                  # I / P, B with spatial and B with temporal direct prediction
    # the chain-table launches (x264hip_slice_sweep_chains: I / P, I / P with the refinement, the extended B kernel) read their arguments
    # from a table entry through the constant address space; they must stay as free of private memory as the kernels they mirror
    tb = {k: v for k, v in md.items() if "k_slice_sweep" in k and TABLE.search(k)}
    assert len(tb) == 3, sorted(tb)
    for k, v in tb.items():
        lim = 512 if REFINE.search(k) else 68
        assert v["private_segment_fixed_size"] <= lim and v["group_segment_fixed_size"] <= 24 * 1024, (k, v)
    # the refinement variant (subme 8-9) is a kernel of its own, so that what it spills (its candidate generator's state on top of the
    # analysis records: a few hundred bytes per lane) costs the default kernels above nothing
    rf = [v for k, v in md.items() if "k_slice_sweep" in k and REFINE.search(k) and not TABLE.search(k)]
    assert len(rf) == 1 and rf[0]["private_segment_fixed_size"] <= 512 and rf[0]["group_segment_fixed_size"] <= 24 * 1024, rf
    for v in rd:
        # (the B instantiation reserves a small frame -- at most a handful of spilled registers, 12 bytes per lane at the time of
        # writing, outside the macroblock loop's hot paths; anything larger would be real private arrays again)
        # (round 3: --direct auto's second prediction in the extended B kernel moved a few more values through its 20-byte frame in the lock-step
        # instantiation -- 46 spill instructions, all in the direct / skip step at the head of a macroblock; the chain-table instantiation the streams
        # use has 2)
        assert v["private_segment_fixed_size"] <= 68 and v["vgpr_spill_count"] <= 64, v
        assert v["group_segment_fixed_size"] <= 24 * 1024, v
    assert min(v["private_segment_fixed_size"] for v in rd) == 0


def test_no_kernel_spills_registers(tmp_path):
    md = kernel_metadata(tmp_path)
    assert len(md) > 20
    md = {k: v for k, v in md.items() if not ("k_slice_sweep" in k and RASTER.search(k))}
    bad = {k: v for k, v in md.items() if v.get("vgpr_spill_count", 0) or v.get("private_segment_fixed_size", 0) > 64}
    assert not bad, bad
