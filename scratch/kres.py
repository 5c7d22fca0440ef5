"""Developer tool: registers, LDS and scratch of every k_slice_sweep instantiation in libx264hip.so (plus the dynamic LDS the launchers add)."""
import os, re, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
tmp = tempfile.mkdtemp()
shutil.copy(os.path.join(ROOT, "x264_vs2008_amd", "libx264hip.so"), os.path.join(tmp, "lib.so"))
subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
for f in sorted(os.listdir(tmp)):
    if "amdgcn" not in f:
        continue
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], cwd=tmp, check=True, capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count")[1:]:
        nm = re.search(r"\.name: +(\S+)", blk)
        if nm and "k_slice_sweep" in nm.group(1):
            g = lambda k: int(re.search(r"\." + k + r": +(\d+)", blk).group(1))
            print(nm.group(1)[:44], "vgpr", g("vgpr_count"), "sgpr", g("sgpr_count"), "static lds", g("group_segment_fixed_size"), "scratch", g("private_segment_fixed_size"),
                  "vgpr spills", g("vgpr_spill_count"), "sgpr spills", g("sgpr_spill_count"))
shutil.rmtree(tmp)
