"""Developer tool (runs on the GPU box): per-dispatch counter sums of the sweep kernel from a rocprofv3 rocpd database."""
import collections
import json
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
acc = collections.defaultdict(dict)
pat = sys.argv[2] if len(sys.argv) > 2 else "%k_slice_sweep%"
for did, cn, val, dur, name, gx in c.execute("select dispatch_id, counter_name, sum(value), max(duration), max(kernel_name), max(grid_size_x) from counters_collection "
                                             "where kernel_name like ? group by dispatch_id, counter_name", (pat,)):
    acc[did][cn] = val
    acc[did]["duration_ms"] = dur / 1e6
    # which instantiation: ...ELb<BS>ELb<TD>ELb<RF>ELb<CH>EEv -- B kernels have BS = 1
    short = name.split("(")[0].replace("void ", "")
    is_b = "ELb1ELb1ELb1ELb0ELb1EEv" in name or "<2, false, true, true, " in name           # k_slice_sweep<WPE, LL, RD, BS = true, ...>
    acc[did]["kernel"] = short if "k_slice_sweep" not in name else ("k_slice_sweep B" if is_b else "k_slice_sweep I/P") + (" (chain table)" if "true>" in short or name.rstrip().endswith("SwDesc") else "")
    acc[did]["name"] = short
    acc[did]["waves"] = gx // 64 if gx else None
print(json.dumps([dict(dispatch=d, **acc[d]) for d in sorted(acc)], indent=1))
