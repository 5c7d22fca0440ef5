"""Developer tool (runs on the GPU box): per-dispatch counter sums of the sweep kernel from a rocprofv3 rocpd database."""
import collections
import json
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
acc = collections.defaultdict(dict)
for did, cn, val, dur in c.execute("select dispatch_id, counter_name, sum(value), max(duration) from counters_collection "
                                   "where kernel_name like '%k_slice_sweep%' group by dispatch_id, counter_name"):
    acc[did][cn] = val
    acc[did]["duration_ms"] = dur / 1e6
print(json.dumps([dict(dispatch=d, **acc[d]) for d in sorted(acc)], indent=1))
