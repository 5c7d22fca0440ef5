set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=24
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python3 $R/bench.py --batch 2048 --drift 1 --steps 6 --warmup 2 --no-cpu > $O/bench_trace.json 2> $O/trace.err || { tail -20 $O/trace.err; exit 1; }
DB=$(find $O/trace -name "*_results.db" | head -1)
python3 - <<PY
import sqlite3
c = sqlite3.connect("$DB")
rows = list(c.execute("select start, end, name, grid_x, queue_id, stream_id from kernels where name like '%k_slice_sweep%' or name like '%k_look_cost%' order by start"))
t0 = rows[0][0]
for s, e, name, gx, q, st in rows:
    kind = 'LOOK' if 'look_cost' in name else ('B ' if 'ELb1ELb1ELb1ELb0ELb1EEv' in name or 'true, true, true, false, true' in name else 'IP')
    print("%8.1f %8.1f %7.1f %s waves %5d q %s st %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, kind, gx // 64, q, st))
PY
rm -rf $O/trace
