"""Developer tool: turn rocprofv3's rocpd databases (ROCm 7.2 writes *_results.db) into the summaries committed here.
usage: make_summaries.py trace <trace.db> <bench.json> <stats.csv> <launches.json>
       make_summaries.py traffic <fetch.db> <write.db> <batch> <out.json> <slice:refs> ..."""
import csv
import json
import sqlite3
import sys


def sweeps(db, what):
    c = sqlite3.connect(db)
    return list(c.execute(what))


if sys.argv[1] == "stream":
    # the stream mode: per step the I / P table kernel and the B table kernel side by side on two streams, and the lookahead's cost kernels
    db, bench, stats, out = sys.argv[2:6]
    c = sqlite3.connect(db)
    with open(stats, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            w.writerow([r[0], r[1], int(r[2] * 1000) if r[2] < 1e9 else int(r[2]), round(r[3] * 1000, 1), round(r[4], 4)])
    rows = list(c.execute("select start, end, name, grid_x from kernels where name like '%k_slice_sweep%' order by start"))
    steps = []
    for s_, e_, name, gx in rows:
        kind = "B" if "ELb1ELb1ELb1ELb0ELb1EEv" in name else "IP"
        if steps and s_ < steps[-1]["end"]:
            steps[-1]["end"] = max(steps[-1]["end"], e_)
        else:
            steps.append({"start": s_, "end": e_, "kernels": []})
        steps[-1]["kernels"].append({"kind": kind, "chains": gx // 64, "ms": round((e_ - s_) / 1e6, 3)})
    b = json.loads(open(bench).read().strip().splitlines()[-1])
    span = [round((s_["end"] - s_["start"]) / 1e6, 3) for s_ in steps]
    timed = span[b["warmup"]:b["warmup"] + b["steps"]]
    look = [round((e_ - s_) / 1e6, 3) for s_, e_ in c.execute("select start, end from kernels where name like '%k_look_cost%' order by start")]
    json.dump({"what": "k_slice_sweep<raster, chain table> launches of `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu` (default options, stream mode) in step order: "
                       "per step the I / P kernel and the B kernel run side by side on two streams; span_ms = first start to last end, which is what bench.py's HIP events "
                       "bracket (plus the table's upload).  Warm-up steps first, then the timed ones.",
               "steps": [{"span_ms": sp, "kernels": s_["kernels"]} for sp, s_ in zip(span, steps)],
               "mean_timed_span_ms": round(sum(timed) / max(len(timed), 1), 3), "bench_json_avg_launch_ms": b["roofline"]["avg_launch_ms"], "bench_json_value": b["value"],
               "k_look_cost_launches": len(look), "k_look_cost_total_ms": round(sum(look), 1), "k_look_cost_mean_ms": round(sum(look) / max(len(look), 1), 3)},
              open(out, "w"), indent=1)
    print("steps", len(steps), "mean timed span", sum(timed) / max(len(timed), 1), "bench", b["roofline"]["avg_launch_ms"], b["value"], "look", len(look), sum(look))
elif sys.argv[1] == "trace":
    db, bench, stats, out = sys.argv[2:6]
    c = sqlite3.connect(db)
    with open(stats, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            w.writerow([r[0], r[1], int(r[2] * 1000) if r[2] < 1e9 else int(r[2]), round(r[3] * 1000, 1), round(r[4], 4)])
    ms = [round((e - s) / 1e6, 3) for s, e in c.execute("select start, end from kernels where name like '%k_slice_sweep%' order by start")]
    res = list(c.execute("select distinct vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x from kernels "
                         "where name like '%k_slice_sweep%'"))
    b = json.loads(open(bench).read().strip().splitlines()[-1])
    timed = ms[b["warmup"]:b["warmup"] + b["steps"]]
    json.dump({"what": "k_slice_sweep<raster> launches of `rocprofv3 --kernel-trace --stats -- python3 bench.py` (default options) in launch order, "
                       "ms: the warm-up steps, then the timed steps.  bench.py's roofline.avg_launch_ms is the mean of the timed ones, measured "
                       "with HIP events in the same run.",
               "launch_ms": ms, "mean_all_ms": round(sum(ms) / len(ms), 3), "mean_timed_ms": round(sum(timed) / len(timed), 3),
               "bench_json_avg_launch_ms": b["roofline"]["avg_launch_ms"], "bench_json_value": b["value"],
               "dispatch": [dict(zip(["vgpr", "agpr", "sgpr", "lds_bytes", "scratch_bytes", "grid_x", "workgroup_x"], r)) for r in res]},
              open(out, "w"), indent=1)
    print("launches", len(ms), "mean timed", sum(timed) / len(timed), "bench", b["roofline"]["avg_launch_ms"], b["value"])
else:
    fdb, wdb, batch, out = sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    names = [a.split(":") for a in sys.argv[6:] if ":" in a]
    bframes = next((int(a[8:]) for a in sys.argv[6:] if a.startswith("bframes=")), 0)

    def per_dispatch(db, counter):
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
        rows = list(c.execute("select * from counters_collection"))
        acc = {}
        for r in rows:
            d = dict(zip(cols, r))
            if "k_slice_sweep" in str(d.get("kernel_name", d.get("name", ""))) and d.get("counter_name") == counter:
                acc[d["dispatch_id"]] = acc.get(d["dispatch_id"], 0.0) + float(d["value"])
        return [acc[k] for k in sorted(acc)]
    fetch, write = per_dispatch(fdb, "FETCH_SIZE"), per_dispatch(wdb, "WRITE_SIZE")
    mbs = batch * 8160
    launches = []
    for (st, nr), f, w in zip(names, fetch, write):
        fb, wb = int(f * 1024), int(w * 1024)           # the counters are reported in KB
        launches.append({"slice": st, "refs": int(nr), "fetch_bytes": fb, "write_bytes": wb,
                         "fetch_bytes_per_macroblock": fb // mbs, "write_bytes_per_macroblock": wb // mbs})
    json.dump({"_comment": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (counters only, no trace domains), over "
               "`python3 bench.py --steps N --warmup 0 --no-cpu` (default options: raster variant, subme 7, trellis 1, psy-rd 1.0, AQ, CABAC, 3 B frames) on "
               "MI355X, round 2.  One k_slice_sweep<raster> launch per frame in coding order; refs = list 0 + list 1 pictures of the launch.  Values are "
               "the counters as reported (KB) converted to bytes: they tally the L2's memory-side requests at request granularity; raw values.",
               "batch": batch, "bframes": bframes, "macroblocks_per_launch": mbs, "launches": launches}, open(out, "w"), indent=1)
    print(json.dumps(launches, indent=1))
