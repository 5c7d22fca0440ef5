"""Developer tool: turn the rocprofv3 outputs of the round-end profile run into the JSON summaries under profiles/.
usage: profile_json.py traffic <fetch_dir> <write_dir> <out.json> | launches <trace_dir> <bench.json> <out.json>"""
import csv
import glob
import json
import sys


def rows(d, pat):
    for f in sorted(glob.glob(d + "/**/*" + pat, recursive=True)):
        yield from csv.DictReader(open(f))


def per_dispatch(d, counter):
    out = {}
    for r in rows(d, "counter_collection.csv"):
        if "k_slice_sweep" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            out[int(r["Dispatch_Id"])] = out.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return [out[k] for k in sorted(out)]


if sys.argv[1] == "traffic":
    fetch, write = per_dispatch(sys.argv[2], "FETCH_SIZE"), per_dispatch(sys.argv[3], "WRITE_SIZE")
    batch, mbs = 240, 240 * 8160
    names = [("I", 0), ("P", 1), ("P", 2), ("P", 3)]
    launches = []
    for (st, nr), f, w in zip(names, fetch, write):
        fb, wb = int(f * 1024), int(w * 1024)          # the counters are reported in KB
        launches.append({"slice": st, "refs": nr, "fetch_bytes": fb, "write_bytes": wb,
                         "fetch_bytes_per_macroblock": fb // mbs, "write_bytes_per_macroblock": wb // mbs})
    json.dump({"_comment": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (counters only, no trace domains), over "
               "`python3 bench.py --steps 3 --warmup 1 --no-cpu` (default options, batch 240, 3 waves/SIMD) on MI355X, round 1, final build "
               "(no scratch memory).  Four k_slice_sweep launches per pass: one I frame, then P frames with 1, 2 and 3 references.  Values are "
               "the counters as reported (KB) converted to bytes: they tally the L2's memory-side requests, so a lane's 1..4-byte store or a "
               "20-byte row read counts as whole 32/64-byte requests; raw values, said to be raw.",
               "batch": batch, "macroblocks_per_launch": mbs, "launches": launches}, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(launches, indent=1))
else:
    ms = []
    for r in rows(sys.argv[2], "kernel_trace.csv"):
        if "k_slice_sweep" in r["Kernel_Name"]:
            ms.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    ms = [round(v, 3) for _, v in sorted(ms)]
    b = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    w = b["warmup"]
    timed = ms[w:w + b["steps"]]
    json.dump({"what": "k_slice_sweep launches of `rocprofv3 --kernel-trace --stats -- python3 bench.py` (default options, batch 240) in launch "
               "order, ms: the warm-up steps, then the timed steps (P with up to 3 references, one I at the keyint).  bench.py's "
               "roofline.avg_launch_ms is the mean of the timed ones, measured with HIP events in the same run.",
               "launch_ms": ms, "mean_all_ms": round(sum(ms) / len(ms), 3), "mean_timed_ms": round(sum(timed) / len(timed), 3),
               "bench_json_avg_launch_ms": b["roofline"]["avg_launch_ms"], "bench_json_value": b["value"]}, open(sys.argv[4], "w"), indent=1)
    print("launches", len(ms), "mean timed", sum(timed) / len(timed), "bench", b["roofline"]["avg_launch_ms"], b["value"])
