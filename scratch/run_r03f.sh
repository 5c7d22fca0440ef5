#!/bin/bash
mkdir -p gpurun_out
run() {  # name args...
  name=$1; shift
  timeout -k 10 420 python bench.py "$@" > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err || { echo "FAILED $name"; tail -c 700 gpurun_out/bench_$name.err; }
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_$name.json")); print("$name", d["value"], d["ms_per_step"], d["config"].get("parity_checked_frames"), d["config"]["slice_types_in_timed_steps"], d["config"].get("scheduler"), d["roofline"]["avg_launch_ms"])
except Exception as e: print("$name", "no json", e)
PY
}
run a64 --batch 64 --steps 4 --warmup 2
run a2048d1 --batch 2048 --drift 1 --steps 8 --warmup 2 --no-cpu
run a1792d2 --batch 1792 --drift 2 --steps 8 --warmup 2 --no-cpu
