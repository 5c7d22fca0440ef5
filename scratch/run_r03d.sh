#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for cfg in "2048 1" "1792 1"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --batch $1 --groups $2 --steps 6 --warmup 2 --no-cpu > gpurun_out/bench_g_$1_$2.json 2> gpurun_out/bench_g_$1_$2.err || { echo "FAILED $cfg"; tail -c 600 gpurun_out/bench_g_$1_$2.err; }
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_g_$1_$2.json")); print("$cfg", d["value"], d["ms_per_step"], d["config"]["slice_types_in_timed_steps"], d["roofline"]["avg_launch_ms"])
except Exception as e: print("$cfg", "no json", e)
PY
done
