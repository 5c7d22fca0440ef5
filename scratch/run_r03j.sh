set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/gpu_full.log 2>&1 || { tail -30 $O/gpu_full.log; exit 1; }
tail -3 $O/gpu_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python bench.py --pcie 1 --no-cpu > $O/bench_pcie.json 2> $O/bench_pcie.err || { tail -20 $O/bench_pcie.err; exit 1; }
python bench.py --stream 0 > $O/bench_lockstep.json 2> $O/bench_lockstep.err || { tail -20 $O/bench_lockstep.err; exit 1; }
for f in bench bench_pcie bench_lockstep; do python - <<PY
import json
d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$f", d["value"], d["ms_per_step"], d["config"].get("parity_checked_frames"), d["config"].get("matches_baseline"), r["achieved"], r["frac"], r["traffic"], d["config"].get("pcie"))
PY
done
