set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/test.log 2>&1 || { tail -30 $O/test.log; exit 1; }
tail -3 $O/test.log
BENCH_LAUNCHES=1 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python3 $R/bench.py --no-cpu > $O/bench_trace.json 2> $O/trace.err || { tail -20 $O/trace.err; exit 1; }
DB=$(find $O/trace -name "*_results.db" | head -1)
python3 $R/profiles/make_summaries.py trace $DB $O/bench_trace.json $O/kernel_stats.csv $O/launches.json
rm -rf $O/trace
