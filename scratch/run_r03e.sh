set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c1-400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python3 $R/bench.py --no-cpu > $O/bench_trace.json 2> $O/trace.err || { tail -20 $O/trace.err; exit 1; }
DB=$(find $O/trace -name "*_results.db" | head -1)
python3 $R/profiles/make_summaries.py stream $DB $O/bench_trace.json $O/kernel_stats.csv $O/launches.json
rm -rf $O/trace
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/sq -o sq -- python3 $R/bench.py --steps 5 --warmup 0 --no-cpu > $O/sq.json 2> $O/sq.err || { tail -30 $O/sq.err; exit 1; }
python3 $R/profiles/pmc_dump.py $(find $O/sq -name "*_results.db" | head -1) "%k_%" > $O/sq_counters.json
rm -rf $O/sq
rocprofv3 --pmc FETCH_SIZE -d $O/f -o f -- python3 $R/bench.py --steps 5 --warmup 0 --no-cpu > $O/f.json 2> $O/f.err || { tail -30 $O/f.err; exit 1; }
python3 $R/profiles/pmc_dump.py $(find $O/f -name "*_results.db" | head -1) "%k_%" > $O/f_counters.json
rm -rf $O/f
rocprofv3 --pmc WRITE_SIZE -d $O/w -o w -- python3 $R/bench.py --steps 5 --warmup 0 --no-cpu > $O/w.json 2> $O/w.err || { tail -30 $O/w.err; exit 1; }
python3 $R/profiles/pmc_dump.py $(find $O/w -name "*_results.db" | head -1) "%k_%" > $O/w_counters.json
rm -rf $O/w
echo done
