set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/sq -o sq -- python3 $R/bench.py --steps 5 --warmup 0 --no-cpu > $O/sq.json 2> $O/sq.err || { tail -30 $O/sq.err; exit 1; }
python3 $R/profiles/pmc_dump.py $(find $O/sq -name "*_results.db" | head -1) > $O/sq_counters.json
rm -rf $O/sq
rocprofv3 --pmc FETCH_SIZE -d $O/f -o f -- python3 $R/bench.py --steps 5 --warmup 0 --no-cpu > $O/f.json 2> $O/f.err || { tail -30 $O/f.err; exit 1; }
python3 $R/profiles/pmc_dump.py $(find $O/f -name "*_results.db" | head -1) > $O/f_counters.json
rm -rf $O/f
rocprofv3 --pmc WRITE_SIZE -d $O/w -o w -- python3 $R/bench.py --steps 5 --warmup 0 --no-cpu > $O/w.json 2> $O/w.err || { tail -30 $O/w.err; exit 1; }
python3 $R/profiles/pmc_dump.py $(find $O/w -name "*_results.db" | head -1) > $O/w_counters.json
rm -rf $O/w
echo done
