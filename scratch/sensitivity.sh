#!/bin/bash
# Developer aid: what each option of the MED flag set costs at the system level (2048 streams, stream mode): bench.py --no-cpu with one option changed.
cd "$(dirname "$0")/.."
out=gpurun_out/sensitivity.txt
: > $out
run() { name="$1"; shift; timeout -k 10 300 python bench.py --no-cpu --steps 6 --warmup 2 "$@" > gpurun_out/sens_tmp.json 2> gpurun_out/sens_tmp.err || { echo "$name FAILED $(tail -c 300 gpurun_out/sens_tmp.err)" >> $out; return 1; }
  python3 -c "
import json,sys; d=json.loads(open('gpurun_out/sens_tmp.json').read().strip().splitlines()[-1]); print('%-28s %8.1f fps  %8.1f ms/step  sweep %8.1f ms  %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['slice_types_in_timed_steps']))" "$name" >> $out; }
run default &&
run "trellis 0" --trellis 0 &&
run "refs 1" --refs 1 &&
run "refs 2" --refs 2 &&
run "subme 6" --subme 6 &&
run "subme 5 (no RD)" --subme 5 &&
run "psy-rd 0" --psy-rd 0 &&
run "mixed-refs 0" --mixed-refs 0 &&
run "no 8x8dct" --dct8 0 --inter 0x111 --intra 0x1 &&
run "no partitions" --inter 0x100 &&
run "bframes 0" --bframes 0 &&
run "me dia" --me 0 &&
run "me umh" --me 2
cat $out
