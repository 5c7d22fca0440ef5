#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/sensitivity2.txt
: > $out
run() { name="$1"; shift; timeout -k 10 300 python bench.py --no-cpu --steps 6 --warmup 2 "$@" > gpurun_out/sens_tmp.json 2> gpurun_out/sens_tmp.err || { echo "$name FAILED $(tail -c 300 gpurun_out/sens_tmp.err)" >> $out; return 1; }
  python3 -c "
import json,sys; d=json.loads(open('gpurun_out/sens_tmp.json').read().strip().splitlines()[-1]); print('%-40s %8.1f fps  %8.1f ms/step  sweep %8.1f ms  %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['slice_types_in_timed_steps']))" "$name" >> $out; }
run "P partitions off, I4x4/I8x8 in P on" --inter 0x103 &&
run "P partitions on, I4x4/I8x8 in P off" --inter 0x110 &&
run "p8x8 off? (inter 0x103) + refs 1" --inter 0x103 --refs 1 &&
run "b8x8 off" --inter 0x13
cat $out
