"""Developer tool: sum rocprofv3 --pmc counter CSVs per kernel.  usage: pmc_sum.py <dir>... [kernel-substring]"""
import csv
import glob
import json
import sys
from collections import defaultdict

dirs = [a for a in sys.argv[1:] if "/" in a or a.startswith("gpurun_out")]
sub = [a for a in sys.argv[1:] if a not in dirs]
sub = sub[0] if sub else "k_slice_sweep"
tot = defaultdict(float)
launches = set()
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if sub in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
                launches.add((f, row["Dispatch_Id"]))
print(json.dumps({"kernel": sub, "counters": dict(sorted(tot.items()))}, indent=1))
