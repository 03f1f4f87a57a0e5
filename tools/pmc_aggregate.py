"""Aggregate rocprofv3 --pmc counter CSVs (gpurun_out/pmc_*/**/*counter_collection.csv) per kernel.
Usage: python tools/pmc_aggregate.py <dir> <COUNTER>"""
import csv, glob, re, sys
from collections import defaultdict

d, counter = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
            name = re.sub(r"\(.*", "", name).replace("void ", "").strip()
            a = acc[name]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
for name, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:40s} calls {n:5d}  {counter} total {v:14.1f}  per call {v / n:12.2f}")
