"""Aggregate rocprofv3 --pmc counter CSVs (<dir>/**/*counter_collection.csv) per kernel and counter.
Usage: python tools/pmc_aggregate.py <dir> [COUNTER]   (no counter: every counter found, per kernel: calls, total, per call)"""
import csv, glob, re, sys
from collections import defaultdict

d = sys.argv[1]
only = sys.argv[2] if len(sys.argv) > 2 else None
acc = defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            c = row.get("Counter_Name")
            if only and c != only:
                continue
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
            name = re.sub(r"\(.*", "", name).replace("void ", "").strip()
            a = acc[(name, c)]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
for (name, c), (n, v) in sorted(acc.items(), key=lambda kv: (kv[0][0] != "f_persist<8, 3, 2>", kv[0][0], kv[0][1])):
    print(f"{name:40s} {c:28s} dispatches {n:5d}  total {v:16.1f}  per dispatch {v / n:14.2f}")
