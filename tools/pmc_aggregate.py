"""Aggregate rocprofv3 --pmc counter CSVs (<dir>/**/*counter_collection.csv) per kernel and counter.
Usage: python tools/pmc_aggregate.py <dir> [COUNTER] [--prefix pmc_C2_]   (no counter: every counter found, per kernel: calls, total, per call;
--prefix: only the pass directories <dir>/<prefix>* -- one workload's passes when several workloads share <dir>)"""
import csv, glob, re, sys
from collections import defaultdict

args = sys.argv[1:]
prefix = ""
if "--prefix" in args:
    k = args.index("--prefix"); prefix = args[k + 1]; args = args[:k] + args[k + 2:]
d = args[0]
only = args[1] if len(args) > 1 else None
acc = defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob(f"{d}/{prefix}*/**/*counter_collection.csv", recursive=True) if prefix else glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)):
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
