#!/usr/bin/env python3
"""Per-kernel averages of the counters of a rocprofv3 `--pmc ... --output-format csv` run: tools/pmc_by_kernel.py <dir> [name filter]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    sys.exit("no *counter_collection.csv under " + sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if flt in r["Kernel_Name"]:
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k[:110])
    for c, v in cs.items():
        v = v[len(v) // 4:]  # skip warm-up launches
        print(f"   {c:32s} n={len(v):3d} avg={sum(v) / len(v):.4e}")
