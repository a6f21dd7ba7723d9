#!/usr/bin/env python3
"""Group the kernels of a rocprofv3 `--kernel-trace --stats --output-format csv` directory: launches and ms per step.
    python tools/kgroups.py DIR STEPS"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
g = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "advhip::" in n:
        k = n.split("advhip::")[1].split("<")[0].split("(")[0]
    elif n.startswith("Cijk"):
        k = "rocBLAS Cijk"
    elif "elementwise" in n:
        k = "torch elementwise"
    elif "reduce_kernel" in n:
        k = "torch reduce"
    elif "multi_tensor" in n:
        k = "torch multi_tensor (Adam)"
    elif "CatArray" in n:
        k = "torch cat"
    elif "Fill" in n or "fill" in n:
        k = "fill"
    else:
        k = n[:50]
    g[k][0] += int(r["Calls"])
    g[k][1] += float(r["TotalDurationNs"]) / 1e6
tot = sum(t for _, t in g.values())
for k, (c, t) in sorted(g.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{k:45s} calls/step {c / steps:7.1f}  ms/step {t / steps:7.3f}")
print(f"{'total':45s} calls/step {sum(c for c, _ in g.values()) / steps:7.1f}  ms/step {tot / steps:7.3f}")
