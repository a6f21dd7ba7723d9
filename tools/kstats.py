#!/usr/bin/env python3
"""Print the per-kernel table of a rocprofv3 `--kernel-trace --stats --output-format csv` output directory."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    sys.exit("no *kernel_stats.csv under " + sys.argv[1])
for r in list(csv.DictReader(open(f[0])))[: int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"]) / 1e3:10.1f} total_ms {float(r["TotalDurationNs"]) / 1e6:9.2f}')
