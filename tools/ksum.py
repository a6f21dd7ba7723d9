#!/usr/bin/env python3
"""Sum of kernel durations (ms) and launch count in a rocprofv3 kernel trace directory, optionally per N steps."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e6
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1
print(f"{len(rows)} launches, kernel time {tot:.2f} ms total; per step (/{n:g}): {len(rows) / n:.0f} launches, {tot / n:.2f} ms")
