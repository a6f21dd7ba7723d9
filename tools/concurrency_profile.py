#!/usr/bin/env python3
"""Concurrency profile of a rocprofv3 --kernel-trace run of bench.py: over the steady part of the stream (the last `--steps` forwards),
the share of wall time with 0 / 1 / 2 / 3+ kernels in flight, and -- for the time with exactly ONE kernel in flight (nothing
overlaps it: its non-MFMA time is exposed) -- which kernels those are.
    python tools/concurrency_profile.py gpurun_out/r5c/trace [--steps 100]"""
import csv
import glob
import re
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 100
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
stems = [i for i, r in enumerate(rows) if "true, 2, 3, false, 0" in r["Kernel_Name"] or "stem" in r["Kernel_Name"] and "merge" not in r["Kernel_Name"]]
first = stems[-steps] if len(stems) > steps else stems[0]
last = stems[-1]
win = rows[first:last]
t0, t1 = int(win[0]["Start_Timestamp"]), int(rows[last]["Start_Timestamp"])
ev = []
for r in win:
    s, e = int(r["Start_Timestamp"]), min(int(r["End_Timestamp"]), t1)
    if e > s:
        ev.append((s, 1, r["Kernel_Name"]))
        ev.append((e, -1, r["Kernel_Name"]))
ev.sort(key=lambda x: (x[0], x[1]))


def short(n):
    m = re.search(r"(\w+)<([^>]*)>", n)
    return (m.group(1) + "<" + m.group(2)[:40] + ">") if m else n[:60]


level = defaultdict(int)
alone = defaultdict(int)
active = {}
prev = t0
for t, d, n in ev:
    dt = t - prev
    k = sum(active.values())
    level[min(k, 4)] += dt
    if k == 1:
        alone[short(next(a for a, c in active.items() if c > 0))] += dt
    prev = t
    active[n] = active.get(n, 0) + d
tot = t1 - t0
print(f"window {tot / 1e6:.2f} ms, {len(win)} launches, {len([1 for i in stems if first <= i < last])} forwards -> {tot / 1e3 / max(1, len([1 for i in stems if first <= i < last])):.1f} us per forward")
for k in sorted(level):
    print(f"  {k}{'+' if k == 4 else ' '} kernels in flight: {100 * level[k] / tot:5.1f} %")
print("alone (exactly one kernel in flight), by kernel:")
for n, t in sorted(alone.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {100 * t / tot:5.2f} %  {n}")
