#!/usr/bin/env python3
"""Every launch of one replayed MGFN training step, in order: kernel, grid (workgroups), duration -- and the share of launches
shorter than a threshold (the latency-bound narrow layers).   python tools/mgfn_replay_launches.py DIR [replay index from the end=1]"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "pack_multi_kernel" in r["Kernel_Name"]]
seg = rows[starts[-back - 1] : starts[-back]]
tot = 0.0
buckets = {}
for r in seg:
    n = r["Kernel_Name"]
    m = re.search(r"advhip::(\w+)(<[^>]*>)?", n)
    name = (m.group(1) + (m.group(2) or "")) if m else n[:50]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    tot += us
    print(f"{us:8.1f} us  {wg:6d} wg  {name[:90]}")
    b = "<15us" if us < 15 else "<30us" if us < 30 else "<100us" if us < 100 else ">=100us"
    buckets.setdefault(b, [0, 0.0])
    buckets[b][0] += 1
    buckets[b][1] += us
print(f"total {tot / 1e3:.3f} ms in {len(seg)} launches")
for b, (c, t) in buckets.items():
    print(f"  {b:8s} {c:4d} launches {t / 1e3:7.3f} ms")
