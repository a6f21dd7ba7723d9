#!/usr/bin/env python3
"""Launches of the LAST scoring pass in a rocprofv3 kernel trace of tools/prof_mgfn_eval.py (a pass = from one add-free amp_combine_fwd to the next)."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "amp_combine_fwd" in r["Kernel_Name"]]
seg = rows[starts[-2] - 1 : starts[-1] - 1]
tot = 0.0
for r in seg:
    n = r["Kernel_Name"]
    m = re.search(r"advhip::(\w+)(<[^>]*>)?", n)
    name = (m.group(1) + (m.group(2) or "")) if m else n[:50]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    tot += us
    print(f"{us:8.1f} us  {wg:6d} wg  {name[:90]}")
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
print(f"sum {tot:.0f} us in {len(seg)} launches; first start .. last end {span:.0f} us")
