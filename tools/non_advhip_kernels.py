#!/usr/bin/env python3
"""Which kernels of a rocprofv3 --kernel-trace run are NOT this package's: name, launches, total us.
    python tools/non_advhip_kernels.py gpurun_out/r5c/trace [--after-first-step]
With --after-first-step only launches after the first conv-stack kernel of the second forward are counted (operand builds and
warm-up done).  VERDICT r4 item 2: no at::native arithmetic and no rocblas_* kernel may remain on the scoring path."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if "--after-first-step" in sys.argv:
    ends = [i for i, r in enumerate(rows) if "conv3d_igemm_dma_kernel<128, 64, 16, false, 2, 6," in r["Kernel_Name"] or "global_avgpool" in r["Kernel_Name"]]
    rows = rows[ends[1] + 1 :] if len(ends) > 1 else rows
acc = {}
for r in rows:
    n = r["Kernel_Name"]
    if "advhip::" in n:
        continue
    a = acc.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"{len(rows)} launches, {tot / 1e3:.0f} us of kernel time in the window; non-advhip kernels:")
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{c:6d} {t / 1e3:10.1f} us  {n[:160]}")
