#!/usr/bin/env python3
"""In the last replay of a traced MGFN training step: time with 1 / 2+ kernels in flight, and the launches that overlap a gemm_kk kernel."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "pack_multi_kernel" in r["Kernel_Name"]]
seg = rows[starts[-2] : starts[-1]]
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in seg]
t0, t1 = iv[0][0], max(e for _s, e, _n in iv)
ev = sorted([(s, 1) for s, e, n in iv] + [(e, -1) for s, e, n in iv])
lvl, prev, acc = 0, t0, {}
for t, d in ev:
    acc[min(lvl, 2)] = acc.get(min(lvl, 2), 0) + t - prev
    prev, lvl = t, lvl + d
print(f"replay span {(t1 - t0) / 1e6:.3f} ms; sum of durations {sum(e - s for s, e, _ in iv) / 1e6:.3f} ms; in flight 0/1/2+: " + " / ".join(f"{acc.get(k, 0) / 1e6:.3f}" for k in (0, 1, 2)) + " ms")
gk = [(s, e) for s, e, n in iv if "gemm_kk_dma_kernel" in n]
ov = [(s, e, n) for s, e, n in iv if "gemm_kk" not in n and any(s < ge and e > gs for gs, ge in gk)]
print(f"{len(ov)} launches overlap a gemm_kk_dma_kernel launch; their durations sum to {sum(e - s for s, e, _ in ov) / 1e3:.0f} us")
