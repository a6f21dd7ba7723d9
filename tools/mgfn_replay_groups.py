#!/usr/bin/env python3
"""Kernel groups of one replayed MGFN training step from a rocprofv3 kernel trace of tools/prof_mgfn_train.py: a replay = the
launches from one `pack_multi_kernel` to the next; averages over the last N replays.
    python tools/mgfn_replay_groups.py DIR [N=10]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "pack_multi_kernel" in r["Kernel_Name"]]
segs = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])][-last:]


def group(n):
    if "advhip::" in n:
        return n.split("advhip::")[1].split("<")[0].split("(")[0]
    if "adam_multi_kernel" in n:
        return "adam_multi_kernel"
    if "fused_dropout" in n:
        return "torch fused_dropout"
    if "FillFunctor" in n:
        return "torch fill"
    if n.startswith("Cijk"):
        return "rocBLAS"
    if "multi_tensor" in n:
        return "torch multi_tensor"
    if "rocclr_copyBuffer" in n:
        return "copyBuffer"
    return "torch " + ("elementwise" if "elementwise" in n else "reduce" if "reduce" in n else "cat" if "CatArray" in n else "other")


g = collections.defaultdict(lambda: [0, 0.0])
for seg in segs:
    for r in seg:
        k = group(r["Kernel_Name"])
        g[k][0] += 1
        g[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
n = len(segs)
span = sum((int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"])) / 1e6 for s in segs) / n
print(f"{n} replays; first start .. last end of a replay: {span:.3f} ms (profiled)")
print("| group | launches / step | ms / step (profiled) |\n|---|---:|---:|")
for k, (c, t) in sorted(g.items(), key=lambda kv: -kv[1][1]):
    print(f"| `{k}` | {c / n:.1f} | {t / n:.3f} |")
print(f"| total | {sum(c for c, _ in g.values()) / n:.1f} | {sum(t for _, t in g.values()) / n:.3f} |")
