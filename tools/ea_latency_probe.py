#!/usr/bin/env python3
"""Where are the conv stack's HBM-side reads served from?  FETCH_SIZE counts every L2 miss, whether the memory-side Infinity
Cache (256 MB) or the DRAM answers it, and the TCC's *_DRAM request counters count both alike (they sit on the L2's side of
that cache).  What does differ is the time a miss takes: TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ = average cycles a read request
is in flight between the L2 and the fabric.  This script gives that figure two calibration points and then the convs:

  1. `sum()` over a 3-GB buffer, three times             -- every miss goes to the DRAM
  2. `sum()` over a 96-MB buffer, 24 times back to back  -- beyond the L2s (8 x 4 MB), inside the Infinity Cache after pass 1
  3. one I3D forward at B = 32 on one stream             -- the plan's 52 conv launches in order

    rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d DIR -- python3 tools/ea_latency_probe.py
    python tools/ea_latency_probe.py --report DIR            (anywhere: reads the csv)
"""
import collections
import csv
import glob
import os
import sys


def run():
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from anomaly_detection_on_video_amd.i3d import I3Res50
    from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_input

    dev = torch.device("cuda:0")
    os.environ.setdefault("ADV_I3D_STREAMS", "1")
    m = I3Res50()
    m.load_state_dict(synth_i3d_state_dict())
    m = m.eval().to(dev)
    x = synth_input((32, 3, 16, 224, 224)).to(dev)
    with torch.no_grad():
        for _ in range(2):
            m(x)
    torch.cuda.synchronize()
    big = torch.ones((768 << 20,), device=dev)       # 3 GB
    small = torch.ones((24 << 20,), device=dev)      # 96 MB
    marker = torch.zeros((64,), device=dev)
    torch.cuda.synchronize()
    marker.cos_()                                    # (markers: the report splits the dispatch list at these)
    for _ in range(3):
        big.sum()
    marker.cos_()
    for _ in range(24):
        small.sum()
    marker.cos_()
    del big
    with torch.no_grad():
        m(x)
    torch.cuda.synchronize()
    marker.cos_()
    torch.cuda.synchronize()


def report(d):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    by = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        e = by.setdefault(k, {"name": r["Kernel_Name"], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = [by[k] for k in sorted(by)]
    marks = [i for i, r in enumerate(rows) if "cos" in r["name"]]
    assert len(marks) >= 4, len(marks)
    a, b, c, e = marks[-4:]

    def line(label, rs):
        req = sum(r.get("TCC_EA0_RDREQ_sum", 0.0) for r in rs)
        lvl = sum(r.get("TCC_EA0_RDREQ_LEVEL_sum", 0.0) for r in rs)
        r32 = sum(r.get("TCC_EA0_RDREQ_32B_sum", 0.0) for r in rs)
        ns = sum(r["ns"] for r in rs)
        if hit_pass:
            hit, miss = (sum(r.get(k, 0.0) for r in rs) for k in ("TCC_HIT_sum", "TCC_MISS_sum"))
            return f"| {label} | {len(rs)} | {ns / 1e3:.0f} | {hit / 1e6:.2f} | {miss / 1e6:.2f} | {100 * hit / max(hit + miss, 1):.1f} % |"
        # (gfx950: wide reads are 128-byte requests tallied as 64-byte ones, the same doubling FETCH_SIZE needs)
        mb = 2 * ((req - r32) * 64 / 1e6 + r32 * 32 / 1e6)
        return f"| {label} | {len(rs)} | {mb:.0f} | {ns / 1e3:.0f} | {mb / max(ns, 1) * 1e3:.2f} | {lvl / max(req, 1):.0f} |"

    hit_pass = any("TCC_HIT_sum" in r for r in rows)
    if hit_pass:
        print("| what | launches | us | L2 hits (M requests) | L2 misses (M) | L2 hit rate |\n|---|---:|---:|---:|---:|---:|")
    else:
        print("| what | launches | EA read MB | us | TB/s at the EA | avg cycles a read is in flight (LEVEL / RDREQ) |\n|---|---:|---:|---:|---:|---:|")
    print(line("sum over 3 GB (DRAM)", rows[a + 1 : b]))
    smalls = rows[b + 1 : c]
    print(line("sum over 96 MB, pass 1", smalls[:1]))
    print(line("sum over 96 MB, passes 2-24 (Infinity Cache)", smalls[1:]))
    convs = [r for r in rows[c + 1 : e] if "conv3d_igemm" in r["name"] or "splitk_reduce" in r["name"]]
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from traffic_by_layer import layers
    L = layers()
    assert len(convs) == len(L), (len(convs), len(L))
    for l, r in zip(L, convs):
        print(line(f"{l['name']} `{r['name'].split('<')[1].split('>')[0]}` (compulsory read {l['read'] / 1e6:.0f} MB)", [r]))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        report(sys.argv[2])
    else:
        run()
