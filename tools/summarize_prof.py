#!/usr/bin/env python3
"""Summarise a rocprofv3 run of bench.py into profiles/: per-kernel table of the advhip kernels,
conv-stack time per forward from the kernel trace (to compare with bench.py's HIP-event figure),
and optionally FETCH_SIZE / WRITE_SIZE PMC passes.

    python tools/summarize_prof.py --trace gpurun_out/prof_x/runc --bench gpurun_out/prof_x.log \
        [--fetch DIR --write DIR] [--mfma DIR] --out profiles/r01_final
"""
import argparse
import collections
import csv
import glob
import json
import os


def load(pattern):
    f = glob.glob(pattern)
    return list(csv.DictReader(open(f[0]))) if f else []


def forward_end(name):
    """The last launch of a forward (or of one stream part): global_avgpool, or the conv whose epilogue takes the mean (EPI_AVG = 6)."""
    return "global_avgpool" in name or "conv3d_igemm_dma_kernel<128, 64, 16, false, 2, 6," in name


def family(name):
    if "conv3d_igemm" in name or "splitk_reduce" in name or "split_w" in name:  # (split_w: the stem's column-parity planes pass)
        return "conv"
    if "maxpool" in name or "avgpool" in name or "stem_pool_merge" in name:
        return "pool"
    if "advhip::" in name:
        return "advhip-other"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", required=True)
    ap.add_argument("--bench", default="")
    ap.add_argument("--fetch", default="")
    ap.add_argument("--write", default="")
    ap.add_argument("--mfma", default="", help="PMC pass with SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA")
    ap.add_argument("--out", required=True)
    ap.add_argument("--parts", type=int, default=2, help="stream parts a forward is split into (global_avgpool / fused-mean launches per forward)")
    a = ap.parse_args()
    rows = load(os.path.join(a.trace, "*_kernel_trace.csv"))
    # dispatch ids follow launch order, and a forward's launches are issued back to back by one host thread, so a
    # forward (or one stream part of it) is a contiguous run of dispatch ids ending in its global_avgpool -- also when
    # consecutive steps run on different HIP streams and overlap in time
    rows.sort(key=lambda r: int(r.get("Dispatch_Id") or r["Start_Timestamp"]))
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    ends = [i for i, r in enumerate(rows) if forward_end(r["Kernel_Name"])][a.parts - 1 :: a.parts]
    fwd = []
    prev = -1
    for e in ends:
        seg = [r for r in rows[prev + 1 : e + 1] if family(r["Kernel_Name"]) in ("conv", "pool")]
        prev = e
        if not seg:
            continue
        conv = sum(dur(r) for r in seg if family(r["Kernel_Name"]) == "conv")
        pool = sum(dur(r) for r in seg if family(r["Kernel_Name"]) == "pool")
        t0, t1 = min(int(r["Start_Timestamp"]) for r in seg), max(int(r["End_Timestamp"]) for r in seg)
        fwd.append((conv / 1e6, pool / 1e6, (t1 - t0) / 1e6, sum(1 for r in seg if family(r["Kernel_Name"]) == "conv"), t0, t1))
    per = collections.OrderedDict()
    for r in rows:
        if "advhip::" not in r["Kernel_Name"]:
            continue
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d = per.setdefault(n, [0, 0])
        d[0] += 1
        d[1] += dur(r)
    bench = None
    if a.bench and os.path.exists(a.bench):
        for line in open(a.bench):
            if line.startswith('{"metric"'):
                bench = json.loads(line)
    bench_line = bench
    out = ["# rocprofv3 summary of `bench.py` (%s)\n" % os.path.basename(a.out)]
    if bench:
        out.append("bench line under the profiler: value %.1f clips/s, %.3f ms/step, conv stack (HIP events) %.3f ms -> %.1f TFLOP/s\n"
                   % (bench["value"], bench["ms_per_step"], bench["roofline"]["avg_ms_per_launch_set"], bench["roofline"]["achieved"]))
    timed = fwd[-(bench["steps"] if bench else len(fwd)):]
    if timed:
        c = sum(f[0] for f in timed) / len(timed)
        window = (max(f[5] for f in timed) - min(f[4] for f in timed)) / 1e6 / len(timed)
        out.append("kernel trace, timed forwards (%d): conv kernel durations sum to %.3f ms per forward (%d launches incl. split-K reduces), pools %.3f ms, "
                   "one forward's first-to-last span %.3f ms; first start .. last end over the timed forwards = %.3f ms per forward "
                   "(forwards overlap when steps alternate between stream lanes)\n"
                   % (len(timed), c, timed[-1][3], sum(f[1] for f in timed) / len(timed), sum(f[2] for f in timed) / len(timed), window))
        if bench:
            overlapped = "lanes" in bench["roofline"]["kernel"] or "split over" in bench["roofline"]["kernel"]
            ref = window if overlapped else c
            out.append("agreement trace (%s) vs HIP events: %.1f %%\n" % ("window per forward" if overlapped else "conv kernel sum", 100 * ref / bench["roofline"]["avg_ms_per_launch_set"]))
    out.append("\n| kernel | calls | total ms | avg us |\n|---|---:|---:|---:|\n")
    for n, (calls, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        out.append("| `%s` | %d | %.2f | %.1f |\n" % (n, calls, t / 1e6, t / calls / 1e3))
    if a.fetch and a.write:
        tot = {}
        for kind, d in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
            agg = collections.defaultdict(float)
            cnt = collections.defaultdict(int)
            for r in load(os.path.join(d, "*_counter_collection.csv")):
                fam = family(r["Kernel_Name"])
                agg[fam] += float(r["Counter_Value"])
                cnt[fam] += 1
            tot[kind] = (agg, cnt)
        nf = len([1 for r in load(os.path.join(a.fetch, "*_counter_collection.csv")) if forward_end(r["Kernel_Name"])]) // a.parts
        out.append("\n## HBM traffic (PMC, separate passes, %d forwards each)\n\n" % nf)
        out.append("| family | FETCH_SIZE KB/forward | x2 (gfx950 fetch correction) GB | WRITE_SIZE KB/forward | GB |\n|---|---:|---:|---:|---:|\n")
        res = {}
        for fam in ("conv", "pool"):
            f = tot["FETCH_SIZE"][0][fam] / max(nf, 1)
            w = tot["WRITE_SIZE"][0][fam] / max(nf, 1)
            out.append("| %s | %.0f | %.2f | %.0f | %.2f |\n" % (fam, f, 2 * f * 1024 / 1e9, w, w * 1024 / 1e9))
            res[fam] = {"fetch_kb": f, "write_kb": w, "bytes_corrected": 2 * f * 1024 + w * 1024}
        with open(a.out + "_traffic.json", "w") as fp:
            json.dump(res, fp, indent=1)
        # the copy bench.py reads for roofline.traffic: tagged with the commit and a hash of the kernel sources it was
        # profiled at, so a later bench line can say whether the number is stale (bench.py:measured_traffic)
        import subprocess
        import sys

        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        import bench

        try:
            commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], text=True).strip()
        except Exception:
            commit = None
        tagged = dict(res, batch=(bench_line or {}).get("config", {}).get("local_batch", 32), commit=commit,
                      kernels_sha16=bench.kernels_sha16(), profile=os.path.basename(a.out) + "_summary.md", forwards=nf)
        with open(os.path.join(root, bench.TRAFFIC_PROFILE), "w") as fp:
            json.dump(tagged, fp, indent=1)
    if a.mfma:
        # matrix-pipe utilisation from counters: SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all
        # 1024 SIMDs, GRBM_GUI_ACTIVE the active cycles of the 8 XCDs (MI355X_MICROARCH.md counters).
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in load(os.path.join(a.mfma, "*_counter_collection.csv")):
            if family(r["Kernel_Name"]) != "conv" or "splitk_reduce" in r["Kernel_Name"]:
                continue
            n = r["Kernel_Name"].split("(")[0].replace("void ", "")
            for key in (n, "all conv kernels"):
                agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    agg[key]["launches"] += 1
        out.append("\n## Matrix-pipe utilisation (PMC pass: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_INSTS_MFMA)\n\n")
        out.append("busy % = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); fp32 16x16x4 MFMA = 32 busy cycles each\n\n")
        out.append("| kernel | launches | MFMA instructions | MFMA busy cycles per SIMD | active cycles | MFMA pipe busy |\n|---|---:|---:|---:|---:|---:|\n")
        mf = {}
        for n, c in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"]):
            busy, act = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0, c["GRBM_GUI_ACTIVE"] / 8.0
            out.append("| `%s` | %d | %.4g | %.4g | %.4g | %.1f %% |\n" % (n, c["launches"], c["SQ_INSTS_MFMA"], busy, act, 100 * busy / max(act, 1)))
            mf[n] = {"launches": c["launches"], "mfma_insts": c["SQ_INSTS_MFMA"], "busy_per_simd": busy, "active": act, "busy_frac": busy / max(act, 1)}
        with open(a.out + "_mfma.json", "w") as fp:
            json.dump(mf, fp, indent=1)
    with open(a.out + "_summary.md", "w") as fp:
        fp.writelines(out)
    print("".join(out))


if __name__ == "__main__":
    main()
