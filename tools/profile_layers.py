#!/usr/bin/env python3
"""Dev tool: per-conv timing of the I3D plan at a given batch, every algorithm variant.

    python tools/profile_layers.py [--batch 32] [--reps 5] [--algos 0,1,2,3,4]
Prints one line per conv: shape, MMAC, and for each algo the time and TFLOP/s.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import ops  # noqa: E402
from anomaly_detection_on_video_amd.i3d import I3Res50  # noqa: E402
from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict  # noqa: E402


def time_fn(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--algos", default="0,1,2,3,4")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    algos = [int(a) for a in args.algos.split(",")]
    dev = torch.device("cuda:0")
    m = I3Res50()
    m.load_state_dict(synth_i3d_state_dict())
    m = m.eval().to(dev)
    m.prepare()
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((args.batch, 3, 16, 224, 224), device=dev, generator=g)
    rows = []
    total = {a: 0.0 for a in algos}
    best_total = 0.0
    tot_flop = 0.0

    def bench_conv(pc, xin, relu, res):
        nonlocal best_total, tot_flop
        y = ops.conv3d_bn_act(xin, pc, relu=relu, residual=res)
        macs = y.numel() * pc.cin * pc.kernel[0] * pc.kernel[1] * pc.kernel[2]
        line = {"name": pc.name, "in": list(xin.shape), "out": list(y.shape), "k": pc.kernel, "s": pc.stride, "mmac": macs / 1e6, "t": {}}
        best = None
        for a in algos:
            if a in (1, 4) and pc.cout % 128:
                continue
            t = time_fn(lambda: ops.conv3d_bn_act(xin, pc, relu=relu, residual=res, algo=a, out=y), args.reps)
            line["t"][a] = t
            total[a] += t
            if a != 0 and (best is None or t < best[1]):
                best = (a, t)
        if best:
            best_total += best[1]
            line["best"] = best[0]
        tot_flop += 2 * macs
        rows.append(line)
        s = " ".join(f"a{a}:{t:7.3f}ms/{2*macs/t/1e9:6.1f}TF" for a, t in line["t"].items())
        print(f"{pc.name:22s} {str(tuple(xin.shape)):28s}->{pc.cout:5d} k{pc.kernel} s{pc.stride} {macs/1e6:9.1f}MMAC {s}", flush=True)
        return y

    cur = x
    for u in m._plan:
        if u.kind == "stem":
            cur = bench_conv(u.convs[0], cur, True, None)
        elif u.kind == "maxpool":
            t = time_fn(lambda: ops.maxpool3d(cur, u.kernel, u.stride), args.reps)
            nxt = ops.maxpool3d(cur, u.kernel, u.stride)
            gb = (cur.numel() + nxt.numel()) * 4 / 1e9
            print(f"maxpool {tuple(cur.shape)} -> {tuple(nxt.shape)}: {t:.3f} ms  {gb/t*1e3:.0f} GB/s")
            for a in total:
                total[a] += t
            best_total += t
            cur = nxt
        elif u.kind == "avgpool":
            t = time_fn(lambda: ops.global_avgpool(cur), args.reps)
            print(f"avgpool: {t:.3f} ms")
            cur = ops.global_avgpool(cur)
        else:
            c1, c2, c3, ds = u.convs
            o1 = bench_conv(c1, cur, True, None)
            o2 = bench_conv(c2, o1, True, None)
            r = bench_conv(ds, cur, False, None) if ds is not None else cur
            cur = bench_conv(c3, o2, True, r)
    for a, t in total.items():
        print(f"algo {a}: total {t:.2f} ms  -> {args.batch / t * 1e3:.0f} clips/s, {tot_flop / t / 1e9:.1f} TFLOP/s")
    print(f"best-per-layer total {best_total:.2f} ms -> {args.batch / best_total * 1e3:.0f} clips/s, {tot_flop / best_total / 1e9:.1f} TFLOP/s")
    t = time_fn(lambda: m(x), args.reps)
    print(f"end-to-end forward: {t:.2f} ms -> {args.batch / t * 1e3:.0f} clips/s ({tot_flop / t / 1e9:.1f} TFLOP/s)")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f)


if __name__ == "__main__":
    main()
