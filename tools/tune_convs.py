#!/usr/bin/env python3
"""Measure every (tile algo, split-K) variant for each conv of the I3D plan at a given batch and
write the winners to anomaly_detection_on_video_amd/tuned/gfx950.json (merged with existing keys).

    python tools/tune_convs.py --batch 32 [--reps 5] [--out path.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import _lib, ops  # noqa: E402
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
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                  "anomaly_detection_on_video_amd", "tuned", "gfx950.json"))
    ap.add_argument("--report", default="")
    ap.add_argument("--quick", action="store_true", help="skip the generic-gather variants")
    ap.add_argument("--algos", default="", help="comma-separated algo ids: time only these (e.g. 162,169 for a tile A/B)")
    ap.add_argument("--max-splits", type=int, default=16)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    os.environ["ADV_NO_TUNED"] = "1"
    m = I3Res50()
    m.load_state_dict(synth_i3d_state_dict())
    m = m.eval().to(dev)
    m.prepare()
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((args.batch, 3, 16, 224, 224), device=dev, generator=g)
    table = {}
    if os.path.exists(args.out):
        with open(args.out) as f:
            table = json.load(f)
    report = []
    total_best = 0.0
    tot_flop = 0.0
    seen = {}

    def tune(pc, xin, relu, res):
        nonlocal total_best, tot_flop
        B, _, T, H, W = xin.shape
        key = pc.key(B, T, H, W)
        y = ops.conv3d_bn_act(xin, pc, relu=relu, residual=res, algo=3, splits=1)
        macs = y.numel() * pc.cin * pc.kernel[0] * pc.kernel[1] * pc.kernel[2]
        tot_flop += 2 * macs
        if key in seen:
            best = seen[key]
        else:
            kpad = pc.w_packed.shape[0]
            cands = []
            taps = pc.kernel[0] * pc.kernel[1] * pc.kernel[2]
            pool = ((3,) if args.quick else _lib.IGEMM_ALGOS) + ((35,) if args.quick else _lib.FAST_ALGOS) + _lib.DMA_ALGOS + _lib.DMA4_ALGOS + _lib.DMA2_ALGOS
            if args.algos:
                pool = tuple(int(a) for a in args.algos.split(","))
            for algo in pool:
                bm, bn, bk = _lib.algo_tile(algo)
                if pc.cout % bn:
                    continue
                tiles = -(-(y.numel() // pc.cout) // bm) * (pc.cout // bn)
                for sp in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16):
                    if sp > args.max_splits:
                        continue
                    if sp > 1 and (tiles * sp > 4096 or kpad // bk < 2 * sp):
                        continue
                    cands.append((algo, sp))
            res_t = []
            for algo, sp in cands:
                try:
                    t = time_fn(lambda: ops.conv3d_bn_act(xin, pc, relu=relu, residual=res, algo=algo, splits=sp, out=y), args.reps)
                except Exception as e:  # pragma: no cover
                    print("skip", pc.name, algo, sp, e)
                    continue
                res_t.append((t, algo, sp))
            res_t.sort()
            best = res_t[0]
            seen[key] = best
            table[key] = [best[1], best[2]]
            top = " ".join(f"a{a}s{s}:{t:.3f}" for t, a, s in res_t[:5])
            print(f"{pc.name:22s} {key:55s} {macs/1e6:9.1f}MMAC best {2*macs/best[0]/1e9:6.1f}TF | {top}", flush=True)
            report.append({"name": pc.name, "key": key, "mmac": macs / 1e6, "ranked": res_t[:8]})
        total_best += best[0]
        return ops.conv3d_bn_act(xin, pc, relu=relu, residual=res, algo=best[1], splits=best[2])

    cur = x
    for u in m._plan:
        if u.kind == "stem":
            cur = tune(u.convs[0], cur, True, None)
        elif u.kind == "maxpool":
            cur = ops.maxpool3d(cur, u.kernel, u.stride)
        elif u.kind == "avgpool":
            cur = ops.global_avgpool(cur)
        else:
            c1, c2, c3, ds = u.convs
            o1 = tune(c1, cur, True, None)
            o2 = tune(c2, o1, True, None)
            if u.cat:  # conv3 + downsample folded into one conv over [x ; h]
                cur = tune(c3, torch.cat([cur, o2], dim=1), True, None)
                continue
            r = tune(ds, cur, False, None) if ds is not None else cur
            cur = tune(c3, o2, True, r)
    print(f"sum of best conv times {total_best:.2f} ms -> {args.batch / total_best * 1e3:.0f} clips/s, {tot_flop / total_best / 1e9:.1f} TFLOP/s")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    if args.report:
        with open(args.report, "w") as f:
            json.dump(report, f)


if __name__ == "__main__":
    main()
