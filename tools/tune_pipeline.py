#!/usr/bin/env python3
"""Second tuning pass at the level of the whole extract -> score stream.

`tools/tune_convs.py` ranks tile / split-K variants per conv launched alone.  The product stream keeps three steps in
flight on three HIP stream lanes, where a launch shares the chip with other launches: the variant that is fastest alone
(often a split-K one that fills 256 CUs by itself) is not always the one that costs the stream least.  This pass does
coordinate descent on the measured step time of the real stream: for every distinct conv shape, try the candidates
(the isolated ranking's top entries when --report is given, else a fixed list) and keep a change only if it lowers the
step time by more than the noise margin twice in a row.

    python tools/tune_pipeline.py [--batch 32] [--report tune_report.json] [--out tuned.json]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import _lib  # noqa: E402
from anomaly_detection_on_video_amd.i3d import I3Res50  # noqa: E402
from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection  # noqa: E402
from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream  # noqa: E402
from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_module_state_dict  # noqa: E402

FIXED = [(a, s) for a in (67, 68, 66, 65, 71, 72, 70, 99, 100, 98, 35) for s in (1, 2, 3)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--margin", type=float, default=0.002)
    ap.add_argument("--report", default="", help="tune_convs.py --report file: candidates = its ranked variants")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                  "anomaly_detection_on_video_amd", "tuned", "gfx950.json"))
    ap.add_argument("--sweeps", type=int, default=1)
    ap.add_argument("--candidates", default="", help="comma list of algo:splits overriding the candidate list, e.g. 133:1,134:1,134:2")
    ap.add_argument("--only-changes", action="store_true", help="write only the entries this run changed (an overlay table)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    bb = I3Res50()
    bb.load_state_dict(synth_i3d_state_dict())
    bb = bb.eval().to(dev)
    sc = MGFNForVideoAnomalyDetection(MGFNConfig())
    sc.load_state_dict(synth_module_state_dict(sc))
    sc = sc.eval().to(dev)
    stream = ExtractScoreStream(bb, sc, clips_per_video=32, ncrops=10, local_batch=args.batch)
    x = torch.randn((args.batch, 3, 16, 224, 224), device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    stream.score_video(torch.rand(32, 10, 2048, device=dev))

    def measure():
        best = 1e9
        for _ in range(2):
            for _ in range(4):
                stream.step_async(x)
            stream.drain()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                stream.step_async(x)
            stream.drain()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / args.steps * 1e3)
        return best

    measure()
    # distinct conv shapes of the plan at this batch, with the dims they run at
    dims = (16, 224, 224)
    groups = {}
    from anomaly_detection_on_video_amd import ops

    for u in bb._plan:
        if u.kind == "maxpool":
            dims = ops.conv_out_dims(dims, u.kernel, u.stride, (0, 0, 0))
        elif u.kind in ("stem", "bottleneck"):
            if u.kind == "bottleneck" and u.convs[3] is not None:
                groups.setdefault(u.convs[3].key(args.batch, *dims), []).append((u.convs[3], dims))
            for c in u.convs[:3]:
                groups.setdefault(c.key(args.batch, *dims), []).append((c, dims))
                dims = ops.conv_out_dims(dims, c.kernel, c.stride, c.padding)
    ranked = {}
    if args.report:
        for r in json.load(open(args.report)):
            ranked[r["key"]] = [(int(a), int(s)) for _t, a, s in r["ranked"]]

    def current(key):
        pc, d = groups[key][0]
        pc.desc(args.batch, *d, relu=True)  # resolves the tuned choice
        return pc.choices[(args.batch, *d)]

    def apply(key, choice):
        for pc, d in groups[key]:
            pc.choices[(args.batch, *d)] = choice

    base = measure()
    print(f"{len(groups)} distinct conv shapes, stream step {base:.4f} ms", flush=True)
    table = json.load(open(args.out)) if os.path.exists(args.out) and not args.only_changes else {}
    for sweep in range(args.sweeps):
        changed = 0
        for key in groups:
            pc, d = groups[key][0]
            cur = current(key)
            cands = [c for c in (ranked.get(key) or FIXED) if c != cur]
            if args.candidates:
                cands = [tuple(int(v) for v in c.split(":")) for c in args.candidates.split(",")]
            elif not ranked.get(key) and cur[1] > 1:
                cands += [(a, cur[1]) for a in (67, 68, 71, 99, 100) if (a, cur[1]) != cur]
            best_c, best_t = cur, base
            for a, s in cands:
                bm, bn, bk = _lib.algo_tile(a)
                kpad = pc.w_packed.shape[0]
                if pc.cout % bn or (s > 1 and kpad // bk < 2 * s):
                    continue
                apply(key, (a, s))
                try:
                    t = measure()
                except Exception as e:  # pragma: no cover
                    print("skip", key, a, s, e, flush=True)
                    continue
                if t < best_t * (1 - args.margin):
                    t2 = measure()  # confirm
                    if t2 < best_t * (1 - args.margin):
                        best_c, best_t = (a, s), max(t, t2)
            apply(key, best_c)
            if best_c != cur:
                changed += 1
                print(f"{pc.name:22s} {key:55s} a{cur[0]}s{cur[1]} -> a{best_c[0]}s{best_c[1]}: {base:.4f} -> {best_t:.4f} ms", flush=True)
                base = best_t
                table[key] = [best_c[0], best_c[1]]
            else:
                print(f"{pc.name:22s} {key:55s} keeps a{cur[0]}s{cur[1]}", flush=True)
        final = measure()
        print(f"sweep {sweep}: {changed} changes, stream step {final:.4f} ms -> {args.batch / final * 1e3:.0f} clips/s", flush=True)
        base = final
        if not changed:
            break
    with open(args.out, "w") as f:
        f.write("{\n")
        items = sorted(table.items())
        for i, (k, v) in enumerate(items):
            f.write(' "%s": [%d, %d]%s\n' % (k, v[0], v[1], "," if i + 1 < len(items) else ""))
        f.write("}\n")


if __name__ == "__main__":
    main()
