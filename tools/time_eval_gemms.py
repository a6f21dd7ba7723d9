#!/usr/bin/env python3
"""Dev tool: the stage-2 GEMMs of a whole-video scoring pass (N = 10 crops x T clips positions) over tile / K-slice choices.
    python tools/time_eval_gemms.py [T=290]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import _lib, mgfn_ops  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 290
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
B = 10


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        s.record()
        graph.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps * 1e3)
    return best


orig = mgfn_ops._desc
D = _lib.ALGO_DMA2_BASE
for cin, cout, k in ((1024, 4096, 1), (4096, 1024, 1), (1024, 1024, 1), (1024, 1024, 3)):
    x = torch.randn((cin, B, T), device=dev, generator=g)
    w = torch.randn((cout, cin, k), device=dev, generator=g) * (cin * k) ** -0.5
    wp = mgfn_ops.pack_kc(w)
    d0 = orig(cin, cout, k, 1 if k == 1 else B, B * T if k == 1 else T, 0)
    flop = 2.0 * cin * k * cout * B * T
    t0 = timed(lambda: mgfn_ops.conv_cn(x, wp, cout, k))
    print(f"{cin}->{cout} k={k} N={B*T}: default algo {d0.algo} splits {d0.splits}: {t0:6.1f} us {flop/t0/1e6:6.1f} TF", flush=True)
    for algo in (D + 1, D + 2, D + 3, D + 4):
        for splits in (1, 2, 3, 4):
            def patched(ci, co, kk, b, t, act, _a=algo, _s=splits):
                d = orig(ci, co, kk, b, t, act)
                d.algo, d.splits = _a, _s
                return d
            mgfn_ops._desc = patched
            try:
                t = timed(lambda: mgfn_ops.conv_cn(x, wp, cout, k))
                print(f"    algo {algo} splits {splits}: {t:6.1f} us {flop/t/1e6:6.1f} TF", flush=True)
            except Exception as ex:  # noqa: BLE001
                print(f"    algo {algo} splits {splits}: {str(ex)[:90]}")
            finally:
                mgfn_ops._desc = orig
