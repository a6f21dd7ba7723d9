#!/usr/bin/env python3
"""Dev tool: the narrow-stage GEMM shapes of one MGFN training step (N = 10 240 positions; stage 0: 64 channels, stage 1: 128) on the 64 x 64 and
128 x 64 tiles over K slices, device-timed through graph replays -- against the (algo, splits) mgfn_ops._desc picks today."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import mgfn_ops, ops  # noqa: E402

dev = torch.device("cuda:0")
N = 10240


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


# (Cout, Cin * k): every GEMM-shaped narrow launch of the step, forward and backward
SHAPES = [(64, 192), (192, 64), (64, 64), (256, 64), (64, 256), (128, 64), (128, 384), (128, 128), (512, 128), (128, 512), (1024, 128), (128, 1024), (64, 128)]
for o, c in SHAPES:
    W = torch.randn(o, c, 1, 1, 1, device=dev)
    one, zero = torch.ones(o, device=dev), torch.zeros(o, device=dev)
    pc = ops.pack_conv(W, one, zero, zero, one, 0.0, (1, 1, 1), (0, 0, 0), name="g")
    X = torch.randn(1, c, 1, 1, N, device=dev)
    Y = torch.empty(1, o, 1, 1, N, device=dev)
    R = torch.randn(1, o, 1, 1, N, device=dev)
    d = mgfn_ops._desc(c, o, 1, 1, N, 0)
    cur = timed(lambda: ops.conv3d_bn_act(X, pc, relu=False, residual=R, algo=d.algo, splits=d.splits, out=Y))
    out = []
    for algo in (163, 162):
        for s in (1, 2, 3, 4, 6, 8):
            if c // 16 < s:
                continue
            try:
                us = timed(lambda: ops.conv3d_bn_act(X, pc, relu=False, residual=R, algo=algo, splits=s, out=Y))
                out.append((us, f"a{algo}s{s}"))
            except Exception:
                pass
    out.sort()
    print(f"o={o:5d} k={c:5d}: today a{d.algo}s{d.splits} {cur:6.1f} us | " + " ".join(f"{n}:{u:.1f}" for u, n in out[:6]), flush=True)
