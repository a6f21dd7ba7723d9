#!/usr/bin/env python3
"""Dev tool: is the in-step GEMM rate (118-120 TFLOP/s) the chip's sustained rate or an artefact of the step?  The stage-2 MGFN
GEMM shapes on the 128x64 LDS-DMA tile: a 10-launch burst vs 1.5 s back to back, on one operand set vs rotating over 8
operand sets (nothing re-used from the Infinity Cache), and torch.matmul (rocBLAS) under the same protocol."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import ops  # noqa: E402


def rate(fns, fl, seconds=None, reps=10):
    for f in fns:
        f()
    torch.cuda.synchronize()
    if seconds is None:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(reps):
            fns[i % len(fns)]()
        e.record()
        torch.cuda.synchronize()
        return fl * reps / s.elapsed_time(e)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for i in range(50):
            fns[(n + i) % len(fns)]()
        n += 50
        torch.cuda.synchronize()
    return fl * n / ((time.perf_counter() - t0) * 1e3)


def main():
    dev = torch.device("cuda:0")
    N = 10240
    for o, c in ((4096, 1024), (1024, 4096), (1024, 1024)):
        fl = 2.0 * o * c * N / 1e9
        W = torch.randn(o, c, 1, 1, 1, device=dev)
        one, zero = torch.ones(o, device=dev), torch.zeros(o, device=dev)
        pc = ops.pack_conv(W, one, zero, zero, one, 0.0, (1, 1, 1), (0, 0, 0), name="g")
        Xs = [torch.randn(1, c, 1, 1, N, device=dev) for _ in range(8)]
        Ys = [torch.empty(1, o, 1, 1, N, device=dev) for _ in range(8)]
        hip = [lambda X=X, Y=Y: ops.conv3d_bn_act(X, pc, relu=False, algo=162, splits=1, out=Y) for X, Y in zip(Xs, Ys)]
        hip128 = [lambda X=X, Y=Y: ops.conv3d_bn_act(X, pc, relu=False, algo=161, splits=1, out=Y) for X, Y in zip(Xs, Ys)]
        blas = [lambda X=X, Y=Y: torch.matmul(W.view(o, c), X.view(c, N), out=Y.view(o, N)) for X, Y in zip(Xs, Ys)]
        dY = [torch.randn(o, N, device=dev) for _ in range(8)]
        nt = [lambda A=A, X=X: ops.gemm_nt(A, X.view(c, N)) for A, X in zip(dY, Xs)]
        print(f"o={o} c={c} ({fl:.0f} GFLOP):", flush=True)
        for name, fns in (("hip 128x64", hip), ("hip 128x128", hip128), ("rocBLAS", blas), ("gemm_nt dW", nt)):
            print(f"  {name:11s} burst/1set {rate(fns[:1], fl):6.1f}  burst/8sets {rate(fns, fl, reps=16):6.1f}  "
                  f"1.5s/1set {rate(fns[:1], fl, 1.5):6.1f}  1.5s/8sets {rate(fns, fl, 1.5):6.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
