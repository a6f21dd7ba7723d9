#!/usr/bin/env python3
"""Dev tool: the stage-2 MGFN forward / dX GEMM shapes (N = 10 240 positions) on every 2-deep LDS-DMA tile, sustained over 8 rotating
operand sets (TFLOP/s)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import ops  # noqa: E402
from time_gemm_sustained import rate  # noqa: E402

dev = torch.device("cuda:0")
N = 10240
for o, c in ((4096, 1024), (1024, 4096), (1024, 1024), (1024, 3072)):
    fl = 2.0 * o * c * N / 1e9
    W = torch.randn(o, c, 1, 1, 1, device=dev)
    one, zero = torch.ones(o, device=dev), torch.zeros(o, device=dev)
    pc = ops.pack_conv(W, one, zero, zero, one, 0.0, (1, 1, 1), (0, 0, 0), name="g")
    Xs = [torch.randn(1, c, 1, 1, N, device=dev) for _ in range(8)]
    Ys = [torch.empty(1, o, 1, 1, N, device=dev) for _ in range(8)]
    out = []
    for algo in (161, 162, 163, 164, 166, 167, 168):
        for s in (1, 2):
            fns = [lambda X=X, Y=Y: ops.conv3d_bn_act(X, pc, relu=False, algo=algo, splits=s, out=Y) for X, Y in zip(Xs, Ys)]
            out.append(f"a{algo}s{s}:{rate(fns, fl, 0.4):.1f}")
    print(f"o={o} c={c} ({fl:.0f} GFLOP): " + " ".join(out), flush=True)
