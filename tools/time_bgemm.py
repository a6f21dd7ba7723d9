#!/usr/bin/env python3
"""Dev tool: advhip_bgemm_f32 vs torch.matmul (rocBLAS) on the MGFN stage-2 GEMM shapes, TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import ops  # noqa: E402


def t(fn, reps=10):
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
    dev = torch.device("cuda:0")
    N = 10240
    for o, c in ((4096, 1024), (1024, 4096), (1024, 1024), (1024, 3072)):
        W = torch.randn(o, c, device=dev)
        X = torch.randn(c, N, device=dev)
        dY = torch.randn(o, N, device=dev)
        fl = 2.0 * o * c * N / 1e9
        rows = [("fwd  W.X      ", lambda: ops.bgemm(W, X), lambda: torch.matmul(W, X)),
                ("dX   W^T.dY   ", lambda: ops.bgemm(W.t(), dY), lambda: torch.matmul(W.t(), dY)),
                ("dW   dY.X^T   ", lambda: ops.bgemm(dY, X.t()), lambda: torch.matmul(dY, X.t())),
                ("dW   gemm_nt  ", lambda: ops.gemm_nt(dY, X), lambda: torch.matmul(dY, X.t())),
                ("dW   gemm_nt/1", lambda: ops.gemm_nt(dY, X, 1), lambda: torch.matmul(dY, X.t()))]
        for name, a, b in rows:
            ta, tb = t(a), t(b)
            print(f"o={o} c={c} {name} advhip {ta:.3f} ms {fl / ta:.1f} TF | torch {tb:.3f} ms {fl / tb:.1f} TF")


if __name__ == "__main__":
    main()
