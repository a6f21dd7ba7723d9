#!/usr/bin/env python3
"""Dev tool: the LDS-DMA conv kernel run as a plain GEMM Y[o, n] = W[o, c] X[c, n] (1x1x1 conv on a (1, c, 1, 1, n) tensor)."""
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
        W = torch.randn(o, c, 1, 1, 1, device=dev)
        one, zero = torch.ones(o, device=dev), torch.zeros(o, device=dev)
        pc = ops.pack_conv(W, one, zero, zero, one, 0.0, (1, 1, 1), (0, 0, 0), name="g")
        X = torch.randn(1, c, 1, 1, N, device=dev)
        fl = 2.0 * o * c * N / 1e9
        out = []
        for algo in (67, 68, 66, 65, 163, 164, 162, 161, 71, 167, 168):
            try:
                ms = t(lambda: ops.conv3d_bn_act(X, pc, relu=False, algo=algo, splits=1))
                out.append(f"a{algo}:{fl / ms:.0f}")
            except Exception as ex:
                out.append(f"a{algo}:err")
        tb = t(lambda: torch.matmul(W.view(o, c), X.view(c, N)))
        print(f"o={o} c={c}: " + " ".join(out) + f" | torch {fl / tb:.0f} TF")
        # the same GEMMs on an x that is only 4-byte aligned: the 16-byte A pieces are off (4-byte LDS-DMA form)
        Xm = torch.empty(c * N + 4, device=dev)[1 : 1 + c * N].view(1, c, 1, 1, N).copy_(X)
        out = []
        for algo in (163, 164, 162, 161):
            ms = t(lambda: ops.conv3d_bn_act(Xm, pc, relu=False, algo=algo, splits=1))
            out.append(f"a{algo}:{fl / ms:.0f}")
        print(f"   4-byte A pieces: " + " ".join(out))


if __name__ == "__main__":
    main()
