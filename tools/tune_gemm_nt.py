#!/usr/bin/env python3
"""Dev tool: (tile, K slices) of advhip_gemm_nt_rowsum_f32 for the weight-gradient shapes of the MGFN scorer at its training batch
(K = 10 240 positions), 0.4 s of back-to-back launches per candidate (burst timings mislead: the clock ramps under load).
Prints a table and the GEMM_NT_TUNED dict for ops.py."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import ops  # noqa: E402

K = 10240
SHAPES = [(4096, 1024), (1024, 4096), (1024, 1024), (1024, 3072), (1024, 128), (512, 128), (128, 512), (128, 128), (128, 384),
          (256, 64), (64, 256), (64, 64), (64, 192), (192, 64), (128, 64)]


def rate(fn, seconds=0.3, per_graph=20):
    """us per call on the DEVICE: `per_graph` calls captured into one HIP graph, replayed back to back (eager launches of the
    small shapes are bound by the host's 30 us per call, which says nothing about the kernel)."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per_graph):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(5):
            g.replay()
        n += 5 * per_graph
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    dev = torch.device("cuda:0")
    best = {}
    for M, N in SHAPES:
        a = torch.randn(M, K, device=dev)
        b = torch.randn(N, K, device=dev)
        fl = 2.0 * M * N * K / 1e6
        res = []
        for tile, (bm, bn) in ((1, (64, 64)), (2, (128, 64)), (3, (128, 128))):
            tiles = -(-M // bm) * -(-N // bn)
            cands = sorted({s for s in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 40, 64) if 200 <= tiles * s <= 4096} | ({1} if tiles >= 128 else set()) | ({4, 8, 16, 40, 64} if tiles < 64 else set()))
            for sp in cands:
                us = rate(lambda: ops.gemm_nt(a, b, sp, rowsum=True, tile=tile))
                res.append((us, tile, sp))
        res.sort()
        us0 = rate(lambda: ops.gemm_nt(a, b, 0, rowsum=True))
        best[(M, N, K)] = (res[0][1], res[0][2])
        print(f"M={M} N={N}: best tile {res[0][1]} splits {res[0][2]} {res[0][0]:.1f} us = {fl / res[0][0]:.1f} TF | current choice {us0:.1f} us = {fl / us0:.1f} TF | "
              + " ".join(f"t{t}s{s}:{fl / u:.0f}" for u, t, s in res[:6]), flush=True)
    print("GEMM_NT_TUNED = {" + ", ".join(f"{k}: {v}" for k, v in best.items()) + "}")


if __name__ == "__main__":
    main()
