#!/usr/bin/env python3
"""Dev tool: the two GEMMs of the 2048 -> 64 token conv in a training step ((32,10,32,2049) batch): forward Wt (192 x 2048) . X^T on the
input rows as stored (advhip_gemm_nt_f32, tile / K-slice sweep) and the weight gradient dZ (192 x 10240) . X as one conv launch
(algo / K-slice sweep), device-timed through graph replays."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import _lib, mgfn_ops, ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
rows = torch.randn((10240, 2049), device=dev, generator=g)
wt = torch.randn((192, 2048), device=dev, generator=g)


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


flop = 2.0 * 192 * 2048 * 10240
print("forward  Wt . X^T  (M=192, N=10240, K=2048): default", ops.gemm_nt_choice(192, 10240, 2048))
for tile in (1, 2, 3):
    for splits in (1, 2, 3, 4):
        try:
            t = timed(lambda: ops.gemm_nt(wt, rows[:, :2048], splits=splits, tile=tile))
            print(f"  tile {tile} splits {splits}: {t:6.1f} us {flop / t / 1e6:6.1f} TF", flush=True)
        except Exception as ex:  # noqa: BLE001
            print(f"  tile {tile} splits {splits}: {str(ex)[:80]}")
dz = torch.randn((192, 10240), device=dev, generator=g)
wp = mgfn_ops.pack_kc(dz.contiguous().view(192, 10240, 1))
x3 = rows.view(10240, 1, 2049)
d0 = mgfn_ops._desc(10240, 192, 1, 1, 2049, 0)
print("weight gradient dZ . X (conv on the rows as stored: Cin=10240, Cout=192, 2049 positions): default algo", d0.algo, "splits", d0.splits)
orig = mgfn_ops._desc
for algo in (mgfn_ops.ALGO_SMALL, mgfn_ops.ALGO, _lib.ALGO_DMA2_BASE + _lib.ALGO_IGEMM_64x128):
    for splits in (4, 8, 12, 16, 24):
        def patched(cin, cout, k, b, t, act, _a=algo, _s=splits):
            d = orig(cin, cout, k, b, t, act)
            d.algo, d.splits = _a, _s
            return d
        mgfn_ops._desc = patched
        try:
            t = timed(lambda: mgfn_ops.conv_cn(x3, wp, 192, 1))
            print(f"  algo {algo} splits {splits}: {t:6.1f} us {flop * 2049 / 2048 / t / 1e6:6.1f} TF", flush=True)
        except Exception as ex:  # noqa: BLE001
            print(f"  algo {algo} splits {splits}: {str(ex)[:100]}")
        finally:
            mgfn_ops._desc = orig
