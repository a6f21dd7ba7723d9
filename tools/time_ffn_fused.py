#!/usr/bin/env python3
"""Dev tool: one `x = x + FFN(LN(x))` step of a narrow MGFN block (C = 64 / 128, N = 10 240 positions) forward and backward, as one fused launch
each (csrc/ffn_fused.hip) against the three + three launches it replaces; device-timed through graph replays; max abs difference of every output."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import mgfn_ops  # noqa: E402
from anomaly_detection_on_video_amd.models.mgfn.modeling_mgfn import MGFNFeedForward  # noqa: E402

dev = "cuda:0"


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


for c in (64, 128):
    torch.manual_seed(c)
    ffn = MGFNFeedForward(c).to(dev)
    x = torch.randn(c, 320, 32, device=dev, requires_grad=True)
    gy = torch.randn(c, 320, 32, device=dev)
    outs = {}
    for fused in (False, True):
        mgfn_ops.FUSED_FFN = fused

        def fwd():
            return mgfn_ops.ffn_block_cn(x, ffn.layer_norm, ffn.in_conv, ffn.out_conv)

        y = fwd()
        for p in list(ffn.parameters()) + [x]:
            p.grad = None
        y.backward(gy)
        outs[fused] = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in ffn.parameters()]
        t_f = timed(lambda: fwd())
        t_fb = timed(lambda: (fwd().backward(gy)))
        print(f"C={c} fused={fused}: forward {t_f:6.1f} us, forward + backward (incl. dW / db launches) {t_fb:6.1f} us", flush=True)
    diffs = [float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(outs[True], outs[False])]
    print(f"C={c}: max rel diff fused vs unfused (y, dx, parameter grads): " + " ".join(f"{d:.1e}" for d in diffs), flush=True)
