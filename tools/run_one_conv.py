#!/usr/bin/env python3
"""Dev tool: run ONE conv configuration N times (for rocprofv3 --pmc / --kernel-trace runs).

    python tools/run_one_conv.py --key 256,256,1,3,3,1,1,1,0,1,1,32,2,14,14 --algo 4 --splits 3 --reps 20
key = Cin,Cout,kt,kh,kw,st,sh,sw,pt,ph,pw,B,T,H,W
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key", required=True)
    ap.add_argument("--algo", type=int, default=0)
    ap.add_argument("--splits", type=int, default=0)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--res", action="store_true")
    ap.add_argument("--relu-input", action="store_true", help="non-negative input with ~50%% exact zeros, like post-ReLU activations")
    a = ap.parse_args()
    cin, cout, kt, kh, kw, st, sh, sw, pt, ph, pw, B, T, H, W = (int(v) for v in a.key.split(","))
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    w = torch.randn((cout, cin, kt, kh, kw), device=dev, generator=g) * (2.0 / (cin * kt * kh * kw)) ** 0.5
    ones = torch.ones(cout, device=dev)
    pc = ops.pack_conv(w, ones, ones * 0.1, ones * 0.05, ones, 1e-5, (st, sh, sw), (pt, ph, pw), name="one")
    x = torch.randn((B, cin, T, H, W), device=dev, generator=g)
    if a.relu_input:
        x = torch.relu(x)
    y = ops.conv3d_bn_act(x, pc, algo=a.algo or None, splits=a.splits or None)
    res = torch.randn_like(y) if a.res else None
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.reps):
        ops.conv3d_bn_act(x, pc, residual=res, algo=a.algo or None, splits=a.splits or None, out=y)
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / a.reps
    macs = y.numel() * cin * kt * kh * kw
    print(f"{a.key} algo={a.algo} splits={a.splits}: {t:.4f} ms  {2*macs/t/1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
