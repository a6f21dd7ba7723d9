#!/usr/bin/env python3
"""Dev tool: what a conv launch costs beside its MFMAs.  One 1x1x1 conv shape (M = 32 x 2 x 14 x 14 positions, Cout = 1024: the
layer3.x.conv3 launch, 1568 tiles of 128 x 64 = one round of workgroups) timed over Cin = 16 .. 1024: the slope is the
MFMA rate of the main loop, the intercept the fixed part of a round (launch, prologue, epilogue, the tail tile).
    python tools/time_fixed_overhead.py [algo]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import ops  # noqa: E402
from time_fused_pool import bench  # noqa: E402

dev = torch.device("cuda:0")
algo = int(sys.argv[1]) if len(sys.argv) > 1 else 162
g = torch.Generator(device=dev).manual_seed(0)
for (B, T, H, W, cout) in ((32, 2, 14, 14, 1024), (32, 2, 7, 7, 2048), (32, 2, 28, 28, 512)):
    rows = []
    for cin in (16, 32, 64, 128, 256, 512, 1024):
        w = torch.randn((cout, cin, 1, 1, 1), device=dev, generator=g) * (2.0 / cin) ** 0.5
        ones = torch.ones(cout, device=dev)
        pc = ops.pack_conv(w, ones, ones * 0.1, ones * 0.05, ones, 1e-5, (1, 1, 1), (0, 0, 0), name="one")
        x = torch.randn((B, cin, T, H, W), device=dev, generator=g)
        y = ops.conv3d_bn_act(x, pc, algo=algo, splits=1)
        res = torch.randn_like(y)
        ts = bench([lambda: ops.conv3d_bn_act(x, pc, relu=True, algo=algo, splits=1, out=y),
                    lambda: ops.conv3d_bn_act(x, pc, relu=True, residual=res, algo=algo, splits=1, out=y)], reps=20, rounds=5)
        flop = 2.0 * y.numel() * cin
        rows.append((cin, ts[0] * 1e3, ts[1] * 1e3, flop))
        print(f"M={B*T*H*W} Cout={cout} Cin={cin:5d}: plain {ts[0]*1e3:7.1f} us ({flop/ts[0]/1e9:6.1f} TF)   +res {ts[1]*1e3:7.1f} us ({flop/ts[1]/1e9:6.1f} TF)", flush=True)
    # least squares over the four largest K
    import numpy as np
    k = np.array([r[0] for r in rows[-4:]], dtype=np.float64)
    for col, name in ((1, "plain"), (2, "+res")):
        t = np.array([r[col] for r in rows[-4:]])
        a, b = np.polyfit(k, t, 1)
        print(f"  {name}: {b:.1f} us fixed + {a*1000:.1f} us per 1000 channels of K -> main-loop rate {2.0*B*T*H*W*cout*1000/(a*1000)/1e6:.1f} TFLOP/s")
