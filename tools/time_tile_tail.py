#!/usr/bin/env python3
"""Dev tool: what the tiles beyond one round of workgroups cost.  The layer3.x.conv3 GEMM (Cin 256 -> Cout 1024, + residual, 128 x 64
tiles: 16 n-tiles) timed alone over M = 128 * m_tiles positions for m_tiles around 96 (= 1536 tiles = 256 CUs x 6 resident workgroups):
the step between 96 and 97 m-tiles is the price of a second round that 16 (... 32: M = 12 544 of the real layer) tiles run alone.
    python tools/time_tile_tail.py [algo]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import ops  # noqa: E402
from time_fused_pool import bench  # noqa: E402

dev = torch.device("cuda:0")
algo = int(sys.argv[1]) if len(sys.argv) > 1 else 162
g = torch.Generator(device=dev).manual_seed(0)
for cin, cout in ((256, 1024), (128, 512), (512, 2048)):
    w = torch.randn((cout, cin, 1, 1, 1), device=dev, generator=g) * (2.0 / cin) ** 0.5
    ones = torch.ones(cout, device=dev)
    pc = ops.pack_conv(w, ones, ones * 0.1, ones * 0.05, ones, 1e-5, (1, 1, 1), (0, 0, 0), name="one")
    per_m = 1536 * 64 // cout  # m-tiles of 128 rows that make one round of 1536 tiles
    for mt in (per_m // 2, per_m - 8, per_m - 2, per_m - 1, per_m, per_m + 1, per_m + 2, per_m + 4, per_m + 8, per_m + per_m // 2, 2 * per_m, 2 * per_m + 2):
        m = 128 * mt
        x = torch.randn((1, cin, 1, 1, m), device=dev, generator=g)
        y = ops.conv3d_bn_act(x, pc, algo=algo, splits=1)
        res = torch.randn_like(y)
        ts = bench([lambda: ops.conv3d_bn_act(x, pc, relu=True, residual=res, algo=algo, splits=1, out=y)], reps=20, rounds=5)
        flop = 2.0 * y.numel() * cin
        tiles = mt * cout // 64
        print(f"Cin={cin} Cout={cout} m_tiles={mt:4d} tiles={tiles:5d} ({tiles / 1536:5.3f} rounds): {ts[0]*1e3:7.1f} us  {flop/ts[0]/1e9:6.1f} TF  {ts[0]*1e6/tiles:6.2f} ns/tile", flush=True)
