#!/usr/bin/env python3
"""Dev tool: the (kt,1,1) convs of layers 2-4 at B=32 over tile / split-K variants of the 2-deep LDS-DMA kernel (ms, TFLOP/s)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import ops  # noqa: E402
from anomaly_detection_on_video_amd.i3d import I3Res50  # noqa: E402
from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict  # noqa: E402
from time_fused_pool import bench  # noqa: E402

dev = torch.device("cuda:0")
B = 32
m = I3Res50()
m.load_state_dict(synth_i3d_state_dict())
m = m.eval().to(dev)
m.prepare()
convs = {u.name: u.convs for u in m._plan if u.kind == "bottleneck"}
for name, (cin, t, hw) in {"layer2.2": (512, 2, 28), "layer3.0": (512, 2, 28), "layer3.2": (1024, 2, 14), "layer4.1": (2048, 2, 7)}.items():
    pc = convs[name][0]
    x = torch.randn((B, cin, t, hw, hw), device=dev)
    macs = B * pc.cout * t * hw * hw * cin * 3
    cands = [(a, s) for a in (161, 162, 163, 164) for s in (1, 2, 3, 5) if pc.cout % (128 if a in (161, 164) else 64) == 0]
    ts = bench([(lambda a=a, s=s: ops.conv3d_bn_act(x, pc, relu=True, algo=a, splits=s)) for a, s in cands], reps=10, rounds=3)
    res = sorted(zip(ts, cands))
    print(f"{name}.conv1 {cin}->{pc.cout} T={t} {hw}x{hw}: " + " ".join(f"a{a}s{s}:{tt * 1e3:.0f}us/{2 * macs / tt / 1e9:.0f}TF" for tt, (a, s) in res[:8]), flush=True)
