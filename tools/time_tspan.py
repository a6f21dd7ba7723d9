#!/usr/bin/env python3
"""Dev tool: (3,1,1) convs of layers 1-2 at B=32: tuned tile vs ADVHIP_ALGO_TSPAN_128x64 (ms, interleaved rounds)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import _lib, ops  # noqa: E402
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
for name, (cin, t, hw) in {"layer1.0": (64, 4, 55), "layer1.1": (256, 4, 55), "layer2.0": (256, 2, 55), "layer2.2": (512, 2, 28), "layer3.0": (512, 2, 28)}.items():
    pc = convs[name][0]
    x = torch.randn((B, cin, t, hw, hw), device=dev)
    ta, tb = bench([lambda: ops.conv3d_bn_act(x, pc, relu=True), lambda: ops.conv3d_bn_act(x, pc, relu=True, algo=_lib.ALGO_TSPAN_128x64)])
    macs = B * pc.cout * t * hw * hw * cin * 3
    print(f"{name}.conv1 {cin}->{pc.cout} T={t} {hw}x{hw}: table {ta:.3f} ms ({2 * macs / ta / 1e9:.1f} TF) | tspan {tb:.3f} ms ({2 * macs / tb / 1e9:.1f} TF)")
