#!/usr/bin/env python3
"""Dev tool: the 1x1x1 stride-1 convs of the B = 32 plan (conv3 + residual, k = 1 conv1): the tuned one-tile-per-workgroup
kernel against every persistent id (ADVHIP_ALGO_PERSIST_BASE + tile + 8 * (wgs per CU - 1)), interleaved rounds, device-timed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import _lib, ops  # noqa: E402
from anomaly_detection_on_video_amd.i3d import I3Res50  # noqa: E402
from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict  # noqa: E402
from time_fused_pool import bench  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = I3Res50()
m.load_state_dict(synth_i3d_state_dict())
m = m.eval().to(dev)
m.prepare()
units = {u.name: u for u in m._plan if u.kind == "bottleneck"}
# (unit, which conv, input dims, residual?)
CASES = [("layer1.0", 2, (128, 4, 55, 55), False), ("layer1.1", 2, (64, 4, 55, 55), True), ("layer2.1", 2, (128, 2, 28, 28), True),
         ("layer2.1", 0, (512, 2, 28, 28), False), ("layer3.1", 2, (256, 2, 14, 14), True), ("layer3.1", 0, (1024, 2, 14, 14), False),
         ("layer4.1", 2, (512, 2, 7, 7), True), ("layer4.0", 0, (1024, 2, 14, 14), False)]
tot_table = tot_best = 0.0
for uname, ci, (cin, t, h, w), use_res in CASES:
    pc = units[uname].convs[ci]
    if pc.kernel != (1, 1, 1) or pc.stride != (1, 1, 1):
        continue
    x = torch.randn((B, cin, t, h, w), device=dev)
    res = torch.randn((B, pc.cout, t, h, w), device=dev) if use_res else None
    ids = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else list(_lib.PERSIST_ALGOS)  # (any algo ids: e.g. the 3-deep ring family 65..72)
    ids = [i for i in ids if pc.cout % _lib.algo_tile(i)[1] == 0]
    fns = [lambda: ops.conv3d_bn_act(x, pc, relu=True, residual=res)] + [lambda a=a: ops.conv3d_bn_act(x, pc, relu=True, residual=res, algo=a) for a in ids]
    ts = bench(fns, reps=20, rounds=5)
    flop = 2.0 * B * t * h * w * cin * pc.cout
    best = min(range(len(ids)), key=lambda i: ts[1 + i])
    tot_table += ts[0]
    tot_best += min(ts[0], ts[1 + best])
    print(f"{pc.name} {cin}->{pc.cout} {t}x{h}x{w} res={use_res}: table {pc.choices.get((B, t, h, w))} {ts[0] * 1e3:.1f} us ({flop / ts[0] / 1e9:.1f} TF) | "
          + " ".join(f"{a}:{tt * 1e3:.1f}" for a, tt in zip(ids, ts[1:])) + f" | best {ids[best]} {ts[1 + best] * 1e3:.1f} us ({flop / ts[1 + best] / 1e9:.1f} TF)", flush=True)
print(f"sum table {tot_table * 1e3:.1f} us, sum best {tot_best * 1e3:.1f} us")
