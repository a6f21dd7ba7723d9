#!/usr/bin/env python3
"""Isolated launch times of the stem from fp32 crops vs from resized uint8 frames (and of the TenCrop pass the latter removes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from anomaly_detection_on_video_amd import _lib, mil_ops, ops

if len(sys.argv) > 2:  # an experimental build of the library (timing studies)
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
from anomaly_detection_on_video_amd.i3d import I3Res50
from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict

dev = torch.device("cuda:0")
m = I3Res50()
m.load_state_dict(synth_i3d_state_dict())
m = m.eval().to(dev)
m.prepare()
pc = m._plan[0].convs[0]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 40
fr = ops.with_slack(torch.randint(0, 256, (64, 256, 340, 3), dtype=torch.uint8, device=dev))
x = mil_ops.tencrop_normalize_u8(fr)[:B].contiguous()


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


gf = 2 * 4720.6e6 * B / 1e12
for name, fn in [("tencrop_normalize_u8 (40 crop-clips)", lambda: mil_ops.tencrop_normalize_u8(fr)),
                 ("stem+pool from fp32 crops", lambda: ops.conv3d_bn_relu_maxpool233(x, pc)),
                 ("stem+pool from uint8 frames (" + ops.U8_STEM_FORM + ")", lambda: ops.conv3d_u8_tencrop_bn_relu_maxpool233(fr, pc, 0, B))]:
    us = t(fn)
    print(f"B={B} {name:40s} {us:9.1f} us" + (f"  {gf / us * 1e6:6.1f} TFLOP/s" if "stem" in name else ""), flush=True)
