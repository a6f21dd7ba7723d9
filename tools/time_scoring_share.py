#!/usr/bin/env python3
"""Dev tool: what the MIL scoring of completed videos costs the extract -> score stream: step time with a video completing every
10 steps (the benchmarked configuration) against a stream whose videos never complete inside the window."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd.i3d import I3Res50  # noqa: E402
from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection  # noqa: E402
from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream  # noqa: E402
from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_module_state_dict  # noqa: E402

dev = torch.device("cuda:0")
bb = I3Res50()
bb.load_state_dict(synth_i3d_state_dict())
bb = bb.eval().to(dev)
sc = MGFNForVideoAnomalyDetection(MGFNConfig())
sc.load_state_dict(synth_module_state_dict(sc))
sc = sc.eval().to(dev)
x = torch.randn((32, 3, 16, 224, 224), device=dev)
streams = {"scoring every 10 steps": ExtractScoreStream(bb, sc, clips_per_video=32, ncrops=10, local_batch=32),
           "no video completes": ExtractScoreStream(bb, sc, clips_per_video=32 * 400, ncrops=10, local_batch=32)}
for s in streams.values():
    for _ in range(12):
        s.step_async(x)
    s.drain()
torch.cuda.synchronize()
for rnd in range(3):
    for name, s in streams.items():
        t0 = time.perf_counter()
        for _ in range(150):
            s.step_async(x)
        s.drain()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 150
        print(f"{name}: {dt * 1e3:.4f} ms/step = {32 / dt:.1f} clips/s", flush=True)
