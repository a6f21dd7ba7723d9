#!/usr/bin/env python3
"""How long does the host take to ISSUE a step of the extract->score stream (all launches of one step_async), against the
8.7 ms the GPU needs for it?  If the two were close the pipeline would be launch-bound."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from anomaly_detection_on_video_amd.i3d import I3Res50
from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection
from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream
from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_module_state_dict

dev = torch.device("cuda:0")
m = I3Res50()
m.load_state_dict(synth_i3d_state_dict())
m = m.eval().to(dev)
sc = MGFNForVideoAnomalyDetection(MGFNConfig())
sc.load_state_dict(synth_module_state_dict(sc))
sc = sc.eval().to(dev)
st = ExtractScoreStream(m, sc, clips_per_video=32, ncrops=10, local_batch=32)
x = torch.randn((32, 3, 16, 224, 224), device=dev)
for _ in range(6):
    st.step_async(x)
st.drain()
torch.cuda.synchronize()
n = 60
t0 = time.perf_counter()
for _ in range(n):
    st.step_async(x)
t1 = time.perf_counter()
st.drain()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time per step {(t1 - t0) / n * 1e3:.3f} ms; wall per step {(t2 - t0) / n * 1e3:.3f} ms", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile
    import pstats

    pr = cProfile.Profile()
    pr.enable()
    for _ in range(30):
        st.step_async(x)
    pr.disable()
    st.drain()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
