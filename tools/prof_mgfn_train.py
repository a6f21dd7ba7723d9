#!/usr/bin/env python3
"""Dev tool: N MGFN training steps (32,10,32,2049) for rocprofv3; prints wall ms per step.
    python tools/prof_mgfn_train.py [N] [eager|graph] [overlap|serial]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection  # noqa: E402
from anomaly_detection_on_video_amd.train_graph import GraphedTrainStep  # noqa: E402
from anomaly_detection_on_video_amd.weights import synth_module_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
mode = sys.argv[2] if len(sys.argv) > 2 else "graph"
overlap = (sys.argv[3] if len(sys.argv) > 3 else "overlap") == "overlap"
dev = "cuda:0"
m = MGFNForVideoAnomalyDetection(MGFNConfig())
m.load_state_dict(synth_module_state_dict(m))
m = m.to(dev).train()
vb = torch.rand(32, 10, 32, 2049, device=dev)
al, nl = torch.ones(16, device=dev), torch.zeros(16, device=dev)
if os.environ.get("ADV_HIP_ADAM", "1") == "1":
    from anomaly_detection_on_video_amd.optim import HipAdam  # noqa: E402

    opt = HipAdam(m.parameters(), lr=1e-3, weight_decay=5e-4)
else:
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-4, fused=True, capturable=True)
step = GraphedTrainStep(m, opt, eager_steps=3 if mode == "graph" else 1 << 30, overlap=overlap)
for _ in range(5):
    step(vb, al, nl)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n):
    step(vb, al, nl)
torch.cuda.synchronize()
print(f"train step ({mode}, {'overlap' if overlap else 'serial'}) wall {(time.perf_counter() - t) / n * 1e3:.2f} ms, replays {step.replays}")
