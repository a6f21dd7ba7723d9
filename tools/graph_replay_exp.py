#!/usr/bin/env python3
"""Experiment: is the 3-lane extract stream host-bound?  Capture one backbone forward per lane in a HIP graph and replay the
three graphs round-robin; compare with the eager three-lane loop (backbone only, no scoring)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from anomaly_detection_on_video_amd.i3d import I3Res50
from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict

dev = torch.device("cuda:0")
m = I3Res50()
m.load_state_dict(synth_i3d_state_dict())
m = m.eval().to(dev)
m.streams = 1
B, LANES, N = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 3, 60
xs = [torch.randn((B, 3, 16, 224, 224), device=dev) for _ in range(LANES)]
with torch.no_grad():
    for x in xs:
        ref = m(x)
torch.cuda.synchronize()
lanes = [torch.cuda.Stream(device=dev) for _ in range(LANES)]


def eager():
    for i in range(N):
        with torch.cuda.stream(lanes[i % LANES]), torch.no_grad():
            m(xs[i % LANES])


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3


hi, wall = timeit(eager)
print(f"eager  {LANES} lanes: host issue {hi:.3f} ms/step, wall {wall:.3f} ms/step = {B / wall * 1e3:.1f} clips/s", flush=True)

graphs, outs = [], []
for i in range(LANES):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=lanes[i]), torch.no_grad():
        outs.append(m(xs[i]))
    graphs.append(g)
torch.cuda.synchronize()


def replay():
    for i in range(N):
        with torch.cuda.stream(lanes[i % LANES]):
            graphs[i % LANES].replay()


hi, wall = timeit(replay)
print(f"graphs {LANES} lanes: host issue {hi:.3f} ms/step, wall {wall:.3f} ms/step = {B / wall * 1e3:.1f} clips/s", flush=True)
with torch.no_grad():
    chk = m(xs[0])
torch.cuda.synchronize()
print("graph output equals eager output:", torch.equal(chk, outs[0]), flush=True)
