#!/usr/bin/env python3
"""Dev tool: the single-pass (non-GEMM) launches of the MGFN training step at the stage-2 shape (1024 channels x 320 sequences x 32
clips), timed alone through HIP-graph replays of 20 launches each (straight C-ABI calls: no autograd in the capture): us per launch
and the bytes of their operands (each counted once) per second.
    python tools/time_mgfn_passes.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import _lib  # noqa: E402
from anomaly_detection_on_video_amd._lib import check, ptr, stream  # noqa: E402

dev = "cuda:0"
Cc, B, T = 1024, 320, 32
N = B * T
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(s, device=dev, generator=g)
x, dy, add = rnd(Cc, B, T), rnd(Cc, B, T), rnd(Cc, B, T)
out = torch.empty_like(x)
MB = Cc * N * 4 / 1e6
eps = C.c_float(1e-5)


def timed(name, fn, mbytes, reps=20):
    fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        s.record()
        graph.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps * 1e3)
    print(f"{name:28s} {best:7.1f} us  {mbytes / best:5.2f} TB/s ({mbytes:.0f} MB of operands)", flush=True)


gam, bet = torch.rand(Cc, device=dev) + 0.5, rnd(Cc)
mu, rs = torch.empty(N, device=dev), torch.empty(N, device=dev)
timed("chan_layernorm fwd", lambda: check(lib.advhip_chan_layernorm_fwd_f32(ptr(x), ptr(gam), ptr(bet), ptr(out), ptr(mu), ptr(rs), Cc, N, eps, stream(x))), 2 * MB)
rows = lib.advhip_chan_layernorm_bwd_partial_rows(N)
pgb = torch.empty((rows, 2 * Cc), device=dev)
timed("chan_layernorm bwd (+add)", lambda: check(lib.advhip_chan_layernorm_bwd_add_f32(ptr(dy), ptr(x), ptr(gam), ptr(mu), ptr(rs), ptr(add), ptr(out), ptr(pgb), Cc, N,
                                                                                        eps, stream(x))), 4 * MB)
pg, pb = torch.empty((rows, Cc), device=dev), torch.empty((rows, Cc), device=dev)
timed("chan_layernorm bwd", lambda: check(lib.advhip_chan_layernorm_bwd_f32(ptr(dy), ptr(x), ptr(gam), ptr(mu), ptr(rs), ptr(out), ptr(pg), ptr(pb), Cc, N, eps,
                                                                              stream(x))), 3 * MB)
fw, fb = rnd(Cc) * 0.03, rnd(1)
xn, mean, rstd, score = torch.empty((N, Cc), device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)
timed("head_ln_fc fwd", lambda: check(lib.advhip_head_ln_fc_fwd_f32(ptr(x), ptr(gam), ptr(bet), ptr(fw), ptr(fb), ptr(xn), ptr(mean), ptr(rstd), ptr(score), Cc, N, eps,
                                                                     stream(x))), 2 * MB)
hrows = lib.advhip_head_ln_fc_partial_rows(N)
hp = torch.empty((hrows, 3 * Cc + 1), device=dev)
dxn, dsc = rnd(N, Cc), rnd(N)
timed("head_ln_fc bwd", lambda: check(lib.advhip_head_ln_fc_bwd_f32(ptr(dxn), ptr(dsc), ptr(x), ptr(gam), ptr(bet), ptr(fw), ptr(mean), ptr(rstd), ptr(score), ptr(out),
                                                                     ptr(hp), Cc, N, stream(x))), 3 * MB)
H, K = 16, 5
w2, b2 = rnd(H, K), rnd(H)
timed("dwconv_t fwd", lambda: check(lib.advhip_dwconv_t_fwd_f32(ptr(x), ptr(w2), ptr(b2), ptr(out), Cc, H, B, T, K, stream(x))), 2 * MB)
chunks = lib.advhip_dwconv_t_bwd_chunks(Cc, B)
part = torch.empty((Cc * chunks, K + 1), device=dev)
timed("dwconv_t bwd", lambda: check(lib.advhip_dwconv_t_bwd_f32(ptr(dy), ptr(x), ptr(w2), ptr(out), ptr(part), Cc, H, B, T, K, stream(x))), 3 * MB)
bm, bv = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
timed("bn_rows fwd (batch stats)", lambda: check(lib.advhip_bn_rows_fwd_f32(ptr(x), ptr(gam), ptr(bet), ptr(out), ptr(bm), ptr(bv), Cc, N, eps, stream(x))), 2 * MB)
dg, db = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
timed("bn_rows bwd (+add)", lambda: check(lib.advhip_bn_rows_bwd_add_f32(ptr(dy), ptr(x), ptr(gam), ptr(bm), ptr(bv), ptr(add), ptr(out), ptr(dg), ptr(db), Cc, N, eps,
                                                                          stream(x))), 4 * MB)
u = torch.empty((3 * Cc, N), device=dev)
timed("unfold3", lambda: check(lib.advhip_unfold3_f32(ptr(x), ptr(u), Cc, B, T, stream(x))), 4 * MB)
