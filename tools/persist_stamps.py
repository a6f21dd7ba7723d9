#!/usr/bin/env python3
"""Dev tool: in-kernel cycle stamps of the persistent conv kernel (diagnostic id ADVHIP_ALGO_PERSIST_BASE + 5 + 8 (W - 1))."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
for (cin, cout, t, h, w, use_res) in [(1024, 512, 2, 14, 14, False), (128, 512, 2, 28, 28, True), (64, 256, 4, 55, 55, True)]:
    wt = torch.randn(cout, cin, 1, 1, 1, device=dev) * 0.05
    one, zero = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    pc = ops.pack_conv(wt, one, zero, zero, one, 0.0, (1, 1, 1), (0, 0, 0), name="dbg")
    x = torch.randn(32, cin, t, h, w, device=dev)
    res = torch.randn(32, cout, t, h, w, device=dev) if use_res else None
    for W in (1, 2):
        algo = _lib.ALGO_PERSIST_BASE + 5 + 8 * (W - 1)
        d = pc.desc(32, t, h, w, True, algo, 1)
        y = torch.empty(32, cout, t, h, w, device=dev)
        dbg = torch.zeros(4096 * 8, device=dev, dtype=torch.int64)
        ep = _lib.ConvEpilogue(_lib.ptr(dbg), None, None, None, None, None, 0, None)
        for _ in range(2):
            _lib.check(lib.advhip_conv3d_bn_act_ex_f32(C.byref(d), _lib.ptr(x), 0, _lib.ptr(pc.w_packed), _lib.ptr(ops.ensure_ktab(pc, (t, h, w))), _lib.ptr(pc.scale),
                                                       _lib.ptr(pc.shift), _lib.ptr(res), _lib.ptr(y), 0, C.byref(ep), None, 0, _lib.stream()), "dbg")
        torch.cuda.synchronize()
        g = dbg.view(-1, 8).cpu()
        g = g[g[:, 2] > 0].double()
        tot, bar, n = g[:, 0], g[:, 1], g[:, 2]
        ltot, lbar, lvm = g[:, 4], g[:, 5], g[:, 6]
        print(f"{cin}->{cout} {t}x{h}x{w} res={use_res} W={W}: WGs {len(g)}, k-tiles/WG {n.mean():.1f} | MFMA wave: {tot.mean() / n.mean():.0f} cyc/k-tile, "
              f"barrier wait {bar.mean() / n.mean():.0f} | loader: {ltot.mean() / n.mean():.0f} cyc/k-tile, barrier wait {lbar.mean() / n.mean():.0f}, vmcnt wait {lvm.mean() / n.mean():.0f}"
              f" | longest WG {tot.max():.0f} cyc", flush=True)
