import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from anomaly_detection_on_video_amd import ops, mil_ops
from test_hip_u8_stem import _stem, _frames
dev = torch.device("cuda:0")
pc, _ = _stem()
for (F, FH, FW, fpc, crop, const) in [(16, 64, 80, 16, 56, 115), (16, 64, 80, 16, 56, None), (8, 40, 52, 8, 32, None)]:
    if const is not None:
        fd = torch.full((F, FH, FW, 3), const, dtype=torch.uint8, device=dev)
    else:
        fd = torch.from_numpy(_frames(3, (F, FH, FW, 3))).to(dev)
    n = F // fpc * 10
    ref = ops.conv3d_bn_relu_maxpool233(mil_ops.tencrop_normalize_u8(fd, fpc, crop), pc)
    got = ops.conv3d_u8_tencrop_bn_relu_maxpool233(fd, pc, 0, n, fpc, crop)
    print("case", F, FH, FW, fpc, crop, const, "shape", tuple(got.shape), "ref max", float(ref.abs().max()))
    d = (got - ref).abs()
    print(" per crop max err:", [round(float(d[i].max()), 4) for i in range(n)])
    print(" per t:", [round(float(d[:, :, t].max()), 4) for t in range(d.shape[2])])
    print(" per h:", [round(float(d[0, :, :, h].max()), 3) for h in range(d.shape[3])])
    print(" per w:", [round(float(d[0, :, :, :, w].max()), 3) for w in range(d.shape[4])])
    print(" got[0,0,0,:3,:6]", got[0, 0, 0, :3, :6].cpu().numpy().round(3))
    print(" ref[0,0,0,:3,:6]", ref[0, 0, 0, :3, :6].cpu().numpy().round(3))
    ktab, cls, corr = ops.ensure_u8_tables(pc, (FH, FW), (fpc, crop, crop))
    print(" cls", cls[:4].tolist(), cls[4:].tolist()[:80])
    print(" corr[:8]", corr[:8].tolist(), "ktab[:8]", ktab[:8].tolist())
