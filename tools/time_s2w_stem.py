#!/usr/bin/env python3
"""Dev tool: the fused stem (conv1 + bn1 + relu + maxpool1) at B = 32 with the 4-byte gather from the NCDHW input vs the
16-byte gather from column-parity planes: the split pass alone, the conv launch alone (planes prepared once), and both."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anomaly_detection_on_video_amd import _lib, ops  # noqa: E402
from anomaly_detection_on_video_amd.i3d import I3Res50  # noqa: E402
from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict  # noqa: E402
from time_fused_pool import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    m = I3Res50()
    m.load_state_dict(synth_i3d_state_dict())
    m = m.eval().to(dev)
    m.prepare()
    x = torch.randn((B, 3, 16, 224, 224), device=dev)
    pc = m._plan[0].convs[0]
    lib = _lib.load()
    xs = ops.split_w(x)
    y = ops.conv3d_bn_relu_maxpool233(x, pc, s2w=False)
    d = pc.desc(B, 16, 224, 224, True, 0, 1)
    need = lib.advhip_conv3d_relu_maxpool233_workspace_bytes(C.byref(d))
    ws = ops.workspace(dev, need)
    tab = ops.ensure_ktab_s2w(pc, (16, 224, 224))

    def conv_only():
        _lib.check(lib.advhip_conv3d_s2w_bn_relu_maxpool233_f32(C.byref(d), xs.data_ptr(), 0, pc.w_packed.data_ptr(), tab.data_ptr(), pc.scale.data_ptr(),
                                                                pc.shift.data_ptr(), y.data_ptr(), 0, ws.data_ptr(), need, _lib.stream()), "s2w")

    t = bench([lambda: ops.conv3d_bn_relu_maxpool233(x, pc, s2w=False), lambda: ops.split_w(x), conv_only,
               lambda: ops.conv3d_bn_relu_maxpool233(x, pc, s2w=True)])
    fl = 2 * B * 4720.6e6 / 1e9
    print(f"stem B={B}: 4-byte gather {t[0]:.3f} ms ({fl / t[0]:.1f} TF) | split pass {t[1]:.3f} | 16-byte gather conv alone {t[2]:.3f} ({fl / t[2]:.1f} TF) | split + conv {t[3]:.3f}")


if __name__ == "__main__":
    main()
