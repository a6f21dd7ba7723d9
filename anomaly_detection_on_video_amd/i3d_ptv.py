"""`i3d_8x8_r50`: the pytorchvideo ResNet-50 that `build_i3d_feature_extractor("i3d_8x8_r50")` builds in the reference
(/root/reference/src/i3d.py:339-350) through `pytorchvideo.models.resnet.create_resnet`, on the HIP kernels.

PARITY UNPINNED.  pytorchvideo (tag 0.1.3, named only by a comment URL at src/i3d.py:14) is a third-party package that
is neither vendored in the reference nor installed in this image, and the reference has no test or fixture for this
branch.  What is restated here is pytorchvideo's published `create_resnet` topology with the arguments the reference
passes (src/i3d.py:340-349) and its documented defaults:

    stem      Conv3d(3, 64, k(5,7,7), s(1,2,2), p(2,3,3), bias=False) + BN + ReLU + MaxPool3d(k(1,3,3), s(1,2,2), p(0,1,1))
    res2..5   depths (3,4,6,3); per block  conv_a k(a,1,1) p(a//2,0,0) + BN + ReLU -> conv_b k(1,3,3) s(1,s,s) p(0,1,1) + BN +
              ReLU -> conv_c 1x1x1 + BN; shortcut conv 1x1x1 s(1,s,s) + BN on the first block of a stage; ReLU(sum);
              conv_a temporal kernels res2: 3,3,3  res3: 3,1,3,1  res4: 3,1,3,1,3,1  res5: 1,3,1 (src/i3d.py:343-348);
              spatial stride 1,2,2,2; widths 64/256, 128/512, 256/1024, 512/2048
    pool      MaxPool3d(k(2,1,1), s(2,1,1)) after res2 (stage1_pool=nn.MaxPool3d)
    head      the reference's own ResNetHead (src/i3d.py:21-57): AvgPool3d(k(4,7,7), s1) -> AdaptiveAvgPool3d(1)

with pytorchvideo's module names, so that a state dict of the reference's `I3D_8x8_R50.pyth` layout loads:
`blocks.0.{conv,norm}`, `blocks.{1,3,4,5}.res_blocks.N.{branch1_conv,branch1_norm,branch2.{conv_a,norm_a,conv_b,norm_b,conv_c,norm_c}}`
(blocks.2 = the stage-1 pool, blocks.6 = the head; neither has parameters).  The arithmetic is the same fused
conv+BN(+residual)(+ReLU) launches as I3Res50; tests check it against a plain-torch restatement of the SAME topology
(oracle/i3d_oracle.py:ptv_forward), which pins the kernels, not pytorchvideo.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import nn

from . import _lib, ops

STAGE_DEPTHS = (3, 4, 6, 3)
CONV_A_KT = ((3, 3, 3), (3, 1, 3, 1), (3, 1, 3, 1, 3, 1), (1, 3, 1))  # src/i3d.py:343-348 cycled over the blocks
HEAD_POOL = (4, 7, 7)


class _Branch2(nn.Module):
    def __init__(self, cin, inner, cout, kt, stride):
        super().__init__()
        self.conv_a = nn.Conv3d(cin, inner, (kt, 1, 1), stride=1, padding=(kt // 2, 0, 0), bias=False)
        self.norm_a = nn.BatchNorm3d(inner)
        self.conv_b = nn.Conv3d(inner, inner, (1, 3, 3), stride=(1, stride, stride), padding=(0, 1, 1), bias=False)
        self.norm_b = nn.BatchNorm3d(inner)
        self.conv_c = nn.Conv3d(inner, cout, 1, bias=False)
        self.norm_c = nn.BatchNorm3d(cout)


class _ResBlock(nn.Module):
    def __init__(self, cin, inner, cout, kt, stride):
        super().__init__()
        if cin != cout or stride != 1:
            self.branch1_conv = nn.Conv3d(cin, cout, 1, stride=(1, stride, stride), bias=False)
            self.branch1_norm = nn.BatchNorm3d(cout)
        self.branch2 = _Branch2(cin, inner, cout, kt, stride)


class _ResStage(nn.Module):
    def __init__(self, cin, inner, cout, kts, stride):
        super().__init__()
        self.res_blocks = nn.ModuleList(_ResBlock(cin if i == 0 else cout, inner, cout, kt, stride if i == 0 else 1) for i, kt in enumerate(kts))


class _Stem(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv3d(3, 64, (5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False)
        self.norm = nn.BatchNorm3d(64)


class I3D8x8R50(nn.Module):
    def __init__(self):
        super().__init__()
        blocks: List[nn.Module] = [_Stem()]
        cin, inner = 64, 64
        for si, kts in enumerate(CONV_A_KT):
            blocks.append(_ResStage(cin, inner, inner * 4, kts, 1 if si == 0 else 2))
            cin, inner = inner * 4, inner * 2
            if si == 0:
                blocks.append(nn.MaxPool3d((2, 1, 1), (2, 1, 1)))
        blocks.append(nn.Identity())  # the head: AvgPool3d((4,7,7)) + AdaptiveAvgPool3d(1), no parameters
        self.blocks = nn.ModuleList(blocks)
        self._plan = None
        self._stamp = None
        self._head_w = {}

    def prepare(self):
        stamp = tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
        if self._plan is not None and stamp == self._stamp:
            return
        if self.blocks[0].conv.weight.device.type != "cuda":
            raise _lib.HipExtensionError("I3D8x8R50 parameters are not on the GPU: the backbone runs only as HIP kernels (no CPU fallback)")

        def pk(conv, bn, name):
            return ops.pack_conv(conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, conv.stride, conv.padding, name=name)

        plan = [("stem", pk(self.blocks[0].conv, self.blocks[0].norm, "blocks.0.conv"))]
        for bi, blk in enumerate(self.blocks[1:-1], start=1):
            if isinstance(blk, nn.MaxPool3d):
                plan.append(("pool211",))
                continue
            for ri, rb in enumerate(blk.res_blocks):
                p = f"blocks.{bi}.res_blocks.{ri}"
                b2 = rb.branch2
                plan.append(("block", pk(b2.conv_a, b2.norm_a, p + ".conv_a"), pk(b2.conv_b, b2.norm_b, p + ".conv_b"),
                             pk(b2.conv_c, b2.norm_c, p + ".conv_c"),
                             pk(rb.branch1_conv, rb.branch1_norm, p + ".branch1") if hasattr(rb, "branch1_conv") else None))
        self._plan, self._stamp = plan, stamp

    def _head_weights(self, thw: Tuple[int, int, int], dev) -> torch.Tensor:
        """AdaptiveAvgPool3d(1)(AvgPool3d(k, stride 1)(x)) as one weighted sum: weight of position i along an axis of
        length n = (number of windows that cover i) / (n - k + 1) / k."""
        key = (thw, dev)
        if key not in self._head_w:
            ws = []
            for n, k in zip(thw, HEAD_POOL):
                if n < k:
                    raise ValueError(f"feature map {thw} smaller than the head's AvgPool3d{HEAD_POOL}")
                nw = n - k + 1
                ws.append(torch.tensor([(min(i, nw - 1) - max(i - k + 1, 0) + 1) / (nw * k) for i in range(n)], dtype=torch.float64))
            w = (ws[0][:, None, None] * ws[1][None, :, None] * ws[2][None, None, :]).reshape(-1, 1).float()
            self._head_w[key] = w.to(dev)
        return self._head_w[key]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            raise _lib.HipExtensionError("I3D8x8R50 HIP path implements eval-mode BatchNorm only; call .eval()")
        if not x.is_cuda or x.dtype != torch.float32:
            raise _lib.HipExtensionError("input must be an fp32 tensor on the GPU; there is no CPU fallback")
        self.prepare()
        with torch.no_grad():
            x = x.detach().contiguous()
            for u in self._plan:
                if u[0] == "stem":
                    x = ops.conv3d_bn_act(x, u[1], relu=True)
                    x = ops.maxpool3d(x, (1, 3, 3), (1, 2, 2), padding=(0, 1, 1))
                elif u[0] == "pool211":
                    x = ops.maxpool3d(x, (2, 1, 1), (2, 1, 1))
                else:
                    _, ca, cb, cc, sc = u
                    h = ops.conv3d_bn_act(x, ca, relu=True)
                    h = ops.conv3d_bn_act(h, cb, relu=True)
                    res = ops.conv3d_bn_act(x, sc, relu=False) if sc is not None else x
                    x = ops.conv3d_bn_act(h, cc, relu=True, residual=res)
            B, Cc = x.shape[:2]
            w = self._head_weights(tuple(x.shape[2:]), x.device)
            return ops.bgemm(x.view(1, B * Cc, -1), w.view(1, -1, 1)).view(B, Cc, 1, 1, 1)
