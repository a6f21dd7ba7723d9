"""I3D-ResNet50 feature extractor executed by hand-written gfx950 HIP kernels.

Drop-in for the reference's `src/i3d.py`: same class names (`I3Res50`, `Bottleneck`), constructor
arguments, state-dict keys (conv1.weight, bn1.*, layerL.B.{conv1,bn1,conv2,bn2,conv3,bn3}.*,
layerL.0.downsample.{0,1}.*) and `forward((B,3,T,H,W)) -> (B,2048,1,1,1)`
(/root/reference/src/i3d.py:60-121, 198-318, 332-364).

The nn.Conv3d / nn.BatchNorm3d children are *parameter holders only* (they give the reference's
state-dict layout, `.cuda()`, `load_state_dict`, ...).  Their `forward` is never called: the
network is run as a flat plan of fused launches

    conv3d (fp32 MFMA implicit GEMM) + folded eval-BN (+ residual) (+ ReLU)     53 launches
    maxpool3d x2, global average pool                                            3 launches

through the C ABI in include/advhip.h.  Eval-mode only (BatchNorm running statistics), no
autograd, CUDA tensors only -- anything else raises; there is no eager fallback.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import _lib, ops

repo_id = "jinmang2/test_video_fe"
model_zoo = {
    "i3d_8x8_r50": "I3D_8x8_R50.pyth",
    "tushar-n-baseline": "converted_ref_i3d.pt",
}


class Bottleneck(nn.Module):
    """Residual unit: (kt,1,1) conv -> (1,3,3) conv -> 1x1x1 conv x4 (+ downsample), src/i3d.py:60-121."""

    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample, temp_conv, temp_stride, use_nl=False):
        super().__init__()
        self.conv1 = nn.Conv3d(inplanes, planes, kernel_size=(1 + temp_conv * 2, 1, 1), stride=(temp_stride, 1, 1),
                               padding=(temp_conv, 0, 0), bias=False)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = nn.Conv3d(planes, planes, kernel_size=(1, 3, 3), stride=(1, stride, stride), padding=(0, 1, 1), bias=False)
        self.bn2 = nn.BatchNorm3d(planes)
        self.conv3 = nn.Conv3d(planes, planes * 4, kernel_size=1, stride=1, padding=0, bias=False)
        self.bn3 = nn.BatchNorm3d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        outplanes = planes * 4
        self.nl = NonLocalBlock(outplanes, outplanes, outplanes // 2) if use_nl else None

    def forward(self, x):  # pragma: no cover - guarded
        raise _lib.HipExtensionError("Bottleneck runs only inside I3Res50's fused HIP plan; call the parent model")


class NonLocalBlock(nn.Module):
    """Embedded-Gaussian non-local block (src/i3d.py:124-195): same children / state-dict keys as the reference
    (theta, phi, g, out: 1x1x1 Conv3d WITH bias; bn; maxpool (1,2,2)).  Dead code under the reference's factory
    (use_nl=False, src/i3d.py:338) but part of I3Res50(use_nl=True).  Runs as HIP launches only:

        mp = maxpool(x)                                   advhip_maxpool3d_f32
        theta = theta(x); [phi ; g] = [phi ; g](mp)       two conv launches (phi and g share one: Cout = 2*inner)
        p = softmax(theta^T.phi * inner**-0.5)            advhip_bgemm_f32 (alpha in the epilogue) + advhip_softmax_rows_f32
        t = g.p^T                                         advhip_bgemm_f32
        y = bn(out(t)) + x                                conv launch, BN folded, residual in the epilogue (no ReLU)
    """

    def __init__(self, dim_in, dim_out, dim_inner):
        super().__init__()
        self.dim_in, self.dim_inner, self.dim_out = dim_in, dim_inner, dim_out
        self.theta = nn.Conv3d(dim_in, dim_inner, kernel_size=(1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0))
        self.maxpool = nn.MaxPool3d(kernel_size=(1, 2, 2), stride=(1, 2, 2), padding=(0, 0, 0))
        self.phi = nn.Conv3d(dim_in, dim_inner, kernel_size=(1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0))
        self.g = nn.Conv3d(dim_in, dim_inner, kernel_size=(1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0))
        self.out = nn.Conv3d(dim_inner, dim_out, kernel_size=(1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0))
        self.bn = nn.BatchNorm3d(dim_out)
        self._packed = None
        self._stamp = None

    def prepare(self, name: str = "nl"):
        stamp = tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
        if self._packed is not None and stamp == self._stamp:
            return self._packed
        if self.theta.weight.device.type != "cuda":
            raise _lib.HipExtensionError("NonLocalBlock runs only as HIP kernels on an AMD GPU (call .cuda()); there is no CPU fallback")

        def biased(w, b, nm):  # conv with bias, no BN: scale 1, shift = bias
            one, zero = torch.ones_like(b), torch.zeros_like(b)
            return ops.pack_conv(w.detach(), one, b.detach(), zero, one, 0.0, (1, 1, 1), (0, 0, 0), name=f"{name}.{nm}")

        theta = biased(self.theta.weight, self.theta.bias, "theta")
        phig = biased(torch.cat([self.phi.weight.detach(), self.g.weight.detach()], dim=0).contiguous(),
                      torch.cat([self.phi.bias.detach(), self.g.bias.detach()]), "phi+g")
        # bn(out(t) + b) = bn'(out(t)) with running_mean' = running_mean - b
        out = ops.pack_conv(self.out.weight.detach(), self.bn.weight.detach(), self.bn.bias.detach(),
                            self.bn.running_mean.detach() - self.out.bias.detach(), self.bn.running_var.detach(), self.bn.eps,
                            (1, 1, 1), (0, 0, 0), name=f"{name}.out")
        self._packed, self._stamp = (theta, phig, out), stamp
        return self._packed

    def run(self, x: torch.Tensor, name: str = "nl") -> torch.Tensor:
        if self.training:
            raise _lib.HipExtensionError("NonLocalBlock HIP path implements eval-mode BatchNorm only; call .eval()")
        theta_pc, phig_pc, out_pc = self.prepare(name)
        B, _, T, H, W = x.shape
        inner = self.dim_inner
        if H < 2 or W < 2:
            raise ValueError(f"NonLocalBlock: input {tuple(x.shape)} smaller than the (1,2,2) pooling window")
        mp = ops.maxpool3d(x, (1, 2, 2), (1, 2, 2))
        theta = ops.conv3d_bn_act(x, theta_pc, relu=False).view(B, inner, -1)               # (B, inner, N)
        pg = ops.conv3d_bn_act(mp, phig_pc, relu=False).view(B, 2 * inner, -1)              # (B, 2*inner, Np)
        att = ops.bgemm(theta.transpose(1, 2), pg[:, :inner], alpha=float(inner) ** -0.5)   # (B, N, Np)
        ops.softmax_rows(att, out=att)
        t = ops.bgemm(pg[:, inner:], att.transpose(1, 2))                                   # (B, inner, N)
        return ops.conv3d_bn_act(t.view(B, inner, T, H, W), out_pc, relu=False, residual=x)

    def forward(self, x):
        with torch.no_grad():
            if not x.is_cuda:
                raise _lib.HipExtensionError("input is not on the GPU; there is no CPU fallback")
            return self.run(x.detach().contiguous())


class I3Res50(nn.Module):
    MIN_PART = 8  # crop-clips per stream below which splitting a batch over streams does not pay (B=8: -5 %, B=16..32: +2-3 %)

    def __init__(self, block=Bottleneck, layers=[3, 4, 6, 3], use_nl=False, in_channels: int = 3):
        """`in_channels`: 3 = the reference's RGB backbone (src/i3d.py:202-209).  Any other count -- 2 for the (x, y) planes of an
        optical-flow stream, BASELINE config 5 -- changes the stem's input width only; that is NOT in the reference (it ships no flow
        stream), so such a model has no reference pin: it is checked against the oracle's generic conv arithmetic."""
        self.inplanes = 64
        super().__init__()
        nonlocal_mod = 2 if use_nl else 1000  # src/i3d.py:219
        self.in_channels = int(in_channels)
        self.conv1 = nn.Conv3d(self.in_channels, 64, kernel_size=(5, 7, 7), stride=(2, 2, 2), padding=(2, 3, 3), bias=False)
        self.bn1 = nn.BatchNorm3d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool1 = nn.MaxPool3d(kernel_size=(2, 3, 3), stride=(2, 2, 2), padding=(0, 0, 0))
        self.maxpool2 = nn.MaxPool3d(kernel_size=(2, 1, 1), stride=(2, 1, 1), padding=(0, 0, 0))
        self.layer1 = self._make_layer(block, 64, layers[0], 1, [1, 1, 1], [1, 1, 1])
        self.layer2 = self._make_layer(block, 128, layers[1], 2, [1, 0, 1, 0], [1, 1, 1, 1], nonlocal_mod)
        self.layer3 = self._make_layer(block, 256, layers[2], 2, [1, 0, 1, 0, 1, 0], [1, 1, 1, 1, 1, 1], nonlocal_mod)
        self.layer4 = self._make_layer(block, 512, layers[3], 2, [0, 1, 0], [1, 1, 1])
        self.avgpool = nn.AdaptiveAvgPool3d((1, 1, 1))
        # same initialisation as the reference (src/i3d.py:246-251)
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
            elif isinstance(m, nn.BatchNorm3d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        self._plan: Optional[List["_Unit"]] = None
        self._plan_stamp: Optional[Tuple] = None
        # fold a stride-1 downsample branch (layer1.0) into conv3 as one conv over [x ; h] (see prepare)
        self.fuse_downsample = os.environ.get("ADV_I3D_FUSE_DS", "1") == "1"
        # run conv1+bn1+relu+maxpool1 and layer1's last conv3(+residual+relu)+maxpool2 as conv launches that pool in
        # their epilogue (src/i3d.py:303-309): the un-pooled activations (822 + 396 MB at B=32) never reach HBM
        self.fuse_pool = os.environ.get("ADV_I3D_FUSE_POOL", "1") == "1"
        # a forward is spread over this many HIP streams, each taking a contiguous part of the batch (see _run_streams)
        self.streams = int(os.environ.get("ADV_I3D_STREAMS", "2"))
        self._side_streams: List[torch.cuda.Stream] = []
        # layer name -> ADVHIP_ALGO_* override (tuning hook)
        self.algo_overrides: Dict[str, int] = {}

    def _make_layer(self, block, planes, blocks, stride, temp_conv, temp_stride, nonlocal_mod=1000):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion or temp_stride[0] != 1:
            downsample = nn.Sequential(
                nn.Conv3d(self.inplanes, planes * block.expansion, kernel_size=(1, 1, 1),
                          stride=(temp_stride[0], stride, stride), padding=(0, 0, 0), bias=False),
                nn.BatchNorm3d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample, temp_conv[0], temp_stride[0], False)]
        self.inplanes = planes * block.expansion
        for i in range(1, blocks):
            layers.append(block(self.inplanes, planes, 1, None, temp_conv[i], temp_stride[i], i % nonlocal_mod == nonlocal_mod - 1))
        return nn.Sequential(*layers)

    # ------------------------------------------------------------------ plan (load-time packing)
    def _stamp(self) -> Tuple:
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def _pack(self, conv: nn.Conv3d, bn: nn.BatchNorm3d, name: str) -> ops.PackedConv:
        return ops.pack_conv(conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                             conv.stride, conv.padding, name=name, algo=self.algo_overrides.get(name, _lib.ALGO_AUTO))

    def _pack_fused(self, conv_x: nn.Conv3d, bn_x: nn.BatchNorm3d, conv_h: nn.Conv3d, bn_h: nn.BatchNorm3d, name: str) -> ops.PackedConv:
        """bn_x(conv_x(x)) + bn_h(conv_h(h)) as one bias-free 1x1x1 conv over [x ; h]: each branch's BN scale goes into
        its weight rows, the shifts add up; the packed conv then carries scale = 1 (gamma 1, var 1, eps 0)."""
        def fold(bn):
            scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
            return scale, bn.bias.detach() - bn.running_mean.detach() * scale

        sx, bx = fold(bn_x)
        sh, bh = fold(bn_h)
        w = torch.cat([conv_x.weight.detach() * sx.view(-1, 1, 1, 1, 1), conv_h.weight.detach() * sh.view(-1, 1, 1, 1, 1)], dim=1).contiguous()
        one, zero = torch.ones_like(sx), torch.zeros_like(sx)
        return ops.pack_conv(w, one, bx + bh, zero, one, 0.0, (1, 1, 1), (0, 0, 0), name=name,
                             algo=self.algo_overrides.get(name, _lib.ALGO_AUTO))

    def prepare(self, force: bool = False) -> None:
        """Fold BN and pack weights into the kernels' layout (once; redone if parameters change)."""
        stamp = self._stamp()
        if not force and self._plan is not None and stamp == self._plan_stamp:
            return
        dev = self.conv1.weight.device
        if dev.type != "cuda":
            raise _lib.HipExtensionError(
                f"I3Res50 parameters are on '{dev}': the backbone runs only as HIP kernels on an AMD GPU "
                "(call .cuda()); there is no CPU fallback"
            )
        _lib.load()
        plan: List[_Unit] = [_Unit("stem", self._pack(self.conv1, self.bn1, "conv1"))]
        plan.append(_Unit("maxpool", kernel=(2, 3, 3), stride=(2, 2, 2)))
        for lname in ("layer1", "layer2", "layer3", "layer4"):
            for bi, blk in enumerate(getattr(self, lname)):
                p = f"{lname}.{bi}"
                ds = None
                if (self.fuse_downsample and blk.downsample is not None and tuple(blk.downsample[0].stride) == (1, 1, 1)
                        and tuple(blk.conv2.stride) == (1, 1, 1) and plan[-1].kind == "maxpool"):
                    # layer1.0: the downsample branch sees the same positions as conv3 (stride 1), so
                    # conv3(h) + downsample(x) is ONE 1x1x1 conv over the channel concatenation [x ; h] with the two
                    # BN scales folded into the weights.  The producers write straight into the two halves of one
                    # buffer (the pool: channels [0, inplanes), conv2: the rest): no torch.cat, and the 64->256
                    # downsample output (396 MB at B=32) is neither written nor read back as a residual.
                    plan[-1].cat_channels = blk.conv3.in_channels
                    plan.append(_Unit(
                        "bottleneck",
                        self._pack(blk.conv1, blk.bn1, f"{p}.conv1"),
                        self._pack(blk.conv2, blk.bn2, f"{p}.conv2"),
                        self._pack_fused(blk.downsample[0], blk.downsample[1], blk.conv3, blk.bn3, f"{p}.conv3+downsample"),
                        None, name=p, cat=blk.downsample[0].in_channels,
                    ))
                    continue
                if blk.downsample is not None:
                    ds = self._pack(blk.downsample[0], blk.downsample[1], f"{p}.downsample")
                plan.append(_Unit(
                    "bottleneck",
                    self._pack(blk.conv1, blk.bn1, f"{p}.conv1"),
                    self._pack(blk.conv2, blk.bn2, f"{p}.conv2"),
                    self._pack(blk.conv3, blk.bn3, f"{p}.conv3"),
                    ds, name=p if blk.nl is None else "",
                ))
                if blk.nl is not None:  # Bottleneck.forward applies the non-local block last (src/i3d.py:118-119)
                    blk.nl.prepare(f"{p}.nl")
                    plan.append(_Unit("nonlocal", name=p, nl=blk.nl))
            if lname == "layer1":
                plan.append(_Unit("maxpool", kernel=(2, 1, 1), stride=(2, 1, 1)))
        plan.append(_Unit("avgpool"))
        # pooling fused into the producing conv: the maxpool unit stays in the plan (shape bookkeeping, the un-fused
        # form used for per-stage taps) and is skipped when its producer has pooled already
        for prev, u in zip(plan, plan[1:]):
            if u.kind != "maxpool":
                continue
            if prev.kind == "stem" and (u.kernel, u.stride) == ((2, 3, 3), (2, 2, 2)):
                prev.pool_unit, u.absorbed = u, True
            elif (prev.kind == "bottleneck" and not prev.cat and (u.kernel, u.stride) == ((2, 1, 1), (2, 1, 1))
                  and prev.convs[2].kernel == (1, 1, 1) and prev.convs[2].stride == (1, 1, 1) and prev.convs[2].cin % 32 == 0):
                prev.pool_unit, u.absorbed = u, True
        # ... and the global mean behind the last bottleneck into its conv3 (src/i3d.py:314): skipped the same way
        if len(plan) >= 2 and plan[-1].kind == "avgpool" and plan[-2].kind == "bottleneck" and not plan[-2].cat and plan[-2].pool_unit is None:
            plan[-2].pool_unit, plan[-1].absorbed = plan[-1], True
        self._plan, self._plan_stamp = plan, stamp

    def packed_convs(self) -> List[ops.PackedConv]:
        self.prepare()
        out = []
        for u in self._plan:
            out.extend(c for c in u.convs if c is not None)
        return out

    # ------------------------------------------------------------------ forward
    def forward_single(self, x: torch.Tensor, taps: Optional[Dict[str, torch.Tensor]] = None,
                       events: Optional[List] = None) -> torch.Tensor:
        """`taps`: filled with every unit's output (tests).  `events`: if a list, HIP events
        (recorded on the launch stream) are appended as [start, (pool_start, pool_end) x3, end] so a
        caller can time the conv stack = (start..end) minus the pool launches (bench.py roofline)."""
        if self.training:
            raise _lib.HipExtensionError(
                "I3Res50 HIP path implements eval-mode BatchNorm (running statistics) only; call .eval() "
                "as extract_features.load_feature_extraction_model does (extract_features.py:36)"
            )
        if x.dim() != 5 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected (B,{self.in_channels},T,H,W), got {tuple(x.shape)}")
        if x.dtype != torch.float32:
            raise _lib.HipExtensionError(f"input dtype {x.dtype}: the backbone computes in fp32")
        if not x.is_cuda:
            raise _lib.HipExtensionError("input is not on the GPU; there is no CPU fallback")
        self.prepare()
        if x.shape[0] == 0:  # empty batch: nothing to launch (torch semantics: empty output)
            return torch.empty((0, 2048, 1, 1, 1), device=x.device, dtype=torch.float32)
        with torch.no_grad():
            x = x.detach().contiguous()

            def mark():
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                events.append(e)

            if events is not None:
                mark()
            n = self._n_streams(x.shape[0]) if taps is None else 1
            fused = self.fuse_pool and taps is None  # taps want every stage's own output: the un-fused launches
            if n > 1:
                x = self._run_streams(x, n)
            else:
                for u in self._plan:
                    if fused and u.absorbed:
                        continue
                    pool = events is not None and u.kind in ("maxpool", "avgpool")
                    if pool:
                        mark()
                    x = u.run(x, fused)
                    if pool:
                        mark()
                    if taps is not None and u.name:
                        taps[u.name] = x[:, : x.shape[1] - u.cat_channels] if u.kind == "maxpool" and u.cat_channels else x
            if events is not None:
                mark()
        return x

    def frames_fused(self) -> bool:
        """True when forward_frames feeds the stem kernel with uint8 pixels directly (stem + maxpool1 fused and present)."""
        self.prepare()
        u = self._plan[0]
        return bool(self.fuse_pool and u.kind == "stem" and u.pool_unit is not None and tuple(u.pool_unit.kernel) == (2, 3, 3)
                    and tuple(u.pool_unit.stride) == (2, 2, 2))

    def ensure_frame_tables(self, frame_hw: Tuple[int, int], frames_per_clip: int = 16, crop: int = 224, batch: Optional[int] = None) -> None:
        """ensure_tables for forward_frames: every lazily built table, on the current stream, before streams fork."""
        self.ensure_tables((frames_per_clip, crop, crop), batch)
        if self.frames_fused():
            pc = self._plan[0].convs[0]
            if self._frames_planes(crop):
                ops.ensure_ktab_s2w(pc, (frames_per_clip, crop, crop))
                return
            build = ops.ensure_u8_taps_tables if ops.U8_STEM_FORM in ("taps", "planes") and pc.cin == 3 and pc.cout == 64 else ops.ensure_u8_tables
            build(pc, tuple(frame_hw), (frames_per_clip, crop, crop))

    def _frames_planes(self, crop: int) -> bool:
        """forward_frames goes through column-parity planes (TenCrop pass writing them + the 16-byte-gather stem)."""
        return ops.U8_STEM_FORM == "planes" and ops.s2w_ok(self._plan[0].convs[0], crop)

    def forward_frames(self, frames: torch.Tensor, first: int, count: int, frames_per_clip: int = 16, crop: int = 224) -> torch.Tensor:
        """Features (count, 2048, 1, 1, 1) of crop-clips [first, first + count) of a video given as resized uint8 frames
        (F, FH, FW, 3), F whole clips; row = clip * 10 + crop in TenCrop order.  What the reference does on the host per clip
        -- GroupTenCrop, ToTensor, GroupNormalize, the (T,C)->(C,T) permute (src/dataset.py:175-195, src/gtransforms.py:29-38,
        57-73, extract_features.py:83-89) -- happens in the load stage of the stem kernel: the fp32 ten-crop tensor never exists."""
        if self.training:
            raise _lib.HipExtensionError("I3Res50 HIP path implements eval-mode BatchNorm only; call .eval()")
        if self.in_channels != 3:
            raise _lib.HipExtensionError("forward_frames takes RGB frames: the backbone was built with in_channels != 3")
        if frames.dtype != torch.uint8 or frames.dim() != 4 or not frames.is_cuda:
            raise _lib.HipExtensionError(f"forward_frames wants uint8 (F,H,W,3) frames on the GPU, got {frames.dtype} {tuple(frames.shape)} on {frames.device}")
        self.prepare()
        with torch.no_grad():
            if not self.frames_fused():  # other stems / ADV_I3D_FUSE_POOL=0: TenCrop + normalise as its own HIP pass
                from . import mil_ops

                return self.forward_single(mil_ops.tencrop_normalize_u8(frames, frames_per_clip, crop)[first : first + count])
            stem, pu = self._plan[0], self._plan[0].pool_unit
            if self._frames_planes(crop):
                # one pass: TenCrop + float + normalise + permutes, written as column-parity planes; then the stem with 16-byte
                # gather pieces.  Same arithmetic per pixel as the fp32 pipeline: the features equal model(tencrop_normalize_u8(..)).
                xs = ops.tencrop_planes_u8(frames, first, count, frames_per_clip, crop)
                stem_fn = lambda out=None: ops.conv3d_s2w_bn_relu_maxpool233(xs, stem.convs[0], out=out)
            else:
                stem_fn = lambda out=None: ops.conv3d_u8_tencrop_bn_relu_maxpool233(frames, stem.convs[0], first, count, frames_per_clip, crop, out=out)
            if pu.cat_channels:  # straight into the [x ; h] buffer of layer1.0, like _Unit.run
                d = ops.conv_pool_out_dims((frames_per_clip, crop, crop), stem.convs[0], pu.kernel, pu.stride)
                x = torch.empty((count, stem.convs[0].cout + pu.cat_channels) + d, device=frames.device, dtype=torch.float32)
                stem_fn(x[:, : stem.convs[0].cout])
            else:
                x = stem_fn()
            for u in self._plan[1:]:
                if not u.absorbed:
                    x = u.run(x, True)
        return x

    def ensure_tables(self, thw: Tuple[int, int, int], batch: Optional[int] = None) -> None:
        """Pack the weights and build every conv's gather table for clips of dims (T,H,W) on the CURRENT stream;
        with `batch`, also the split-bf16 weight images of every conv whose resolved choice for (batch,T,H,W) is a
        split-bf16 kernel (ADV_ARITH=mixed / bf16x3).  All of these are created lazily on whatever stream first
        needs them; a caller that is about to run forwards on other streams (stream parts, pipeline lanes) calls
        this first and orders those streams after it."""
        self.prepare()
        dims = tuple(thw)
        for u in self._plan:
            if u.kind == "maxpool":
                dims = ops.conv_out_dims(dims, u.kernel, u.stride, (0, 0, 0))
            elif u.kind == "nonlocal":
                theta_pc, phig_pc, out_pc = u.nl.prepare(u.name + ".nl")
                ops.ensure_ktab(theta_pc, dims, batch)
                ops.ensure_ktab(phig_pc, ops.conv_out_dims(dims, (1, 2, 2), (1, 2, 2), (0, 0, 0)), batch)
                ops.ensure_ktab(out_pc, dims, batch)
            elif u.kind in ("stem", "bottleneck"):
                if u.kind == "bottleneck" and u.convs[3] is not None:
                    ops.ensure_ktab(u.convs[3], dims, batch)  # the downsample branch reads the unit's input
                if u.kind == "stem" and ops.STEM_S2W and ops.s2w_ok(u.convs[0], dims[2]):
                    ops.ensure_ktab_s2w(u.convs[0], dims)  # the fused stem's column-parity gather table
                for c in u.convs[:3]:
                    ops.ensure_ktab(c, dims, batch)
                    dims = ops.conv_out_dims(dims, c.kernel, c.stride, c.padding)

    def _n_streams(self, batch: int) -> int:
        """Streams a forward of `batch` crop-clips is spread over (each part keeps >= MIN_PART clips)."""
        return max(1, min(self.streams, batch // self.MIN_PART))

    def _run_streams(self, x: torch.Tensor, n: int) -> torch.Tensor:
        """The whole plan with the batch cut in `n` parts, one HIP stream each.  Crop-clips are independent
        (extract_features.py:85-89), and the net alternates matrix-pipe-bound convs with HBM-bound launches
        (the K=64 `64->256` convs of layer1, pools, split-K reduces): with two half-batches in flight one
        part's memory-bound launches run beside the other's MFMA-bound ones and kernel tails overlap the next
        kernel's head.  Measured at B=32: 10.73 -> 10.12 ms per step with 2 streams (4: 10.58, 8: 13.2) on the first LDS-DMA kernels; 9.97 -> 9.71
        with the final ones."""
        B = x.shape[0]
        bounds = [(B * i) // n for i in range(n + 1)]
        for part_b in sorted({bounds[i + 1] - bounds[i] for i in range(n)}):
            self.ensure_tables(tuple(x.shape[2:]), part_b)
        main = torch.cuda.current_stream(x.device)
        while len(self._side_streams) < n - 1:
            self._side_streams.append(torch.cuda.Stream(device=x.device))
        fork = torch.cuda.Event()
        fork.record(main)

        def chain(part: torch.Tensor) -> torch.Tensor:
            for u in self._plan:
                if not (self.fuse_pool and u.absorbed):
                    part = u.run(part, self.fuse_pool)
            return part

        parts = [chain(x[bounds[0]:bounds[1]])]
        for i in range(1, n):
            side = self._side_streams[i - 1]
            with torch.cuda.stream(side):
                side.wait_event(fork)
                parts.append(chain(x[bounds[i]:bounds[i + 1]]))
                join = torch.cuda.Event()
                join.record(side)
            main.wait_event(join)
            parts[-1].record_stream(main)
        return torch.cat(parts, dim=0)

    def forward(self, batch):
        return self.forward_single(batch)


class _Unit:
    """One step of the flat execution plan."""

    def __init__(self, kind, *convs, name: str = "", kernel=None, stride=None, cat: int = 0, nl=None):
        self.kind = kind
        self.convs = convs
        self.name = name or (kind if kind != "bottleneck" else "")
        self.nl = nl
        self.kernel, self.stride = kernel, stride
        self.cat = cat            # bottleneck: > 0 = input is the [x ; h] buffer, x = its first `cat` channels
        self.cat_channels = 0     # maxpool: > 0 = allocate that many extra channels behind the pooled ones
        self.pool_unit: Optional["_Unit"] = None  # stem / bottleneck: the maxpool unit that follows and can be fused into this one
        self.absorbed = False     # maxpool: its producer can pool in its own epilogue

    def run(self, x: torch.Tensor, fused: bool = False) -> torch.Tensor:
        fused = fused and self.pool_unit is not None
        if self.kind == "stem":
            if fused:  # conv1 + bn1 + relu + maxpool1 (src/i3d.py:303-306), straight into the [x ; h] buffer of layer1.0
                pu = self.pool_unit
                if pu.cat_channels:
                    d = ops.conv_pool_out_dims(tuple(x.shape[2:]), self.convs[0], pu.kernel, pu.stride)
                    wide = torch.empty((x.shape[0], self.convs[0].cout + pu.cat_channels) + d, device=x.device, dtype=torch.float32)
                    ops.conv3d_bn_relu_maxpool233(x, self.convs[0], out=wide[:, : self.convs[0].cout])
                    return wide
                return ops.conv3d_bn_relu_maxpool233(x, self.convs[0])
            return ops.conv3d_bn_act(x, self.convs[0], relu=True)
        if self.kind == "maxpool":
            if self.cat_channels:
                to, ho, wo = ops.conv_out_dims(tuple(x.shape[2:]), self.kernel, self.stride, (0, 0, 0))
                wide = torch.empty((x.shape[0], x.shape[1] + self.cat_channels, to, ho, wo), device=x.device, dtype=torch.float32)
                ops.maxpool3d(x, self.kernel, self.stride, out=wide[:, : x.shape[1]])
                return wide
            return ops.maxpool3d(x, self.kernel, self.stride)
        if self.kind == "avgpool":
            return ops.global_avgpool(x)
        if self.kind == "nonlocal":
            return self.nl.run(x, self.name + ".nl")
        c1, c2, c3, ds = self.convs
        if self.cat:  # x is the wide buffer [x ; room for h]: conv2 writes h into its second half, c3 is the fused conv
            h = ops.conv3d_bn_act(x[:, : self.cat], c1, relu=True)
            ops.conv3d_bn_act(h, c2, relu=True, out=x[:, self.cat :])
            return ops.conv3d_bn_act(x, c3, relu=True)
        h = ops.conv3d_bn_act(x, c1, relu=True)
        h = ops.conv3d_bn_act(h, c2, relu=True)
        res = ops.conv3d_bn_act(x, ds, relu=False) if ds is not None else x
        if fused and self.pool_unit.kind == "avgpool":  # conv3 + bn3 + residual + relu + avgpool (src/i3d.py:111-121, 314) in one launch
            if ops.FUSE_AVGPOOL and ops.avgpool_fusable(c3, tuple(h.shape[2:])):
                return ops.conv3d_bn_act_avgpool(h, c3, relu=True, residual=res)
            return ops.global_avgpool(ops.conv3d_bn_act(h, c3, relu=True, residual=res))  # (the pool unit itself is skipped)
        if fused:  # conv3 + bn3 + residual + relu + maxpool2 (src/i3d.py:111-121, 309) in one launch
            return ops.conv3d_bn_act_maxpool211(h, c3, relu=True, residual=res)
        return ops.conv3d_bn_act(h, c3, relu=True, residual=res)


def print_model_size(model):
    bits = 0
    for p in model.parameters():
        info = torch.finfo(p.dtype) if p.is_floating_point() else torch.iinfo(p.dtype)
        bits += p.numel() * info.bits
    print(f"model size: {bits} / bit | {bits / 8e6:.2f} / MB")


def build_i3d_feature_extractor(
    model_name: str = "tushar-n-baseline",
    check_model_size: bool = True,
    strict: bool = False,
    state_dict_path: Optional[str] = None,
):
    """Factory with the reference's signature (src/i3d.py:332-364) plus `state_dict_path`.

    Weight source, in order: `state_dict_path`, $ADV_I3D_WEIGHTS, the HF hub file the reference
    uses (needs network), and -- only when $ADV_I3D_SYNTHETIC=1 -- the deterministic synthetic
    weights of `weights.synth_i3d_state_dict` (benchmarks / tests; never silently).
    """
    if model_name == "tushar-n-baseline":
        model = I3Res50(use_nl=False)
    elif model_name == "i3d_8x8_r50":
        # pytorchvideo's create_resnet topology on the same kernels; PARITY UNPINNED (third-party arithmetic that is
        # neither vendored in the reference nor installed here): see i3d_ptv.py
        from .i3d_ptv import I3D8x8R50

        model = I3D8x8R50()
    else:
        raise AttributeError

    path = state_dict_path or os.environ.get("ADV_I3D_WEIGHTS")
    if path is None and os.environ.get("ADV_I3D_SYNTHETIC") == "1":
        from .weights import synth_i3d_state_dict, synth_module_state_dict

        sd = synth_i3d_state_dict() if model_name == "tushar-n-baseline" else synth_module_state_dict(model, gain=2.0)
    else:
        if path is None:
            from huggingface_hub import hf_hub_download

            path = hf_hub_download(repo_id=repo_id, filename=model_zoo[model_name])
        sd = torch.load(path, map_location="cpu")
    model.load_state_dict(state_dict=sd, strict=strict)
    if check_model_size:
        print_model_size(model)
    return model
