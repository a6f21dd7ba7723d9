"""ORACLE (test infrastructure only) -- CPU restatement of the I3D-ResNet50 forward.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this module; the shipped path (`anomaly_detection_on_video_amd.i3d`) never does.

This is a *functional* restatement (no module tree) of the reference's arithmetic in plain
fp32 torch CPU ops, driven directly by a state dict with the reference's key layout:

    stem      conv1 k(5,7,7) s2 p(2,3,3) -> bn1 -> relu        /root/reference/src/i3d.py:202-211, 303-305
    maxpool1  k(2,3,3) s2 p0                                   src/i3d.py:212-214, 306
    layer1..4 Bottleneck x [3,4,6,3]                           src/i3d.py:220-243, 308-312
      conv1 k(1+2*tc,1,1) p(tc,0,0) -> bn -> relu              src/i3d.py:67-75, 101-103
      conv2 k(1,3,3) s(1,s,s) p(0,1,1) -> bn -> relu           src/i3d.py:76-84, 105-107
      conv3 k1 x4 -> bn ; (+ downsample(x) = conv k1 s(1,s,s) + bn) ; += ; relu   src/i3d.py:85-88, 109-116, 262-272
    maxpool2  k(2,1,1) s(2,1,1) between layer1 and layer2      src/i3d.py:215-217, 309
    avgpool   AdaptiveAvgPool3d(1)                             src/i3d.py:244, 314

Parity pinning: `tests/golden/make_golden.py` runs the *reference's own* `I3Res50`
(imported from /root/reference with the absent third-party `pytorchvideo` import stubbed)
on the same deterministic weights/inputs and commits its outputs under `tests/golden/`;
`tests/test_oracle_golden.py` checks this restatement against those vectors.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm3d default, src/i3d.py:75

# (name, planes, spatial stride, temporal-conv flags)  src/i3d.py:220-243
STAGES: List[Tuple[str, int, int, List[int]]] = [
    ("layer1", 64, 1, [1, 1, 1]),
    ("layer2", 128, 2, [1, 0, 1, 0]),
    ("layer3", 256, 2, [1, 0, 1, 0, 1, 0]),
    ("layer4", 512, 2, [0, 1, 0]),
]


def _bn(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str) -> torch.Tensor:
    return F.batch_norm(
        x, sd[f"{p}.running_mean"], sd[f"{p}.running_var"], sd[f"{p}.weight"], sd[f"{p}.bias"],
        training=False, eps=BN_EPS,
    )


def bottleneck(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str, stride: int, temp_conv: int, has_ds: bool) -> torch.Tensor:
    out = F.conv3d(x, sd[f"{p}.conv1.weight"], None, stride=1, padding=(temp_conv, 0, 0))
    out = F.relu(_bn(out, sd, f"{p}.bn1"))
    out = F.conv3d(out, sd[f"{p}.conv2.weight"], None, stride=(1, stride, stride), padding=(0, 1, 1))
    out = F.relu(_bn(out, sd, f"{p}.bn2"))
    out = F.conv3d(out, sd[f"{p}.conv3.weight"], None)
    out = _bn(out, sd, f"{p}.bn3")
    if has_ds:
        res = F.conv3d(x, sd[f"{p}.downsample.0.weight"], None, stride=(1, stride, stride))
        res = _bn(res, sd, f"{p}.downsample.1")
    else:
        res = x
    return F.relu(out + res)


def nonlocal_block(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str) -> torch.Tensor:
    """NonLocalBlock.forward, src/i3d.py:159-195: embedded-Gaussian attention of every position over the (1,2,2)-pooled
    positions; theta/phi/g/out are 1x1x1 convs WITH bias (nn.Conv3d default), `out` is followed by eval BatchNorm, the
    residual is added without a ReLU."""
    b = x.shape[0]
    inner = sd[f"{p}.theta.weight"].shape[0]
    mp = F.max_pool3d(x, kernel_size=(1, 2, 2), stride=(1, 2, 2))                       # :163
    theta = F.conv3d(x, sd[f"{p}.theta.weight"], sd[f"{p}.theta.bias"])                  # :164
    phi = F.conv3d(mp, sd[f"{p}.phi.weight"], sd[f"{p}.phi.bias"])                       # :165
    g = F.conv3d(mp, sd[f"{p}.g.weight"], sd[f"{p}.g.bias"])                             # :166
    shape5 = theta.shape
    theta, phi, g = theta.reshape(b, inner, -1), phi.reshape(b, inner, -1), g.reshape(b, inner, -1)
    att = torch.bmm(theta.transpose(1, 2), phi) * (inner ** -0.5)                        # :171-174
    att = F.softmax(att, dim=-1)                                                         # :175
    t = torch.bmm(g, att.transpose(1, 2)).reshape(shape5)                                # :178-179
    out = F.conv3d(t, sd[f"{p}.out.weight"], sd[f"{p}.out.bias"])                        # :181
    return _bn(out, sd, f"{p}.bn") + x                                                   # :182-184


@torch.no_grad()
def i3d_forward(
    x: torch.Tensor,
    sd: Dict[str, torch.Tensor],
    tap: Optional[Callable[[str, torch.Tensor], None]] = None,
) -> torch.Tensor:
    """(B,3,T,H,W) fp32 -> (B,2048,1,1,1).  `tap(name, tensor)` sees every stage output.  Blocks whose state dict
    carries `<block>.nl.*` entries (I3Res50(use_nl=True)) are followed by their NonLocalBlock (src/i3d.py:118-119)."""
    t = tap or (lambda n, v: None)
    x = F.conv3d(x, sd["conv1.weight"], None, stride=(2, 2, 2), padding=(2, 3, 3))
    x = F.relu(_bn(x, sd, "bn1"))
    t("stem", x)
    x = F.max_pool3d(x, kernel_size=(2, 3, 3), stride=(2, 2, 2))
    t("maxpool1", x)
    for name, _planes, stride, temps in STAGES:
        for i, tc in enumerate(temps):
            x = bottleneck(x, sd, f"{name}.{i}", stride if i == 0 else 1, tc, i == 0)
            if f"{name}.{i}.nl.theta.weight" in sd:
                x = nonlocal_block(x, sd, f"{name}.{i}.nl")
            t(f"{name}.{i}", x)
        t(name, x)
        if name == "layer1":
            x = F.max_pool3d(x, kernel_size=(2, 1, 1), stride=(2, 1, 1))
            t("maxpool2", x)
    x = F.adaptive_avg_pool3d(x, 1)
    t("avgpool", x)
    return x


@torch.no_grad()
def conv_bn_act(
    x: torch.Tensor,
    w: torch.Tensor,
    gamma: torch.Tensor,
    beta: torch.Tensor,
    mean: torch.Tensor,
    var: torch.Tensor,
    stride: Tuple[int, int, int],
    padding: Tuple[int, int, int],
    residual: Optional[torch.Tensor] = None,
    relu: bool = True,
) -> torch.Tensor:
    """One conv -> eval-BN (-> +residual) (-> relu) unit, the granularity of the HIP kernels."""
    y = F.conv3d(x, w, None, stride=stride, padding=padding)
    y = F.batch_norm(y, mean, var, gamma, beta, training=False, eps=BN_EPS)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def conv_macs(in_shape: Tuple[int, int, int, int, int] = (1, 3, 16, 224, 224)) -> int:
    """Multiply-accumulates of all 53 convs for one forward (16.415 G per crop-clip at 16x224^2)."""
    b, c, t, h, w = in_shape
    total = 0

    def conv(cin, cout, k, s, p, t, h, w):
        nonlocal total
        to = (t + 2 * p[0] - k[0]) // s[0] + 1
        ho = (h + 2 * p[1] - k[1]) // s[1] + 1
        wo = (w + 2 * p[2] - k[2]) // s[2] + 1
        total += b * cout * to * ho * wo * cin * k[0] * k[1] * k[2]
        return to, ho, wo

    t, h, w = conv(3, 64, (5, 7, 7), (2, 2, 2), (2, 3, 3), t, h, w)
    t, h, w = (t - 2) // 2 + 1, (h - 3) // 2 + 1, (w - 3) // 2 + 1
    inpl = 64
    for name, planes, stride, temps in STAGES:
        for i, tc in enumerate(temps):
            s = stride if i == 0 else 1
            conv(inpl, planes, (1 + 2 * tc, 1, 1), (1, 1, 1), (tc, 0, 0), t, h, w)
            t2, h2, w2 = conv(planes, planes, (1, 3, 3), (1, s, s), (0, 1, 1), t, h, w)
            conv(planes, planes * 4, (1, 1, 1), (1, 1, 1), (0, 0, 0), t2, h2, w2)
            if i == 0:
                conv(inpl, planes * 4, (1, 1, 1), (1, s, s), (0, 0, 0), t, h, w)
            t, h, w = t2, h2, w2
            inpl = planes * 4
        if name == "layer1":
            t = (t - 2) // 2 + 1
    return total


# ---- `i3d_8x8_r50` (pytorchvideo create_resnet as called at src/i3d.py:339-350): PARITY UNPINNED ------------------------
# pytorchvideo 0.1.3 is third-party, not vendored in the reference and not installed here; this restates its published
# topology (see anomaly_detection_on_video_amd/i3d_ptv.py's header) in plain torch ops with pytorchvideo's state-dict
# keys.  It pins the HIP kernels on that topology, not pytorchvideo itself.
PTV_CONV_A_KT = ((3, 3, 3), (3, 1, 3, 1), (3, 1, 3, 1, 3, 1), (1, 3, 1))


@torch.no_grad()
def ptv_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    def bn(v, p):
        return F.batch_norm(v, sd[f"{p}.running_mean"], sd[f"{p}.running_var"], sd[f"{p}.weight"], sd[f"{p}.bias"], training=False, eps=BN_EPS)

    x = F.relu(bn(F.conv3d(x, sd["blocks.0.conv.weight"], None, stride=(1, 2, 2), padding=(2, 3, 3)), "blocks.0.norm"))
    x = F.max_pool3d(x, (1, 3, 3), (1, 2, 2), padding=(0, 1, 1))
    bi = 1
    for si, kts in enumerate(PTV_CONV_A_KT):
        for ri, kt in enumerate(kts):
            p = f"blocks.{bi}.res_blocks.{ri}"
            s = 2 if (si > 0 and ri == 0) else 1
            h = F.relu(bn(F.conv3d(x, sd[f"{p}.branch2.conv_a.weight"], None, padding=(kt // 2, 0, 0)), f"{p}.branch2.norm_a"))
            h = F.relu(bn(F.conv3d(h, sd[f"{p}.branch2.conv_b.weight"], None, stride=(1, s, s), padding=(0, 1, 1)), f"{p}.branch2.norm_b"))
            h = bn(F.conv3d(h, sd[f"{p}.branch2.conv_c.weight"], None), f"{p}.branch2.norm_c")
            if f"{p}.branch1_conv.weight" in sd:
                x = bn(F.conv3d(x, sd[f"{p}.branch1_conv.weight"], None, stride=(1, s, s)), f"{p}.branch1_norm")
            x = F.relu(h + x)
        bi += 1
        if si == 0:
            x = F.max_pool3d(x, (2, 1, 1), (2, 1, 1))
            bi += 1
    return F.adaptive_avg_pool3d(F.avg_pool3d(x, (4, 7, 7), stride=1), 1)
