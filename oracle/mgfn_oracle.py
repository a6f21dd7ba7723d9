"""ORACLE (test infrastructure only) -- CPU restatement of the MGFN MIL scorer and its losses.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.

A functional (state-dict driven, autograd-friendly) restatement in plain fp32 torch ops of

    MGFNFeatureAmplifier        /root/reference/src/models/mgfn/modeling_mgfn.py:67-93
    MGFNLayerNorm (std+eps)     modeling_mgfn.py:36-46
    MGFNFeedForward             modeling_mgfn.py:49-64
    GlanceBlock/GlanceAttention modeling_mgfn.py:96-147
    FocusBlock/FocusAttention   modeling_mgfn.py:150-205
    MGFNIntermediate            modeling_mgfn.py:208-216
    head + magnitude_selection_and_score_prediction   modeling_mgfn.py:302-427
    TemporalSmoothnessLoss / SparsityLoss / ContrastiveLoss   /root/reference/src/loss/base.py:7-48
    MGFNLoss                    /root/reference/src/loss/mgfn.py:7-47

Pinned by `tests/golden/mgfn_*.npz`, produced by running the reference's own classes
(`tests/golden/make_golden.py`).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


@dataclass
class MGFNSpec:
    """Hyper-parameters, defaults = reference `MGFNConfig` (configuration_mgfn.py:5-21)."""

    dims: Sequence[int] = (64, 128, 1024)
    depths: Sequence[int] = (3, 3, 2)
    mgfn_types: Sequence[str] = ("gb", "fb", "fb")
    channels: int = 2048
    ff_repe: int = 4
    dim_head: int = 64
    local_aggr_kernel: int = 5
    dropout_rate: float = 0.7
    mag_ratio: float = 0.1
    k: int = 3


# ----------------------------------------------------------------------------- blocks
def chan_layer_norm(x, g, b, eps=1e-5):
    # modeling_mgfn.py:43-46 -- biased variance, divides by (std + eps), NOT sqrt(var+eps)
    std = torch.var(x, dim=1, unbiased=False, keepdim=True).sqrt()
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) / (std + eps) * g + b


def feed_forward(x, sd, p):
    # modeling_mgfn.py:58-64 (dropout p=0.0 in every shipped config -> identity)
    h = chan_layer_norm(x, sd[f"{p}.layer_norm.g"], sd[f"{p}.layer_norm.b"])
    h = F.conv1d(h, sd[f"{p}.in_conv.weight"], sd[f"{p}.in_conv.bias"])
    h = F.gelu(h)
    return F.conv1d(h, sd[f"{p}.out_conv.weight"], sd[f"{p}.out_conv.bias"])


def glance_attention(x, sd, p, heads, dim_head):
    # modeling_mgfn.py:107-123
    h = chan_layer_norm(x, sd[f"{p}.norm.g"], sd[f"{p}.norm.b"])
    b, _, n = h.shape
    qkv = F.conv1d(h, sd[f"{p}.to_qkv.weight"], None)
    q, k, v = qkv.chunk(3, dim=1)
    # "b (h d) n -> b h n d"
    q, k, v = (t.reshape(b, heads, dim_head, n).permute(0, 1, 3, 2) for t in (q, k, v))
    q = q * (dim_head ** -0.5)
    sim = torch.matmul(q, k.transpose(-1, -2))  # b h i j
    attn = sim.softmax(dim=-1)
    out = torch.matmul(attn, v)  # b h i d
    out = out.permute(0, 1, 3, 2).reshape(b, heads * dim_head, n)  # "b h n d -> b (h d) n"
    return F.conv1d(out, sd[f"{p}.to_out.weight"], sd[f"{p}.to_out.bias"])


def focus_attention(x, sd, p, heads, training, kernel):
    # modeling_mgfn.py:173-180
    h = F.batch_norm(
        x, sd[f"{p}.norm.running_mean"].detach().clone(), sd[f"{p}.norm.running_var"].detach().clone(),
        sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], training=training, momentum=0.1, eps=1e-5,
    )
    b, _, n = h.shape
    v = F.conv1d(h, sd[f"{p}.to_v.weight"], None)
    inner = v.shape[1]
    c = inner // heads
    # "b (c h) n -> (b c) h n": channel index = c_idx * heads + h_idx
    v = v.reshape(b, c, heads, n).reshape(b * c, heads, n)
    out = F.conv1d(v, sd[f"{p}.rel_pos.weight"], sd[f"{p}.rel_pos.bias"], padding=kernel // 2, groups=heads)
    out = out.reshape(b, c, heads, n).reshape(b, inner, n)
    return F.conv1d(out, sd[f"{p}.to_out.weight"], sd[f"{p}.to_out.bias"])


def backbone(video, sd, spec: MGFNSpec, training: bool, prefix="backbone"):
    # amplifier, modeling_mgfn.py:81-93
    bs, ncrops, t, c = video.shape
    x = video.reshape(bs * ncrops, t, c).permute(0, 2, 1)
    x_f, x_m = x[:, : spec.channels, :], x[:, spec.channels :, :]
    a = f"{prefix}.amplifier"
    x_f = F.conv1d(x_f, sd[f"{a}.to_tokens.weight"], sd[f"{a}.to_tokens.bias"], padding=1)
    x_m = F.conv1d(x_m, sd[f"{a}.to_mag.weight"], sd[f"{a}.to_mag.bias"], padding=1)
    x = x_f + spec.mag_ratio * x_m
    # stages, modeling_mgfn.py:242-267
    for s, (depth, kind) in enumerate(zip(spec.depths, spec.mgfn_types)):
        dim = spec.dims[s]
        heads = dim // spec.dim_head
        for i in range(depth):
            p = f"{prefix}.layers.{s}.{i}"
            x = F.conv1d(x, sd[f"{p}.scc.weight"], sd[f"{p}.scc.bias"], padding=1) + x
            if kind == "gb":
                x = glance_attention(x, sd, f"{p}.attention", heads, spec.dim_head) + x
            elif kind == "fb":
                x = focus_attention(x, sd, f"{p}.attention", heads, training, spec.local_aggr_kernel) + x
            else:
                raise AttributeError("The type of mgfn block must be either `gb` or `fb`.")
            x = feed_forward(x, sd, f"{p}.ffn") + x
        if s != len(spec.depths) - 1:
            p = f"{prefix}.layers.{s}.{depth}"
            x = chan_layer_norm(x, sd[f"{p}.layer_norm.g"], sd[f"{p}.layer_norm.b"])
            x = F.conv1d(x, sd[f"{p}.conv.weight"], sd[f"{p}.conv.bias"])
    return x  # (bs*ncrops, dims[-1], T)


# ----------------------------------------------------------------------------- MIL top-k
def mil_select(features, scores, batch_size, ncrops, k, split, keep_abn=None, keep_nor=None):
    """modeling_mgfn.py:302-374.

    features (bs*ncrops, T, F); scores (bs*ncrops, T, 1).  `keep_*` = post-dropout multiplier
    (n, T) (ones in eval; Bernoulli(1-p)/(1-p) in training, modeling_mgfn.py:342-345) -- passed
    in explicitly so both sides of a parity test use the same mask.
    Returns (score_abn, score_nor, feat_abn, feat_nor, scores(bs,T,1), idx_abn, idx_nor).
    """
    _, t, f = features.shape
    mag = torch.norm(features, p=2, dim=2).view(batch_size, ncrops, -1).mean(dim=1)
    sc = scores.view(batch_size, ncrops, -1).mean(dim=1).unsqueeze(2)
    if split:
        h = batch_size // 2
        nf, af = features[: h * ncrops], features[h * ncrops :]
        ns, as_ = sc[:h], sc[h:]
        nm, am = mag[:h], mag[h:]
    else:
        nf = af = features
        ns = as_ = sc
        nm = am = mag
    n = nm.shape[0]

    def select(m, feats, keep):
        if keep is None:
            keep = torch.ones_like(m)
        idx = torch.topk(m * keep, k, dim=1)[1]
        fe = feats.view(n, ncrops, t, f).permute(1, 0, 2, 3)  # crop-major
        gi = idx.unsqueeze(2).expand(-1, -1, f)
        sel = torch.cat([torch.gather(fc, 1, gi) for fc in fe], dim=0)  # (ncrops*n, k, f)
        return idx, sel

    def score_of(idx, s):
        return torch.gather(s, 1, idx.unsqueeze(2)).mean(dim=1)

    ia, fa = select(am, af, keep_abn)
    sa = score_of(ia, as_)
    in_, fn = select(nm, nf, keep_nor)
    sn = score_of(in_, ns)
    return sa, sn, fa, fn, sc, ia, in_


# ----------------------------------------------------------------------------- losses
def smoothness_loss(scores, lambda1=8e-4):
    # loss/base.py:16-18
    return lambda1 * torch.sum((scores[:, 1:, :] - scores[:, :-1, :]) ** 2)


def sparsity_loss(x, lambda2=8e-3):
    # loss/base.py:30-31 (x is 1-D at the only call site, modeling_mgfn.py:409)
    return lambda2 * torch.mean(torch.norm(x, dim=0))


def contrastive_loss(o1, o2, label, margin=200.0):
    # loss/base.py:42-48; pairwise_distance adds eps=1e-6 to the difference
    d = torch.sqrt(torch.sum((o1 - o2 + 1e-6) ** 2, dim=-1, keepdim=True))
    return torch.mean((1 - label) * d ** 2 + label * torch.clamp(margin - d, min=0.0) ** 2)


def mgfn_loss(abn_scores, nor_scores, a_feat, n_feat, abn_labels, nor_labels, alpha=0.001):
    # loss/mgfn.py:23-47
    labels = torch.cat((nor_labels, abn_labels), 0)
    scores = torch.cat((nor_scores, abn_scores), 0).squeeze()
    sep = int(len(n_feat) / 2)
    l1 = lambda v: torch.norm(v, p=1, dim=2)
    loss_cls = F.binary_cross_entropy(scores, labels)
    loss_con = contrastive_loss(l1(a_feat), l1(n_feat), 1)
    loss_con_n = contrastive_loss(l1(n_feat[sep:]), l1(n_feat[:sep]), 0)
    loss_con_a = contrastive_loss(l1(a_feat[sep:]), l1(a_feat[:sep]), 0)
    total = loss_cls + alpha * (alpha * loss_con + loss_con_a + loss_con_n)
    return total, dict(cls=loss_cls, con=loss_con, con_n=loss_con_n, con_a=loss_con_a)


# ----------------------------------------------------------------------------- whole model
@dataclass
class MGFNOut:
    loss: Optional[torch.Tensor]
    abnormal_scores: torch.Tensor
    normal_scores: torch.Tensor
    a_feat_magnitude: torch.Tensor
    n_feat_magnitude: torch.Tensor
    scores: torch.Tensor
    idx_abn: torch.Tensor
    idx_nor: torch.Tensor
    terms: Dict[str, torch.Tensor] = field(default_factory=dict)
    features: Optional[torch.Tensor] = None


def mgfn_forward(
    video: torch.Tensor,
    sd: Dict[str, torch.Tensor],
    spec: MGFNSpec = MGFNSpec(),
    abnormal_labels: Optional[torch.Tensor] = None,
    normal_labels: Optional[torch.Tensor] = None,
    training: bool = False,
    force_split: bool = False,
    keep_abn: Optional[torch.Tensor] = None,
    keep_nor: Optional[torch.Tensor] = None,
) -> MGFNOut:
    """modeling_mgfn.py:376-427."""
    bs, ncrops = video.shape[:2]
    x_f = backbone(video, sd, spec, training).permute(0, 2, 1)
    x = F.layer_norm(x_f, (spec.dims[-1],), sd["layer_norm.weight"], sd["layer_norm.bias"], eps=1e-5)
    scores = torch.sigmoid(F.linear(x, sd["fc.weight"], sd["fc.bias"]))
    sa, sn, fa, fn, sc, ia, in_ = mil_select(
        x, scores, bs, ncrops, spec.k, split=(force_split or training), keep_abn=keep_abn, keep_nor=keep_nor
    )
    loss, terms = None, {}
    if abnormal_labels is not None and normal_labels is not None:
        l_smooth = smoothness_loss(sc)
        l_sparse = sparsity_loss(sc[: bs // 2].reshape(-1))
        l_mgfn, terms = mgfn_loss(sa, sn, fa, fn, abnormal_labels, normal_labels)
        terms = dict(terms, smooth=l_smooth, sparse=l_sparse, mgfn=l_mgfn)
        loss = l_mgfn + l_smooth + l_sparse
    return MGFNOut(loss, sa, sn, fa, fn, sc, ia, in_, terms, x)
