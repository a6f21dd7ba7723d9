"""MGFN multiple-instance-learning scorer: PyTorch-ROCm body, HIP head and losses.

Drop-in for `/root/reference/src/models/mgfn/modeling_mgfn.py`: same class names, constructor
(`MGFNForVideoAnomalyDetection(config)`), `forward(video, abnormal_labels, normal_labels)`,
output dataclass fields, `force_split` property and state-dict keys (145 entries).

What runs where
  * backbone (amplifier, Glance/Focus blocks: Conv1d / attention / GELU, with autograd): the fp32-MFMA GEMM kernels and fused
    norm / attention / depth-wise-conv launches of csrc/conv_igemm.hip + csrc/mgfn.hip through `mgfn_ops` (see "Body layout" below);
  * the MIL head `magnitude_selection_and_score_prediction` (modeling_mgfn.py:302-374) and the
    four loss terms (loss/base.py, loss/mgfn.py): hand-written wavefront-shuffle HIP kernels with
    custom backward (`mil_ops.py`; csrc/mil.hip, csrc/loss.hip).
CUDA tensors only for the head/loss: there is no CPU fallback (calls raise on CPU tensors).
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from ... import mgfn_ops, mil_ops
from .configuration_mgfn import MGFNConfig

try:
    from transformers import PreTrainedModel as _PreTrainedModel
    from transformers.utils import ModelOutput as _ModelOutput
except Exception:  # pragma: no cover - transformers absent: minimal stand-ins
    _PreTrainedModel = None
    _ModelOutput = None


if _ModelOutput is not None:
    @dataclass
    class MGFNModelOutput(_ModelOutput):
        outputs: torch.FloatTensor = None

    @dataclass
    class MGFNVideoAnomalyDetectionOutput(_ModelOutput):
        loss: torch.FloatTensor = None
        abnormal_scores: torch.FloatTensor = None
        normal_scores: torch.FloatTensor = None
        a_feat_magnitude: torch.FloatTensor = None
        n_feat_magnitude: torch.FloatTensor = None
        scores: torch.FloatTensor = None
else:  # pragma: no cover
    @dataclass
    class MGFNModelOutput:
        outputs: torch.Tensor = None

    @dataclass
    class MGFNVideoAnomalyDetectionOutput:
        loss: torch.Tensor = None
        abnormal_scores: torch.Tensor = None
        normal_scores: torch.Tensor = None
        a_feat_magnitude: torch.Tensor = None
        n_feat_magnitude: torch.Tensor = None
        scores: torch.Tensor = None


# ------------------------------------------------------------------------------------------------
# Body layout.  The reference feeds (B, C, T) tensors through nn.Conv1d.  Every convolution of the scorer is a GEMM over channels
# (k = 1) or over (tap, channel) (k = 3), so the body keeps activations as (C, B, T) -- channels outermost, N = B*T positions
# contiguous -- and every GEMM-shaped layer is Y[o, n] = W[o, c] X[c, n] with X in exactly the LDS-DMA conv kernel's A layout: no
# layout copies between layers.  The nn.Conv1d / nn.BatchNorm1d children remain the parameter holders (the reference's state-dict
# keys); their own forward is not used.
#
# Every layer of the default architecture runs on the hand-written fp32-MFMA kernels through `mgfn_ops`, forward AND backward, at
# any T: the conv kernel as a GEMM with bias / GELU / residual / LayerNorm-fold / GELU-backward epilogues, advhip_gemm_nt_f32 for the
# weight gradients, the 2048 -> 64 token conv on the input rows as stored (_TokenTaps), GlanceAttention's core as one launch (T = 32:
# everything in LDS; any other T: key tiles + online softmax), eval-mode BatchNorm1d folded into to_v's operand, the norms / the
# depth-wise conv / the head as single fused launches.  The torch expressions below (`_pointwise_torch`, `_conv_k_torch`, the einsum
# attention, `var_mean` norms) remain only for layers outside the kernels' shape rules (mgfn_ops.eligible: channel counts that are
# not multiples of 64 / 32, dim_head != 64, B*T % 16 != 0 with autograd, tensors off the current device, eval-mode BatchNorm with
# autograd); each announces itself through `mgfn_ops.torch_path`, which raises under ADV_MGFN_STRICT=1 (tests/test_hip_strict.py).
# ------------------------------------------------------------------------------------------------
def _hip(conv: nn.Conv1d, x: torch.Tensor) -> bool:
    return conv.weight.shape[2] in (1, 3) and mgfn_ops.eligible(conv.weight.shape[1], conv.weight.shape[0], x)


def _pointwise(conv: nn.Conv1d, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """1x1 Conv1d on a (C, B, T) tensor (+ residual)."""
    if _hip(conv, x):
        return mgfn_ops.linear_cn(x, conv, residual)
    y = _pointwise_torch(conv, x)
    return y if residual is None else y + residual


def _pointwise_torch(conv: nn.Conv1d, x: torch.Tensor) -> torch.Tensor:
    mgfn_ops.torch_path(x, f"1x1 Conv1d {conv.weight.shape[1]} -> {conv.weight.shape[0]}")
    c, b, t = x.shape
    y = torch.matmul(conv.weight[:, :, 0], x.reshape(c, b * t))
    if conv.bias is not None:
        y = y + conv.bias[:, None]
    return y.view(-1, b, t)


def _conv_k(conv: nn.Conv1d, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Conv1d with odd kernel k, padding k//2, stride 1 on a (C, B, T) tensor (+ residual): GEMM over (tap, channel)."""
    o, c, k = conv.weight.shape
    if k == 1 or _hip(conv, x):
        return _pointwise(conv, x, residual)
    y = _conv_k_torch(conv, x)
    return y if residual is None else y + residual


def _conv_k_torch(conv: nn.Conv1d, x: torch.Tensor) -> torch.Tensor:
    mgfn_ops.torch_path(x, f"k = {conv.weight.shape[2]} Conv1d {conv.weight.shape[1]} -> {conv.weight.shape[0]}")
    o, c, k = conv.weight.shape
    _, b, t = x.shape
    xp = F.pad(x, (k // 2, k // 2))
    xu = torch.cat([xp[:, :, j : j + t] for j in range(k)], dim=0)  # ((tap, c), B, T)
    w2 = conv.weight.permute(0, 2, 1).reshape(o, k * c)
    y = torch.matmul(w2, xu.reshape(k * c, b * t))
    if conv.bias is not None:
        y = y + conv.bias[:, None]
    return y.view(o, b, t)


class MGFNLayerNorm(nn.Module):
    """Channel-dim norm dividing by (std + eps) -- not sqrt(var + eps) (modeling_mgfn.py:43-46)."""

    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.g = nn.Parameter(torch.ones(1, dim, 1))
        self.b = nn.Parameter(torch.zeros(1, dim, 1))

    def forward(self, x):  # x: (C, B, T)
        if mgfn_ops.fused_ok(x):  # one HIP launch forward, one backward (csrc/mgfn.hip)
            return mgfn_ops.chan_layernorm(x, self.g, self.b, self.eps)
        mgfn_ops.torch_path(x, "channel LayerNorm")
        var, mean = torch.var_mean(x, dim=0, unbiased=False, keepdim=True)
        return (x - mean) / (var.sqrt() + self.eps) * self.g.view(-1, 1, 1) + self.b.view(-1, 1, 1)


class MGFNFeedForward(nn.Module):
    def __init__(self, dim: int, repe: int = 4, dropout: float = 0.0):
        super().__init__()
        self.layer_norm = MGFNLayerNorm(dim)
        self.in_conv = nn.Conv1d(dim, dim * repe, 1)
        self.gelu = nn.GELU()
        self.dropout = nn.Dropout(dropout)
        self.out_conv = nn.Conv1d(dim * repe, dim, 1)

    def forward(self, x, residual: Optional[torch.Tensor] = None):
        """ffn(x) (+ residual).  On the HIP path the two GEMMs carry bias, GELU (+ its backward) and the residual in their
        epilogues; without autograd the LayerNorm is folded into the first one as well."""
        if (_hip(self.in_conv, x) and _hip(self.out_conv, x) and residual is not None
                and not (self.training and self.dropout.p > 0)):
            if not torch.is_grad_enabled():
                if residual is x:
                    return mgfn_ops.ffn_cn_folded_ln(x, self.layer_norm, self.in_conv, self.out_conv)
            elif residual is x and mgfn_ops.fused_ok(x) and self.in_conv.bias is not None and self.out_conv.bias is not None:
                return mgfn_ops.ffn_block_cn(x, self.layer_norm, self.in_conv, self.out_conv)  # LN + FFN + skip: one autograd node
            return mgfn_ops.ffn_cn(self.layer_norm(x), residual, self.in_conv, self.out_conv)
        mgfn_ops.torch_path(x, "FFN off the fused GEMM pair (channel rules, no residual, or dropout > 0 in training)")
        y = _pointwise(self.out_conv, self.dropout(self.gelu(_pointwise(self.in_conv, self.layer_norm(x)))))
        return y if residual is None else y + residual


class MGFNFeatureAmplifier(nn.Module):
    """tokens = Conv1d_k3(features) + mag_ratio * Conv1d_k3(magnitude) (modeling_mgfn.py:81-93).
    Input (bs, ncrops, T, channels+1); output (dims[0], bs*ncrops, T)."""

    def __init__(self, config):
        super().__init__()
        self.channels = config.channels
        self.mag_ratio = config.mag_ratio
        self.to_tokens = nn.Conv1d(config.channels, config.dims[0], kernel_size=3, stride=1, padding=1)
        self.to_mag = nn.Conv1d(1, config.dims[0], kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        bs, ncrops, t, c = x.shape
        rows = x.reshape(bs * ncrops * t, c)               # the input as stored: (positions, channels + magnitude)
        x = x.reshape(bs * ncrops, t, c).permute(2, 0, 1)  # (C+1, B, T) view
        if (torch.is_grad_enabled() and self.to_tokens.weight.requires_grad) or mgfn_ops.fused_ok(x):
            # (with autograd, and on the GPU without it too: no unfolded input, no torch GEMM -- eval's 1 -> 64 magnitude conv included)
            tokens, whole = self._tokens_by_taps(x, bs * ncrops, t, rows)
            if whole:  # (the magnitude conv, the scale and the sum went into the launch that finishes the token conv)
                return tokens
        else:
            tokens = _conv_k(self.to_tokens, x[: self.channels])
        return tokens + self.mag_ratio * _conv_k(self.to_mag, x[self.channels :])

    def _tokens_by_taps(self, x, b, t, rows=None):
        """The 2048 -> 64 token conv with autograd, without unfolding its input: a k-tap conv is linear in its taps,
        conv_k(x)[o, t] = sum_j (W_j x)[o, t + j - k/2], so ONE GEMM of the stacked tap matrices (k*64 x 2048) with the input AS
        STORED (positions x channels, read through its strides: no transposed / padded / tap-stacked copy, 84 + 252 MB at the
        training batch) gives the k per-tap products, and a shifted add over (k*64, B, T) -- 1/32 of the input's size --
        finishes the conv.  Backward (autograd): dW = dZ X, again on the input as stored; no input gradient."""
        conv = self.to_tokens
        o, c, k = conv.weight.shape
        wt = conv.weight.permute(2, 0, 1).reshape(k * o, c)              # the k tap matrices, stacked
        rows = rows.contiguous() if rows is not None else None
        if rows is not None and rows.shape[1] == c + 1 and mgfn_ops.fused_ok(x) and mgfn_ops.token_taps_ok(wt, rows):
            # (B*T, C + 1) rows read in place: advhip_gemm_nt_f32 forward, one conv launch on the same rows for the weight gradient
            z = mgfn_ops.token_taps(wt, rows).view(k, o, b, t)
        else:
            mgfn_ops.torch_path(x, "token conv tap GEMM (channels % 16 != 0, taps x dims[0] % 64 != 0, or an input that carries a gradient)")
            xv = x[:c].reshape(c, b * t)                                  # (C, B*T) view of the (B*T, C+1) rows: strides (1, C+1)
            z = torch.matmul(wt, xv).view(k, o, b, t)
        mag = x[c:]
        if k == 3 and mgfn_ops.amp_combine_ok(z, conv, self.to_mag, mag):
            # the shifted add, the bias AND mag_ratio * to_mag(magnitude): one HIP launch forward, one backward -> (tokens, True)
            return mgfn_ops.amp_combine(z, conv, self.to_mag, mag, self.mag_ratio), True
        mgfn_ops.torch_path(x, "token conv tap sum (k != 3, a conv without bias, or a magnitude channel that is not a view of the input rows)")
        zp = F.pad(z, (k // 2, k // 2))
        y = zp[0, :, :, 0:t]
        for j in range(1, k):
            y = y + zp[j, :, :, j : j + t]
        return (y + conv.bias.view(-1, 1, 1) if conv.bias is not None else y), False


class GlanceAttention(nn.Module):
    """Multi-head self-attention over the T clips (modeling_mgfn.py:107-123)."""

    def __init__(self, dim: int, heads: int, dim_head: int):
        super().__init__()
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head ** -0.5
        self.norm = MGFNLayerNorm(dim)
        self.to_qkv = nn.Conv1d(dim, dim_head * heads * 3, 1, bias=False)
        self.to_out = nn.Conv1d(dim_head * heads, dim, 1)

    def forward(self, x, residual: Optional[torch.Tensor] = None):  # (C, B, T)
        _, b, n = x.shape
        if (mgfn_ops.GLANCE_BLOCK and residual is x and torch.is_grad_enabled() and mgfn_ops.fused_ok(x) and _hip(self.to_qkv, x) and _hip(self.to_out, x)
                and self.to_qkv.bias is None and self.to_out.bias is not None and self.dim_head == 64
                and self.to_qkv.weight.shape[0] == 3 * self.heads * self.dim_head):
            # LN + to_qkv + attention core + to_out + skip as one autograd node (mgfn_ops._GlanceAttnBlockCN)
            return mgfn_ops.glance_attention_block_cn(x, self.norm, self.to_qkv, self.to_out, self.heads, self.dim_head, self.scale)
        qkv = _pointwise(self.to_qkv, self.norm(x))
        if mgfn_ops.glance_attention_ok(qkv, self.heads, self.dim_head):  # scale, sim, softmax, v attn^T, layout: one HIP launch
            return _pointwise(self.to_out, mgfn_ops.glance_attention_core(qkv, self.heads, self.dim_head, self.scale), residual)
        mgfn_ops.torch_path(x, "Glance attention core (dim_head != 64)")
        qkv = qkv.view(3, self.heads, self.dim_head, b, n)
        q, k, v = (t.permute(2, 0, 1, 3) for t in qkv.unbind(0))  # (b, h, d, n)
        sim = torch.matmul((q * self.scale).transpose(-1, -2), k)  # (b, h, i, j)
        out = torch.matmul(v, sim.softmax(dim=-1).transpose(-1, -2))  # (b, h, d, i)
        return _pointwise(self.to_out, out.permute(1, 2, 0, 3).reshape(self.heads * self.dim_head, b, n), residual)


class FocusAttention(nn.Module):
    """BN1d -> 1x1 conv -> per-head depth-wise k=5 temporal conv -> 1x1 conv (modeling_mgfn.py:173-180)."""

    def __init__(self, dim: int, heads: int, dim_head: int, local_aggr_kernel: int):
        super().__init__()
        self.heads = heads
        inner = dim_head * heads
        self.norm = nn.BatchNorm1d(dim)
        self.to_v = nn.Conv1d(dim, inner, 1, bias=False)
        self.rel_pos = nn.Conv1d(heads, heads, local_aggr_kernel, padding=local_aggr_kernel // 2, groups=heads)
        self.to_out = nn.Conv1d(inner, dim, 1)

    def _batch_norm(self, x):  # nn.BatchNorm1d semantics on a (C, B, T) tensor
        bn = self.norm
        if bn.training and mgfn_ops.fused_ok(x):  # batch statistics + normalisation: one HIP launch (fwd) / one (bwd)
            y, mean, var = mgfn_ops.bn_rows_train(x, bn.weight, bn.bias, bn.eps)
            if bn.track_running_stats:
                with torch.no_grad():
                    n = x.shape[1] * x.shape[2]
                    mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                    bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
                    bn.running_var.mul_(1 - mom).add_(var * (n / max(n - 1, 1)), alpha=mom)
                    bn.num_batches_tracked += 1
            return y
        mgfn_ops.torch_path(x, "BatchNorm1d (batch statistics off the fused kernel, or running statistics with autograd)")
        if bn.training:
            var, mean = torch.var_mean(x, dim=(1, 2), unbiased=False)
            if bn.track_running_stats:
                with torch.no_grad():
                    n = x.shape[1] * x.shape[2]
                    mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                    bn.running_mean.mul_(1 - mom).add_(mean.detach(), alpha=mom)
                    bn.running_var.mul_(1 - mom).add_(var.detach() * (n / max(n - 1, 1)), alpha=mom)
                    bn.num_batches_tracked += 1
        else:
            mean, var = bn.running_mean, bn.running_var
        scale = bn.weight * torch.rsqrt(var + bn.eps)
        return x * scale[:, None, None] + (bn.bias - mean * scale)[:, None, None]

    def forward(self, x, residual: Optional[torch.Tensor] = None):  # (C, B, T)
        _, b, n = x.shape
        bn = self.norm
        if (residual is x and torch.is_grad_enabled() and bn.training and bn.momentum is not None and bn.affine
                and mgfn_ops.fused_ok(x) and _hip(self.to_v, x) and _hip(self.to_out, x) and self.to_v.bias is None
                and self.to_out.bias is not None and self.rel_pos.weight.shape[-1] in (3, 5)):
            # BN + to_v + rel_pos + to_out + skip as one autograd node (mgfn_ops._FocusAttnBlockCN)
            return mgfn_ops.focus_attention_block_cn(x, bn, self.to_v, self.rel_pos, self.to_out, self.heads)
        if mgfn_ops.bn_eval_fold_ok(x, bn, self.to_v):  # eval-mode BN is an affine map: folded into to_v's operand, no launch of its own
            v = mgfn_ops.to_v_folded_bn(x, bn, self.to_v)
        else:
            v = _pointwise(self.to_v, self._batch_norm(x))
        inner = v.shape[0]
        h = self.heads
        v = v.view(inner // h, h, b, n)  # channel = c_idx*heads + h_idx  ("b (c h) n -> (b c) h n")
        k = self.rel_pos.weight.shape[-1]
        if mgfn_ops.fused_ok(x) and k in (3, 5):  # the per-head depth-wise temporal conv as one HIP launch (fwd) / one (bwd)
            out = mgfn_ops.dwconv_t(v.reshape(inner, b, n), self.rel_pos.weight, self.rel_pos.bias)
            return _pointwise(self.to_out, out, residual)
        mgfn_ops.torch_path(x, f"depth-wise temporal conv with k = {k} (the kernel is built for 3 and 5)")
        vp = F.pad(v, (k // 2, k // 2))
        w = self.rel_pos.weight[:, 0]  # (h, k): one temporal filter per head
        out = self.rel_pos.bias.view(1, h, 1, 1)
        for j in range(k):
            out = out + w[:, j].view(1, h, 1, 1) * vp[..., j : j + n]
        return _pointwise(self.to_out, out.reshape(inner, b, n), residual)


class _Block(nn.Module):
    def forward(self, x):
        # x = scc(x) + x;  x = attention(x) + x;  x = ffn(x) + x  (modeling_mgfn.py:143-147, 201-205): the residual adds ride
        # in the epilogue of the GEMM that ends each branch
        x = _conv_k(self.scc, x, residual=x)
        x = self.attention(x, residual=x)
        return self.ffn(x, residual=x)


class GlanceBlock(_Block):
    def __init__(self, config, dim: int, heads: int):
        super().__init__()
        self.scc = nn.Conv1d(dim, dim, 3, padding=1)
        self.attention = GlanceAttention(dim=dim, heads=heads, dim_head=config.dim_head)
        self.ffn = MGFNFeedForward(dim, repe=config.ff_repe, dropout=config.dropout)


class FocusBlock(_Block):
    def __init__(self, config, dim, heads):
        super().__init__()
        self.scc = nn.Conv1d(dim, dim, 3, padding=1)
        self.attention = FocusAttention(dim=dim, heads=heads, dim_head=config.dim_head, local_aggr_kernel=config.local_aggr_kernel)
        self.ffn = MGFNFeedForward(dim, repe=config.ff_repe, dropout=config.dropout)


class MGFNIntermediate(nn.Module):
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.layer_norm = MGFNLayerNorm(in_dim)
        self.conv = nn.Conv1d(in_dim, out_dim, 1, stride=1)

    def forward(self, x):
        return _pointwise(self.conv, self.layer_norm(x))


if _PreTrainedModel is not None:
    class MGFNPreTrainedModel(_PreTrainedModel):
        config_class = MGFNConfig
        base_model_prefix = "backbone"

        def _init_weights(self, module):  # the reference leaves torch's default init untouched
            return None

        @property
        def dummy_inputs(self):
            return torch.randn(32, 10, 32, 2049)
else:  # pragma: no cover
    class MGFNPreTrainedModel(nn.Module):
        config_class = MGFNConfig

        def __init__(self, config):
            super().__init__()
            self.config = config


class MGFNModel(MGFNPreTrainedModel):
    def __init__(self, config):
        super().__init__(config)
        self.amplifier = MGFNFeatureAmplifier(config)
        stages = []
        n_stages = len(config.depths)
        for ind, (depth, kind) in enumerate(zip(config.depths, config.mgfn_types)):
            dim = config.dims[ind]
            heads = dim // config.dim_head
            if kind == "gb":
                cls = GlanceBlock
            elif kind == "fb":
                cls = FocusBlock
            else:
                raise AttributeError("The type of mgfn block must be either `gb` or `fb`.")
            blocks = [cls(config, dim=dim, heads=heads) for _ in range(depth)]
            if ind != n_stages - 1:
                blocks.append(MGFNIntermediate(dim, config.dims[ind + 1]))
            stages.append(nn.Sequential(*blocks))
        self.layers = nn.Sequential(*stages)

    def forward(self, x: torch.Tensor) -> MGFNModelOutput:
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            mgfn_ops.invalidate_caches()  # the weights are about to be updated: no cached packed copy may outlive this pass
            if x.is_cuda and x.dtype == torch.float32 and x.device.index == torch.cuda.current_device():
                mgfn_ops.step_packs(self._gemm_convs())  # ... and every GEMM layer's packed operand comes from one launch
        ok = False
        try:
            # internal layout (C, B, T); `outputs` is returned in the reference's (B, C, T) shape (a view)
            y = self.layers(self.amplifier(x))
            ok = True
        finally:
            mgfn_ops.end_step_packs()
            mgfn_ops.flush_counters(discard=not ok)  # (the BatchNorm layers' num_batches_tracked += 1, all in one launch)
        return MGFNModelOutput(outputs=y.permute(1, 0, 2))

    def _gemm_convs(self):
        """The Conv1d layers whose training forward runs on the HIP GEMMs (mgfn_ops.eligible's channel rules): their packed
        operands are produced together."""
        out = []
        for m in self.modules():
            if isinstance(m, nn.Conv1d) and m.groups == 1 and m.kernel_size[0] in (1, 3) and m.weight.requires_grad:
                cout, cin, _k = m.weight.shape
                w = m.weight
                if (min(cin, cout) >= mgfn_ops.MIN_CHANNELS_TRAIN and cout % 64 == 0 and cin % 64 == 0 and w.is_cuda and w.dtype == torch.float32
                        and w.is_contiguous() and w.device.index == torch.cuda.current_device()):
                    out.append(m)
        return out


class MGFNForVideoAnomalyDetection(MGFNPreTrainedModel):
    def __init__(self, config):
        super().__init__(config)
        self.k = config.k
        last_dim = config.dims[-1]
        self.backbone = MGFNModel(config)
        self.layer_norm = nn.LayerNorm(last_dim)
        self.fc = nn.Linear(last_dim, 1)
        self.sigmoid = nn.Sigmoid()
        self._force_split = False
        self.dropout = nn.Dropout(config.dropout_rate)
        # test hook: (keep_abnormal, keep_normal) multipliers used instead of drawing dropout masks
        self.injected_keep: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self.last_loss_terms: Optional[torch.Tensor] = None
        self.last_indices: Optional[Tuple[torch.Tensor, torch.Tensor]] = None

    @property
    def force_split(self) -> bool:
        """Split the batch into normal/abnormal halves in eval mode too (modeling_mgfn.py:290-300)."""
        return self._force_split

    @force_split.setter
    def force_split(self, val: bool):
        self._force_split = val

    def magnitude_selection_and_score_prediction(self, features, scores, batch_size, ncrops):
        """HIP restatement of modeling_mgfn.py:302-374; features (bs*ncrops,T,F), scores (bs*ncrops,T,1)."""
        mag, sc = mil_ops.mil_magnitude(features, scores.squeeze(-1), batch_size, ncrops)
        split = self.force_split or self.training
        if self.injected_keep is not None:
            keep_a, keep_n = self.injected_keep
        elif self.training:
            # same call order as the reference: abnormal first, then normal (modeling_mgfn.py:364-372)
            h = batch_size // 2 if split else batch_size
            # (ONE dropout call for both masks, rows [0, h) the abnormal half's and [h, 2h) the normal half's -- two launches instead of
            # four.  Same distribution as the reference's two calls (iid Bernoulli(0.3) / 0.3 per element), NOT the same draws: one
            # 2h x T Philox call maps elements to random numbers differently from two h x T calls, so a seed-for-seed comparison of
            # training masks with the reference is not offered; parity of this branch is pinned with `injected_keep` instead.)
            keep = self.dropout(torch.ones((2 * h, mag.shape[1]), device=mag.device, dtype=mag.dtype))
            keep_a, keep_n = keep[:h], keep[h:]
        else:
            keep_a = keep_n = None
        if split and batch_size % 2 == 0:
            # both halves as one autograd node (one zero-filled feature-gradient buffer instead of two padded and added ones)
            idx_a, feat_a, score_a, idx_n, feat_n, score_n = mil_ops.mil_topk_select_split(mag, keep_a, keep_n, sc, features, ncrops, self.k)
        else:
            if split:
                h = batch_size // 2
                nf, af = features[: h * ncrops], features[h * ncrops :]
                nm, am = mag[:h], mag[h:]
                ns, as_ = sc[:h], sc[h:]
            else:
                nf = af = features
                nm = am = mag
                ns = as_ = sc
            idx_a, feat_a, score_a = mil_ops.mil_topk_select(am, keep_a, as_, af, ncrops, self.k)
            idx_n, feat_n, score_n = mil_ops.mil_topk_select(nm, keep_n, ns, nf, ncrops, self.k)
        self.last_indices = (idx_a, idx_n)
        return score_a, score_n, feat_a, feat_n, sc.unsqueeze(2)

    def forward(self, video: torch.Tensor, abnormal_labels: Optional[torch.Tensor] = None,
                normal_labels: Optional[torch.Tensor] = None) -> MGFNVideoAnomalyDetectionOutput:
        bs, ncrops = video.shape[:2]
        body = self.backbone(video).outputs              # (bs*ncrops, C, T): a view of the body's (C, bs*ncrops, T) activation
        y = body.permute(1, 0, 2)
        if mgfn_ops.head_ok(y, self.layer_norm, self.fc):  # LayerNorm + Linear + sigmoid on the body's layout: one HIP launch
            x, scores = mgfn_ops.head_ln_fc(y, self.layer_norm, self.fc)
        else:
            mgfn_ops.torch_path(y, "head (LayerNorm + Linear + sigmoid) off the fused kernel")
            x = self.layer_norm(body.permute(0, 2, 1))   # (bs*ncrops, T, last_dim)
            scores = self.sigmoid(self.fc(x))
        abn_s, nor_s, a_feat, n_feat, sc = self.magnitude_selection_and_score_prediction(x, scores, bs, ncrops)
        loss = None
        if abnormal_labels is not None and normal_labels is not None:
            loss, terms = mil_ops.mgfn_loss(sc, abn_s, nor_s, a_feat, n_feat, abnormal_labels, normal_labels, ncrops)
            self.last_loss_terms = terms
        return MGFNVideoAnomalyDetectionOutput(
            loss=loss, abnormal_scores=abn_s, normal_scores=nor_s,
            a_feat_magnitude=a_feat, n_feat_magnitude=n_feat, scores=sc,
        )


def mgfn_param_shapes(config: Optional[MGFNConfig] = None) -> "OrderedDict[str, Tuple[int, ...]]":
    """state-dict key -> shape (lets tests build deterministic weights without a reference model)."""
    m = MGFNForVideoAnomalyDetection(config or MGFNConfig())
    return OrderedDict((k, tuple(v.shape)) for k, v in m.state_dict().items())
