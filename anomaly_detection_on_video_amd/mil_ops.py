"""Autograd bindings of the MIL head / loss HIP kernels (csrc/mil.hip, csrc/loss.hip).

Three differentiable ops, each one C-ABI call forward and one backward:

    mil_magnitude(features, scores, bs, ncrops)          -> mag (bs,T), sc (bs,T)
    mil_topk_select(mag, keep, sc, features, ncrops, k)  -> idx (n,k), sel (ncrops*n,k,F), score (n,1)
    mgfn_loss(sc, abn_score, nor_score, a_feat, n_feat, abn_labels, nor_labels) -> total, terms(8)

CUDA tensors only; no fallback.  Reference semantics: modeling_mgfn.py:302-374, loss/*.py.
"""
from __future__ import annotations

from ctypes import c_float as C_float
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check, ptr, require_gpu, stream


class _MilMagnitude(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, scores: torch.Tensor, bs: int, ncrops: int):
        ctx.set_materialize_grads(False)  # (unused output gradients arrive as None, not as zero-filled tensors)
        features = features.contiguous()
        scores = scores.contiguous()
        require_gpu(features, scores)
        rows, T, F = features.shape
        if rows != bs * ncrops or scores.numel() != rows * T:
            raise ValueError(f"mil_magnitude: features {tuple(features.shape)} / scores {tuple(scores.shape)} vs bs={bs} ncrops={ncrops}")
        mag = torch.empty((bs, T), device=features.device, dtype=torch.float32)
        sc = torch.empty_like(mag)
        check(_lib.load().advhip_mil_magnitude_f32(ptr(features), ptr(scores), ptr(mag), ptr(sc), bs, ncrops, T, F, stream()), "mil_magnitude")
        ctx.save_for_backward(features)
        ctx.dims = (bs, ncrops, T, F, scores.shape)
        ctx.mark_non_differentiable(mag)  # only feeds topk indices (modeling_mgfn.py:345-346)
        return mag, sc

    @staticmethod
    def backward(ctx, _d_mag, d_sc):
        (features,) = ctx.saved_tensors
        bs, ncrops, T, F, sshape = ctx.dims
        d_scores = None
        if d_sc is not None and ctx.needs_input_grad[1]:
            d_sc = d_sc.contiguous()
            d_scores = torch.zeros((bs * ncrops, T), device=features.device, dtype=torch.float32)
            check(_lib.load().advhip_mil_magnitude_bwd_f32(ptr(features), None, ptr(d_sc), None, ptr(d_scores), bs, ncrops, T, F, stream()), "mil_magnitude_bwd")
            d_scores = d_scores.view(sshape)
        return None, d_scores, None, None


class _MilTopkSelect(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mag, keep, sc, features, ncrops: int, k: int):
        ctx.set_materialize_grads(False)  # (unused output gradients arrive as None, not as zero-filled tensors)
        mag, sc, features = mag.contiguous(), sc.contiguous(), features.contiguous()
        keep = None if keep is None else keep.contiguous()
        require_gpu(mag, keep, sc, features)
        n, T = mag.shape
        rows, T2, F = features.shape
        if rows != n * ncrops or T2 != T or sc.shape != mag.shape or (keep is not None and keep.shape != mag.shape):
            raise ValueError("mil_topk_select: inconsistent shapes")
        dev = mag.device
        idx = torch.empty((n, k), device=dev, dtype=torch.int64)
        sel = torch.empty((ncrops * n, k, F), device=dev, dtype=torch.float32)
        score = torch.empty((n,), device=dev, dtype=torch.float32)
        check(_lib.load().advhip_mil_topk_select_f32(ptr(mag), ptr(keep), ptr(sc), ptr(features), ptr(idx), ptr(sel), ptr(score),
                                                     n, ncrops, T, F, k, stream()), "mil_topk_select")
        ctx.save_for_backward(idx)
        ctx.dims = (n, ncrops, T, F, k)
        ctx.mark_non_differentiable(idx)
        return idx, sel, score.view(n, 1)

    @staticmethod
    def backward(ctx, _d_idx, d_sel, d_score):
        (idx,) = ctx.saved_tensors
        n, ncrops, T, F, k = ctx.dims
        dev = idx.device
        d_feat = d_sc = None
        want_f, want_s = ctx.needs_input_grad[3], ctx.needs_input_grad[2]
        if want_f:
            d_feat = torch.zeros((n * ncrops, T, F), device=dev, dtype=torch.float32)
        if want_s:
            d_sc = torch.zeros((n, T), device=dev, dtype=torch.float32)
        ds = d_sel.contiguous() if (d_sel is not None and want_f) else None
        dscore = d_score.contiguous() if (d_score is not None and want_s) else None
        if ds is not None or dscore is not None:
            check(_lib.load().advhip_mil_topk_select_bwd_f32(ptr(idx), ptr(ds), ptr(dscore), ptr(d_feat), ptr(d_sc), n, ncrops, T, F, k, stream()),
                  "mil_topk_select_bwd")
        return None, None, d_sc, d_feat, None, None


class _MilTopkSelectSplit(torch.autograd.Function):
    """mil_topk_select of the normal half (videos [0, h)) and of the abnormal half (videos [h, 2h)) of one batch
    (modeling_mgfn.py:324-332, 364-372) as ONE autograd node: the two halves' feature gradients are scattered into one
    zero-filled (bs*ncrops, T, F) buffer -- two separate nodes on slices of `features` make autograd zero-fill and add two
    full-size tensors (five launches over 42 MB at the training batch)."""

    @staticmethod
    def forward(ctx, mag, keep_a, keep_n, sc, features, ncrops: int, k: int):
        ctx.set_materialize_grads(False)  # (unused output gradients arrive as None, not as zero-filled tensors)
        mag, sc, features = mag.contiguous(), sc.contiguous(), features.contiguous()
        require_gpu(mag, keep_a, keep_n, sc, features)
        bs, T = mag.shape
        rows, T2, F = features.shape
        h = bs // 2
        if bs != 2 * h or rows != bs * ncrops or T2 != T or sc.shape != mag.shape:
            raise ValueError("mil_topk_select_split: inconsistent shapes")
        dev = mag.device
        lib = _lib.load()
        outs = []
        for half, keep in ((1, keep_a), (0, keep_n)):  # abnormal first, as the reference calls them
            keep = None if keep is None else keep.contiguous()
            if keep is not None and tuple(keep.shape) != (h, T):
                raise ValueError("mil_topk_select_split: keep mask must be (bs/2, T)")
            idx = torch.empty((h, k), device=dev, dtype=torch.int64)
            sel = torch.empty((ncrops * h, k, F), device=dev, dtype=torch.float32)
            score = torch.empty((h,), device=dev, dtype=torch.float32)
            lo = half * h
            check(lib.advhip_mil_topk_select_f32(ptr(mag[lo : lo + h]), ptr(keep), ptr(sc[lo : lo + h]), ptr(features[lo * ncrops : (lo + h) * ncrops]),
                                                 ptr(idx), ptr(sel), ptr(score), h, ncrops, T, F, k, stream()), "mil_topk_select")
            outs += [idx, sel, score.view(h, 1)]
        ctx.save_for_backward(outs[0], outs[3])
        ctx.dims = (h, ncrops, T, F, k)
        ctx.mark_non_differentiable(outs[0], outs[3])
        return tuple(outs)  # idx_a, sel_a, score_a, idx_n, sel_n, score_n

    @staticmethod
    def backward(ctx, _dia, d_sel_a, d_score_a, _din, d_sel_n, d_score_n):
        idx_a, idx_n = ctx.saved_tensors
        h, ncrops, T, F, k = ctx.dims
        dev = idx_a.device
        want_f, want_s = ctx.needs_input_grad[4], ctx.needs_input_grad[3]
        d_feat = torch.zeros((2 * h * ncrops, T, F), device=dev, dtype=torch.float32) if want_f else None
        d_sc = torch.zeros((2 * h, T), device=dev, dtype=torch.float32) if want_s else None
        lib = _lib.load()
        for half, idx, d_sel, d_score in ((1, idx_a, d_sel_a, d_score_a), (0, idx_n, d_sel_n, d_score_n)):
            ds = d_sel.contiguous() if (d_sel is not None and want_f) else None
            dscore = d_score.contiguous() if (d_score is not None and want_s) else None
            if ds is None and dscore is None:
                continue
            lo = half * h
            check(lib.advhip_mil_topk_select_bwd_f32(ptr(idx), ptr(ds), ptr(dscore), ptr(d_feat[lo * ncrops : (lo + h) * ncrops]) if want_f else None,
                                                     ptr(d_sc[lo : lo + h]) if want_s else None, h, ncrops, T, F, k, stream()), "mil_topk_select_bwd")
        return None, None, None, d_sc, d_feat, None, None


def mil_topk_select_split(mag, keep_a: Optional[torch.Tensor], keep_n: Optional[torch.Tensor], sc, features, ncrops: int, k: int):
    """-> (idx_a, sel_a, score_a, idx_n, sel_n, score_n): videos [bs/2, bs) are the abnormal half, [0, bs/2) the normal one."""
    return _MilTopkSelectSplit.apply(mag, keep_a, keep_n, sc, features, ncrops, k)


class _MgfnLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sc, abn_score, nor_score, a_feat, n_feat, abn_labels, nor_labels, ncrops: int):
        ctx.set_materialize_grads(False)  # (unused output gradients arrive as None, not as zero-filled tensors)
        args = [t.contiguous().float() for t in (sc, abn_score, nor_score, a_feat, n_feat, abn_labels, nor_labels)]
        require_gpu(*args)
        sc, abn_score, nor_score, a_feat, n_feat, abn_labels, nor_labels = args
        bs, T = sc.shape[0], sc.shape[1]
        R, k, F = a_feat.shape
        n = bs // 2
        if R != n * ncrops or n_feat.shape != a_feat.shape or abn_score.numel() != n or nor_score.numel() != n:
            raise ValueError("mgfn_loss: inconsistent shapes")
        if abn_labels.numel() != n or nor_labels.numel() != n:
            raise ValueError("mgfn_loss: need bs/2 labels per class")
        lib = _lib.load()
        ws = torch.empty((lib.advhip_mgfn_loss_ws_floats(n, ncrops, k),), device=sc.device, dtype=torch.float32)
        out = torch.empty((8,), device=sc.device, dtype=torch.float32)
        check(lib.advhip_mgfn_loss_fwd_f32(ptr(sc), ptr(abn_score), ptr(nor_score), ptr(a_feat), ptr(n_feat), ptr(abn_labels), ptr(nor_labels),
                                           ptr(ws), ptr(out), bs, T, ncrops, k, F, stream()), "mgfn_loss_fwd")
        ctx.save_for_backward(sc, abn_score, nor_score, a_feat, n_feat, abn_labels, nor_labels, ws)
        ctx.dims = (bs, T, ncrops, k, F)
        terms = out.detach().clone()
        ctx.mark_non_differentiable(terms)
        return out[0].clone(), terms

    @staticmethod
    def backward(ctx, d_loss, _d_terms):
        sc, abn_score, nor_score, a_feat, n_feat, abn_labels, nor_labels, ws = ctx.saved_tensors
        bs, T, ncrops, k, F = ctx.dims
        d_loss = d_loss.contiguous().float().reshape(1)
        d_sc = torch.empty_like(sc)
        d_abn = torch.empty_like(abn_score)
        d_nor = torch.empty_like(nor_score)
        d_a = torch.empty_like(a_feat)
        d_n = torch.empty_like(n_feat)
        check(_lib.load().advhip_mgfn_loss_bwd_f32(ptr(d_loss), ptr(sc), ptr(abn_score), ptr(nor_score), ptr(a_feat), ptr(n_feat), ptr(abn_labels),
                                                   ptr(nor_labels), ptr(ws), ptr(d_sc), ptr(d_abn), ptr(d_nor), ptr(d_a), ptr(d_n),
                                                   bs, T, ncrops, k, F, stream()), "mgfn_loss_bwd")
        return d_sc, d_abn, d_nor, d_a, d_n, None, None, None


def mil_magnitude(features: torch.Tensor, scores: torch.Tensor, bs: int, ncrops: int) -> Tuple[torch.Tensor, torch.Tensor]:
    return _MilMagnitude.apply(features, scores, bs, ncrops)


def mil_topk_select(mag, keep: Optional[torch.Tensor], sc, features, ncrops: int, k: int):
    return _MilTopkSelect.apply(mag, keep, sc, features, ncrops, k)


LOSS_TERMS = ("total", "bce", "con", "con_a", "con_n", "smooth", "sparse", "mgfn")


def mgfn_loss(sc, abn_score, nor_score, a_feat, n_feat, abn_labels, nor_labels, ncrops: int):
    """-> (total loss [differentiable], terms tensor(8) in LOSS_TERMS order)."""
    if sc.dim() == 3:
        sc = sc.squeeze(-1)
    return _MgfnLoss.apply(sc, abn_score.reshape(-1), nor_score.reshape(-1), a_feat, n_feat, abn_labels, nor_labels, ncrops)


# ------------------------------------------------------------------- feature post-processing
def segment_features(feats: torch.Tensor, seg_length: int = 32) -> torch.Tensor:
    """(n_clips, ncrops, C) -> (ncrops, seg_length, C), extract_features.py:171-183, on device."""
    feats = feats.contiguous()
    require_gpu(feats)
    n, ncrops, C = feats.shape
    out = torch.empty((ncrops, seg_length, C), device=feats.device, dtype=torch.float32)
    check(_lib.load().advhip_segment_features_f32(ptr(feats), ptr(out), n, ncrops, C, seg_length, stream()), "segment_features")
    return out


def add_magnitude(feats: torch.Tensor) -> torch.Tensor:
    """(..., C) -> (..., C+1) with the L2 norm appended (dataset.py:121-124), on device."""
    feats = feats.contiguous()
    require_gpu(feats)
    C = feats.shape[-1]
    rows = feats.numel() // C
    out = torch.empty(feats.shape[:-1] + (C + 1,), device=feats.device, dtype=torch.float32)
    check(_lib.load().advhip_add_magnitude_f32(ptr(feats), ptr(out), rows, C, stream()), "add_magnitude")
    return out


PIXEL_MEAN, PIXEL_STD = 114.75, 57.375  # GroupNormalize constants, src/dataset.py:180-181


def normalize_permute_u8(frames: torch.Tensor, mean: float = PIXEL_MEAN, std: float = PIXEL_STD) -> torch.Tensor:
    """uint8 (N, T, C, H, W) -> fp32 (N, C, T, H, W), (x - mean) / std, one HIP pass."""
    frames = frames.contiguous()
    require_gpu(frames)
    if frames.dtype != torch.uint8 or frames.dim() != 5:
        raise ValueError(f"expected uint8 (N,T,C,H,W), got {frames.dtype} {tuple(frames.shape)}")
    n, t, c, h, w = frames.shape
    out = torch.empty((n, c, t, h, w), device=frames.device, dtype=torch.float32)
    check(_lib.load().advhip_normalize_permute_u8(ptr(frames), ptr(out), n, t, c, h, w, C_float(mean), C_float(std), stream()),
          "normalize_permute_u8")
    return out


def tencrop_normalize_u8(frames: torch.Tensor, frames_per_clip: int = 16, crop: int = 224, mean: float = PIXEL_MEAN,
                         std: float = PIXEL_STD) -> torch.Tensor:
    """Resized uint8 frames (F, H, W, C) of one video -> the backbone's input (n_clips * 10, C, frames_per_clip, crop, crop)
    fp32: TenCrop, float, normalise, LoopPad and the layout permutes of TenCropVideoFrameDataset / _extract
    (src/dataset.py:175-195, src/gtransforms.py, extract_features.py:83) in one HIP pass.  Row = clip * 10 + crop."""
    frames = frames.contiguous()
    require_gpu(frames)
    if frames.dtype != torch.uint8 or frames.dim() != 4:
        raise ValueError(f"expected uint8 (F,H,W,C), got {frames.dtype} {tuple(frames.shape)}")
    f, h, w, c = frames.shape
    if h < crop or w < crop:
        raise ValueError(f"frames {h}x{w} smaller than the {crop} crop")
    n_clips = -(-f // frames_per_clip)
    out = torch.empty((n_clips * 10, c, frames_per_clip, crop, crop), device=frames.device, dtype=torch.float32)
    check(_lib.load().advhip_tencrop_normalize_u8(ptr(frames), ptr(out), f, h, w, c, frames_per_clip, crop, C_float(mean), C_float(std),
                                                  stream()), "tencrop_normalize_u8")
    return out
