"""ORACLE (test infrastructure only) -- numpy restatements of the small host-side functions
that sit either side of the hot path.

    segment_features   /root/reference/extract_features.py:159-185  (per-file body :171-183)
    add_magnitude      /root/reference/src/dataset.py:121-124
    stack_crop_outputs /root/reference/extract_features.py:93-100
    gt_from_annotation /root/reference/make_gt_ucf.py:36-50
    frame_level_auc    /root/reference/src/runner.py:66-76 (sklearn roc_curve/auc, precision_recall_curve/auc)

    ten_crop_clips     /root/reference/src/dataset.py:175-195 + src/gtransforms.py:20-73,115-132 + extract_features.py:83
                       (TenCrop itself is torchvision.transforms.TenCrop -- third-party, not in the reference tree and not
                       installed here: its published algorithm is restated, "parity unpinned" for the crop geometry)

Written as explicit python loops on purpose: this is the checker, not the product.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def segment_features(features: np.ndarray, seg_length: int = 32) -> np.ndarray:
    """(n_clips, 10, C) -> (10, seg_length, C); bucket means over linspace boundaries."""
    feats = features.transpose(1, 0, 2)
    out = []
    for f in feats:
        new = np.zeros((seg_length, f.shape[1])).astype(np.float32)
        r = np.linspace(0, len(f), seg_length + 1, dtype=int)
        for i in range(seg_length):
            if r[i] != r[i + 1]:
                new[i, :] = np.mean(f[r[i] : r[i + 1], :], 0)
            else:
                new[i, :] = f[r[i], :]
        out.append(new)
    return np.array(out, dtype=np.float32)


def add_magnitude(feature: np.ndarray) -> np.ndarray:
    mag = np.linalg.norm(feature, axis=2)[:, :, np.newaxis]
    return np.concatenate((feature, mag), axis=2)


def stack_crop_outputs(per_batch: List[List[np.ndarray]]) -> np.ndarray:
    """[[ (B,2048,1,1,1) x ncrops ] x n_batches] -> (n_clips, ncrops, 2048)."""
    stacked = [np.stack(crops, axis=1) for crops in per_batch]
    return np.squeeze(np.vstack(stacked))


def gt_from_annotation(n_clips: int, first_event: Tuple[int, int], second_event: Tuple[int, int], frames_per_clip: int = 16) -> List[float]:
    num_frame = n_clips * frames_per_clip
    gt = [0.0] * num_frame
    # reference tests first_event[0] twice (make_gt_ucf.py:44) -- reproduced as observed
    if first_event[0] > 0 and first_event[0] > 0:
        for i in range(first_event[0], min(first_event[1] + 1, num_frame)):
            gt[i] = 1.0
    if second_event[0] > 0 and second_event[1] > 0:
        for i in range(second_event[0], min(second_event[1] + 1, num_frame)):
            gt[i] = 1.0
    return gt


def _trapz(x: np.ndarray, y: np.ndarray) -> float:
    return float(np.sum((x[1:] - x[:-1]) * (y[1:] + y[:-1]) * 0.5))


def roc_auc(labels: Sequence[float], preds: Sequence[float]) -> float:
    """Area under the ROC curve (trapezoid over distinct thresholds), as sklearn's roc_curve+auc."""
    y = np.asarray(labels, dtype=np.float64) > 0.5
    s = np.asarray(preds, dtype=np.float64)
    order = np.argsort(-s, kind="mergesort")
    y, s = y[order], s[order]
    distinct = np.where(np.diff(s))[0]
    idx = np.r_[distinct, y.size - 1]
    tps = np.cumsum(y)[idx].astype(np.float64)
    fps = (1 + idx - tps).astype(np.float64)
    tps = np.r_[0.0, tps]
    fps = np.r_[0.0, fps]
    return _trapz(fps / fps[-1], tps / tps[-1])


def pr_auc(labels: Sequence[float], preds: Sequence[float]) -> float:
    """auc(recall, precision) of sklearn's precision_recall_curve (runner.py:75-76)."""
    y = np.asarray(labels, dtype=np.float64) > 0.5
    s = np.asarray(preds, dtype=np.float64)
    order = np.argsort(-s, kind="mergesort")
    y, s = y[order], s[order]
    distinct = np.where(np.diff(s))[0]
    idx = np.r_[distinct, y.size - 1]
    tps = np.cumsum(y)[idx].astype(np.float64)
    fps = (1 + idx - tps).astype(np.float64)
    precision = tps / (tps + fps)
    recall = tps / tps[-1]
    # sklearn reverses and appends the (recall=0, precision=1) end point
    precision = np.r_[precision[::-1], 1.0]
    recall = np.r_[recall[::-1], 0.0]
    return -_trapz(recall, precision)


def frame_level_auc(preds_per_video: List[np.ndarray], labels_per_video: List[np.ndarray], frames_per_clip: int = 16) -> Tuple[float, float]:
    preds = np.repeat(np.concatenate(preds_per_video), frames_per_clip)
    labels = np.concatenate(labels_per_video)
    return roc_auc(labels, preds), pr_auc(labels, preds)


def ten_crop_clips(frames: np.ndarray, frames_per_clip: int = 16, crop: int = 224, mean: float = 114.75, std: float = 57.375) -> np.ndarray:
    """uint8 (F, H, W, C) resized frames of one video -> float32 (n_clips, 10, C, frames_per_clip, crop, crop): what
    `TenCropVideoFrameDataset[i]` (src/dataset.py:187-195) followed by `inputs.permute(0,1,3,2,4,5)` (extract_features.py:83)
    hands to the backbone, for every clip i.

    torchvision.transforms.TenCrop(size) (vertical_flip=False) = five_crop(img) + five_crop(hflip(img));
    five_crop = (top-left, top-right, bottom-left, bottom-right, center_crop);
    center_crop offsets = int(round((H - size) / 2.0)), int(round((W - size) / 2.0))  (Python round: half to even).
    Then PILToTensor().float() (HWC bytes -> CHW float), `t.sub_(mean).div_(std)` per channel in fp32
    (src/gtransforms.py:69-72), LoopPad (src/gtransforms.py:119-132)."""
    F, H, W, C = frames.shape
    m32, s32 = np.float32(mean), np.float32(std)

    def five(img):  # img: (H, W, C)
        top_c, left_c = int(round((H - crop) / 2.0)), int(round((W - crop) / 2.0))
        offs = [(0, 0), (0, W - crop), (H - crop, 0), (H - crop, W - crop), (top_c, left_c)]
        return [img[t : t + crop, l : l + crop] for t, l in offs]

    clips = []
    for start in range(0, F, frames_per_clip):
        chunk = frames[start : start + frames_per_clip]
        per_frame = []
        for img in chunk:
            crops = five(img) + five(img[:, ::-1])
            t = np.stack([c.transpose(2, 0, 1) for c in crops]).astype(np.float32)  # (10, C, crop, crop)
            per_frame.append((t - m32) / s32)
        tensor = np.stack(per_frame)  # (len, 10, C, crop, crop)
        length = tensor.shape[0]
        if length != frames_per_clip:  # LoopPad
            n_pad = frames_per_clip - length
            pad = [tensor] * (n_pad // length)
            if n_pad % length > 0:
                pad.append(tensor[0 : n_pad % length])
            tensor = np.concatenate([tensor] + pad, axis=0)
        clips.append(tensor.transpose(1, 2, 0, 3, 4))  # (10, C, T, crop, crop)
    return np.stack(clips).astype(np.float32)
