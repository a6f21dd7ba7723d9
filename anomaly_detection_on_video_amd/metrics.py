"""Frame-level ROC-AUC / PR-AUC of `on_validation_epoch_end` (/root/reference/src/runner.py:62-79)
restated in numpy (sklearn is optional; pinned against sklearn known-answers in the tests)."""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np


def _ranked(labels, preds):
    y = np.asarray(labels, dtype=np.float64).ravel() > 0.5
    s = np.asarray(preds, dtype=np.float64).ravel()
    order = np.argsort(-s, kind="stable")
    y, s = y[order], s[order]
    cut = np.flatnonzero(np.diff(s))
    idx = np.concatenate([cut, [y.size - 1]])
    tps = np.cumsum(y)[idx].astype(np.float64)
    fps = 1.0 + idx - tps
    return tps, fps


def _area(x: np.ndarray, y: np.ndarray) -> float:
    return float(np.sum(np.diff(x) * (y[1:] + y[:-1]) / 2.0))


def roc_auc(labels: Sequence[float], preds: Sequence[float]) -> float:
    tps, fps = _ranked(labels, preds)
    tpr = np.concatenate([[0.0], tps]) / tps[-1]
    fpr = np.concatenate([[0.0], fps]) / fps[-1]
    return _area(fpr, tpr)


def pr_auc(labels: Sequence[float], preds: Sequence[float]) -> float:
    tps, fps = _ranked(labels, preds)
    precision = np.concatenate([(tps / (tps + fps))[::-1], [1.0]])
    recall = np.concatenate([(tps / tps[-1])[::-1], [0.0]])
    return -_area(recall, precision)


def frame_level_auc(preds_per_video, labels_per_video, frames_per_clip: int = 16) -> Tuple[float, float]:
    """np.repeat(clip scores, 16) vs the frame-level ground truth (runner.py:66-76)."""
    preds = np.repeat(np.concatenate([np.asarray(p).ravel() for p in preds_per_video]), frames_per_clip)
    labels = np.concatenate([np.asarray(l).ravel() for l in labels_per_video])
    if preds.shape != labels.shape:
        raise ValueError(f"{preds.shape[0]} repeated predictions vs {labels.shape[0]} frame labels")
    return roc_auc(labels, preds), pr_auc(labels, preds)
