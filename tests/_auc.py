"""Frame-level ROC-AUC of two scorers whose clip scores agree to `tol`: the AUC is a count of (positive, negative) frame pairs
in the right order, so the other scorer's value lies between the AUC with every positive frame's score lowered by 2 tol and
the one with it raised by 2 tol.  On a corpus of a few dozen clips one near-tie swapping costs 1 / (pairs) -- a fixed absolute
tolerance on the AUC itself is either loose or flaky; this band is neither."""
import numpy as np

from anomaly_detection_on_video_amd import metrics


def auc_band(preds_per_video, labels_per_video, frames_per_clip: int = 16, tol: float = 1e-5):
    preds = np.repeat(np.concatenate([np.asarray(p, dtype=np.float64).ravel() for p in preds_per_video]), frames_per_clip)
    labels = np.concatenate([np.asarray(l).ravel() for l in labels_per_video])
    assert preds.shape == labels.shape
    pos = labels > 0.5
    lo = metrics.roc_auc(labels, np.where(pos, preds - 2 * tol, preds))
    hi = metrics.roc_auc(labels, np.where(pos, preds + 2 * tol, preds))
    return lo, metrics.roc_auc(labels, preds), hi
