"""Ground-truth builder: the pure function inside `/root/reference/make_gt_ucf.py:36-50`
(annotation (s1,e1,s2,e2) -> frame-level 0/1 vector of length n_clips*16) plus the parser of the
double-space separated annotation file (:17-24)."""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np


def parse_temporal_annotations(text: str) -> Dict[str, Dict[str, Tuple[int, int]]]:
    out = {}
    for line in text.splitlines():
        line = line.strip()
        if not line:
            continue
        filename, _cls, s1, e1, s2, e2 = line.split("  ")
        out[filename.split(".")[0]] = {"first_event": (int(s1), int(e1)), "second_event": (int(s2), int(e2))}
    return out


def frame_ground_truth(n_clips: int, first_event: Tuple[int, int], second_event: Tuple[int, int], frames_per_clip: int = 16) -> List[float]:
    n = n_clips * frames_per_clip
    gt = np.zeros(n, dtype=np.float64)
    # the reference tests first_event[0] twice and never e1 (make_gt_ucf.py:44); kept as observed
    if first_event[0] > 0:
        gt[first_event[0] : min(first_event[1] + 1, n)] = 1.0
    if second_event[0] > 0 and second_event[1] > 0:
        gt[second_event[0] : min(second_event[1] + 1, n)] = 1.0
    return gt.tolist()
