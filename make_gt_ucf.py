#!/usr/bin/env python3
"""Frame-level ground truth from temporal annotations (the reference's make_gt_ucf.py, without its
import-time hub downloads):

    python make_gt_ucf.py --annotations Temporal_Anomaly_Annotation_for_Testing_Videos.txt \
                          --test-zip test.zip --out ground_truth_ucf_crime.json
"""
import argparse
import json
import os
import sys
import zipfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from anomaly_detection_on_video_amd.gt import frame_ground_truth, parse_temporal_annotations  # noqa: E402


def main(annotations: str, test_zip: str, out: str) -> None:
    with open(annotations) as f:
        annots = parse_temporal_annotations(f.read())
    gts = {}
    with zipfile.ZipFile(test_zip) as z:
        for member in z.infolist():
            if member.is_dir():
                continue
            n_clips = np.load(z.open(member)).shape[0]
            name = member.filename.split("/")[-1].replace("_i3d.npy", "")
            gts[name] = frame_ground_truth(n_clips, annots[name]["first_event"], annots[name]["second_event"])
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with open(out, "w") as f:
        json.dump(gts, f)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--annotations", required=True)
    ap.add_argument("--test-zip", required=True)
    ap.add_argument("--out", default="ground_truth_ucf_crime.json")
    a = ap.parse_args()
    main(a.annotations, a.test_zip, a.out)
