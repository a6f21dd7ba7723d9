#!/usr/bin/env python3
"""I3D feature extraction entry point (the reference's extract_features.py surface).

    python extract_features.py --outdir OUT [--videos N] [--weights path.pt | --synthetic-weights]

The reference decodes the UCF-Crime videos with decord + torchvision TenCrop (not available in
the MI355X image, and outside the hot path).  Here the video source is synthetic TenCrop'd clip
tensors of the same layout; plug a real decoder in by passing (name, loader) pairs to
`anomaly_detection_on_video_amd.extract.extract`.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from anomaly_detection_on_video_amd.extract import extract, load_feature_extraction_model, segment  # noqa: E402,F401


def synthetic_sources(n_videos: int, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    for i in range(n_videos):
        n_clips = int(torch.randint(2, 6, (1,), generator=g))
        name = ("Normal_Videos_%03d_x264" if i % 2 == 0 else "Abuse%03d_x264") % i
        yield name, (lambda n=n_clips, s=seed + i: torch.randn((n, 10, 16, 3, 224, 224), generator=torch.Generator().manual_seed(s)))


def main(outdir: str = "ucf_crime", videos: int = 4, weights: str = None, synthetic_weights: bool = False,
         model_name: str = "i3d_8x8_r50"):
    """`model_name` defaults to the reference's (extract_features.py:34,46); that variant is parity-unpinned here (a warning
    says so) -- `--model-name tushar-n-baseline` is the I3Res50 pinned against the reference."""
    if synthetic_weights:
        os.environ["ADV_I3D_SYNTHETIC"] = "1"
    model, _device = load_feature_extraction_model(model_name, state_dict_path=weights, check_model_size=True)
    outpath = os.path.join(outdir, "anomaly_features", "train")
    extract(synthetic_sources(videos), model, outpath)
    seg_length = 32
    segment(outpath, os.path.join(outdir, f"segment_features_{seg_length}"), seg_length)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--outdir", default="ucf_crime")
    ap.add_argument("--videos", type=int, default=4)
    ap.add_argument("--weights", default=None)
    ap.add_argument("--synthetic-weights", action="store_true")
    ap.add_argument("--model-name", default="i3d_8x8_r50", choices=["i3d_8x8_r50", "tushar-n-baseline"],
                    help="the reference's default is i3d_8x8_r50 (parity-unpinned here); tushar-n-baseline = the pinned in-repo I3Res50")
    a = ap.parse_args()
    main(a.outdir, a.videos, a.weights, a.synthetic_weights, a.model_name)
