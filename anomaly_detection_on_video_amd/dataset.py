"""Pre-extracted feature datasets: the `FeatureDataset` / `build_feature_dataset` surface of
`/root/reference/src/dataset.py:24-142` (zip of `<video>_i3d.npy` files, normal/abnormal split by
"Normal" in the file name, L2-magnitude channel appended per item, frame-level labels from
`ground_truth.json` in test mode).  Hub download is kept as the fallback when no `local_path` is
given (needs network); everything else works from local files, and
`write_synthetic_feature_zips` fabricates a UCF-Crime-shaped corpus for tests / benchmarks.

Video decoding + TenCrop (`TenCropVideoFrameDataset`, decord / torchvision) is outside the hot
path: extraction sources are tensors (see extract.py).
"""
from __future__ import annotations

import io
import json
import os
import zipfile
from typing import Callable, Dict, List, Optional, Union

import numpy as np
from torch.utils.data import Dataset

DEFAULT_FEATURE_HUB = "jinmang2/ucf_crime_tencrop_i3d_seg32"
DEFAULT_FILENAMES = {"train": "train.zip", "test": "test.zip"}


class FeatureDataset(Dataset):
    def __init__(self, filenames: List[str], values: Dict[str, Union[zipfile.ZipInfo, np.ndarray]],
                 labels: Optional[Dict[str, List[float]]] = None, open_func: Optional[Callable] = None):
        self.filenames = filenames
        self.values = values
        self.labels = labels
        self.open_func = open_func

    def __len__(self) -> int:
        return len(self.values)

    def open(self, value):
        if self.open_func is None:
            return value
        return np.load(self.open_func(value))  # dynamic loading straight from the zip member

    def add_magnitude(self, feature: np.ndarray) -> np.ndarray:
        # (a, b, C) -> (a, b, C+1): append ||f||_2 (dataset.py:121-124)
        return np.concatenate((feature, np.linalg.norm(feature, axis=2)[:, :, np.newaxis]), axis=2)

    def get_filename(self, idx: int) -> str:
        return self.filenames[idx]

    def __getitem__(self, idx: int) -> Dict[str, np.ndarray]:
        fname = self.get_filename(idx)
        feature = self.open(self.values[fname])
        item = {
            "feature": self.add_magnitude(feature),
            "anomaly": np.array(0.0 if "Normal" in fname else 1.0, dtype=np.float32),
        }
        if self.labels is not None:
            key = fname if fname in self.labels else fname.replace("_i3d.npy", "")
            item["label"] = np.array(self.labels[key], dtype=np.float32)
        return item


def _build_feature_dataset(filepath: str, mode: str, dynamic_load: bool, ground_truth: Optional[Dict] = None):
    assert mode in ("train", "test")
    zipf = zipfile.ZipFile(filepath)
    filenames, values = [], {}
    for member in zipf.infolist():
        if member.is_dir():
            continue
        name = member.filename.split("/")[-1]
        filenames.append(name)
        values[name] = member if dynamic_load else np.load(zipf.open(member))
    opener = zipf.open if dynamic_load else None
    if mode == "test":
        if ground_truth is None:
            from huggingface_hub import hf_hub_download

            with open(hf_hub_download(repo_id=DEFAULT_FEATURE_HUB, filename="ground_truth.json", repo_type="dataset")) as f:
                ground_truth = json.load(f)
        return FeatureDataset(filenames=filenames, values=values, labels=ground_truth, open_func=opener)
    out = {}
    for split, pick in (("normal", lambda n: "Normal" in n), ("abnormal", lambda n: "Normal" not in n)):
        names = [n for n in filenames if pick(n)]
        out[split] = FeatureDataset(filenames=names, values={n: values[n] for n in names}, open_func=opener)
    return out


def build_feature_dataset(mode: str = "train", local_path: Optional[str] = None, filename: Optional[str] = None,
                          cache_dir: Optional[str] = None, revision: str = "main", dynamic_load: bool = True):
    """Reference signature (dataset.py:73-95).  With `local_path`+`filename` the zip (and, in test
    mode, `<local_path>/ground_truth.json`) is read locally; otherwise it is fetched from the hub."""
    assert mode in ("train", "test")
    assert sum([local_path is None, filename is None]) != 1
    gt = None
    if local_path is None:
        from huggingface_hub import hf_hub_download

        filepath = hf_hub_download(repo_id=DEFAULT_FEATURE_HUB, filename=DEFAULT_FILENAMES[mode], cache_dir=cache_dir,
                                   revision=revision, repo_type="dataset")
    else:
        filepath = os.path.join(local_path, filename)
        gt_path = os.path.join(local_path, "ground_truth.json")
        if mode == "test" and os.path.exists(gt_path):
            with open(gt_path) as f:
                gt = json.load(f)
    return _build_feature_dataset(filepath, mode, dynamic_load, gt)


def write_synthetic_feature_zips(outdir: str, n_normal: int = 8, n_abnormal: int = 8, n_test: int = 6, seg: int = 32,
                                 channels: int = 2048, ncrops: int = 10, seed: int = 0) -> str:
    """A small UCF-Crime-shaped feature corpus: train.zip ((10, seg, C) per video), test.zip
    ((n_clips, 10, C) per video) and ground_truth.json built with the make_gt_ucf rule.  Abnormal
    videos carry a burst of larger-magnitude features on the annotated clips."""
    from .gt import frame_ground_truth

    rng = np.random.default_rng(seed)
    os.makedirs(outdir, exist_ok=True)

    def put(z, name, arr):
        buf = io.BytesIO()
        np.save(buf, arr.astype(np.float32))
        z.writestr(name, buf.getvalue())

    with zipfile.ZipFile(os.path.join(outdir, "train.zip"), "w") as z:
        for i in range(n_normal):
            put(z, f"train/Normal_Videos{i:03d}_x264_i3d.npy", np.abs(rng.standard_normal((ncrops, seg, channels))))
        for i in range(n_abnormal):
            f = np.abs(rng.standard_normal((ncrops, seg, channels)))
            s = int(rng.integers(0, seg - 6))
            f[:, s : s + 6] *= 2.5
            put(z, f"train/Abuse{i:03d}_x264_i3d.npy", f)
    gt = {}
    with zipfile.ZipFile(os.path.join(outdir, "test.zip"), "w") as z:
        for i in range(n_test):
            n_clips = int(rng.integers(20, 60))
            f = np.abs(rng.standard_normal((n_clips, ncrops, channels)))
            if i % 2 == 0:
                name, ev = f"Normal_Videos_{900 + i}_x264", ((-1, -1), (-1, -1))
            else:
                c0 = int(rng.integers(2, n_clips - 8))
                f[c0 : c0 + 6] *= 2.5
                name, ev = f"Burglary{i:03d}_x264", ((c0 * 16, (c0 + 6) * 16 - 1), (-1, -1))
            put(z, f"test/{name}_i3d.npy", f)
            gt[name] = frame_ground_truth(n_clips, ev[0], ev[1])
    with open(os.path.join(outdir, "ground_truth.json"), "w") as f:
        json.dump(gt, f)
    return outdir
