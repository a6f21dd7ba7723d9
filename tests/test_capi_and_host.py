"""CPU-only checks: the C-ABI library loads and exports every symbol include/advhip.h declares
(no compute calls), host-side logic (config composer, datasets, metrics, ground truth, sharding
arithmetic, stream ring) against the oracle / reference-generated goldens."""
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO
from oracle import host_oracle


# ------------------------------------------------------------------------------ C ABI
def _declared_functions():
    text = open(os.path.join(REPO, "include", "advhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(advhip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__

    __graft_entry__.build()
    from anomaly_detection_on_video_amd import _lib

    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/advhip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == declared
    assert lib.advhip_abi_version() == 2
    assert lib.advhip_target_arch() == b"gfx950"


def test_capi_argument_validation_without_gpu():
    """Pure host-side entry points and error reporting (nothing is launched)."""
    import ctypes as C

    from anomaly_detection_on_video_amd import _lib

    lib = _lib.load()
    d = _lib.ConvDesc(2, 64, 4, 55, 55, 256, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 0, 0)
    to, ho, wo = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.advhip_conv3d_out_dims(C.byref(d), C.byref(to), C.byref(ho), C.byref(wo)) == 0
    assert (to.value, ho.value, wo.value) == (4, 55, 55)
    assert lib.advhip_conv3d_packed_rows(C.byref(d)) == 64
    stem = _lib.ConvDesc(1, 3, 16, 224, 224, 64, 5, 7, 7, 2, 2, 2, 2, 3, 3, 1, 0, 0)
    assert lib.advhip_conv3d_out_dims(C.byref(stem), C.byref(to), C.byref(ho), C.byref(wo)) == 0
    assert (to.value, ho.value, wo.value) == (8, 112, 112)
    assert lib.advhip_conv3d_packed_rows(C.byref(stem)) == 736  # 735 -> next multiple of 32
    bad = _lib.ConvDesc(1, 3, 16, 224, 224, 60, 5, 7, 7, 2, 2, 2, 2, 3, 3, 1, 0, 0)
    assert lib.advhip_conv3d_packed_rows(C.byref(bad)) == -1
    assert b"multiple of 64" in lib.advhip_last_error()
    assert lib.advhip_mgfn_loss_ws_floats(16, 10, 3) == 4 * 16 * 10 * 3
    # null pointers are rejected before any launch
    assert lib.advhip_maxpool3d_f32(None, None, 1, 1, 2, 3, 3, 2, 3, 3, 2, 2, 2, None) == -1


def test_u8_stem_entry_points_validate_before_any_launch():
    """The uint8-frame stem: table sizes are pure host arithmetic, and bad frames / ranges / slack are refused with a message
    (every call below fails validation, so nothing is launched; the pointers are never dereferenced on the host)."""
    import ctypes as C

    from anomaly_detection_on_video_amd import _lib

    lib = _lib.load()
    stem = _lib.ConvDesc(8, 3, 16, 224, 224, 64, 5, 7, 7, 2, 2, 2, 2, 3, 3, 1, 0, 0)
    nk, nf, nw = C.c_int64(), C.c_int64(), C.c_int64()
    assert lib.advhip_conv3d_u8_table_sizes(C.byref(stem), C.byref(nk), C.byref(nf)) == 0
    assert (nk.value, nf.value) == (4 * 736, 9 * 16 * 16 * 64)   # two {offset, bits} tables; (pt+1)^2 (ph+1)^2 (pw+1)^2 classes x Cout
    assert lib.advhip_conv3d_u8_taps_table_sizes(C.byref(stem), C.byref(nk), C.byref(nf), C.byref(nw)) == 0
    assert (nk.value, nw.value) == (4 * 248, 248 * 3 * 64)        # 245 taps -> 248 (k-tiles of 8 taps)
    p = C.c_void_p(4096)  # stands for a device pointer
    F, FH, FW = 32, 256, 340
    nbytes = F * FH * FW * 3
    args = lambda d, frames_F, readable, first: (C.byref(d), p, frames_F, FH, FW, readable, first, p, p, p, p, p, C.c_float(57.375), p, 0, p, 1 << 40, None)
    assert lib.advhip_conv3d_u8_taps_tencrop_bn_relu_maxpool233_f32(*args(stem, F, nbytes, 0)) == -1
    assert b"one byte past the last pixel" in lib.advhip_last_error()
    assert lib.advhip_conv3d_u8_taps_tencrop_bn_relu_maxpool233_f32(*args(stem, F, nbytes + 4, 13)) == -1
    assert b"outside the 2 clips x 10 crops" in lib.advhip_last_error()
    assert lib.advhip_conv3d_u8_taps_tencrop_bn_relu_maxpool233_f32(*args(stem, F - 1, nbytes + 4, 0)) == -1
    assert b"not whole clips" in lib.advhip_last_error()
    wide = _lib.ConvDesc(8, 4, 16, 224, 224, 64, 5, 7, 7, 2, 2, 2, 2, 3, 3, 1, 0, 0)
    assert lib.advhip_conv3d_u8_taps_tencrop_bn_relu_maxpool233_f32(*args(wide, F, nbytes + 4, 0)) == -1
    assert b"3-channel pixels" in lib.advhip_last_error()
    big = _lib.ConvDesc(8, 3, 16, 300, 224, 64, 5, 7, 7, 2, 2, 2, 2, 3, 3, 1, 0, 0)
    assert lib.advhip_conv3d_u8_tencrop_bn_relu_maxpool233_f32(C.byref(big), p, F, FH, FW, 0, p, p, p, p, p, C.c_float(57.375), p, 0, p, 1 << 40, None) == -1
    assert b"smaller than the 300 x 224 crop" in lib.advhip_last_error()


def test_uninstantiated_algo_ids_are_rejected_without_gpu():
    """Ids inside a family's numeric range that have no kernel (DMA 69, DMA4 97/101-104, split-bf16 129-132/135/136,
    DMA2 165, and anything outside every family) must be an error, never a silent no-op (host-side query only)."""
    import ctypes as C

    from anomaly_detection_on_video_amd import _lib

    lib = _lib.load()
    ok = set([0, _lib.ALGO_TSPAN_128x64, _lib.ALGO_MIXED_128x64]) | set(_lib.IGEMM_ALGOS) | set(_lib.FAST_ALGOS) | set(_lib.DMA_ALGOS) | set(_lib.DMA4_ALGOS) | set(_lib.BF16X3_ALGOS) | set(_lib.DMA2_ALGOS) | set(_lib.PERSIST_ALGOS)
    for algo in range(0, 300):
        d = _lib.ConvDesc(2, 64, 4, 13, 11, 128, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, algo, 1)
        rc = lib.advhip_conv3d_workspace_bytes(C.byref(d))
        if algo in ok:
            assert rc == 0, (algo, lib.advhip_last_error())
        else:
            assert rc == -1, algo
            assert b"not instantiated" in lib.advhip_last_error(), (algo, lib.advhip_last_error())
    for algo in (69, 97, 101, 104, 129, 132, 135, 136, 165):
        assert algo not in ok


def test_product_ops_refuse_cpu_tensors():
    from anomaly_detection_on_video_amd import _lib, mil_ops, ops

    with pytest.raises(_lib.HipExtensionError):
        ops.maxpool3d(torch.zeros(1, 1, 2, 3, 3), (2, 3, 3), (2, 2, 2))
    with pytest.raises(_lib.HipExtensionError):
        mil_ops.add_magnitude(torch.zeros(2, 4))
    with pytest.raises(_lib.HipExtensionError):
        mil_ops.mil_magnitude(torch.zeros(10, 4, 8), torch.zeros(10, 4), 1, 10)


def test_missing_extension_fails_loudly(tmp_path):
    from anomaly_detection_on_video_amd import _lib

    with pytest.raises(_lib.HipExtensionError, match="not found"):
        _lib.load(str(tmp_path / "nope.so"))


# ------------------------------------------------------------------------------ config composer
def test_compose_defaults_and_overrides():
    from anomaly_detection_on_video_amd.config import compose

    cfg = compose(os.path.join(REPO, "configs"), "default", [])
    assert cfg.runner.cls == "src.runner.VideoAnomalyDetectionRunner"
    assert cfg.runner.model_class.endswith("MGFNForVideoAnomalyDetection")
    assert cfg.runner.model_config.dims == [64, 128, 1024] and cfg.runner.model_config.k == 3
    assert float(cfg.runner.optimizer.learning_rate) == 1e-3 and cfg.runner.optimizer.weight_decay == 0.0005
    assert cfg.data.batch_size == 16 and cfg.data.frames_per_clip == 16 and cfg.data.revision == "tushar-n"
    assert cfg.trainer.cls.max_epochs == 1000 and cfg.trainer.cls.precision == "32-true"
    assert set(cfg.trainer.callbacks) == {"lrmonitor", "model_checkpoint"}
    assert cfg.wandb_key is None
    cfg = compose(os.path.join(REPO, "configs"), "default", ["runner=default", "data=synthetic", "data.batch_size=2", "trainer.cls.max_epochs=3", "+extra.flag=true", "~wandb_key"])
    assert cfg.runner.model_class is None and "model_config" not in cfg.runner
    assert cfg.data.batch_size == 2 and cfg.data.local_path and cfg.trainer.cls.max_epochs == 3
    assert cfg.extra.flag is True and "wandb_key" not in cfg
    assert "${" not in cfg.trainer.logger.jsonl.path and "synthetic-default" in cfg.trainer.logger.jsonl.path


def test_instantiate_and_locate_through_src_alias():
    from anomaly_detection_on_video_amd.config import compose, instantiate, locate

    cfg = compose(os.path.join(REPO, "configs"), "default", [])
    mc = instantiate(cfg.runner.model_config)
    assert type(mc).__name__ == "MGFNConfig" and tuple(mc.dims) == (64, 128, 1024) and mc.dropout_rate == 0.7
    cls = locate(cfg.runner.model_class)
    from anomaly_detection_on_video_amd.models.mgfn import MGFNForVideoAnomalyDetection

    assert cls is MGFNForVideoAnomalyDetection
    assert locate(cfg.runner.cls).__name__ == "VideoAnomalyDetectionRunner"
    cb = instantiate(cfg.trainer.callbacks.model_checkpoint)
    assert cb.monitor == "rec_auc" and cb.save_top_k == 10


# ------------------------------------------------------------------------------ datasets / metrics / gt
def test_synthetic_feature_zips_and_feature_dataset(tmp_path):
    from anomaly_detection_on_video_amd.dataset import build_feature_dataset, write_synthetic_feature_zips

    d = write_synthetic_feature_zips(str(tmp_path), n_normal=3, n_abnormal=2, n_test=4, channels=32)
    for dyn in (False, True):
        tr = build_feature_dataset("train", local_path=d, filename="train.zip", dynamic_load=dyn)
        assert len(tr["normal"]) == 3 and len(tr["abnormal"]) == 2
        item = tr["abnormal"][0]
        assert item["feature"].shape == (10, 32, 33) and item["anomaly"] == 1.0
        np.testing.assert_allclose(item["feature"][..., -1], np.linalg.norm(item["feature"][..., :-1], axis=2), rtol=1e-6)
        assert tr["normal"][1]["anomaly"] == 0.0
        te = build_feature_dataset("test", local_path=d, filename="test.zip", dynamic_load=dyn)
        it = te[1]
        assert it["feature"].shape[1:] == (10, 33) and it["label"].shape == (it["feature"].shape[0] * 16,)
        assert it["label"].sum() == 96  # 6 annotated clips * 16 frames
    with pytest.raises(AssertionError):
        build_feature_dataset("train", local_path=d)


def test_add_magnitude_matches_reference_golden():
    from anomaly_detection_on_video_amd.dataset import FeatureDataset
    from anomaly_detection_on_video_amd.weights import synth_tensor

    g = np.load(os.path.join(GOLDEN, "host.npz"))
    f = synth_tensor("addmag", (10, 32, 48), scale=2.0).numpy()
    ds = FeatureDataset(["a_Normal.npy"], {"a_Normal.npy": f})
    np.testing.assert_array_equal(ds[0]["feature"], g["addmag"])
    assert ds[0]["anomaly"] == g["addmag_anomaly"]


def test_metrics_match_sklearn_known_answer_and_oracle():
    from anomaly_detection_on_video_amd import metrics

    g = np.load(os.path.join(GOLDEN, "auc.npz"))
    assert abs(metrics.roc_auc(g["labels"], g["preds"]) - float(g["roc_auc"])) < 1e-12
    assert abs(metrics.pr_auc(g["labels"], g["preds"]) - float(g["pr_auc"])) < 1e-12
    rng = np.random.default_rng(0)
    for _ in range(5):
        p = [np.round(rng.random(7), 1), np.round(rng.random(11), 1)]
        l = [(rng.random(7 * 16) < 0.3).astype(float), (rng.random(11 * 16) < 0.3).astype(float)]
        a, b = metrics.frame_level_auc(p, l), host_oracle.frame_level_auc(p, l)
        assert abs(a[0] - b[0]) < 1e-12 and abs(a[1] - b[1]) < 1e-12
    with pytest.raises(ValueError):
        metrics.frame_level_auc([np.zeros(3)], [np.zeros(40)])


def test_ground_truth_rule_matches_oracle():
    from anomaly_detection_on_video_amd.gt import frame_ground_truth, parse_temporal_annotations

    cases = [(4, (10, 20), (-1, -1)), (2, (5, 100), (30, 31)), (3, (-1, -1), (-1, -1)), (3, (0, 5), (7, 9)), (5, (3, -1), (60, 79))]
    for n, e1, e2 in cases:
        assert frame_ground_truth(n, e1, e2) == host_oracle.gt_from_annotation(n, e1, e2)
    txt = "Abuse028_x264.mp4  Abuse  165  240  -1  -1\nNormal_Videos_003_x264.mp4  Normal  -1  -1  -1  -1\n"
    a = parse_temporal_annotations(txt)
    assert a["Abuse028_x264"]["first_event"] == (165, 240) and a["Normal_Videos_003_x264"]["second_event"] == (-1, -1)


def test_weights_are_a_pure_function():
    from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_tensor

    a, b = synth_tensor("x", (3, 5)), synth_tensor("x", (3, 5))
    assert torch.equal(a, b) and not torch.equal(a, synth_tensor("y", (3, 5)))
    sd = synth_i3d_state_dict()
    assert abs(float(sd["layer3.2.conv2.weight"][5, 7, 0, 1, 2]) - (-0.0101)) < 5e-2  # stable across runs
    assert torch.equal(sd["conv1.weight"], synth_i3d_state_dict()["conv1.weight"])


# ------------------------------------------------------------------------------ sharding arithmetic + stream ring
def test_shard_bounds_cover_and_balance():
    from anomaly_detection_on_video_amd.dist import padded_local_rows, shard_bounds

    for n in (0, 1, 7, 32, 33, 320):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == padded_local_rows(n, world)


class _FakeBackbone(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))


def test_stream_ring_orders_videos_like_the_reference():
    """Feed global batches of feature rows whose value is the crop-clip's stream position: every
    video must come out exactly once, on its owner rank, as rows [v*P, (v+1)*P) viewed (clips, crops)
    -- the (n_clips, 10, C) layout of extract_features.py:93-100 -- for step sizes that do and do not
    divide the video length."""
    from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream

    class S(ExtractScoreStream):
        def score_video(self, feats):
            self.videos_scored += 1
            return feats[:, :, 0].clone()

    for world in (1, 2, 4, 8):
        for local_batch in (6, 5, 20):
            streams = [S(_FakeBackbone(), None, clips_per_video=4, ncrops=5, local_batch=local_batch, world=world, rank=r, feat_dim=8)
                       for r in range(world)]
            got, pos, gb = {}, 0, local_batch * world
            for _step in range(23):
                rows = torch.arange(pos, pos + gb, dtype=torch.float32).unsqueeze(1).expand(-1, 8).contiguous()
                for r, s in enumerate(streams):
                    for v, ids in s.ingest(rows):
                        assert v % world == r and v not in got
                        got[v] = ids
                pos += gb
            assert sorted(got) == list(range(pos // 20))
            for v, ids in got.items():
                assert torch.equal(ids, torch.arange(20 * v, 20 * v + 20, dtype=torch.float32).view(4, 5))
            with pytest.raises(ValueError):
                streams[0].ingest(torch.zeros(gb + 1, 8))


def test_stream_ring_variable_length_videos():
    """A UCF-Crime-shaped stream (SURVEY 8(d) cfg 3): every video has its own clip count, a global batch may end one video and
    begin the next (or hold several whole ones), the ring wraps many times.  Every video must come out exactly once, on rank
    v % world, as ONE contiguous window of rows [start_v, start_v + n_v * crops) viewed (n_v clips, crops) --
    /root/reference/extract_features.py:93-100 builds exactly that array per video."""
    import random

    from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream

    class S(ExtractScoreStream):
        def score_video(self, feats):
            self.videos_scored += 1
            assert feats.is_contiguous()
            return feats[:, :, 0].clone()

    rng = random.Random(5)
    for world, local_batch, crops, lo, hi in ((1, 32, 10, 50, 500), (2, 16, 10, 5, 60), (3, 7, 5, 1, 9), (8, 32, 10, 50, 500), (1, 64, 2, 1, 3)):
        clips = [rng.randint(lo, hi) for _ in range(11)]
        streams = [S(_FakeBackbone(), None, clips_per_video=clips, ncrops=crops, local_batch=local_batch, world=world, rank=r, feat_dim=4)
                   for r in range(world)]
        gb = local_batch * world
        assert streams[0].ring_rows % gb == 0 and streams[0].ring_rows >= max(clips) * crops + gb
        total = 3 * sum(clips) * crops + gb  # the clip-count list is walked three times: the ring wraps, the list cycles
        got, pos = {}, 0
        while pos < total:
            rows = torch.arange(pos, pos + gb, dtype=torch.float32).unsqueeze(1).expand(-1, 4).contiguous()
            for r, st in enumerate(streams):
                for v, ids in st.ingest(rows):
                    assert v % world == r and v not in got
                    got[v] = ids
            pos += gb
        starts, v, s0 = {}, 0, 0
        while s0 + clips[v % 11] * crops <= pos:
            starts[v] = s0
            s0 += clips[v % 11] * crops
            v += 1
        assert sorted(got) == sorted(starts) and len(got) >= 33
        for v, ids in got.items():
            n = clips[v % 11]
            assert torch.equal(ids, torch.arange(starts[v], starts[v] + n * crops, dtype=torch.float32).view(n, crops))
    # seek: a stream continued in the middle of video 1 scores video 1 when its last row arrives, then video 2
    st = S(_FakeBackbone(), None, clips_per_video=[3, 4, 2], ncrops=2, local_batch=2, feat_dim=4)
    st.seek(8)  # video 0 = rows 0..5, video 1 = rows 6..13
    seen = []
    for pos in range(8, 20, 2):
        rows = torch.arange(pos, pos + 2, dtype=torch.float32).unsqueeze(1).expand(-1, 4).contiguous()
        seen += st.ingest(rows)
    assert [v for v, _ in seen] == [1, 2]
    assert torch.equal(seen[0][1].reshape(-1)[2:], torch.arange(8, 14, dtype=torch.float32))  # rows 6, 7 were never fed: the ring's zeros
    assert torch.equal(seen[1][1].reshape(-1), torch.arange(14, 18, dtype=torch.float32))
    with pytest.raises(ValueError):
        st.seek(3)
    with pytest.raises(ValueError):
        S(_FakeBackbone(), None, clips_per_video=[4, 0], ncrops=2, local_batch=2, feat_dim=4)


def test_checkpoint_carries_the_keys_lightning_reads():
    """A last.ckpt written here must be loadable by the reference's Lightning trainer: its loader indexes
    `pytorch-lightning_version` unconditionally (migration step), then reads `state_dict` (keys prefixed `model.`, the
    LightningModule holds the net as `self.model`: /root/reference/src/runner.py:21-24), `optimizer_states`,
    `lr_schedulers`, `epoch`, `global_step`, `loops`, `callbacks`.  Lightning itself is absent from this image, so the key
    set is what is checked."""
    from anomaly_detection_on_video_amd.runner import VideoAnomalyDetectionRunner, checkpoint_state

    net = torch.nn.Linear(3, 2)
    runner = VideoAnomalyDetectionRunner(net, {"learning_rate": 1e-3, "weight_decay": 5e-4}, {"frames_per_clip": 16})
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    st = checkpoint_state(runner, opt, epoch=3, global_step=17, metrics_={"valid/rec_auc": 0.5})
    for key in ("pytorch-lightning_version", "state_dict", "optimizer_states", "lr_schedulers", "epoch", "global_step", "loops", "callbacks",
                "hyper_parameters"):
        assert key in st, key
    assert isinstance(st["pytorch-lightning_version"], str) and st["pytorch-lightning_version"].split(".")[0] == "2"
    assert set(st["state_dict"]) == {"model.weight", "model.bias"} and st["epoch"] == 3 and st["global_step"] == 17
    assert isinstance(st["callbacks"], dict) and isinstance(st["loops"], dict) and len(st["optimizer_states"]) == 1


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """Every struct that crosses the C ABI: size and field offsets of the ctypes mirror (_lib.py) equal what a C compiler makes of
    include/advhip.h -- a field added on one side only would otherwise shift every later operand silently."""
    import ctypes as C
    import shutil
    import subprocess

    from anomaly_detection_on_video_amd import _lib

    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = {"advhip_conv3d_desc": _lib.ConvDesc, "advhip_conv3d_epilogue": _lib.ConvEpilogue, "advhip_gemm_desc": _lib.GemmDesc,
             "advhip_nt_item": _lib.NtItem, "advhip_colsum_item": _lib.ColsumItem}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "advhip.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _t in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = {}
    for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        cname, what, val = ln.split()
        got[(cname, what)] = int(val)
    for cname, cls in pairs.items():
        assert got[(cname, "size")] == C.sizeof(cls), cname
        for fname, _t in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)
    assert C.sizeof(C.c_void_p) == 8
    # the multi-pack item table is built as a numpy record array (mgfn_ops.step_packs): the same check against its dtype
    import numpy as np

    from anomaly_detection_on_video_amd import mgfn_ops

    dt = np.dtype(mgfn_ops.PACK_ITEM_FIELDS)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "advhip.h"', "int main(void) {", '  printf("size %zu\\n", sizeof(advhip_pack_item));']
    lines += [f'  printf("{n} %zu\\n", offsetof(advhip_pack_item, {n}));' for n in dt.names]
    lines += ["  return 0;", "}"]
    src.write_text("\n".join(lines))
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(ln.split() for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    assert int(out["size"]) == dt.itemsize
    for n in dt.names:
        assert int(out[n]) == dt.fields[n][1], n
