"""The CPU oracle (oracle/*.py) is pinned against outputs of the reference's own code
(tests/golden/*.npz, made by tests/golden/make_golden.py).  CPU only; runs everywhere."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from anomaly_detection_on_video_amd.weights import (
    synth_i3d_state_dict,
    synth_input,
    synth_module_state_dict,
    synth_tensor,
)
from oracle import host_oracle, i3d_oracle, mgfn_oracle

TOL = 2e-5  # oracle vs reference on the same CPU ops: only reduction-order noise is allowed


@pytest.fixture(scope="module")
def i3d_sd():
    return synth_i3d_state_dict()


def test_conv_macs_matches_survey():
    assert i3d_oracle.conv_macs() == 16_414_572_544  # 16.415 GMAC per crop-clip


def test_i3d_state_dict_layout(i3d_sd):
    assert len(i3d_sd) == 318
    n_params = sum(v.numel() for k, v in i3d_sd.items() if v.is_floating_point() and "running" not in k)
    assert abs(n_params - 27.22e6) < 0.01e6  # 27.22 M parameters
    convs = [k for k in i3d_sd if k.endswith("conv1.weight") or k.endswith("conv2.weight") or k.endswith("conv3.weight") or k.endswith("downsample.0.weight")]
    assert len(convs) == 53


def test_i3d_fullnet_small_clip(i3d_sd):
    g = np.load(os.path.join(GOLDEN, "i3d_fullnet.npz"))
    x = synth_input((1, 3, 8, 112, 96), 7)
    y = i3d_oracle.i3d_forward(x, i3d_sd).reshape(1, 2048)
    assert rel_err(y, g["feat_small"]) < TOL


def test_i3d_fullnet_224_and_stage_stats(i3d_sd):
    g = np.load(os.path.join(GOLDEN, "i3d_fullnet.npz"))
    x = synth_input((2, 3, 16, 224, 224), 0)
    taps = {}
    y = i3d_oracle.i3d_forward(x, i3d_sd, lambda n, v: taps.__setitem__(n, v)).reshape(2, 2048)
    assert rel_err(y, g["feat_seed0"]) < TOL
    for name, v in taps.items():
        key = f"stat_{name}"
        if key not in g.files:
            continue
        ref = g[key]
        assert tuple(g[f"shape_{name}"]) == tuple(v.shape), name
        flat = v.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, 64).long()
        assert abs(v.mean().item() - ref[0]) <= 1e-4 * max(1.0, abs(ref[0])), name
        assert abs(v.std().item() - ref[1]) <= 1e-4 * max(1.0, abs(ref[1])), name
        assert rel_err(flat[idx], ref[3:]) < TOL, name


def _block_sd(name, blk_keys_shapes):
    sd = {}
    for k, shape, is_float in blk_keys_shapes:
        key = f"micro.{name}.{k}"
        leaf = k.rsplit(".", 1)[-1]
        if not is_float:
            sd[k] = torch.zeros(shape, dtype=torch.long)
        elif len(shape) == 5:
            fan = int(np.prod(shape[1:]))
            sd[k] = synth_tensor(key, shape, scale=float(np.sqrt(6.0 / fan)))
        elif leaf == "running_var":
            sd[k] = synth_tensor(key, shape, scale=0.5, offset=1.0)
        elif leaf == "weight":
            sd[k] = synth_tensor(key, shape, scale=0.5, offset=1.0)
        else:
            sd[k] = synth_tensor(key, shape, scale=0.25)
    return sd


def block_keys(inpl, planes, tc, has_ds):
    def bn(p, c):
        return [(f"{p}.weight", (c,), True), (f"{p}.bias", (c,), True), (f"{p}.running_mean", (c,), True),
                (f"{p}.running_var", (c,), True), (f"{p}.num_batches_tracked", (), False)]

    ks = [("conv1.weight", (planes, inpl, 1 + 2 * tc, 1, 1), True)] + bn("bn1", planes)
    ks += [("conv2.weight", (planes, planes, 1, 3, 3), True)] + bn("bn2", planes)
    ks += [("conv3.weight", (planes * 4, planes, 1, 1, 1), True)] + bn("bn3", planes * 4)
    if has_ds:
        ks += [("downsample.0.weight", (planes * 4, inpl, 1, 1, 1), True)] + bn("downsample.1", planes * 4)
    return ks


BLOCK_CASES = ["l1b0", "l1b1", "l2b0", "l2b1", "l3b0", "l4b0", "l4b1"]


def micro_block_case(name):
    g = np.load(os.path.join(GOLDEN, "i3d_blocks.npz"))
    inpl, planes, stride, tc, has_ds, b, t, h, w = (int(v) for v in g[f"{name}_cfg"])
    sd = _block_sd(name, block_keys(inpl, planes, tc, bool(has_ds)))
    x = synth_tensor(f"micro.{name}.x", (b, inpl, t, h, w), scale=2.0)
    return dict(inpl=inpl, planes=planes, stride=stride, tc=tc, has_ds=bool(has_ds), sd=sd, x=x, y=torch.from_numpy(g[f"{name}_y"]))


@pytest.mark.parametrize("name", BLOCK_CASES)
def test_bottleneck_micro(name):
    c = micro_block_case(name)
    sd = {f"blk.{k}": v for k, v in c["sd"].items()}
    with torch.no_grad():
        y = i3d_oracle.bottleneck(c["x"], sd, "blk", c["stride"], c["tc"], c["has_ds"])
    assert rel_err(y, c["y"]) < TOL


# ------------------------------------------------------------------------------ MGFN
def mgfn_inputs(bs, t, seed):
    feats = synth_tensor(f"mgfn.x/{seed}", (bs, 10, t, 2048), scale=1.0).abs() * 2.0
    bump = synth_tensor(f"mgfn.bump/{seed}", (bs, 1, t, 1), scale=1.0).abs()
    feats = feats * (1.0 + bump)
    mag = torch.linalg.norm(feats, dim=3, keepdim=True)
    return torch.cat([feats, mag], dim=3)


def mgfn_state_dict(keys=None):
    """Deterministic MGFN weights; needs only the key->shape map, which the product model provides."""
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, mgfn_param_shapes

    shapes = mgfn_param_shapes(MGFNConfig())
    if keys is not None:
        assert list(shapes.keys()) == list(keys)

    class _Shim(torch.nn.Module):
        def state_dict(self_inner):
            return {k: torch.zeros(s, dtype=(torch.long if k.endswith("num_batches_tracked") else torch.float32)) for k, s in shapes.items()}

    return synth_module_state_dict(_Shim(), gain=1.0)


def test_mgfn_eval_split_and_losses():
    g = np.load(os.path.join(GOLDEN, "mgfn.npz"))
    sd = mgfn_state_dict(list(g["state_keys"]))
    video = mgfn_inputs(4, 32, 0)
    nl, al = torch.zeros(2), torch.ones(2)
    params = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in sd.items()}
    o = mgfn_oracle.mgfn_forward(video, params, abnormal_labels=al, normal_labels=nl, training=False, force_split=True)
    assert rel_err(o.scores, g["evalsplit_scores"]) < TOL
    assert rel_err(o.abnormal_scores, g["evalsplit_abn_scores"]) < TOL
    assert rel_err(o.normal_scores, g["evalsplit_nor_scores"]) < TOL
    assert rel_err(o.a_feat_magnitude.norm(p=1, dim=2), g["evalsplit_a_feat_l1"]) < TOL
    assert rel_err(o.n_feat_magnitude[..., :32], g["evalsplit_n_feat_head"]) < TOL
    assert rel_err(o.loss, g["evalsplit_loss"]) < TOL
    assert rel_err(o.terms["smooth"], g["evalsplit_loss_smooth"]) < TOL
    assert rel_err(o.terms["sparse"], g["evalsplit_loss_sparse"]) < TOL
    assert rel_err(o.terms["mgfn"], g["evalsplit_loss_mgfn"]) < TOL
    o.loss.backward()
    assert rel_err(params["fc.weight"].grad, g["evalsplit_grad_fc_w"]) < 1e-4
    gt = params["backbone.amplifier.to_tokens.weight"].grad
    assert rel_err(gt.norm(), g["evalsplit_grad_to_tokens_w_norm"]) < 1e-4


def test_mgfn_training_branch_with_injected_mask():
    g = np.load(os.path.join(GOLDEN, "mgfn.npz"))
    sd = mgfn_state_dict()
    video = mgfn_inputs(4, 32, 0)
    nl, al = torch.zeros(2), torch.ones(2)
    o = mgfn_oracle.mgfn_forward(
        video, sd, abnormal_labels=al, normal_labels=nl, training=True,
        keep_abn=torch.from_numpy(g["train_keep_abn"]), keep_nor=torch.from_numpy(g["train_keep_nor"]),
    )
    assert rel_err(o.scores, g["train_scores"]) < TOL
    assert rel_err(o.abnormal_scores, g["train_abn_scores"]) < TOL
    assert rel_err(o.a_feat_magnitude.norm(p=1, dim=2), g["train_a_feat_l1"]) < TOL
    assert rel_err(o.loss, g["train_loss"]) < TOL


def test_mgfn_eval_no_split_odd_T():
    g = np.load(os.path.join(GOLDEN, "mgfn.npz"))
    sd = mgfn_state_dict()
    o = mgfn_oracle.mgfn_forward(mgfn_inputs(1, 57, 3), sd)
    assert o.loss is None
    assert rel_err(o.scores, g["eval57_scores"]) < TOL
    assert rel_err(o.abnormal_scores, g["eval57_abn_scores"]) < TOL
    assert torch.equal(o.abnormal_scores, o.normal_scores)


def test_loss_known_answers():
    g = np.load(os.path.join(GOLDEN, "loss.npz"))
    s = synth_tensor("loss.scores", (6, 32, 1), scale=0.5, offset=0.5)
    assert rel_err(mgfn_oracle.smoothness_loss(s), g["smooth"]) < 1e-6
    assert rel_err(mgfn_oracle.sparsity_loss(s[:3].reshape(-1)), g["sparse"]) < 1e-6
    a = synth_tensor("loss.a", (30, 3), scale=100.0, offset=150.0)
    b = synth_tensor("loss.b", (30, 3), scale=100.0, offset=120.0)
    assert rel_err(mgfn_oracle.contrastive_loss(a, b, 1), g["con1"]) < 1e-6
    assert rel_err(mgfn_oracle.contrastive_loss(a, b, 0), g["con0"]) < 1e-6


# ------------------------------------------------------------------------------ host functions
@pytest.mark.parametrize("n", [5, 32, 33, 100])
def test_segment_oracle(n):
    g = np.load(os.path.join(GOLDEN, "host.npz"))
    feats = synth_tensor(f"segment/{n}", (n, 10, 64), scale=3.0).numpy()
    np.testing.assert_array_equal(host_oracle.segment_features(feats, 32), g[f"segment_{n}"])


def test_add_magnitude_oracle():
    g = np.load(os.path.join(GOLDEN, "host.npz"))
    f = synth_tensor("addmag", (10, 32, 48), scale=2.0).numpy()
    np.testing.assert_array_equal(host_oracle.add_magnitude(f), g["addmag"])


def test_auc_oracle_matches_sklearn_known_answer():
    g = np.load(os.path.join(GOLDEN, "auc.npz"))
    assert abs(host_oracle.roc_auc(g["labels"], g["preds"]) - float(g["roc_auc"])) < 1e-12
    assert abs(host_oracle.pr_auc(g["labels"], g["preds"]) - float(g["pr_auc"])) < 1e-12


def test_gt_rule():
    gt = host_oracle.gt_from_annotation(4, (10, 20), (-1, -1))
    assert len(gt) == 64 and sum(gt) == 11 and gt[10] == 1.0 and gt[21] == 0.0
    gt = host_oracle.gt_from_annotation(2, (5, 100), (30, 31))
    assert sum(gt) == 27  # clipped at num_frame
    assert sum(host_oracle.gt_from_annotation(2, (-1, -1), (-1, -1))) == 0


# ------------------------------------------------------------------------------ NonLocalBlock (I3Res50(use_nl=True))
def test_nonlocal_block_oracle_vs_reference_golden():
    """oracle.nonlocal_block vs the reference's own NonLocalBlock outputs (src/i3d.py:124-195)."""
    from anomaly_detection_on_video_amd.weights import NONLOCAL_CASES, synth_nonlocal_case

    g = np.load(os.path.join(GOLDEN, "nonlocal.npz"))
    for name in NONLOCAL_CASES:
        _dim, _inner, sd, x = synth_nonlocal_case(name)
        y = i3d_oracle.nonlocal_block(x, {f"nl.{k}": v for k, v in sd.items()}, "nl")
        assert rel_err(y, g[f"{name}_y"]) < 1e-5, name


def test_i3d_with_nonlocal_blocks_oracle_vs_reference_golden():
    from anomaly_detection_on_video_amd.weights import nonlocal_positions, synth_i3d_state_dict

    sd = synth_i3d_state_dict(use_nl=True)
    assert sum(1 for k in sd if k.endswith(".nl.theta.weight")) == len(nonlocal_positions(True)) == 5
    g = np.load(os.path.join(GOLDEN, "nonlocal.npz"))
    y = i3d_oracle.i3d_forward(synth_input((1, 3, 8, 112, 96), 7), sd)
    assert rel_err(y.reshape(1, 2048), g["feat_nl_small"]) < 1e-5
    assert np.isfinite(g["feat_nl_64"]).all() and float(np.abs(g["feat_nl_64"]).max()) > 0


@pytest.mark.parametrize("length", [5, 8, 16])
def test_clip_preprocessing_oracle_vs_reference_golden(length):
    """host_oracle.ten_crop_clips against the reference's own GroupStandardizationTenCrop + LoopPad + the two permutes
    (src/gtransforms.py:57-73,115-132; src/dataset.py:195; extract_features.py:83), bit for bit.  Crop-sized frames: no crop
    geometry (torchvision's, absent) enters, only the arithmetic, the padding rule and the layout."""
    from oracle import host_oracle

    g = np.load(os.path.join(GOLDEN, "preproc.npz"))
    got = host_oracle.ten_crop_clips(g[f"frames_{length}"], 16, 8)
    assert got.shape == g[f"clip_{length}"].shape == (1, 10, 3, 16, 8, 8)
    assert np.array_equal(got, g[f"clip_{length}"])
    # LoopPad: slot i of the padded clip holds source frame i % length (lengths that divide 16 or not)
    assert np.array_equal(g[f"looppad_index_{length}"], np.arange(16) % length)


def test_standardisation_order_vs_reference_golden():
    """(x - mean) / std as two fp32 operations in that order (`t.sub_(m).div_(s)`, gtransforms.py:69-72) -- not one fused
    multiply-add, which rounds differently."""
    g = np.load(os.path.join(GOLDEN, "preproc.npz"))
    x = g["std_in"]
    assert np.array_equal((x - np.float32(114.75)) / np.float32(57.375), g["std_out"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree is absent (GPU box): the committed fixtures stand alone")
def test_committed_goldens_regenerate_from_the_reference(tmp_path):
    """The oracle's pin checks itself: tests/golden/make_golden.py, run the documented way (no target arguments), imports
    the reference and must reproduce every committed array bit for bit.  A drift of the generator, of the synthetic
    inputs (weights.py) or of the import order its placeholders depend on fails here instead of going unnoticed."""
    import subprocess
    import sys

    script = os.path.join(GOLDEN, "make_golden.py")
    r = subprocess.run([sys.executable, "-B", script, "--out", str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    committed = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert committed == sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz")), "the default invocation does not write every committed fixture"
    arrays = 0
    for f in committed:
        a, b = np.load(os.path.join(GOLDEN, f)), np.load(os.path.join(tmp_path, f))
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (f, k)
            assert np.array_equal(a[k], b[k]), f"{f}:{k} differs from the committed fixture"
            arrays += 1
    assert arrays >= 131
