#!/usr/bin/env python3
"""Generate the golden parity vectors under tests/golden/ by running the REFERENCE's own code.

Dev-only: needs /root/reference (read-only), which exists in the build container and NOT on
the GPU box.  The outputs (*.npz, small) are committed; nothing at test/bench time reads the
reference.  Run from anywhere:

    python -B tests/golden/make_golden.py [--out DIR] [i3d nonlocal mgfn host preproc]

With no target every fixture is regenerated; `--out DIR` writes them somewhere else than tests/golden/ (what
tests/test_oracle_golden.py::test_committed_goldens_regenerate_from_the_reference does, to compare array for array
with the committed files).  Whatever the order on the command line, the targets run in one fixed order: the ones that
import `transformers` (through the reference's modeling_mgfn) first, the ones that register the torchvision / decord
placeholders last.

The reference imports three third-party packages that are absent from this image and are not
used by the code under test (pytorchvideo: only the `i3d_8x8_r50` factory branch; decord /
torchvision: only real-video decoding).  Empty placeholder modules are registered for them so
the reference's modules import; no reference arithmetic is replaced.

Weights/inputs come from `anomaly_detection_on_video_amd.weights` (a pure hash function), so
tests regenerate the identical tensors without the reference.
"""
from __future__ import annotations

import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = HERE  # where the .npz files go (--out)
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

# reference first (both trees have a top-level `src` package; here we want the reference's)
sys.path.insert(0, REF)
sys.path.append(REPO)
sys.dont_write_bytecode = True


def _placeholder(name, **attrs):
    import importlib.machinery

    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_placeholder("pytorchvideo")
_placeholder("pytorchvideo.models")
_placeholder("pytorchvideo.models.resnet", create_resnet=None)


def _video_io_placeholders():
    # only for importing the reference's extract_features.py / src/dataset.py / src/gtransforms.py (golden_host,
    # golden_preproc).  `transformers` probes torchvision's import spec when its lazy modules load, and dies on a
    # placeholder: so load what the reference's modeling_mgfn needs from it BEFORE the placeholder exists, and run the
    # targets in TARGETS' order (placeholders last).
    if "torchvision" in sys.modules:
        return
    try:
        from transformers import PreTrainedModel, PretrainedConfig  # noqa: F401
    except Exception:  # pragma: no cover - transformers absent: nothing probes torchvision then
        pass
    _placeholder("decord")
    tv = _placeholder("torchvision")
    tv.transforms = _placeholder("torchvision.transforms")


from anomaly_detection_on_video_amd.weights import (  # noqa: E402
    NONLOCAL_CASES,
    synth_nonlocal_case,
    synth_i3d_state_dict,
    synth_input,
    synth_module_state_dict,
    synth_tensor,
)

torch.manual_seed(0)
torch.set_num_threads(8)


def sample64(v: torch.Tensor) -> np.ndarray:
    flat = v.reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, 64).long()
    return flat[idx].numpy().copy()


# ------------------------------------------------------------------------------- I3D
def golden_i3d():
    from src.i3d import I3Res50, Bottleneck  # reference
    import torch.nn as nn

    model = I3Res50(use_nl=False)
    sd = synth_i3d_state_dict()
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert list(model.state_dict().keys()) == list(sd.keys()), "state-dict key order differs"
    model.eval()

    out = {}
    # (i) full net, two seeds, (2,3,16,224,224)
    for seed in (0, 1):
        x = synth_input((2, 3, 16, 224, 224), seed)
        stats = {}

        def hook(name):
            def f(_m, _i, o):
                stats[name] = o.detach()
            return f

        hs = []
        if seed == 0:
            hs.append(model.relu.register_forward_hook(lambda m, i, o: stats.setdefault("stem", o.detach().clone())))
            hs.append(model.maxpool1.register_forward_hook(hook("maxpool1")))
            hs.append(model.maxpool2.register_forward_hook(hook("maxpool2")))
            for ln in ("layer1", "layer2", "layer3", "layer4"):
                hs.append(getattr(model, ln).register_forward_hook(hook(ln)))
                for bi, blk in enumerate(getattr(model, ln)):
                    hs.append(blk.register_forward_hook(hook(f"{ln}.{bi}")))
        with torch.no_grad():
            y = model(x)
        for h in hs:
            h.remove()
        assert y.shape == (2, 2048, 1, 1, 1)
        out[f"feat_seed{seed}"] = y.reshape(2, 2048).numpy()
        for name, v in stats.items():
            out[f"stat_{name}"] = np.concatenate(
                [np.array([v.mean().item(), v.std().item(), v.abs().max().item()], dtype=np.float32), sample64(v)]
            )
            out[f"shape_{name}"] = np.array(v.shape, dtype=np.int64)
    # a (1,3,16,224,224) and odd-size clip to pin the non-224 path: (1,3,8,112,96)
    x = synth_input((1, 3, 8, 112, 96), 7)
    with torch.no_grad():
        out["feat_small"] = model(x).reshape(1, 2048).numpy()
    np.savez_compressed(os.path.join(OUT, "i3d_fullnet.npz"), **out)
    print("i3d_fullnet.npz", {k: v.shape for k, v in out.items() if k.startswith("feat")})

    # (iii) block-level micro-goldens through the reference's own Bottleneck on reduced shapes
    micro = {}
    cases = [
        # name, inplanes, planes, stride, temp_conv, has_ds, input (B,T,H,W)
        ("l1b0", 64, 64, 1, 1, True, (2, 4, 7, 5)),
        ("l1b1", 256, 64, 1, 1, False, (1, 4, 6, 7)),
        ("l2b0", 256, 128, 2, 1, True, (2, 2, 9, 11)),
        ("l2b1", 512, 128, 1, 0, False, (1, 2, 5, 6)),
        ("l3b0", 512, 256, 2, 1, True, (1, 2, 10, 12)),
        ("l4b0", 1024, 512, 2, 0, True, (3, 2, 7, 7)),
        ("l4b1", 2048, 512, 1, 1, False, (2, 2, 4, 4)),
    ]
    for name, inpl, planes, stride, tc, has_ds, (b, t, h, w) in cases:
        ds = None
        if has_ds:
            ds = nn.Sequential(
                nn.Conv3d(inpl, planes * 4, kernel_size=1, stride=(1, stride, stride), bias=False),
                nn.BatchNorm3d(planes * 4),
            )
        blk = Bottleneck(inpl, planes, stride, ds, tc, 1, False).eval()
        bsd = {}
        for k, ref in blk.state_dict().items():
            key = f"micro.{name}.{k}"
            leaf = k.rsplit(".", 1)[-1]
            if not ref.is_floating_point():
                bsd[k] = torch.zeros_like(ref)
            elif ref.dim() == 5:
                fan = ref[0].numel()
                bsd[k] = synth_tensor(key, tuple(ref.shape), scale=float(np.sqrt(6.0 / fan)))
            elif leaf == "running_var":
                bsd[k] = synth_tensor(key, tuple(ref.shape), scale=0.5, offset=1.0)
            elif leaf == "weight":
                bsd[k] = synth_tensor(key, tuple(ref.shape), scale=0.5, offset=1.0)
            else:
                bsd[k] = synth_tensor(key, tuple(ref.shape), scale=0.25)
        blk.load_state_dict(bsd, strict=True)
        x = synth_tensor(f"micro.{name}.x", (b, inpl, t, h, w), scale=2.0)
        with torch.no_grad():
            y = blk(x)
        micro[f"{name}_y"] = y.numpy()
        micro[f"{name}_cfg"] = np.array([inpl, planes, stride, tc, int(has_ds), b, t, h, w], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "i3d_blocks.npz"), **micro)
    print("i3d_blocks.npz", {k: v.shape for k, v in micro.items() if k.endswith("_y")})


def golden_nonlocal():
    """The reference's own NonLocalBlock (src/i3d.py:124-195) and I3Res50(use_nl=True) -- dead code under the factory's
    use_nl=False, but importable, so it can be pinned."""
    from src.i3d import I3Res50, NonLocalBlock  # reference

    out = {}
    for name in NONLOCAL_CASES:
        dim, inner, sd, x = synth_nonlocal_case(name)
        blk = NonLocalBlock(dim, dim, inner).eval()
        blk.load_state_dict(sd, strict=True)
        with torch.no_grad():
            out[f"{name}_y"] = blk(x).numpy()
    model = I3Res50(use_nl=True)
    sd = synth_i3d_state_dict(use_nl=True)
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model.eval()
    with torch.no_grad():
        out["feat_nl_small"] = model(synth_input((1, 3, 8, 112, 96), 7)).reshape(1, 2048).numpy()
        out["feat_nl_64"] = model(synth_input((2, 3, 16, 64, 64), 3)).reshape(2, 2048).numpy()
    np.savez_compressed(os.path.join(OUT, "nonlocal.npz"), **out)
    print("nonlocal.npz", {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------- MGFN
class _InjectedDropout(torch.nn.Module):
    """Stands in for `model.dropout` so the training branch's Bernoulli mask is a known input.
    The reference calls it abnormal-first, then normal (modeling_mgfn.py:364-372)."""

    def __init__(self, masks):
        super().__init__()
        self.masks = list(masks)
        self.calls = 0

    def forward(self, ones):
        m = self.masks[self.calls]
        self.calls += 1
        assert m.shape == ones.shape
        return ones * m


def mgfn_inputs(bs, t, seed):
    feats = synth_tensor(f"mgfn.x/{seed}", (bs, 10, t, 2048), scale=1.0).abs() * 2.0
    # make the abnormal half (second half) a bit "louder" on a few segments so top-k is non-trivial
    bump = synth_tensor(f"mgfn.bump/{seed}", (bs, 1, t, 1), scale=1.0).abs()
    feats = feats * (1.0 + bump)
    mag = torch.linalg.norm(feats, dim=3, keepdim=True)
    return torch.cat([feats, mag], dim=3)


def golden_mgfn():
    from src.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection  # reference
    from src.loss import TemporalSmoothnessLoss, SparsityLoss, MGFNLoss, ContrastiveLoss

    model = MGFNForVideoAnomalyDetection(MGFNConfig())
    sd = synth_module_state_dict(model, gain=1.0)
    model.load_state_dict(sd, strict=True)
    out = {"state_keys": np.array(list(model.state_dict().keys()))}

    def run(video, al, nl, tag, training=False, force_split=False, masks=None):
        model.train(training)
        model.force_split = force_split
        orig = model.dropout
        if masks is not None:
            model.dropout = _InjectedDropout(masks)
        video = video.clone().requires_grad_(False)
        for p in model.parameters():
            p.grad = None
        o = model(video=video, abnormal_labels=al, normal_labels=nl)
        model.dropout = orig
        out[f"{tag}_scores"] = o.scores.detach().numpy()
        out[f"{tag}_abn_scores"] = o.abnormal_scores.detach().numpy()
        out[f"{tag}_nor_scores"] = o.normal_scores.detach().numpy()
        # selected features: keep the L1 norms the loss consumes plus a 32-column head (size)
        out[f"{tag}_a_feat_l1"] = o.a_feat_magnitude.detach().norm(p=1, dim=2).numpy()
        out[f"{tag}_n_feat_l1"] = o.n_feat_magnitude.detach().norm(p=1, dim=2).numpy()
        out[f"{tag}_a_feat_head"] = o.a_feat_magnitude.detach()[..., :32].numpy().copy()
        out[f"{tag}_n_feat_head"] = o.n_feat_magnitude.detach()[..., :32].numpy().copy()
        if o.loss is not None:
            out[f"{tag}_loss"] = o.loss.detach().numpy()
            bs = video.shape[0]
            out[f"{tag}_loss_smooth"] = TemporalSmoothnessLoss()(o.scores).detach().numpy()
            out[f"{tag}_loss_sparse"] = SparsityLoss()(o.scores[: bs // 2].view(-1)).detach().numpy()
            out[f"{tag}_loss_mgfn"] = MGFNLoss()(
                abnormal_scores=o.abnormal_scores, normal_scores=o.normal_scores,
                abnormal_labels=al, normal_labels=nl,
                a_feat_magnitude=o.a_feat_magnitude, n_feat_magnitude=o.n_feat_magnitude,
            ).detach().numpy()
            o.loss.backward()
            out[f"{tag}_grad_fc_w"] = model.fc.weight.grad.detach().numpy().copy()
            out[f"{tag}_grad_to_tokens_w_sample"] = sample64(model.backbone.amplifier.to_tokens.weight.grad.detach())
            out[f"{tag}_grad_to_tokens_w_norm"] = model.backbone.amplifier.to_tokens.weight.grad.norm().detach().numpy()
        return o

    bs, t = 4, 32
    video = mgfn_inputs(bs, t, 0)
    nl = torch.zeros(bs // 2)
    al = torch.ones(bs // 2)
    # (iv) eval + force_split (deterministic), with losses and gradients
    run(video, al, nl, "evalsplit", training=False, force_split=True)
    # (v) training branch with an injected keep-mask (p_drop 0.7 -> multiplier 1/0.3 or 0)
    keep = (synth_tensor("mgfn.keep", (2, bs // 2, t), scale=0.5, offset=0.5) < 0.3).float() / 0.3
    out["train_keep_abn"] = keep[0].numpy()
    out["train_keep_nor"] = keep[1].numpy()
    run(video, al, nl, "train", training=True, force_split=False, masks=[keep[0], keep[1]])
    # BatchNorm1d running stats after that one training step (momentum 0.1)
    model.load_state_dict(sd, strict=True)
    # eval, no split, odd T (validation shape (1,10,T,2049))
    video57 = mgfn_inputs(1, 57, 3)
    run(video57, None, None, "eval57", training=False, force_split=False)
    np.savez_compressed(os.path.join(OUT, "mgfn.npz"), **out)
    print("mgfn.npz", {k: v.shape for k, v in out.items() if "scores" in k or "loss" in k})

    # stand-alone loss known-answers
    lo = {}
    s = synth_tensor("loss.scores", (6, 32, 1), scale=0.5, offset=0.5)
    lo["smooth"] = TemporalSmoothnessLoss()(s).numpy()
    lo["sparse"] = SparsityLoss()(s[:3].reshape(-1)).numpy()
    a = synth_tensor("loss.a", (30, 3), scale=100.0, offset=150.0)
    b = synth_tensor("loss.b", (30, 3), scale=100.0, offset=120.0)
    lo["con1"] = ContrastiveLoss()(a, b, 1).numpy()
    lo["con0"] = ContrastiveLoss()(a, b, 0).numpy()
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **lo)
    print("loss.npz", {k: float(v) for k, v in lo.items()})


# ------------------------------------------------------------------------------- host funcs
def golden_host():
    _video_io_placeholders()
    import extract_features as ref_extract  # reference (module-level imports satisfied by placeholders)
    from src.dataset import FeatureDataset

    out = {}
    for n in (5, 32, 33, 100):
        feats = synth_tensor(f"segment/{n}", (n, 10, 64), scale=3.0).numpy()
        with tempfile.TemporaryDirectory() as d:
            src_dir, dst_dir = os.path.join(d, "in"), os.path.join(d, "out")
            os.makedirs(src_dir)
            os.makedirs(dst_dir)
            np.save(os.path.join(src_dir, "v_i3d.npy"), feats)
            ref_extract.segment(src_dir, dst_dir, 32)
            out[f"segment_{n}"] = np.load(os.path.join(dst_dir, "v_i3d.npy"))
    f = synth_tensor("addmag", (10, 32, 48), scale=2.0).numpy()
    ds = FeatureDataset(["a_Normal.npy"], {"a_Normal.npy": f})
    item = ds[0]
    out["addmag"] = item["feature"]
    out["addmag_anomaly"] = item["anomaly"]
    np.savez_compressed(os.path.join(OUT, "host.npz"), **out)
    print("host.npz", {k: v.shape for k, v in out.items()})

    # sklearn known-answer for the AUC restatement (runner.py:73-76)
    from sklearn.metrics import auc, roc_curve, precision_recall_curve

    preds = np.repeat((synth_tensor("auc.p", (97,), scale=0.5, offset=0.5).numpy() * 20).round() / 20, 16)
    labels = (synth_tensor("auc.l", (97 * 16,), scale=0.5, offset=0.5).numpy() < 0.2).astype(np.float32)
    fpr, tpr, _ = roc_curve(labels.tolist(), preds)
    prec, rec, _ = precision_recall_curve(labels.tolist(), preds)
    np.savez_compressed(
        os.path.join(OUT, "auc.npz"), preds=preds, labels=labels,
        roc_auc=np.array(auc(fpr, tpr)), pr_auc=np.array(auc(rec, prec)),
    )
    print("auc.npz", auc(fpr, tpr), auc(rec, prec))


# ------------------------------------------------------------------------------- clip pre-processing
def golden_preproc():
    """The reference's own GroupStandardizationTenCrop / LoopPad (src/gtransforms.py:57-73, 115-132) and the two layout
    permutes (src/dataset.py:195, extract_features.py:83) on uint8-valued input.  The frames are exactly crop-sized, so
    torchvision's five crops all coincide with the frame and the other five with its mirror image: no crop GEOMETRY (which
    is torchvision's, absent here) enters -- what is pinned is the arithmetic (sub then div, fp32, in place), the padding
    rule and the (clip, crop, C, T, H, W) layout the backbone receives."""
    _video_io_placeholders()
    from src import gtransforms as g  # reference

    std, pad = g.GroupStandardizationTenCrop(), g.LoopPad(max_len=16)
    out = {}
    for length in (5, 8, 16):
        frames = (synth_tensor(f"preproc.frames/{length}", (length, 8, 8, 3), scale=0.5, offset=0.5) * 255).round().clamp(0, 255).to(torch.uint8)
        chw = frames.permute(0, 3, 1, 2)
        # what GroupTenCrop + ToTensorTenCrop (gtransforms.py:20-38) hand on for a crop-sized frame: (len, 10, C, h, w) floats
        crops = torch.stack([chw] * 5 + [chw.flip(-1)] * 5, dim=1).float()
        item = pad(std(crops.clone())).permute(1, 0, 2, 3, 4)        # TenCropVideoFrameDataset.__getitem__, dataset.py:191-195
        batch = item.unsqueeze(0).permute(0, 1, 3, 2, 4, 5)          # _extract, extract_features.py:81-83
        out[f"frames_{length}"] = frames.numpy()
        out[f"clip_{length}"] = batch.contiguous().numpy()           # (1, 10, 3, 16, 8, 8)
        out[f"looppad_index_{length}"] = pad(torch.arange(length, dtype=torch.float32).view(length, 1)).view(-1).numpy()
    x = (synth_tensor("preproc.std_in", (5, 10, 3, 8, 8), scale=0.5, offset=0.5) * 255).round().clamp(0, 255)
    out["std_in"] = x.numpy().copy()
    out["std_out"] = std(x.clone()).numpy()
    np.savez_compressed(os.path.join(OUT, "preproc.npz"), **out)
    print("preproc.npz", {k: v.shape for k, v in out.items()})


# fixed run order: everything that imports `transformers` before the torchvision / decord placeholders are registered
TARGETS = (("i3d", golden_i3d), ("nonlocal", golden_nonlocal), ("mgfn", golden_mgfn), ("host", golden_host), ("preproc", golden_preproc))


def main(argv=None) -> int:
    import argparse

    global OUT
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", default=HERE, help="directory the .npz fixtures are written to (default: tests/golden/)")
    ap.add_argument("targets", nargs="*", help="any of: " + " ".join(n for n, _ in TARGETS) + " (default: all)")
    ns = ap.parse_args(argv)
    unknown = set(ns.targets) - {n for n, _ in TARGETS}
    if unknown:
        ap.error(f"unknown target(s) {sorted(unknown)}")
    OUT = os.path.abspath(ns.out)
    os.makedirs(OUT, exist_ok=True)
    which = set(ns.targets) or {n for n, _ in TARGETS}
    for name, fn in TARGETS:
        if name in which:
            fn()
    return 0


if __name__ == "__main__":
    sys.exit(main())
