"""GPU tests of the callers either side of the kernels: MIL training steps (BASELINE config 4),
the run.py entry point on a synthetic corpus, and the extraction driver (extract -> segment)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import REPO, rel_err
from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_module_state_dict, synth_tensor
from test_oracle_golden import mgfn_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3


def test_eval_after_fused_optimizer_steps_sees_the_new_weights(monkeypatch):
    """torch's fused Adam updates parameters WITHOUT moving their version counters: the packed / folded weight copies the
    inference path caches must not survive a training step.  eval -> 2 fused steps (an eval in between) -> eval: the logits
    of the HIP path must equal those of the plain-torch path on the SAME updated parameters (a stale packed copy would
    reproduce the logits from before the step)."""
    from anomaly_detection_on_video_amd import mgfn_ops
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection

    model = MGFNForVideoAnomalyDetection(MGFNConfig())
    model.load_state_dict(synth_module_state_dict(model))
    model = model.to(DEV)
    video = mgfn_inputs(4, 32, 3).to(DEV)
    nl, al = torch.zeros(2, device=DEV), torch.ones(2, device=DEV)
    grabbed = []
    # the body's output (the scores saturate; the head may run as one fused launch that never calls model.fc)
    model.backbone.register_forward_hook(lambda m, i, o: grabbed.append(o.outputs.detach().cpu()))

    def logits(torch_path=False):
        model.eval()
        grabbed.clear()
        with torch.no_grad(), monkeypatch.context() as mp:
            if torch_path:
                mp.setattr(mgfn_ops, "eligible", lambda *a: False)
                mp.setattr(mgfn_ops, "fused_ok", lambda x: False)
            model(video=video)
        return grabbed[-1]

    before = logits()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4, fused=True)
    versions = [p._version for p in model.parameters()]
    for step in range(2):
        model.train()
        opt.zero_grad()
        model(video=mgfn_inputs(4, 32, 20 + step).to(DEV), abnormal_labels=al, normal_labels=nl).loss.backward()
        opt.step()
        logits()  # an eval between the steps re-creates the cached copies: the next step must drop them again
    assert versions == [p._version for p in model.parameters()], "fused Adam moved the version counters after all"
    after, ref = logits(), logits(torch_path=True)
    assert rel_err(after, before) > 1e-2, "two optimizer steps changed nothing"
    assert rel_err(after, ref) < 1e-4, rel_err(after, ref)


def test_adam_training_steps_track_the_cpu_oracle():
    """3 Adam steps (lr 1e-3, weight decay 5e-4: runner.py:53-59) in train mode (BatchNorm1d batch
    statistics) with the top-k keep-mask pinned to ones: loss curve and weights vs the oracle."""
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection
    from oracle import mgfn_oracle

    model = MGFNForVideoAnomalyDetection(MGFNConfig())
    sd = synth_module_state_dict(model)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    ones = torch.ones(2, 32)
    model.injected_keep = (ones.to(DEV), ones.to(DEV))
    params = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k and "num_batches" not in k) for k, v in sd.items()}
    learn = [k for k, v in params.items() if v.requires_grad]
    assert learn == [n for n, _ in model.named_parameters()]
    opt_g = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4)
    opt_c = torch.optim.Adam([params[k] for k in learn], lr=1e-3, weight_decay=5e-4)
    nl, al = torch.zeros(2), torch.ones(2)
    losses_g, losses_c = [], []
    for step in range(3):
        video = mgfn_inputs(4, 32, 10 + step)
        opt_g.zero_grad()
        out = model(video=video.to(DEV), abnormal_labels=al.to(DEV), normal_labels=nl.to(DEV))
        out.loss.backward()
        opt_g.step()
        losses_g.append(float(out.loss))
        opt_c.zero_grad()
        # the oracle's BatchNorm1d running stats are not carried over between steps (train mode
        # normalises with batch statistics, so they do not influence the forward)
        o = mgfn_oracle.mgfn_forward(video, params, abnormal_labels=al, normal_labels=nl, training=True, keep_abn=ones, keep_nor=ones)
        o.loss.backward()
        opt_c.step()
        losses_c.append(float(o.loss))
    assert rel_err(torch.tensor(losses_g), torch.tensor(losses_c)) < TOL, (losses_g, losses_c)
    assert losses_g[0] != losses_g[1]
    # Adam's first steps are sign-like (|update| ~ lr), so compare the parameter *change*
    d_g = model.fc.weight.detach().cpu() - sd["fc.weight"]
    d_c = params["fc.weight"].detach() - sd["fc.weight"]
    assert rel_err(d_g, d_c) < 5e-2
    assert rel_err(model.fc.weight.detach().cpu(), params["fc.weight"].detach()) < TOL


def test_run_py_trains_on_synthetic_corpus(tmp_path):
    from anomaly_detection_on_video_amd.dataset import write_synthetic_feature_zips
    import run

    data_dir = write_synthetic_feature_zips(str(tmp_path / "feat"), n_normal=4, n_abnormal=4, n_test=4, seed=1)
    ckpt = str(tmp_path / "ckpt")
    logp = str(tmp_path / "log.jsonl")
    torch.manual_seed(0)
    trainer = run.main([
        "data=synthetic", f"data.local_path={data_dir}", "data.batch_size=2", "trainer.cls.max_epochs=2",
        f"trainer.callbacks.model_checkpoint.dirpath={ckpt}", "trainer.callbacks.model_checkpoint.every_n_epochs=1",
        f"trainer.logger.jsonl.path={logp}",
    ])
    steps = [h for h in trainer.history if "train_loss" in h]
    vals = [h for h in trainer.history if "valid/rec_auc" in h]
    assert len(steps) == 4 and len(vals) == 2  # 4 videos per class / batch 2 = 2 steps x 2 epochs
    assert all(np.isfinite(h["train_loss"]) for h in steps)
    assert all(0.0 <= h["valid/rec_auc"] <= 1.0 and 0.0 <= h["valid/pr_auc"] <= 1.0 for h in vals)
    assert os.path.exists(os.path.join(ckpt, "last.ckpt"))
    state = torch.load(os.path.join(ckpt, "last.ckpt"), map_location="cpu", weights_only=False)
    # Lightning's checkpoint layout (the reference's LightningModule holds the net as `self.model`, runner.py:21-24)
    assert state["epoch"] == 1 and state["global_step"] == 4 and "model.fc.weight" in state["state_dict"]
    assert len(state["optimizer_states"]) == 1 and "optimizer" in state["hyper_parameters"]
    lines = [json.loads(l) for l in open(logp)]
    assert any("lr-Adam" in l for l in lines) and any("valid/pr_auc" in l for l in lines)
    # resume: one more epoch from last.ckpt continues the step count and starts from the saved weights
    trainer2 = run.main([
        "data=synthetic", f"data.local_path={data_dir}", "data.batch_size=2", "trainer.cls.max_epochs=3",
        f"trainer.callbacks.model_checkpoint.dirpath={ckpt}", "trainer.callbacks.model_checkpoint.every_n_epochs=1",
        f"trainer.logger.jsonl.path={logp}", f"ckpt_path={os.path.join(ckpt, 'last.ckpt')}",
    ])
    steps2 = [h for h in trainer2.history if "train_loss" in h]
    assert len(steps2) == 2 and steps2[0]["step"] == 5 and steps2[-1]["epoch"] == 2
    # a reference-style checkpoint (bare Lightning keys) and the round-1 layout both load
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection
    from anomaly_detection_on_video_amd.runner import VideoAnomalyDetectionRunner, load_checkpoint

    r = VideoAnomalyDetectionRunner(MGFNForVideoAnomalyDetection(MGFNConfig()), {"learning_rate": 1e-3, "weight_decay": 5e-4}, {"frames_per_clip": 16})
    state = torch.load(os.path.join(ckpt, "last.ckpt"), map_location="cpu", weights_only=False)  # rewritten by the resumed run
    assert state["epoch"] == 2 and state["global_step"] == 6
    load_checkpoint(os.path.join(ckpt, "last.ckpt"), r)
    assert torch.equal(r.model.fc.weight, state["state_dict"]["model.fc.weight"])
    old = str(tmp_path / "old.pt")
    torch.save({"model": {k[len("model."):]: v for k, v in state["state_dict"].items()}, "optimizer": state["optimizer_states"][0], "epoch": 0}, old)
    load_checkpoint(old, r)


def test_extract_driver_layout_resume_and_segment(tmp_path):
    """extract_video == per-crop oracle forwards stacked the reference's way; skip-if-exists resume;
    segment() file driver == host oracle."""
    from anomaly_detection_on_video_amd import extract
    from anomaly_detection_on_video_amd.i3d import I3Res50
    from oracle import host_oracle, i3d_oracle

    sd = synth_i3d_state_dict()
    model = I3Res50()
    model.load_state_dict(sd)
    model = model.eval().to(DEV)
    # 3 clips x 10 crops of a small frame size keep the CPU oracle fast: (n, 10, T=16, 3, 64, 64)
    clips = synth_tensor("extract.clips", (3, 10, 16, 3, 64, 64), scale=2.0)
    feats = extract.extract_video(model, clips, batch_size=2, max_crop_clips=8)
    assert feats.shape == (3, 10, 2048) and feats.dtype == np.float32
    # reference layout: outputs[batch][crop] = model(inputs[:, crop]) (extract_features.py:83-100)
    x = clips.permute(0, 1, 3, 2, 4, 5)
    per_batch = []
    for b0 in (0, 2):
        xb = x[b0 : b0 + 2]
        per_batch.append([i3d_oracle.i3d_forward(xb[:, c].contiguous(), sd).numpy() for c in range(10)])
    ref = host_oracle.stack_crop_outputs(per_batch)
    assert rel_err(feats, ref) < TOL

    # uint8 source: pixels in, same features as normalising on the host first
    pix = torch.randint(0, 256, (2, 10, 16, 3, 64, 64), generator=torch.Generator().manual_seed(5), dtype=torch.uint8)
    f_u8 = extract.extract_video(model, pix, batch_size=2, max_crop_clips=8)
    f_f32 = extract.extract_video(model, (pix.float() - 114.75) / 57.375, batch_size=2, max_crop_clips=8)
    np.testing.assert_array_equal(f_u8, f_f32)

    out = str(tmp_path / "feats")
    calls = []

    def loader():
        calls.append(1)
        return clips

    written = extract.extract([("vidA", loader), ("vidB", loader)], model, out, batch_size=4, max_crop_clips=16)
    assert set(written) == {"vidA", "vidB"} and len(calls) == 2
    again = extract.extract([("vidA", loader), ("vidC", loader)], model, out, batch_size=4, max_crop_clips=16)
    assert set(again) == {"vidC"} and len(calls) == 3  # vidA skipped: file exists
    np.testing.assert_array_equal(np.load(os.path.join(out, "vidA_i3d.npy")), np.load(os.path.join(out, "vidC_i3d.npy")))
    seg = str(tmp_path / "seg32")
    extract.segment(out, seg, 32)
    got = np.load(os.path.join(seg, "vidA_i3d.npy"))
    np.testing.assert_array_equal(got, host_oracle.segment_features(np.load(os.path.join(out, "vidA_i3d.npy")), 32))
    assert got.shape == (10, 32, 2048)


def test_stream_scores_match_oracle_end_to_end():
    """ExtractScoreStream (backbone -> ring -> add_magnitude -> MGFN eval) on 2 tiny videos vs the
    CPU oracle chain."""
    from anomaly_detection_on_video_amd.i3d import I3Res50
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection
    from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream
    from oracle import host_oracle, i3d_oracle, mgfn_oracle

    sd = synth_i3d_state_dict()
    bb = I3Res50()
    bb.load_state_dict(sd)
    bb = bb.eval().to(DEV)
    sc = MGFNForVideoAnomalyDetection(MGFNConfig())
    msd = synth_module_state_dict(sc)
    sc.load_state_dict(msd)
    sc = sc.eval().to(DEV)
    stream = ExtractScoreStream(bb, sc, clips_per_video=3, ncrops=2, local_batch=4)
    x = synth_tensor("stream.x", (12, 3, 16, 48, 48), scale=2.0)  # 2 videos x 3 clips x 2 crops
    scored = []
    for i in range(0, 12, 4):
        _g, s = stream.step(x[i : i + 4].to(DEV))
        scored += s
    assert [v for v, _ in scored] == [0, 1]
    feats = i3d_oracle.i3d_forward(x, sd).reshape(12, 2048)
    for v, s in scored:
        f = feats[v * 6 : (v + 1) * 6].reshape(3, 2, 2048).numpy()
        video = torch.from_numpy(host_oracle.add_magnitude(f)).unsqueeze(0).permute(0, 2, 1, 3)
        ref = mgfn_oracle.mgfn_forward(video, msd).scores.reshape(-1)
        assert rel_err(s.cpu(), ref) < TOL
    # the same stream through step_async (consecutive steps on alternating HIP stream lanes, only the ring
    # update ordered): identical launches, so bit-identical features and scores
    lanes = ExtractScoreStream(bb, sc, clips_per_video=3, ncrops=2, local_batch=4)
    assert lanes.lanes == 3
    handles = [lanes.step_async(x[i : i + 4].to(DEV)) for i in range(0, 12, 4)]
    lanes.drain()
    got = []
    for k, h in enumerate(handles):
        g, sc_list = h.result()
        assert torch.equal(g, stream.ring[4 * k : 4 * k + 4])
        got += sc_list
    assert [v for v, _ in got] == [0, 1]
    for (_v, a), (_w, b) in zip(got, scored):
        assert torch.equal(a, b)
    # host buffers handed to the lanes: the H2D copy runs on the step's lane (`prepare`)
    hosted = ExtractScoreStream(bb, sc, clips_per_video=3, ncrops=2, local_batch=4)
    pinned = x.pin_memory()
    hs = [hosted.step_async(pinned[i : i + 4], prepare=lambda h: h.to(DEV, non_blocking=True)) for i in range(0, 12, 4)]
    hosted.drain()
    torch.cuda.synchronize()
    assert torch.equal(hosted.ring, stream.ring)
    assert [v for h in hs for v, _ in h.result()[1]] == [0, 1]


def test_end_to_end_extract_segment_train_auc(tmp_path):
    """BASELINE config 5 in miniature: synthetic uint8 videos -> I3D extract (HIP) -> segment(32) for
    train / raw for test + make_gt rule -> MGFN training through run.py -> frame-level AUC; the AUC of
    the trained weights is then recomputed with the CPU oracle from the same features."""
    import io
    import json
    import zipfile

    import run
    from anomaly_detection_on_video_amd import extract, metrics
    from anomaly_detection_on_video_amd.dataset import build_feature_dataset
    from anomaly_detection_on_video_amd.gt import frame_ground_truth
    from anomaly_detection_on_video_amd.i3d import I3Res50
    from oracle import host_oracle, mgfn_oracle
    from _auc import auc_band

    torch.manual_seed(0)
    bb = I3Res50()
    bb.load_state_dict(synth_i3d_state_dict())
    bb = bb.eval().to(DEV)
    g = torch.Generator().manual_seed(11)

    def video(n_clips, burst=None):
        v = torch.randint(60, 140, (n_clips, 10, 16, 3, 32, 32), generator=g, dtype=torch.uint8)
        if burst is not None:  # "anomalous" clips: saturated frames -> different feature statistics
            v[burst[0] : burst[1]] = torch.randint(200, 256, v[burst[0] : burst[1]].shape, generator=g, dtype=torch.uint8)
        return v

    root = tmp_path / "corpus"
    os.makedirs(root)

    def put(z, name, arr):
        buf = io.BytesIO()
        np.save(buf, arr.astype(np.float32))
        z.writestr(name, buf.getvalue())

    with zipfile.ZipFile(root / "train.zip", "w") as z:
        for i in range(4):
            f = extract.extract_video(bb, video(5), batch_size=5, max_crop_clips=50)
            put(z, f"train/Normal_Videos{i:03d}_x264_i3d.npy", host_oracle.segment_features(f, 32))
        for i in range(4):
            f = extract.extract_video(bb, video(6, burst=(2, 4)), batch_size=6, max_crop_clips=60)
            put(z, f"train/Abuse{i:03d}_x264_i3d.npy", extract.segment_array(f, 32))
    gt, test_feats = {}, {}
    with zipfile.ZipFile(root / "test.zip", "w") as z:
        for i in range(4):
            n = 5 + i
            if i % 2:
                f = extract.extract_video(bb, video(n, burst=(1, 3)), batch_size=8, max_crop_clips=40)
                name, ev = f"Burglary{i:03d}_x264", ((16, 47), (-1, -1))
            else:
                f = extract.extract_video(bb, video(n), batch_size=8, max_crop_clips=40)
                name, ev = f"Normal_Videos_{900 + i}_x264", ((-1, -1), (-1, -1))
            put(z, f"test/{name}_i3d.npy", f)
            gt[name] = frame_ground_truth(n, *ev)
            test_feats[name] = f
    with open(root / "ground_truth.json", "w") as f:
        json.dump(gt, f)

    trainer = run.main([
        "data=synthetic", f"data.local_path={root}", "data.batch_size=2", "trainer.cls.max_epochs=3",
        f"trainer.callbacks.model_checkpoint.dirpath={tmp_path / 'ckpt'}", "trainer.callbacks.model_checkpoint.every_n_epochs=1",
        f"trainer.logger.jsonl.path={tmp_path / 'log.jsonl'}",
    ])
    vals = [h for h in trainer.history if "valid/rec_auc" in h]
    assert len(vals) == 3
    gpu_auc = vals[-1]["valid/rec_auc"]
    # oracle recomputation from the trained weights
    state = torch.load(tmp_path / "ckpt" / "last.ckpt", map_location="cpu", weights_only=False)["state_dict"]
    state = {k[len("model."):]: v for k, v in state.items()}
    ds = build_feature_dataset("test", local_path=str(root), filename="test.zip", dynamic_load=False)
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection

    scorer = MGFNForVideoAnomalyDetection(MGFNConfig())
    scorer.load_state_dict(state)
    scorer = scorer.eval().to(DEV)
    preds, labels = [], []
    for i in range(len(ds)):
        item = ds[i]
        video_t = torch.from_numpy(item["feature"]).unsqueeze(0).permute(0, 2, 1, 3)
        with torch.no_grad():
            preds.append(mgfn_oracle.mgfn_forward(video_t.float(), state).scores.reshape(-1).numpy())
            got = scorer(video=video_t.float().contiguous().to(DEV)).scores.reshape(-1).cpu().numpy()
        # the scores themselves, clip by clip: the HIP scorer on the trained weights against the oracle on the same weights
        assert np.abs(got - preds[-1]).max() < 1e-5, (i, np.abs(got - preds[-1]).max())
        labels.append(item["label"])
    # the GPU scores agree with the oracle's to ~1e-6 (forward only: the weights are the trained ones in both); a near-tie between
    # a positive and a negative clip may rank either way, so the GPU's AUC lies in the band the oracle's scores +- 1e-5 span
    lo, cpu_auc, hi = auc_band(preds, labels, 16, tol=1e-5)
    # (after two or three epochs on a few dozen clips the scores can sit within 1e-5 of each other: the band is then wide and the
    # clip-by-clip comparison above is the check)
    assert lo - 1e-9 <= gpu_auc <= hi + 1e-9, (lo, cpu_auc, hi, gpu_auc)
    assert 0.0 <= gpu_auc <= 1.0


@pytest.mark.parametrize("optimizer", ["hip", "torch-fused"])
def test_graphed_training_steps_equal_eager_steps_bit_for_bit(optimizer):
    """train_graph.GraphedTrainStep (the step the Trainer and bench.py run: 3 eager steps, then ONE HIP-graph replay per step)
    against the plain eager loop on an identical model: same batches, keep mask pinned -> after 6 steps the losses and every
    parameter, BatchNorm running statistic and Adam moment must be IDENTICAL (same kernels, same order, same arithmetic;
    the graph only removes the host from the loop).  Also: a batch of another shape falls back to the eager step."""
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection
    from anomaly_detection_on_video_amd.train_graph import GraphedTrainStep

    def make():
        m = MGFNForVideoAnomalyDetection(MGFNConfig())
        m.load_state_dict(synth_module_state_dict(m))
        m = m.to(DEV).train()
        ones = torch.ones(2, 32, device=DEV)
        m.injected_keep = (ones, ones)
        if optimizer == "hip":  # what runner.configure_optimizers builds on GPU parameters
            from anomaly_detection_on_video_amd.optim import HipAdam

            opt = HipAdam(m.parameters(), lr=1e-3, weight_decay=5e-4)
        else:
            opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-4, fused=True, capturable=True)
        return m, opt

    nl, al = torch.zeros(2, device=DEV), torch.ones(2, device=DEV)
    batches = [mgfn_inputs(4, 32, 40 + i).to(DEV) for i in range(6)]
    m_e, opt_e = make()
    eager = GraphedTrainStep(m_e, opt_e, eager_steps=1 << 30)
    losses_e = [float(eager(b, al, nl)) for b in batches]
    assert eager.graph is None and eager.replays == 0
    m_g, opt_g = make()
    graphed = GraphedTrainStep(m_g, opt_g, eager_steps=3)
    losses_g = [float(graphed(b, al, nl)) for b in batches]
    assert graphed.graph is not None and graphed.replays == 3
    assert losses_g == losses_e, (losses_g, losses_e)
    assert losses_e[0] != losses_e[-1]
    for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
        assert torch.equal(a, b), k
    for pe, pg in zip(m_e.parameters(), m_g.parameters()):
        se, sg = opt_e.state[pe], opt_g.state[pg]
        assert torch.equal(se["exp_avg"], sg["exp_avg"]) and torch.equal(se["exp_avg_sq"], sg["exp_avg_sq"]) and torch.equal(se["step"], sg["step"])
    # another batch shape: eager fallback, the captured graph stays valid for the original shape afterwards
    odd = mgfn_inputs(2, 32, 77).to(DEV)
    m_g.injected_keep = (torch.ones(1, 32, device=DEV), torch.ones(1, 32, device=DEV))
    loss_odd = graphed(odd, al[:1], nl[:1])
    assert torch.isfinite(loss_odd) and graphed.replays == 3
    ones = torch.ones(2, 32, device=DEV)
    m_g.injected_keep = (ones, ones)
    # (the captured graph holds the ORIGINAL keep tensors: pinned masks are part of the capture)
    assert torch.isfinite(graphed(batches[0], al, nl)) and graphed.replays == 4


def test_graphed_step_recaptures_after_lr_change_parameter_move_and_state_reload():
    """What a captured graph bakes in besides shapes (train_graph.GraphedTrainStep._baked): optimizer scalars (lr is a kernel
    argument of the Adam launch), parameter addresses (`p.data = p.data.clone()` moves the storage: new pack plans, whose item
    tables are built with a host -> device copy -- illegal inside a capture, so ONE eager step runs first) and the optimizer's
    moment / step tensors (`load_state_dict` replaces them).  After each such change the graphed loop must (i) notice, (ii) run one
    live eager step, (iii) capture again, and stay IDENTICAL to the plain eager loop going through the same changes."""
    import copy

    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection
    from anomaly_detection_on_video_amd.optim import HipAdam
    from anomaly_detection_on_video_amd.train_graph import GraphedTrainStep

    def make():
        m = MGFNForVideoAnomalyDetection(MGFNConfig())
        m.load_state_dict(synth_module_state_dict(m))
        m = m.to(DEV).train()
        ones = torch.ones(2, 32, device=DEV)
        m.injected_keep = (ones, ones)
        return m, HipAdam(m.parameters(), lr=1e-3, weight_decay=5e-4)

    nl, al = torch.zeros(2, device=DEV), torch.ones(2, device=DEV)
    batches = [mgfn_inputs(4, 32, 60 + i).to(DEV) for i in range(14)]

    def run(eager_steps):
        m, opt = make()
        step = GraphedTrainStep(m, opt, eager_steps=eager_steps)
        losses, marks = [], []
        for i, b in enumerate(batches):
            if i == 4:
                for g in opt.param_groups:
                    g["lr"] = 5e-4
            if i == 7:
                w = m.backbone.layers[2][0].ffn.in_conv.weight
                w.data = w.data.clone()  # same values, new storage
            if i == 10:
                opt.load_state_dict(copy.deepcopy(opt.state_dict()))  # every moment / step tensor replaced by a copy
            losses.append(float(step(b, al, nl)))
            marks.append((step.captures, step.replays, step.graph is not None, len(step.held_plans)))
        return m, opt, step, losses, marks

    m_e, opt_e, _s, losses_e, _ = run(1 << 30)
    m_g, opt_g, step, losses_g, marks = run(2)
    assert losses_g == losses_e, (losses_g, losses_e)
    for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
        assert torch.equal(a, b), k
    for pe, pg in zip(m_e.parameters(), m_g.parameters()):
        se, sg = opt_e.state[pe], opt_g.state[pg]
        assert torch.equal(se["exp_avg"], sg["exp_avg"]) and torch.equal(se["exp_avg_sq"], sg["exp_avg_sq"]) and float(se["step"]) == float(sg["step"]) == 14.0
    # steps 0-1 eager, 2 captures (#1) and replays, 3 replays; 4: stale (lr) -> eager, 5 captures (#2), 6 replays; 7: stale (moved
    # parameter) -> eager, 8 captures (#3), 9 replays; 10: stale (optimizer state) -> eager, 11 captures (#4), 12-13 replay
    assert [c for c, _r, _g, _h in marks] == [0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4], marks
    assert [g for _c, _r, g, _h in marks] == [False, False, True, True, False, True, True, False, True, True, False, True, True, True], marks
    assert step.replays == 9 and all(h >= 1 for _c, _r, g, h in marks if g) and all(h == 0 for _c, _r, g, h in marks if not g)
    # a fifth change: MAX_CAPTURES reached -> the step stays eager from here on (and still tracks the eager loop: same code path)
    for g in opt_g.param_groups:
        g["lr"] = 2.5e-4
    step(batches[0], al, nl)
    step(batches[1], al, nl)
    assert step.graph is None and step.captures == GraphedTrainStep.MAX_CAPTURES


def test_trainer_feeds_the_captured_steps_input_buffers_in_place():
    """runner.Trainer._feed_graph_inputs + GraphedTrainStep.inputs(): once the step is a graph, a (normal, abnormal) loader batch on the
    HOST is copied straight into the graph's input buffers -- normal rows first, as src/runner.py:29-39's torch.cat orders them -- and the
    step is replayed on them: same losses and parameters, bit for bit, as the eager loop fed through training_batch()."""
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection
    from anomaly_detection_on_video_amd.optim import HipAdam
    from anomaly_detection_on_video_amd.runner import Trainer, VideoAnomalyDetectionRunner
    from anomaly_detection_on_video_amd.train_graph import GraphedTrainStep

    def make():
        m = MGFNForVideoAnomalyDetection(MGFNConfig())
        m.load_state_dict(synth_module_state_dict(m))
        m = m.to(DEV).train()
        ones = torch.ones(2, 32, device=DEV)
        m.injected_keep = (ones, ones)
        return m, HipAdam(m.parameters(), lr=1e-3, weight_decay=5e-4)

    def host_batch(i):  # what the two train loaders yield: dicts of CPU tensors
        v = mgfn_inputs(4, 32, 90 + i)
        return ({"feature": v[:2], "anomaly": torch.zeros(2)}, {"feature": v[2:], "anomaly": torch.ones(2)})

    m_e, opt_e = make()
    eager = GraphedTrainStep(m_e, opt_e, eager_steps=1 << 30)
    m_g, opt_g = make()
    graphed = GraphedTrainStep(m_g, opt_g, eager_steps=2)
    losses_e, losses_g, fed = [], [], []
    for i in range(6):
        batch = host_batch(i)
        dev_batch = tuple({k: t.to(DEV) for k, t in d.items()} for d in batch)
        losses_e.append(float(eager(*VideoAnomalyDetectionRunner.training_batch(dev_batch))))
        ok = Trainer._feed_graph_inputs(graphed, batch)
        fed.append(ok)
        if ok:
            losses_g.append(float(graphed(*graphed.inputs())))
        else:
            losses_g.append(float(graphed(*VideoAnomalyDetectionRunner.training_batch(dev_batch))))
    assert fed == [False, False, False, True, True, True]  # (no graph before the capture in step 2's call)
    assert graphed.replays == 4 and losses_g == losses_e, (losses_g, losses_e)
    for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
        assert torch.equal(a, b), k
    # another batch shape is not written anywhere: the caller falls back to the device batch
    odd = ({"feature": mgfn_inputs(1, 32, 5), "anomaly": torch.zeros(1)}, {"feature": mgfn_inputs(1, 32, 6), "anomaly": torch.ones(1)})
    before = graphed.inputs()[0].clone()
    assert Trainer._feed_graph_inputs(graphed, odd) is False and torch.equal(graphed.inputs()[0], before)


def test_hip_adam_creates_state_only_for_parameters_with_gradients():
    """torch.optim.Adam initialises `state[p]` the first time p has a gradient; a frozen parameter never gets an entry, so the
    state_dict of a partly frozen model has the same keys under both optimizers (a checkpoint interchange requirement)."""
    from anomaly_detection_on_video_amd.optim import HipAdam

    ps_ref = [torch.randn(5, device=DEV, requires_grad=True) for _ in range(4)]
    ps_hip = [p.detach().clone().requires_grad_() for p in ps_ref]
    ref, hip = torch.optim.Adam(ps_ref, lr=1e-3), HipAdam(ps_hip, lr=1e-3)
    for step in range(3):
        for i, (a, b) in enumerate(zip(ps_ref, ps_hip)):
            if i == 2 or (i == 1 and step == 0):  # parameter 2 never has a gradient, parameter 1 gets its first one in step 1
                a.grad = b.grad = None
                continue
            g = torch.full((5,), 0.1 * (i + 1) * (step + 1), device=DEV)
            a.grad, b.grad = g.clone(), g.clone()
        ref.step()
        hip.step()
    sr, sh = ref.state_dict()["state"], hip.state_dict()["state"]
    assert sorted(sr) == sorted(sh) == [0, 1, 3]
    for k in sr:
        assert float(sr[k]["step"]) == float(sh[k]["step"])
        assert rel_err(sh[k]["exp_avg"].cpu(), sr[k]["exp_avg"].cpu()) < 2e-6
    for a, b in zip(ps_ref, ps_hip):
        assert rel_err(b.detach().cpu(), a.detach().cpu()) < 2e-6
    with pytest.raises(ValueError):
        HipAdam(ps_hip, lr=torch.tensor(1e-3))


def test_hip_adam_tracks_torch_adam_and_shares_its_state_layout():
    """optim.HipAdam (one launch per 80 tensors) against torch.optim.Adam, the reference's optimizer (/root/reference/src/runner.py:53-59):
    same update after several steps on tensors of ragged sizes (vector body + scalar tails, > 80 tensors = two launches), a
    parameter without a gradient keeps its step count, and state_dict() round-trips between the two classes."""
    from anomaly_detection_on_video_amd.optim import HipAdam

    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    shapes = [(64, 2048, 3), (64,), (1024, 1024, 1), (7,), (1, 33, 5), (4099,), (3, 3)] + [(5 + i,) for i in range(90)]
    ref_p = [torch.randn(s, generator=g).to(dev).requires_grad_() for s in shapes]
    hip_p = [p.detach().clone().requires_grad_() for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=1e-3, weight_decay=5e-4)
    hip = HipAdam(hip_p, lr=1e-3, weight_decay=5e-4)
    for step in range(4):
        for i, (a, b) in enumerate(zip(ref_p, hip_p)):
            if step == 2 and i == 3:  # no gradient for this one in this step
                a.grad = b.grad = None
                continue
            gr = torch.randn(a.shape, generator=g).to(dev) * (1.0 + i % 3)
            a.grad, b.grad = gr.clone(), gr.clone()
        ref.step()
        hip.step()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(ref_p, hip_p)):
        assert rel_err(b.detach().cpu(), a.detach().cpu()) < 2e-6, i
        sa, sb = ref.state[a], hip.state[b]
        assert float(sb["step"]) == float(sa["step"]) == (3.0 if i == 3 else 4.0)
        assert rel_err(sb["exp_avg"].cpu(), sa["exp_avg"].cpu()) < 2e-6 and rel_err(sb["exp_avg_sq"].cpu(), sa["exp_avg_sq"].cpu()) < 2e-6
    # state interchange: torch's state into HipAdam (what a resumed reference checkpoint does) and back
    hip2_p = [p.detach().clone().requires_grad_() for p in ref_p]
    hip2 = HipAdam(hip2_p, lr=1e-3, weight_decay=5e-4)
    import copy

    # (deep copies, as a checkpoint file gives: Optimizer.load_state_dict keeps the very `step` tensors of the dict it is handed)
    hip2.load_state_dict(copy.deepcopy(ref.state_dict()))
    ref2_p = [p.detach().clone().requires_grad_() for p in ref_p]
    ref2 = torch.optim.Adam(ref2_p, lr=1e-3, weight_decay=5e-4)
    ref2.load_state_dict(copy.deepcopy(hip.state_dict()))
    for a, b, c in zip(ref_p, hip2_p, ref2_p):
        gr = torch.randn(a.shape, generator=g).to(dev)
        a.grad, b.grad, c.grad = gr.clone(), gr.clone(), gr.clone()
    ref.step()
    hip2.step()
    ref2.step()
    torch.cuda.synchronize()
    for i, (a, b, c) in enumerate(zip(ref_p, hip2_p, ref2_p)):
        assert rel_err(b.detach().cpu(), a.detach().cpu()) < 2e-6, i
        assert rel_err(c.detach().cpu(), a.detach().cpu()) < 3e-6, i
