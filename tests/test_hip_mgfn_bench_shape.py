"""BASELINE config 4 at the size bench.py times: one MGFN training step on (32,10,32,2049) -- forward, the four loss
terms, total loss and the gradient of EVERY parameter against the CPU oracle (oracle/mgfn_oracle.py, pinned by the
reference-made goldens), and a proof that the step ran on the benchmarked kernels: stage-2 layers on the 128 x 64 tile
without split-K, weight gradients by advhip_gemm_nt_f32 at K = 10 240 (the bs = 4 tests resolve to 64 x 64 tiles + in-kernel
split-K, a different code path).

Follows /root/reference/src/models/mgfn/modeling_mgfn.py:376-427 and src/runner.py:29-39 (normal half first)."""
import pytest
import torch

from conftest import assert_close_elementwise, rel_err
from anomaly_detection_on_video_amd.weights import synth_module_state_dict, synth_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3
BS, NCROPS, T = 32, 10, 32


def bench_shape_video():
    """bench.py's recipe (uniform features * 3 + their L2 magnitude as channel 2049), seeded through synth_tensor so the
    CPU oracle sees the very same values."""
    feats = (synth_tensor("mgfn.bench.x", (BS, NCROPS, T, 2048), scale=0.5, offset=0.5) * 3.0).contiguous()
    return torch.cat([feats, feats.norm(dim=3, keepdim=True)], dim=3)


@pytest.fixture(scope="module")
def step_records():
    """One HIP training step + one oracle step on the same weights / input / keep mask; the launch log of the HIP step."""
    from anomaly_detection_on_video_amd import mgfn_ops, ops
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection
    from oracle import mgfn_oracle

    model = MGFNForVideoAnomalyDetection(MGFNConfig())
    sd = synth_module_state_dict(model)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    video = bench_shape_video()
    ones = torch.ones(BS // 2, T)
    model.injected_keep = (ones.to(DEV), ones.to(DEV))
    nl, al = torch.zeros(BS // 2), torch.ones(BS // 2)

    descs, nts = [], []
    orig_desc, orig_nt = mgfn_ops._desc, ops.gemm_nt

    def rec_desc(cin, cout, k, b, t, act):
        d = orig_desc(cin, cout, k, b, t, act)
        descs.append((cin, cout, k, b * t, int(d.algo), int(d.splits)))
        return d

    def rec_nt(a, b, *args, **kw):
        nts.append((a.shape[0], b.shape[0], a.shape[1]))
        return orig_nt(a, b, *args, **kw)

    mgfn_ops._desc, ops.gemm_nt = rec_desc, rec_nt
    try:
        out = model(video=video.to(DEV), abnormal_labels=al.to(DEV), normal_labels=nl.to(DEV))
        out.loss.backward()
        torch.cuda.synchronize()
    finally:
        mgfn_ops._desc, ops.gemm_nt = orig_desc, orig_nt

    params = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k and "num_batches" not in k) for k, v in sd.items()}
    ref = mgfn_oracle.mgfn_forward(video, params, abnormal_labels=al, normal_labels=nl, training=True, keep_abn=ones, keep_nor=ones)
    ref.loss.backward()
    return model, out, ref, params, descs, nts


def test_forward_scores_and_loss_terms_vs_oracle(step_records):
    model, out, ref, _params, _d, _n = step_records
    assert out.scores.shape == (BS, T, 1) and out.a_feat_magnitude.shape == (BS // 2 * NCROPS, 3, 1024)
    assert rel_err(out.scores.detach().cpu(), ref.scores.detach()) < TOL
    assert_close_elementwise(out.scores.detach().cpu(), ref.scores.detach())
    assert rel_err(out.abnormal_scores.detach().cpu(), ref.abnormal_scores.detach()) < TOL
    assert rel_err(out.normal_scores.detach().cpu(), ref.normal_scores.detach()) < TOL
    assert rel_err(out.a_feat_magnitude.detach().cpu(), ref.a_feat_magnitude.detach()) < TOL
    assert rel_err(out.n_feat_magnitude.detach().cpu(), ref.n_feat_magnitude.detach()) < TOL
    assert rel_err(out.loss.detach().cpu(), ref.loss.detach()) < TOL
    from anomaly_detection_on_video_amd.mil_ops import LOSS_TERMS

    t = dict(zip(LOSS_TERMS, model.last_loss_terms.cpu()))
    # the four loss terms of the step (loss/base.py:7-48, loss/mgfn.py:7-47) and the pieces of the MGFN term
    for mine, theirs in (("smooth", "smooth"), ("sparse", "sparse"), ("mgfn", "mgfn"), ("bce", "cls"), ("con", "con"), ("con_a", "con_a"), ("con_n", "con_n")):
        assert rel_err(t[mine], ref.terms[theirs].detach()) < TOL, (mine, float(t[mine]), float(ref.terms[theirs]))
    assert rel_err(t["total"], ref.loss.detach()) < TOL


def test_every_parameter_gradient_vs_oracle(step_records):
    model, _out, _ref, params, _d, _n = step_records
    checked = 0
    for name, p in model.named_parameters():
        want = params[name].grad
        assert p.grad is not None and want is not None, name
        got = p.grad.detach().cpu()
        assert rel_err(got, want) < TOL, (name, rel_err(got, want))
        assert_close_elementwise(got, want)
        checked += 1
    assert checked == len([k for k, v in params.items() if v.requires_grad]) == 130


def test_the_step_ran_on_the_benchmarked_kernels(step_records):
    from anomaly_detection_on_video_amd import mgfn_ops

    _m, _o, _r, _p, descs, nts = step_records
    n = BS * NCROPS * T
    assert n == 10240
    stage2 = [d for d in descs if min(d[0], d[1]) >= 1024 and d[3] > 1]  # (npos == 1: the descriptor of a weight re-pack)
    assert len(stage2) == 2 * (5 + 5), len(stage2)  # 2 blocks x (scc, to_v, to_out, in_conv, out_conv) x (forward, dX)
    for cin, cout, k, npos, algo, splits in stage2:  # (128 x 128 tiles for the 4096-wide outputs, 128 x 64 for the 1024-wide)
        assert npos == n and algo == (mgfn_ops.ALGO_WIDE if cout == 4096 else mgfn_ops.ALGO) and splits == 1, (cin, cout, k, npos, algo, splits)
    # weight gradients dW = dY X^T: contraction over all 10 240 positions, stage-2 shapes present
    # (and ONE product over the 2 048 input channels: the token conv's tap GEMM on the input rows as stored, modeling_mgfn._tokens_by_taps)
    assert (192, n, 2048) in nts and all(kk == n for a_, b_, kk in nts if (a_, b_, kk) != (192, n, 2048)), nts[:4]
    assert (4096, 1024, n) in nts and (1024, 4096, n) in nts and (1024, 3072, n) in nts


def test_deferred_param_grads_equal_the_per_layer_launches(step_records):
    """mgfn_ops.deferred_param_grads (what GraphedTrainStep turns on): the narrow layers' weight / bias gradients from ONE
    grouped launch at the end of the backward pass, written to .grad by the engine callback -- bit for bit the gradients of the
    per-layer launches, for all 130 parameters, at the benchmarked shape; a second pass accumulates (+=) as autograd does."""
    from anomaly_detection_on_video_amd import mgfn_ops, ops

    model, _o, _r, _p, _d, _n = step_records
    want = {k: p.grad.clone() for k, p in model.named_parameters()}
    video = bench_shape_video().to(DEV)
    nl, al = torch.zeros(BS // 2, device=DEV), torch.ones(BS // 2, device=DEV)
    calls = []
    orig = ops.gemm_nt_group
    ops.gemm_nt_group = lambda items: (calls.append(len(items)), orig(items))[1]
    try:
        for passes in (1, 2):
            for p in model.parameters():
                p.grad = None
            for _ in range(passes):
                with mgfn_ops.deferred_param_grads():
                    model(video=video, abnormal_labels=al, normal_labels=nl).loss.backward()
            torch.cuda.synchronize()
            for k, p in model.named_parameters():
                assert p.grad is not None, k
                if passes == 1:
                    assert torch.equal(p.grad, want[k]), (k, float((p.grad - want[k]).abs().max()))
                else:
                    assert rel_err(p.grad.cpu(), (2 * want[k]).cpu()) < 1e-6, k
    finally:
        ops.gemm_nt_group = orig
        for k, p in model.named_parameters():
            p.grad = want[k]
    assert calls and all(c >= 25 for c in calls), calls  # one grouped launch per backward pass, ~31 layers in it
    assert not mgfn_ops._DEFER["items"] and not mgfn_ops._DEFER["queued"] and not mgfn_ops._DEFER["on"]
