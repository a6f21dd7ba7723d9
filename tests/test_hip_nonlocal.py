"""NonLocalBlock / I3Res50(use_nl=True) on the HIP path vs the reference's own outputs (tests/golden/nonlocal.npz, made
by the reference's NonLocalBlock and I3Res50(use_nl=True), src/i3d.py:124-195) and the CPU oracle; plus the batched GEMM
and row-softmax entry points they are built from.  Run with -m gpu."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_close_elementwise, rel_err
from anomaly_detection_on_video_amd.weights import NONLOCAL_CASES, synth_i3d_state_dict, synth_input, synth_nonlocal_case, synth_tensor

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


GEMM_CASES = [
    # batch, M, N, K, A transposed view?, B transposed view?
    (2, 96, 48, 256, True, False),     # theta^T . phi      (A row-contiguous in m, B row-contiguous in n)
    (2, 256, 96, 48, False, True),     # g . p^T            (both k-contiguous)
    (3, 70, 130, 37, False, False),    # ragged everything, K not a multiple of 4
    (1, 300, 65, 128, True, True),
    (1, 1, 7, 5, False, False),
    (4, 128, 64, 64, False, False),
    (1, 1024, 4096, 96, False, False),  # big tile path (128 x 64)
    (1, 2048, 200, 1024, False, True),
]


@pytest.mark.parametrize("case", GEMM_CASES, ids=[str(c) for c in GEMM_CASES])
def test_bgemm_vs_torch_fp64(case):
    from anomaly_detection_on_video_amd import ops

    batch, M, N, K, ta, tb = case
    a = synth_tensor(f"gemm.a{case}", (batch, K, M) if ta else (batch, M, K), scale=1.0)
    b = synth_tensor(f"gemm.b{case}", (batch, N, K) if tb else (batch, K, N), scale=1.0)
    av, bv = (a.transpose(1, 2) if ta else a), (b.transpose(1, 2) if tb else b)
    ref = torch.bmm(av.double(), bv.double())
    ad, bd = a.to(_dev()), b.to(_dev())
    adv, bdv = (ad.transpose(1, 2) if ta else ad), (bd.transpose(1, 2) if tb else bd)
    out = ops.bgemm(adv, bdv)
    assert rel_err(out.cpu(), ref) < 2e-6
    # epilogue: alpha, biases, GELU / ReLU, residual, output written through a transposed view
    bm = synth_tensor(f"gemm.bm{case}", (M,))
    bn = synth_tensor(f"gemm.bn{case}", (N,))
    res = synth_tensor(f"gemm.r{case}", (batch, M, N))
    ref2 = torch.nn.functional.gelu(0.37 * ref + bm.double()[None, :, None] + bn.double()[None, None, :]) + 0.5 * res.double()
    out2 = ops.bgemm(adv, bdv, alpha=0.37, bias_m=bm.to(_dev()), bias_n=bn.to(_dev()), act=2, residual=res.to(_dev()), beta=0.5)
    assert rel_err(out2.cpu(), ref2) < 5e-6
    out_t = torch.empty((batch, N, M), device=_dev())
    ops.bgemm(adv, bdv, act=1, out=out_t.transpose(1, 2))
    assert rel_err(out_t.transpose(1, 2).cpu(), torch.relu(ref)) < 2e-6
    # broadcast batch (stride 0) on B
    if batch > 1:
        out3 = ops.bgemm(adv, bdv[:1])
        assert rel_err(out3.cpu(), torch.matmul(av.double(), bv[:1].double())) < 2e-6


def test_bgemm_layernorm_fold_and_softmax_rows():
    """W.LN(x) with MGFNLayerNorm semantics ((x - mean) / (sqrt(var_biased) + eps) * g + b over channels,
    modeling_mgfn.py:36-46) as ONE GEMM on the raw x: W.diag(g) as the A operand, u = rowsum(W.diag(g)), column
    statistics mu / rs in the epilogue, W.b as the row bias."""
    from anomaly_detection_on_video_amd import ops

    Cout, Cin, Ncol = 192, 128, 330
    w = synth_tensor("lnf.w", (Cout, Cin), scale=0.2)
    x = synth_tensor("lnf.x", (Cin, Ncol), scale=2.0, offset=1.5)
    g = synth_tensor("lnf.g", (Cin,), scale=0.25, offset=1.0)
    bb = synth_tensor("lnf.b", (Cin,), scale=0.1)
    xd = x.double()
    mu = xd.mean(0)
    std = xd.var(0, unbiased=False).sqrt()
    ln = (xd - mu) / (std + 1e-5) * g.double()[:, None] + bb.double()[:, None]
    ref = w.double() @ ln
    dev = _dev()
    wg = (w * g[None, :]).to(dev)
    out = ops.bgemm(wg, x.to(dev), bias_m=(w.double() @ bb.double()).float().to(dev),
                    ln=(wg.sum(1), mu.float().to(dev), (1.0 / (std + 1e-5)).float().to(dev)))
    assert rel_err(out[0].cpu(), ref) < 1e-5
    s = synth_tensor("sm.x", (7, 33, 392), scale=6.0)
    got = ops.softmax_rows(s.to(dev), scale=0.25)
    assert rel_err(got.cpu(), torch.softmax(s.double() * 0.25, dim=-1)) < 1e-6
    assert torch.allclose(got.sum(-1).cpu(), torch.ones(7, 33), atol=1e-5)


@pytest.mark.parametrize("name", NONLOCAL_CASES)
def test_nonlocal_block_vs_reference_golden(name):
    from anomaly_detection_on_video_amd.i3d import NonLocalBlock
    from oracle import i3d_oracle

    dim, inner, sd, x = synth_nonlocal_case(name)
    blk = NonLocalBlock(dim, dim, inner)
    blk.load_state_dict(sd, strict=True)
    blk = blk.eval().to(_dev())
    y = blk(x.to(_dev())).cpu()
    g = np.load(os.path.join(GOLDEN, "nonlocal.npz"))
    assert rel_err(y, g[f"{name}_y"]) < 2e-5
    assert_close_elementwise(y, g[f"{name}_y"], 1e-3, 1e-3)
    assert rel_err(y, i3d_oracle.nonlocal_block(x, {f"nl.{k}": v for k, v in sd.items()}, "nl")) < 2e-5
    assert torch.equal(y, blk(x.to(_dev())).cpu())


def test_i3res50_with_nonlocal_blocks_vs_reference_golden():
    from anomaly_detection_on_video_amd.i3d import I3Res50, NonLocalBlock

    m = I3Res50(use_nl=True)
    sd = synth_i3d_state_dict(use_nl=True)
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd, strict=True)
    m = m.eval().to(_dev())
    assert sum(1 for mod in m.modules() if isinstance(mod, NonLocalBlock)) == 5
    g = np.load(os.path.join(GOLDEN, "nonlocal.npz"))
    y = m(synth_input((1, 3, 8, 112, 96), 7).to(_dev())).reshape(1, 2048).cpu()
    assert rel_err(y, g["feat_nl_small"]) < 1e-4
    y2 = m(synth_input((2, 3, 16, 64, 64), 3).to(_dev())).reshape(2, 2048).cpu()
    assert rel_err(y2, g["feat_nl_64"]) < 1e-4
    assert_close_elementwise(y2, g["feat_nl_64"], 1e-3, 1e-3)
    # a batch that splits over two streams (first-use tables of the non-local convs are built ahead of the fork)
    x16 = synth_input((2, 3, 16, 64, 64), 3).repeat(8, 1, 1, 1, 1).to(_dev())
    y16 = m(x16).reshape(16, 2048).cpu()
    assert rel_err(y16, np.tile(g["feat_nl_64"], (8, 1))) < 1e-4


# ------------------------------------------------------------------------------ clip pre-processing on the device
@pytest.mark.parametrize("shape,fpc,crop", [((21, 40, 53, 3), 16, 32), ((16, 256, 341, 3), 16, 224), ((5, 225, 224, 3), 4, 224), ((3, 33, 34, 3), 16, 32)])
def test_tencrop_normalize_on_device_is_bit_exact(shape, fpc, crop):
    """advhip_tencrop_normalize_u8 vs the numpy restatement of TenCropVideoFrameDataset + the driver's permute
    (oracle/host_oracle.py:ten_crop_clips): uint8 in, pure selection + one fp32 subtract and divide -> bit exact.
    Covers odd (H - crop) / (W - crop) (Python round-half-to-even centre offsets), a short last clip (LoopPad) and a
    video shorter than one clip."""
    from anomaly_detection_on_video_amd import mil_ops
    from oracle import host_oracle

    rng = np.random.default_rng(sum(shape))
    frames = rng.integers(0, 256, shape, dtype=np.uint8)
    ref = host_oracle.ten_crop_clips(frames, fpc, crop)
    got = mil_ops.tencrop_normalize_u8(torch.from_numpy(frames).to(_dev()), fpc, crop).cpu().numpy()
    n_clips = -(-shape[0] // fpc)
    assert got.shape == (n_clips * 10, 3, fpc, crop, crop)
    assert np.array_equal(got.reshape(ref.shape), ref)
    # crop 0 is the top-left window, crop 5 its mirror image's top-left = the original's top-right, flipped
    assert np.array_equal(got.reshape(ref.shape)[:, 5], got.reshape(ref.shape)[:, 1][..., ::-1])


def test_extract_from_resized_uint8_frames_matches_tencrop_tensor_path():
    """extract.extract_video_frames: resized uint8 frames -> (n_clips, 10, 2048), the same features as feeding the
    oracle's ten-crop fp32 tensor through extract_video (the reference's data path)."""
    from anomaly_detection_on_video_amd import extract
    from anomaly_detection_on_video_amd.i3d import I3Res50
    from oracle import host_oracle

    m = I3Res50()
    m.load_state_dict(synth_i3d_state_dict())
    m = m.eval().to(_dev())
    frames = np.random.default_rng(5).integers(0, 256, (24, 72, 90, 3), dtype=np.uint8)
    a = extract.extract_video_frames(m, torch.from_numpy(frames), crop=64)
    clips = host_oracle.ten_crop_clips(frames, 16, 64)  # (n_clips, 10, C, T, h, w)
    b = extract.extract_video(m, torch.from_numpy(clips).permute(0, 1, 3, 2, 4, 5).contiguous())
    assert a.shape == b.shape == (2, 10, 2048)
    assert rel_err(a, b) < 1e-5


# ------------------------------------------------------------------------------ i3d_8x8_r50 (parity unpinned)
def test_i3d_8x8_r50_topology_vs_torch_restatement(monkeypatch):
    """`build_i3d_feature_extractor("i3d_8x8_r50")`: pytorchvideo's create_resnet topology on the HIP kernels, against a
    plain-torch restatement of the same topology (oracle.ptv_forward).  PARITY UNPINNED against pytorchvideo itself
    (third-party, absent from the reference tree and from this image): this pins the kernels on that topology only."""
    from anomaly_detection_on_video_amd.i3d import build_i3d_feature_extractor
    from anomaly_detection_on_video_amd.weights import synth_module_state_dict
    from oracle import i3d_oracle

    monkeypatch.setenv("ADV_I3D_SYNTHETIC", "1")
    m = build_i3d_feature_extractor("i3d_8x8_r50", check_model_size=False, strict=True)
    keys = list(m.state_dict().keys())
    assert keys[0] == "blocks.0.conv.weight" and "blocks.1.res_blocks.0.branch1_conv.weight" in keys
    assert "blocks.3.res_blocks.3.branch2.conv_c.weight" in keys and "blocks.5.res_blocks.2.branch2.norm_c.running_var" in keys
    assert sum(1 for k in keys if k.endswith("conv.weight") or "conv_" in k and k.endswith(".weight") or k.endswith("branch1_conv.weight")) == 53
    sd = synth_module_state_dict(m, gain=2.0)
    m = m.eval().to(_dev())
    for shape, seed in (((1, 3, 8, 224, 224), 4), ((2, 3, 16, 232, 240), 9)):
        x = synth_input(shape, seed)
        y = m(x.to(_dev())).cpu()
        ref = i3d_oracle.ptv_forward(x, sd)
        assert y.shape == ref.shape == (shape[0], 2048, 1, 1, 1)
        assert rel_err(y, ref) < 1e-4, shape
    with pytest.raises(ValueError):
        m(synth_input((1, 3, 4, 224, 224), 0).to(_dev()))  # T = 2 after the stage-1 pool: smaller than the head's (4,7,7) window


@pytest.mark.parametrize("M,N,K,splits", [(64, 64, 16, 1), (100, 70, 160, 1), (256, 192, 1024, 4), (1024, 128, 10240, 0), (130, 1024, 2048, 3), (1, 5, 32, 2)])
def test_gemm_nt_vs_torch_fp64(M, N, K, splits):
    """advhip_gemm_nt_f32 (both operands k-contiguous, LDS-DMA row copies): dW = dY . X^T shapes, ragged M / N, K slices,
    operands that are row slices of wider matrices (row pitch > K)."""
    from anomaly_detection_on_video_amd import ops

    a = synth_tensor(f"nt.a{M}{K}", (M, K + 32), scale=1.0).to(_dev())[:, 16 : 16 + K]
    b = synth_tensor(f"nt.b{N}{K}", (N, K), scale=1.0).to(_dev())
    out = ops.gemm_nt(a, b, splits)
    ref = a.double().cpu() @ b.double().cpu().t()
    assert out.shape == (M, N)
    assert rel_err(out.cpu(), ref) < 3e-6
    assert torch.equal(out, ops.gemm_nt(a, b, splits))


@pytest.mark.parametrize("tile", [1, 2, 3])
@pytest.mark.parametrize("M,N,K,splits", [(64, 64, 16, 1), (100, 70, 160, 1), (256, 192, 1024, 4), (1024, 384, 10240, 2), (130, 1024, 2048, 3), (1, 5, 32, 2)])
def test_gemm_nt_row_sums_and_tiles_vs_torch_fp64(M, N, K, splits, tile):
    """advhip_gemm_nt_rowsum_f32: every output tile size, with the row sums of A (the bias gradient beside dW = dY X^T) taken
    from the fragments of the n-tile-0 workgroups: both against fp64, ragged shapes, K slices; run-to-run bit-identical."""
    from anomaly_detection_on_video_amd import ops

    a = synth_tensor(f"ntr.a{M}{K}", (M, K + 32), scale=1.0).to(_dev())[:, 16 : 16 + K]
    b = synth_tensor(f"ntr.b{N}{K}", (N, K), scale=1.0).to(_dev())
    out, rs = ops.gemm_nt(a, b, splits, rowsum=True, tile=tile)
    assert out.shape == (M, N) and rs.shape == (M,)
    assert rel_err(out.cpu(), a.double().cpu() @ b.double().cpu().t()) < 3e-6
    want = a.double().cpu().sum(1)
    assert float((rs.cpu().double() - want).abs().max()) < 3e-6 * float(a.double().abs().sum(1).max())
    out2, rs2 = ops.gemm_nt(a, b, splits, rowsum=True, tile=tile)
    assert torch.equal(out, out2) and torch.equal(rs, rs2)
    assert torch.equal(out, ops.gemm_nt(a, b, splits, tile=tile))  # the product does not depend on the row sums being taken


@pytest.mark.parametrize("tile", [1, 2, 3])
def test_gemm_nt_in_kernel_slice_reduction_equals_the_ordered_slab_sum_under_load(tile):
    """advhip_gemm_nt_reduced_f32 (partial tiles published write-through, last arriver sums in slice order, counters left
    zero) against the slab form of the same kernel summed in slice order on the host side of the ABI: bit for bit, C and the
    row sums, launch after launch on two streams at once (every launch re-uses the counters the previous one left)."""
    import ctypes as C

    from anomaly_detection_on_video_amd import _lib, ops

    lib = _lib.load()
    dev = _dev()
    M, N, K, splits = 320, 448, 4096, 5
    a = synth_tensor("ntk.a", (M, K), scale=1.0).to(dev)
    b = synth_tensor("ntk.b", (N, K), scale=1.0).to(dev)
    slabs = torch.empty((splits, M, N), device=dev)
    rs_slabs = torch.empty((splits, M), device=dev)
    _lib.check(lib.advhip_gemm_nt_rowsum_f32(a.data_ptr(), b.data_ptr(), slabs.data_ptr(), rs_slabs.data_ptr(), M, N, K, K, K, N, splits, M * N, tile,
                                             torch.cuda.current_stream().cuda_stream), "slabs")
    want, want_rs = slabs[0].clone(), rs_slabs[0].clone()
    for sl in range(1, splits):
        want += slabs[sl]
        want_rs += rs_slabs[sl]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    load = torch.randn(4096, 4096, device=dev)
    results = []
    torch.cuda.synchronize()
    for rep in range(6):
        for st in streams:
            with torch.cuda.stream(st):
                load @ load  # uneven load beside the reductions
                results.append(ops.gemm_nt(a, b, splits, rowsum=True, tile=tile, reduce_in_kernel=True))
    torch.cuda.synchronize()
    for out, rs in results:
        assert torch.equal(out, want) and torch.equal(rs, want_rs)
    out, rs = ops.gemm_nt(a, b, splits, rowsum=True, tile=tile, reduce_in_kernel=False)  # slabs + advhip_sum_slabs_f32: the same sums
    assert torch.equal(out, want) and torch.equal(rs, want_rs)
    for key, ws in ops._ZERO_WORKSPACES.items():  # every counter (the workspace's first 64 KiB) back at zero
        assert int(ws[: 65536 // 4].view(torch.int32).abs().sum()) == 0


def test_long_video_segment_cache_and_resume(tmp_path):
    """extract_features.py:116-148: long videos are extracted per segment of frames, each segment cached as
    <out>/<name>/<name>_<seg>.npy and re-used when the run is repeated; the stacked result equals the un-segmented one."""
    from anomaly_detection_on_video_amd import extract
    from anomaly_detection_on_video_amd.i3d import I3Res50

    m = I3Res50()
    m.load_state_dict(synth_i3d_state_dict())
    m = m.eval().to(_dev())
    frames = torch.from_numpy(np.random.default_rng(11).integers(0, 256, (72, 70, 80, 3), dtype=np.uint8))
    reads = []

    def read(lo, hi):
        reads.append((lo, hi))
        return frames[lo:hi]

    out = str(tmp_path / "feat")
    whole = extract.extract_video_frames(m, frames, crop=64)
    written = extract.extract_frames([("vid", 72, read)], m, out, long_video_frames=32, seg_len=32, crop=64)
    got = np.load(written["vid"])
    assert got.shape == whole.shape == (5, 10, 2048) and rel_err(got, whole) < 1e-5
    assert reads == [(0, 32), (32, 64), (64, 72)]
    assert sorted(os.listdir(os.path.join(out, "vid"))) == ["vid_0.npy", "vid_1.npy", "vid_2.npy"]
    # resume: the final file is gone, segment 1 comes from its cache (made recognisable), nothing is read again for it
    os.remove(written["vid"])
    os.remove(os.path.join(out, "vid", "vid_2.npy"))
    seg1 = np.load(os.path.join(out, "vid", "vid_1.npy"))
    np.save(os.path.join(out, "vid", "vid_1.npy"), seg1 + 1000.0)
    reads.clear()
    again = np.load(extract.extract_frames([("vid", 72, read)], m, out, long_video_frames=32, seg_len=32, crop=64)["vid"])
    assert reads == [(64, 72)]
    assert np.array_equal(again[2:4], seg1 + 1000.0) and rel_err(again[4:], whole[4:]) < 1e-5
    assert extract.extract_frames([("vid", 72, read)], m, out, long_video_frames=32, seg_len=32, crop=64) == {}  # skip-if-exists
