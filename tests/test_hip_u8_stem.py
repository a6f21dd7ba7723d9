"""The stem fed by resized uint8 frames (TenCrop + float + normalise in the conv's load stage) -- run with -m gpu.

Parity chain: the numpy restatement of TenCropVideoFrameDataset (oracle/host_oracle.ten_crop_clips, torchvision's published
TenCrop geometry; /root/reference/src/dataset.py:175-195, src/gtransforms.py:29-38,57-73) -> the oracle's conv + BN + ReLU
-> max_pool3d (/root/reference/src/i3d.py:303-306).  The uint8 kernel accumulates sum w * pixel, subtracts mean * sum w (tabulated per border class) and
applies 1/std with the BN scale, the reference rounds (pixel - mean) / std first: 2e-5 of the output scale, element-wise
|a - b| <= 1e-3 |b| + 1e-3 rms(b)."""
import numpy as np
import pytest
import torch

from conftest import assert_close_elementwise, rel_err
from anomaly_detection_on_video_amd.weights import synth_tensor

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _stem(name="u8stem"):
    from anomaly_detection_on_video_amd import ops

    dev = _dev()
    k, s, p = (5, 7, 7), (2, 2, 2), (2, 3, 3)
    wt = synth_tensor(f"{name}.w", (64, 3) + k, scale=float(np.sqrt(6.0 / (3 * 5 * 7 * 7))))
    g = synth_tensor(f"{name}.g", (64,), scale=0.5, offset=1.0)
    be = synth_tensor(f"{name}.b", (64,), scale=0.25)
    mu = synth_tensor(f"{name}.m", (64,), scale=0.25)
    var = synth_tensor(f"{name}.v", (64,), scale=0.5, offset=1.0)
    pc = ops.pack_conv(wt.to(dev), g.to(dev), be.to(dev), mu.to(dev), var.to(dev), 1e-5, s, p, name=name)
    return pc, (wt, g, be, mu, var)


def _frames(seed, shape):
    rng = np.random.default_rng(seed)
    f = rng.integers(0, 256, size=shape, dtype=np.uint8)
    f[0, :3, :5] = 0      # extreme pixel values in a corner every corner crop / its mirror sees
    f[-1, -3:, -5:] = 255
    return f


# (frames F, FH, FW), frames per clip, crop, (first, count) ranges of crop-clips
CASES = [
    ((32, 256, 340), 16, 224, [(0, 20), (7, 11)]),          # the reference's geometry; a range that starts mid-clip
    ((16, 40, 52), 8, 32, [(0, 20), (3, 1), (15, 5)]),      # small: checked against the CPU oracle too
    ((12, 37, 45), 4, 30, [(0, 30), (9, 13)]),              # odd margins (centre crop rounds half to even), crop not a multiple of the brick
    ((6, 24, 24), 6, 24, [(0, 10)]),                        # crop == frame: all five crops coincide
    ((10, 33, 61), 5, 23, [(4, 16)]),                       # odd crop and clip length
]


@pytest.fixture(params=["taps", "bytes", "planes"])
def form(request, monkeypatch):
    """The uint8 stems: whole pixels per gather, one byte per (channel, tap), and (forward_frames only; the stem entry point
    itself then runs the whole-pixel form) the TenCrop pass writing column-parity planes + the 16-byte-gather stem."""
    from anomaly_detection_on_video_amd import ops

    monkeypatch.setattr(ops, "U8_STEM_FORM", request.param)
    return request.param


@pytest.mark.parametrize("case", CASES, ids=[str(c[0]) + f"/{c[1]}/{c[2]}" for c in CASES])
def test_u8_stem_matches_float_pipeline_and_oracle(case, form):
    from anomaly_detection_on_video_amd import mil_ops, ops
    from oracle import host_oracle, i3d_oracle

    (F, FH, FW), fpc, crop, ranges = case
    pc, (wt, g, be, mu, var) = _stem()
    frames = _frames(F * 1000 + FH, (F, FH, FW, 3))
    fd = torch.from_numpy(frames).to(_dev())
    crops = mil_ops.tencrop_normalize_u8(fd, fpc, crop)            # (n_clips * 10, 3, fpc, crop, crop) fp32 on the device
    ref_all = ops.conv3d_bn_relu_maxpool233(crops, pc)             # the fp32-input product path (bit-exact vs conv + pool)
    scale = float(ref_all.abs().max())
    for first, count in ranges:
        got = ops.conv3d_u8_tencrop_bn_relu_maxpool233(fd, pc, first, count, fpc, crop)
        ref = ref_all[first : first + count]
        assert got.shape == ref.shape
        err = float((got - ref).abs().max()) / scale
        assert err < 2e-5, f"crop-clips [{first},{first + count}): {err:.3e}"
        assert_close_elementwise(got.cpu(), ref.cpu())
    if F * FH * FW <= 16 * 40 * 52:  # the CPU oracle end to end: numpy TenCrop -> torch conv / BN / ReLU / max_pool3d
        x = torch.from_numpy(host_oracle.ten_crop_clips(frames, fpc, crop)).reshape(-1, 3, fpc, crop, crop)
        assert torch.equal(crops.cpu(), x)
        want = torch.nn.functional.max_pool3d(i3d_oracle.conv_bn_act(x, wt, g, be, mu, var, (2, 2, 2), (2, 3, 3), None, True), (2, 3, 3), (2, 2, 2))
        got = ops.conv3d_u8_tencrop_bn_relu_maxpool233(fd, pc, 0, x.shape[0], fpc, crop)
        assert rel_err(got.cpu(), want) < 2e-5
        assert_close_elementwise(got.cpu(), want)


def test_u8_stem_constant_frames_hit_only_the_border_table(form):
    """All pixels = 115 -> (pixel - mean) = 0.25 everywhere: interior outputs are 0.25 * sum(w) and the border outputs differ
    from them only through the taps outside the clip -- a direct check of the per-class correction (a missing correction
    would be off by 114.75 * the outside weights, ~100x the signal)."""
    from anomaly_detection_on_video_amd import mil_ops, ops

    pc, _ = _stem("u8const")
    fd = torch.full((16, 64, 80, 3), 115, dtype=torch.uint8, device=_dev())
    got = ops.conv3d_u8_tencrop_bn_relu_maxpool233(fd, pc, 0, 10, 16, 56)
    ref = ops.conv3d_bn_relu_maxpool233(mil_ops.tencrop_normalize_u8(fd, 16, 56), pc)
    assert float((got - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_u8_stem_writes_into_a_channel_slice_and_rejects_bad_ranges(form):
    from anomaly_detection_on_video_amd import ops

    pc, _ = _stem()
    fd = torch.from_numpy(_frames(5, (16, 40, 52, 3))).to(_dev())
    dense = ops.conv3d_u8_tencrop_bn_relu_maxpool233(fd, pc, 2, 6, 8, 32)
    wide = torch.full((6, 128) + tuple(dense.shape[2:]), -3.0, device=_dev())
    ops.conv3d_u8_tencrop_bn_relu_maxpool233(fd, pc, 2, 6, 8, 32, out=wide[:, :64])
    assert torch.equal(wide[:, :64], dense) and (wide[:, 64:] == -3.0).all()
    with pytest.raises(ValueError):
        ops.conv3d_u8_tencrop_bn_relu_maxpool233(fd, pc, 15, 6, 8, 32)      # past the video's 20 crop-clips
    with pytest.raises(ValueError):
        ops.conv3d_u8_tencrop_bn_relu_maxpool233(fd[:15], pc, 0, 5, 8, 32)  # not whole clips
    with pytest.raises(ValueError):
        ops.conv3d_u8_tencrop_bn_relu_maxpool233(fd, pc, 0, 5, 8, 64)       # crop larger than the frame


def test_forward_frames_whole_backbone_and_fallback(form):
    """I3Res50.forward_frames (uint8 frames -> features) vs forward_single on the fp32 ten-crop tensor: 40 crop-clips of the
    reference's geometry in the chunks the extraction driver uses, and the separate-pass fallback when the pools are not fused."""
    from anomaly_detection_on_video_amd import mil_ops
    from anomaly_detection_on_video_amd.i3d import I3Res50
    from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict

    m = I3Res50()
    m.load_state_dict(synth_i3d_state_dict())
    m = m.eval().to(_dev())
    fd = torch.from_numpy(_frames(11, (64, 256, 340, 3))).to(_dev())
    assert m.frames_fused()
    ref = m(mil_ops.tencrop_normalize_u8(fd, 16, 224))
    got = torch.cat([m.forward_frames(fd, 0, 32), m.forward_frames(fd, 32, 8)])
    assert got.shape == ref.shape == (40, 2048, 1, 1, 1)
    assert rel_err(got.cpu(), ref.cpu()) < 1e-5
    assert_close_elementwise(got.cpu(), ref.cpu())
    if form == "planes":  # same arithmetic per pixel, same K order: the fp32 pipeline's features bit for bit, launch shape for
        assert m._frames_planes(224)  # launch shape (a direct model(x) call cuts its batch over two streams: other tile choices)
        m.streams = 1
        assert torch.equal(m.forward_frames(fd, 0, 40), m.forward_single(mil_ops.tencrop_normalize_u8(fd, 16, 224)))
    m.fuse_pool = False
    assert not m.frames_fused()
    assert torch.equal(m.forward_frames(fd, 5, 4), m(mil_ops.tencrop_normalize_u8(fd, 16, 224)[5:9]))


def test_stream_steps_from_frames_match_steps_from_crops(form):
    """ExtractScoreStream on three lanes fed with FrameCrops (uint8 frames, H2D copy inside the step's `prepare`) vs the same
    steps fed with the fp32 ten-crop tensor: features within 1e-5, first step on every lane equal to a later one bit for bit
    (the lazily built tables are ordered before the lanes that read them), same videos scored."""
    from anomaly_detection_on_video_amd import mil_ops
    from anomaly_detection_on_video_amd.i3d import I3Res50
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection
    from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream, FrameCrops
    from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_module_state_dict

    dev = _dev()
    sc = MGFNForVideoAnomalyDetection(MGFNConfig())
    sc.load_state_dict(synth_module_state_dict(sc))
    sc = sc.eval().to(dev)

    def fresh():
        m = I3Res50()
        m.load_state_dict(synth_i3d_state_dict())
        return m.eval().to(dev)

    host = torch.from_numpy(_frames(21, (32, 256, 340, 3))).pin_memory()  # 2 clips = 20 crop-clips per step
    a = ExtractScoreStream(fresh(), sc, clips_per_video=4, ncrops=10, local_batch=20)
    ha = [a.step_async(host, prepare=lambda h: FrameCrops(h.to(dev, non_blocking=True), 0, 20)) for _ in range(6)]
    a.drain()
    fa = [h.result()[0].cpu() for h in ha]
    b = ExtractScoreStream(fresh(), sc, clips_per_video=4, ncrops=10, local_batch=20)
    hb = [b.step_async(host, prepare=lambda h: mil_ops.tencrop_normalize_u8(h.to(dev, non_blocking=True))) for _ in range(6)]
    b.drain()
    fb = [h.result()[0].cpu() for h in hb]
    torch.cuda.synchronize()
    for i in range(6):
        assert rel_err(fa[i], fb[i]) < 1e-5, f"step {i}"
        assert torch.equal(fa[i], fa[(i + 3) % 6])
    assert a.videos_scored == b.videos_scored == 3  # 6 steps x 2 clips = 12 clips = 3 videos of 4


def test_host_feeder_copies_ahead_of_the_lanes_and_changes_no_result():
    """pipeline.HostFeeder (H2D on a copy stream of its own, a ring of device buffers re-used only after the step that read them)
    feeding a three-lane stream with DIFFERENT frames every step, ring depth 2 < lanes so that buffers are re-used while steps
    are in flight: every step's features equal those of the same frames fed one by one on the lane itself."""
    from anomaly_detection_on_video_amd.i3d import I3Res50
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection
    from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream, FrameCrops, HostFeeder
    from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_module_state_dict

    dev = _dev()
    sc = MGFNForVideoAnomalyDetection(MGFNConfig())
    sc.load_state_dict(synth_module_state_dict(sc))
    sc = sc.eval().to(dev)
    m = I3Res50()
    m.load_state_dict(synth_i3d_state_dict())
    m = m.eval().to(dev)
    hosts = [torch.from_numpy(_frames(100 + i, (16, 72, 88, 3))).pin_memory() for i in range(7)]  # one clip = 10 crop-clips per step
    a = ExtractScoreStream(m, sc, clips_per_video=4, ncrops=10, local_batch=10)
    feeder = HostFeeder(dev, depth=2)
    handles = []
    for h in hosts:
        p = feeder.feed(h, lambda d: FrameCrops(d, 0, 10, 16, 64))
        handle = a.step_async(h, prepare=p)
        feeder.done(p, handle)
        handles.append(handle)
    a.drain()
    got = [h.result()[0].cpu() for h in handles]
    torch.cuda.synchronize()
    for h, g in zip(hosts, got):
        want = m.forward_frames(h.to(dev), 0, 10, 16, 64).reshape(10, -1).cpu()
        assert torch.equal(g, want)


@pytest.mark.parametrize("length", [5, 8, 16])
def test_tencrop_normalize_pass_vs_reference_golden(length):
    """mil_ops.tencrop_normalize_u8 against the reference's own GroupStandardizationTenCrop + LoopPad + permutes (goldens made
    by tests/golden/make_golden.py from src/gtransforms.py:57-73,115-132, src/dataset.py:195, extract_features.py:83), bit
    for bit.  Crop-sized frames, so the (unpinned, torchvision) crop offsets play no part."""
    import os

    from conftest import GOLDEN
    from anomaly_detection_on_video_amd import mil_ops

    g = np.load(os.path.join(GOLDEN, "preproc.npz"))
    got = mil_ops.tencrop_normalize_u8(torch.from_numpy(g[f"frames_{length}"]).to(_dev()), 16, 8)
    assert got.shape == (10, 3, 16, 8, 8)
    assert np.array_equal(got.cpu().numpy(), g[f"clip_{length}"][0])


def test_u8_stem_at_the_reference_geometry_vs_oracle(form):
    """The border-class table at full size against the CPU oracle (not against another HIP path): 2 clips of 256 x 340
    frames, crop 224 -> 20 crop-clips through numpy TenCrop -> torch conv / BN / ReLU -> max_pool3d
    (/root/reference/src/i3d.py:303-306)."""
    from anomaly_detection_on_video_amd import ops
    from oracle import host_oracle, i3d_oracle

    pc, (wt, g, be, mu, var) = _stem()
    frames = _frames(77, (32, 256, 340, 3))
    fd = torch.from_numpy(frames).to(_dev())
    got = ops.conv3d_u8_tencrop_bn_relu_maxpool233(fd, pc, 0, 20, 16, 224).cpu()
    x = torch.from_numpy(host_oracle.ten_crop_clips(frames, 16, 224)).reshape(20, 3, 16, 224, 224)
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    want = torch.cat([torch.nn.functional.max_pool3d(i3d_oracle.conv_bn_act(x[i : i + 5], wt, g, be, mu, var, (2, 2, 2), (2, 3, 3), None, True),
                                                     (2, 3, 3), (2, 2, 2)) for i in range(0, 20, 5)])
    assert got.shape == want.shape == (20, 64, 4, 55, 55)
    assert rel_err(got, want) < 2e-5
    assert_close_elementwise(got, want)


@pytest.mark.parametrize("case", [((32, 256, 340), 16, 224, (0, 20)), ((20, 40, 52), 8, 32, (7, 11)), ((6, 24, 24), 6, 24, (0, 10)), ((13, 30, 44), 4, 16, (5, 33))],
                         ids=["reference-geometry", "mid-clip-range", "crop==frame", "short-last-clip"])
def test_tencrop_planes_pass_equals_the_two_pass_form(case):
    """advhip_tencrop_normalize_planes_u8 (TenCrop + float + normalise + LoopPad + permutes written as column-parity planes for a
    range of crop-clips) against advhip_tencrop_normalize_u8 followed by advhip_split_w_f32: bit for bit, padding columns zero."""
    from anomaly_detection_on_video_amd import mil_ops, ops

    (F, FH, FW), fpc, crop, (first, count) = case
    fd = torch.from_numpy(_frames(F + FH, (F, FH, FW, 3))).to(_dev())
    want = ops.split_w(mil_ops.tencrop_normalize_u8(fd, fpc, crop)[first : first + count].contiguous())
    got = ops.tencrop_planes_u8(fd, first, count, fpc, crop)
    assert got.shape == want.shape == (count, 3, fpc, crop, 2, crop // 2 + 4)
    assert torch.equal(got, want)
    with pytest.raises(ValueError):
        ops.tencrop_planes_u8(fd, first, 10 * (-(-F // fpc)) - first + 1, fpc, crop)
