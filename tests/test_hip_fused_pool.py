"""conv + max-pool fused launches (brick-ordered LDS-DMA kernel) vs the unfused pair and the oracle (run with -m gpu).

The fused forms compute exactly the fp32 conv values of advhip_conv3d_bn_act_f32 and pool them inside the launch, so
the bar is bit-exact against conv3d_bn_act + maxpool3d of this library, and the usual 2e-5 against the CPU oracle's
max_pool3d(conv_bn_act(...)) (/root/reference/src/i3d.py:303-309)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from anomaly_detection_on_video_amd.weights import synth_tensor

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _pack(name, cin, cout, k, s, p):
    from anomaly_detection_on_video_amd import ops

    dev = _dev()
    fan = cin * k[0] * k[1] * k[2]
    wt = synth_tensor(f"fp.{name}.w", (cout, cin) + tuple(k), scale=float(np.sqrt(6.0 / fan)))
    g = synth_tensor(f"fp.{name}.g", (cout,), scale=0.5, offset=1.0)
    be = synth_tensor(f"fp.{name}.b", (cout,), scale=0.25)
    mu = synth_tensor(f"fp.{name}.m", (cout,), scale=0.25)
    var = synth_tensor(f"fp.{name}.v", (cout,), scale=0.5, offset=1.0)
    pc = ops.pack_conv(wt.to(dev), g.to(dev), be.to(dev), mu.to(dev), var.to(dev), 1e-5, s, p, name=name)
    return pc, (wt, g, be, mu, var)


# (B, T, H, W) of the stem input: full size, sizes whose conv output is not a multiple of the brick (4 x 16), odd extents,
# the smallest input with one pooled output, and one where whole bricks fall outside the tensor
STEM_SHAPES = [(2, 16, 224, 224), (1, 8, 112, 96), (3, 6, 50, 70), (1, 4, 38, 38), (2, 5, 11, 13), (1, 16, 64, 230)]


@pytest.mark.parametrize("shape", STEM_SHAPES, ids=[str(s) for s in STEM_SHAPES])
def test_stem_conv_relu_maxpool233_is_bit_exact(shape):
    from anomaly_detection_on_video_amd import ops
    from oracle import i3d_oracle

    b, t, h, w = shape
    pc, (wt, g, be, mu, var) = _pack("stem", 3, 64, (5, 7, 7), (2, 2, 2), (2, 3, 3))
    x = synth_tensor(f"fp.stem.x{shape}", (b, 3, t, h, w), scale=2.0)
    xd = x.to(_dev())
    fused = ops.conv3d_bn_relu_maxpool233(xd, pc)
    for algo in (162, 163, 35):
        unfused = ops.maxpool3d(ops.conv3d_bn_act(xd, pc, relu=True, algo=algo), (2, 3, 3), (2, 2, 2))
        assert fused.shape == unfused.shape
        assert torch.equal(fused, unfused), f"algo {algo}: max diff {float((fused - unfused).abs().max()):.3e}"
    if b * t * h * w <= 3 * 6 * 50 * 70:
        ref = torch.nn.functional.max_pool3d(i3d_oracle.conv_bn_act(x, wt, g, be, mu, var, (2, 2, 2), (2, 3, 3), None, True), (2, 3, 3), (2, 2, 2))
        assert rel_err(fused.cpu(), ref) < 2e-5
    # into a channel slice of a wider buffer (how layer1.0 consumes it), neighbours untouched
    wide = torch.full((b, 64 + 64) + tuple(fused.shape[2:]), -7.0, device=_dev())
    ops.conv3d_bn_relu_maxpool233(xd, pc, out=wide[:, :64])
    assert torch.equal(wide[:, :64], fused) and (wide[:, 64:] == -7.0).all()


S2W_CASES = [  # (name, cin, cout, kernel, stride, padding, (B, T, H, W)): stride 2 / odd kernel / 'same' padding along w, W % 8 == 0
    ("stem", 3, 64, (5, 7, 7), (2, 2, 2), (2, 3, 3), (2, 16, 224, 224)),
    ("stem", 3, 64, (5, 7, 7), (2, 2, 2), (2, 3, 3), (1, 4, 38, 40)),     # H not a multiple of the brick, whole bricks outside
    ("stem", 3, 64, (5, 7, 7), (2, 2, 2), (2, 3, 3), (2, 5, 11, 16)),     # the smallest widths
    ("stem", 3, 64, (5, 7, 7), (2, 2, 2), (2, 3, 3), (1, 16, 64, 232)),   # Wo = 116: bricks past the last column
    ("s555", 8, 64, (5, 5, 5), (2, 2, 2), (2, 2, 2), (2, 6, 30, 48)),     # kw = 5, pw = 2
    ("s133", 16, 128, (1, 3, 3), (1, 1, 2), (0, 1, 1), (1, 4, 21, 64)),   # kw = 3, stride 2 along w only, two n-tiles
    ("s379", 4, 64, (3, 7, 9), (1, 2, 2), (1, 3, 4), (1, 4, 20, 24)),     # kw = 9, pw = 4 (the widest the planes' padding admits)
]


@pytest.mark.parametrize("case", S2W_CASES, ids=[f"{c[0]}-{c[6]}" for c in S2W_CASES])
def test_column_parity_gather_is_bit_identical_to_the_plain_gather(case):
    """advhip_conv3d_s2w_bn_relu_maxpool233_f32 (16-byte pieces from column-parity planes of the input) against the 4-byte
    gather from the NCDHW input: same K order, operands and accumulation -> torch.equal; and against conv + maxpool3d."""
    from anomaly_detection_on_video_amd import ops

    name, cin, cout, k, st, pd, (b, t, h, w) = case
    pc, _ = _pack(name, cin, cout, k, st, pd)
    assert ops.s2w_ok(pc, w)
    x = synth_tensor(f"fp.s2w.{name}.x{(b, t, h, w)}", (b, cin, t, h, w), scale=2.0).to(_dev())
    xs = ops.split_w(x)
    assert xs.shape == (b, cin, t, h, 2, w // 2 + 4)
    assert torch.equal(xs[..., 0, 2 : 2 + w // 2], x[..., 0::2]) and torch.equal(xs[..., 1, 2 : 2 + w // 2], x[..., 1::2])
    assert float(xs[..., :2].abs().max()) == 0.0 and float(xs[..., 2 + w // 2 :].abs().max()) == 0.0
    plain = ops.conv3d_bn_relu_maxpool233(x, pc, s2w=False)
    planes = ops.conv3d_bn_relu_maxpool233(x, pc, s2w=True)
    assert torch.equal(planes, plain), f"max diff {float((planes - plain).abs().max()):.3e}"
    unfused = ops.maxpool3d(ops.conv3d_bn_act(x, pc, relu=True, algo=162), (2, 3, 3), (2, 2, 2))
    assert torch.equal(planes, unfused)
    wide = torch.full((b, cout + 32) + tuple(plain.shape[2:]), -7.0, device=_dev())
    ops.conv3d_bn_relu_maxpool233(x, pc, out=wide[:, :cout], s2w=True)
    assert torch.equal(wide[:, :cout], plain) and (wide[:, cout:] == -7.0).all()


def test_conv_relu_maxpool233_other_convs():
    """Not only the stem: any conv the LDS-DMA kernel runs (here a padded 3x3x3 and an un-padded 1x1x1, Cout 128)."""
    from anomaly_detection_on_video_amd import ops

    for name, cin, cout, k, s, p, shape in [("c333", 32, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 4, 21, 37)),
                                            ("c111", 64, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 6, 9, 35))]:
        pc, _ = _pack(name, cin, cout, k, s, p)
        x = synth_tensor(f"fp.{name}.x", (shape[0], cin) + shape[1:], scale=2.0).to(_dev())
        fused = ops.conv3d_bn_relu_maxpool233(x, pc)
        unfused = ops.maxpool3d(ops.conv3d_bn_act(x, pc, relu=True, algo=162), (2, 3, 3), (2, 2, 2))
        assert torch.equal(fused, unfused)


TPOOL_SHAPES = [(2, 4, 55, 55), (1, 2, 7, 9), (3, 5, 13, 11), (2, 4, 8, 8), (1, 6, 1, 1)]


@pytest.mark.parametrize("shape", TPOOL_SHAPES, ids=[str(s) for s in TPOOL_SHAPES])
@pytest.mark.parametrize("cin,cout", [(64, 256), (128, 64)])
def test_conv_act_maxpool211_is_bit_exact(shape, cin, cout):
    from anomaly_detection_on_video_amd import ops
    from oracle import i3d_oracle

    b, t, h, w = shape
    pc, (wt, g, be, mu, var) = _pack(f"tp{cin}", cin, cout, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    x = synth_tensor(f"fp.tp.x{shape}{cin}", (b, cin, t, h, w), scale=2.0)
    res = synth_tensor(f"fp.tp.r{shape}{cout}", (b, cout, t, h, w), scale=1.0)
    xd, rd = x.to(_dev()), res.to(_dev())
    for use_res, relu in ((True, True), (False, True), (True, False)):
        fused = ops.conv3d_bn_act_maxpool211(xd, pc, relu=relu, residual=rd if use_res else None)
        unfused = ops.maxpool3d(ops.conv3d_bn_act(xd, pc, relu=relu, residual=rd if use_res else None, algo=162), (2, 1, 1), (2, 1, 1))
        assert fused.shape == unfused.shape == (b, cout, t // 2, h, w)
        assert torch.equal(fused, unfused), f"res={use_res} relu={relu}: max diff {float((fused - unfused).abs().max()):.3e}"
        ref = torch.nn.functional.max_pool3d(i3d_oracle.conv_bn_act(x, wt, g, be, mu, var, (1, 1, 1), (0, 0, 0), res if use_res else None, relu), (2, 1, 1), (2, 1, 1))
        assert rel_err(fused.cpu(), ref) < 2e-5
    wide_in = torch.full((b, cin + 32, t, h, w), 3.0, device=_dev())
    wide_in[:, 32:] = xd
    wide_out = torch.full((b, cout + 8, t // 2, h, w), -5.0, device=_dev())
    ops.conv3d_bn_act_maxpool211(wide_in[:, 32:], pc, relu=True, residual=rd, out=wide_out[:, 8:])
    assert torch.equal(wide_out[:, 8:], ops.conv3d_bn_act_maxpool211(xd, pc, relu=True, residual=rd)) and (wide_out[:, :8] == -5.0).all()


def test_fused_pool_argument_checks():
    from anomaly_detection_on_video_amd import _lib, ops

    pc, _ = _pack("chk", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    x = synth_tensor("fp.chk.x", (1, 64, 4, 9, 9)).to(_dev())
    with pytest.raises(_lib.HipExtensionError, match="1x1x1"):
        ops.conv3d_bn_act_maxpool211(x, pc)
    with pytest.raises(ValueError):
        ops.conv3d_bn_relu_maxpool233(x[:, :, :1, :2, :2].contiguous(), pc)
    with pytest.raises(ValueError):
        ops.conv3d_bn_relu_maxpool233(x, pc, out=torch.empty((1, 64, 2, 4, 5), device=_dev()))


def test_fullnet_fused_pools_equal_unfused_plan():
    """I3Res50 with conv1+maxpool1 and layer1.2.conv3+maxpool2 fused (the default) vs the same plan with the pools as
    their own launches: identical bits at full size (the fused launches pool the very same fp32 conv values), on the
    direct forward and per stream part; per-stage taps still come from the un-fused launches."""
    from anomaly_detection_on_video_amd.i3d import I3Res50
    from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_input

    m = I3Res50(use_nl=False)
    m.load_state_dict(synth_i3d_state_dict(), strict=True)
    m = m.eval().to(_dev())
    from anomaly_detection_on_video_amd import ops

    assert m.fuse_pool and ops.FUSE_AVGPOOL
    for shape, seed in (((2, 3, 16, 224, 224), 0), ((3, 3, 8, 112, 96), 7), ((16, 3, 16, 64, 80), 2)):
        x = synth_input(shape, seed).to(_dev())
        try:
            ops.FUSE_AVGPOOL = False  # the max-pool fusions alone: every other launch is the same one
            m.fuse_pool = True
            y_f = m(x)
            m.fuse_pool = False
            y_u = m(x)
            m.fuse_pool = True
        finally:
            ops.FUSE_AVGPOOL = True
        # small inputs: the un-fused convs may run split-K (another fp32 summation order); otherwise the same bits
        assert torch.equal(y_f, y_u) if shape[-1] == 224 else rel_err(y_f.cpu(), y_u.cpu()) < 1e-5, f"{shape}: max diff {float((y_f - y_u).abs().max()):.3e}"
        # ... and with the global mean folded into layer4.2.conv3 (always unsplit on its 128 x 64 tile; the conv it replaces
        # may run split-K at small batches: same values up to the fp32 summation order)
        y_a = m(x)
        assert rel_err(y_a.cpu(), y_f.cpu()) < 1e-6, f"{shape}: max diff {float((y_a - y_f).abs().max()):.3e}"
    assert sum(1 for u in m._plan if u.absorbed) == 3
    taps = {}
    m.forward_single(synth_input((1, 3, 16, 224, 224), 1).to(_dev()), taps)
    assert tuple(taps["stem"].shape) == (1, 64, 8, 112, 112) and tuple(taps["layer1.2"].shape) == (1, 256, 4, 55, 55)


AVG_CASES = [(32, 512, 2048, (2, 7, 7), True), (5, 512, 2048, (2, 7, 7), False), (3, 64, 128, (2, 8, 8), True), (2, 96, 64, (1, 8, 8), True),
             (4, 32, 64, (1, 2, 2), True), (1, 256, 192, (3, 5, 7), False)]


@pytest.mark.parametrize("case", AVG_CASES, ids=[str(c) for c in AVG_CASES])
def test_conv_avgpool_in_one_launch_is_bit_exact(case):
    """conv3 + bn3 + residual + ReLU + AdaptiveAvgPool3d((1,1,1)) (/root/reference/src/i3d.py:111-121, 314) in one launch
    (advhip_conv3d_epilogue.avgpool_out) vs the conv launch (unsplit) followed by advhip_global_avgpool_f32: the same bits; 98,
    128, 64, 4 and 105 positions per sample; and within 1e-5 of torch's mean."""
    from anomaly_detection_on_video_amd import _lib, ops

    b, cin, cout, thw, with_res = case
    pc, _ = _pack(f"avg{cin}x{cout}", cin, cout, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    x = synth_tensor(f"fp.avg.x.{case}", (b, cin) + thw, scale=2.0).to(_dev())
    res = synth_tensor(f"fp.avg.r.{case}", (b, cout) + thw, scale=1.0).to(_dev()) if with_res else None
    assert ops.avgpool_fusable(pc, thw)
    got = ops.conv3d_bn_act_avgpool(x, pc, relu=True, residual=res)
    full = ops.conv3d_bn_act(x, pc, relu=True, residual=res, algo=_lib.ALGO_DMA2_BASE + _lib.ALGO_IGEMM_64x64, splits=1)
    want = ops.global_avgpool(full)
    assert tuple(got.shape) == (b, cout, 1, 1, 1)
    assert torch.equal(got, want), f"max diff {float((got - want).abs().max()):.3e}"
    assert rel_err(got.cpu(), full.mean(dim=(2, 3, 4), keepdim=True).cpu()) < 1e-5
    with pytest.raises(ValueError):
        ops.conv3d_bn_act_avgpool(torch.zeros((1, cin, 3, 7, 7), device=_dev()), pc)  # 147 positions: more than one tile
