"""GPU parity tests for the I3D backbone kernels (run with -m gpu on an MI355X).

Every comparison is HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs, or
vs the golden vectors made by the reference's own code.  Tolerance: north_star's 1e-3 relative
(norm-wise, max|a-b|/max|b|); the fp32-MFMA path is expected to land around 1e-6.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_input, synth_tensor

pytestmark = pytest.mark.gpu

TOL = 1e-3        # the contract
TIGHT = 2e-5      # what exact-fp32 MFMA should actually achieve per conv


def _dev():
    return torch.device("cuda:0")


# (name, Cin, Cout, kernel, stride, padding, (B,T,H,W))  -- every distinct conv config of I3Res50
#  on reduced batch / spatial extents, plus edge cases
CONV_CASES = [
    ("stem", 3, 64, (5, 7, 7), (2, 2, 2), (2, 3, 3), (2, 8, 36, 28)),
    ("l1.conv1.t3", 64, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 4, 13, 11)),
    ("l1.conv2", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 4, 13, 11)),
    ("l1.conv3", 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 13, 11)),
    ("l1.conv1b", 256, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 4, 9, 7)),
    ("l2.conv1", 256, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 2, 11, 11)),
    ("l2.conv2.s2", 128, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 2, 11, 11)),
    ("l2.conv3", 128, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 2, 6, 6)),
    ("l2.ds.s2", 256, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0), (2, 2, 11, 11)),
    ("l2.conv1.k1", 512, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 2, 6, 6)),
    ("l2.conv2", 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 2, 6, 6)),
    ("l2.conv1.t3", 512, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 2, 6, 6)),
    ("l3.conv1", 512, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 2, 6, 6)),
    ("l3.conv2.s2", 256, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 2, 6, 6)),
    ("l3.conv3", 256, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0), (3, 2, 3, 3)),
    ("l3.ds.s2", 512, 1024, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 2, 6, 6)),
    ("l3.conv1.k1", 1024, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 2, 3, 3)),
    ("l3.conv2", 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 2, 3, 3)),
    ("l3.conv1.t3", 1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 2, 3, 3)),
    ("l4.conv1", 1024, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 2, 3, 3)),
    ("l4.conv2.s2", 512, 512, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 2, 3, 3)),
    ("l4.conv3", 512, 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 2, 2, 2)),
    ("l4.ds.s2", 1024, 2048, (1, 1, 1), (1, 2, 2), (0, 0, 0), (2, 2, 3, 3)),
    ("l4.conv1.t3", 2048, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 2, 2, 2)),
    ("l4.conv2", 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 2, 2, 2)),
    ("l4.conv1.k1", 2048, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 2, 2, 2)),
    # edge cases: full-size 7x7 frames (THW = 98, not a multiple of 4), single output position,
    # odd extents everywhere, output smaller than one tile
    ("edge.7x7", 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), (3, 2, 7, 7)),
    ("edge.1pos", 64, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1, 1)),
    ("edge.odd", 64, 128, (3, 3, 3), (1, 2, 1), (1, 0, 2), (1, 3, 5, 7)),
    ("edge.55", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 55, 55)),
]


def _conv_case(name, cin, cout, k, s, p, bthw):
    from oracle import i3d_oracle

    b, t, h, w = bthw
    fan = cin * k[0] * k[1] * k[2]
    x = synth_tensor(f"conv.{name}.x", (b, cin, t, h, w), scale=2.0)
    wt = synth_tensor(f"conv.{name}.w", (cout, cin) + tuple(k), scale=float(np.sqrt(6.0 / fan)))
    g = synth_tensor(f"conv.{name}.g", (cout,), scale=0.5, offset=1.0)
    be = synth_tensor(f"conv.{name}.b", (cout,), scale=0.25)
    mu = synth_tensor(f"conv.{name}.m", (cout,), scale=0.25)
    var = synth_tensor(f"conv.{name}.v", (cout,), scale=0.5, offset=1.0)
    y0 = i3d_oracle.conv_bn_act(x, wt, g, be, mu, var, s, p, None, relu=False)
    res = synth_tensor(f"conv.{name}.r", tuple(y0.shape), scale=1.0)
    return x, wt, g, be, mu, var, res


DMA = [65, 66, 67, 68, 70, 71, 72]
DMA4 = [98, 99, 100]
DMA2 = [161, 162, 163, 164, 166, 167, 168, 169]  # 169: the 256 x 64 tile (four waves along M)
ALGOS = [0] + list(range(1, 9)) + list(range(33, 41)) + DMA + DMA4 + DMA2
TILE_IDS = ["128x128", "128x64", "64x64", "64x128", "128x128x32", "128x64x32", "64x64x32", "64x128x32", "256x64"]


def _skip_algo(algo, cout, k):
    base = algo % 32
    if base in (1, 4, 5, 8) and cout % 128:
        pytest.skip("Cout not a multiple of the 128-wide N tile")


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
@pytest.mark.parametrize("algo", ALGOS, ids=["auto"] + TILE_IDS[:8] + ["fast" + t for t in TILE_IDS[:8]] + ["dma" + TILE_IDS[a - 65] for a in DMA] + ["dma4_" + TILE_IDS[a - 97] for a in DMA4] + ["dma2_" + TILE_IDS[a - 161] for a in DMA2])
def test_conv_bn_act_vs_oracle(case, algo):
    from anomaly_detection_on_video_amd import ops
    from oracle import i3d_oracle

    name, cin, cout, k, s, p, bthw = case
    _skip_algo(algo, cout, k)
    x, wt, g, be, mu, var, res = _conv_case(*case)
    dev = _dev()
    pc = ops.pack_conv(wt.to(dev), g.to(dev), be.to(dev), mu.to(dev), var.to(dev), 1e-5, s, p, name=name)
    for use_res, relu in ((False, True), (True, True), (True, False)):
        ref = i3d_oracle.conv_bn_act(x, wt, g, be, mu, var, s, p, res if use_res else None, relu)
        out = ops.conv3d_bn_act(x.to(dev), pc, relu=relu, residual=res.to(dev) if use_res else None, algo=algo)
        torch.cuda.synchronize()
        assert out.shape == ref.shape
        e = rel_err(out.cpu(), ref)
        assert e < TIGHT, f"{name} algo={algo} res={use_res} relu={relu}: rel err {e:.3e}"
        assert e < TOL


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[0] in ("l1.conv3", "l4.conv2", "edge.7x7", "edge.odd")], ids=lambda c: c[0])
@pytest.mark.parametrize("algo", [0, 161, 162, 163, 164, 169, 66, 3], ids=["auto", "dma2_128x128", "dma2_128x64", "dma2_64x64", "dma2_64x128", "dma2_256x64", "dma_128x64", "igemm_64x64"])
def test_conv_residual_may_be_the_output_buffer(case, algo):
    """y = act(conv(x) + res) written over `res` itself (the Bottleneck's `out += residual`, src/i3d.py:118-121, done in place): the
    epilogue keeps several residual pieces in flight before it stores anything, so every piece must have been read by the lane that
    overwrites it -- the in-place launch equals the out-of-place one bit for bit, on rows of a multiple of 4 positions and on odd ones."""
    from anomaly_detection_on_video_amd import ops

    name, cin, cout, k, s, p, bthw = case
    _skip_algo(algo, cout, k)
    x, wt, g, be, mu, var, res = _conv_case(*case)
    dev = _dev()
    pc = ops.pack_conv(wt.to(dev), g.to(dev), be.to(dev), mu.to(dev), var.to(dev), 1e-5, s, p, name=name)
    xd, rd = x.to(dev), res.to(dev)
    want = ops.conv3d_bn_act(xd, pc, relu=True, residual=rd, algo=algo, splits=1)
    buf = rd.clone()
    got = ops.conv3d_bn_act(xd, pc, relu=True, residual=buf, algo=algo, splits=1, out=buf)
    torch.cuda.synchronize()
    assert got.data_ptr() == buf.data_ptr()
    assert torch.equal(got, want)


def test_conv_and_pool_on_channel_slices():
    """x / y as channel slices of wider NCDHW buffers (batch stride > one sample): how layer1.0's downsample branch is
    folded into conv3 -- the pool writes channels [0,64) and conv2 channels [64,128) of one buffer, and a 128-channel
    1x1x1 conv reads it.  Same numbers as on dense tensors, neighbours of the slices untouched."""
    from anomaly_detection_on_video_amd import ops
    from oracle import i3d_oracle

    dev = _dev()
    name, cin, cout, k, s, p, bthw = next(c for c in CONV_CASES if c[0] == "l1.conv2")
    x, wt, g, be, mu, var, res = _conv_case(name, cin, cout, k, s, p, bthw)
    pc = ops.pack_conv(wt.to(dev), g.to(dev), be.to(dev), mu.to(dev), var.to(dev), 1e-5, s, p, name=name)
    b, t, h, w = bthw
    wide_in = torch.full((b, cin + 24, t, h, w), 7.0, device=dev)
    wide_in[:, 8 : 8 + cin] = x.to(dev)
    wide_out = torch.full((b, cout + 40, t, h, w), -3.0, device=dev)
    dense = ops.conv3d_bn_act(x.to(dev), pc, relu=True, residual=res.to(dev))
    for algo in (None, 67, 163, 35, 3, 134):
        wide_out.fill_(-3.0)
        got = ops.conv3d_bn_act(wide_in[:, 8 : 8 + cin], pc, relu=True, residual=res.to(dev), out=wide_out[:, 16 : 16 + cout], algo=algo)
        if algo in (None, 67):
            assert torch.equal(got, dense) or rel_err(got.cpu(), dense.cpu()) < TIGHT
        assert rel_err(got.cpu(), i3d_oracle.conv_bn_act(x, wt, g, be, mu, var, s, p, res, True)) < (BF16X3_TOL if algo == 134 else TIGHT)
        assert (wide_out[:, :16] == -3.0).all() and (wide_out[:, 16 + cout :] == -3.0).all()
    with pytest.raises(ValueError):
        ops.conv3d_bn_act(wide_in[:, :, :, :, ::2][:, :cin], pc)
    xp = synth_tensor("slice.pool", (2, 6, 4, 12, 10), scale=2.0).to(dev)
    for kk, ss in (((2, 3, 3), (2, 2, 2)), ((2, 1, 1), (2, 1, 1)), ((1, 2, 2), (1, 1, 1))):
        ref = ops.maxpool3d(xp, kk, ss)
        wide = torch.full((2, 11) + tuple(ref.shape[2:]), 5.0, device=dev)
        ops.maxpool3d(xp, kk, ss, out=wide[:, 3:9])
        assert torch.equal(wide[:, 3:9], ref) and (wide[:, :3] == 5.0).all() and (wide[:, 9:] == 5.0).all()


BF16X3_TOL = 1e-4  # split-bf16 arithmetic (hi*hi + hi*lo + lo*hi): ~2^-16 per product; the contract is TOL = 1e-3


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
@pytest.mark.parametrize("algo,splits", [(133, 1), (134, 1), (134, 2), (133, 3)], ids=["x3_128x128", "x3_128x64", "x3_128x64_s2", "x3_128x128_s3"])
def test_conv_split_bf16_vs_oracle(case, algo, splits):
    """The opt-in split-bf16 kernels (ADVHIP_ALGO_BF16X3_*) against the fp32 oracle: not bit-comparable with an fp32
    FMA chain by construction, so the bar is 1e-4 of the output range (10x inside the 1e-3 contract), on every conv
    shape of the net and the edge cases, with and without residual / ReLU / split-K; deterministic run to run."""
    from anomaly_detection_on_video_amd import ops
    from oracle import i3d_oracle

    name, cin, cout, k, s, p, bthw = case
    if algo == 133 and cout % 128:
        pytest.skip("Cout not a multiple of the 128-wide N tile")
    x, wt, g, be, mu, var, res = _conv_case(*case)
    dev = _dev()
    pc = ops.pack_conv(wt.to(dev), g.to(dev), be.to(dev), mu.to(dev), var.to(dev), 1e-5, s, p, name=name)
    if splits > pc.w_packed.shape[0] // 32:
        pytest.skip("fewer k-tiles than splits")
    for use_res, relu in ((False, True), (True, False)):
        ref = i3d_oracle.conv_bn_act(x, wt, g, be, mu, var, s, p, res if use_res else None, relu)
        out = ops.conv3d_bn_act(x.to(dev), pc, relu=relu, residual=res.to(dev) if use_res else None, algo=algo, splits=splits)
        again = ops.conv3d_bn_act(x.to(dev), pc, relu=relu, residual=res.to(dev) if use_res else None, algo=algo, splits=splits)
        assert out.shape == ref.shape
        e = rel_err(out.cpu(), ref)
        assert e < BF16X3_TOL, f"{name} algo={algo} splits={splits} res={use_res} relu={relu}: rel err {e:.3e}"
        assert torch.equal(out, again)


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[0] in ("l2.conv2", "l3.conv1.t3", "l4.conv2", "edge.7x7", "l1.conv3", "edge.odd")],
                         ids=lambda c: c[0])
@pytest.mark.parametrize("algo,splits", [(3, 2), (1, 3), (7, 4), (6, 2), (0, 0), (35, 2), (36, 3), (39, 4), (67, 2), (68, 3), (71, 4), (99, 3), (100, 2), (163, 2), (164, 3), (167, 4), (169, 2), (169, 3)])
def test_conv_split_k_vs_oracle(case, algo, splits):
    """split-K slabs + fixed-order reduce pass: same parity bar, and bit-identical run to run."""
    from anomaly_detection_on_video_amd import ops
    from oracle import i3d_oracle

    name, cin, cout, k, s, p, bthw = case
    _skip_algo(algo, cout, k)
    x, wt, g, be, mu, var, res = _conv_case(*case)
    dev = _dev()
    pc = ops.pack_conv(wt.to(dev), g.to(dev), be.to(dev), mu.to(dev), var.to(dev), 1e-5, s, p, name=name)
    kpad = pc.w_packed.shape[0]
    bk = 32 if (algo % 32) >= 5 else 16
    if splits > kpad // bk:
        pytest.skip("fewer k-tiles than splits")
    ref = i3d_oracle.conv_bn_act(x, wt, g, be, mu, var, s, p, res, True)
    out = ops.conv3d_bn_act(x.to(dev), pc, relu=True, residual=res.to(dev), algo=algo, splits=splits)
    out2 = ops.conv3d_bn_act(x.to(dev), pc, relu=True, residual=res.to(dev), algo=algo, splits=splits)
    assert rel_err(out.cpu(), ref) < TIGHT
    assert torch.equal(out, out2)


TSPAN_CASES = [c for c in CONV_CASES if c[3][1:] == (1, 1) and c[4] == (1, 1, 1) and c[6][1] % 2 == 0] + [
    ("tspan.t6", 64, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 6, 9, 10)),     # T = 6: two-frame bricks, three per column
    ("tspan.t8.k5", 32, 64, (5, 1, 1), (1, 1, 1), (2, 0, 0), (1, 8, 7, 5)),    # five temporal taps, four-frame bricks
    ("tspan.55", 256, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 4, 55, 55)),    # layer1.x.conv1 at its real plane size
]


@pytest.mark.parametrize("case", TSPAN_CASES, ids=[c[0] for c in TSPAN_CASES])
def test_conv_tspan_tiles_vs_oracle(case):
    """ADVHIP_ALGO_TSPAN_128x64: (kt,1,1) convs on m-tiles that span T (brick-ordered rows, same MFMA loop).  Same fp32
    k-order as the plain 128x64 tile, so bit-identical to it, and within 2e-5 of the oracle."""
    from anomaly_detection_on_video_amd import _lib, ops
    from oracle import i3d_oracle

    name, cin, cout, k, s, p, bthw = case
    x, wt, g, be, mu, var, res = _conv_case(*case)
    dev = _dev()
    pc = ops.pack_conv(wt.to(dev), g.to(dev), be.to(dev), mu.to(dev), var.to(dev), 1e-5, s, p, name=name)
    for use_res, relu in ((False, True), (True, True), (True, False)):
        ref = i3d_oracle.conv_bn_act(x, wt, g, be, mu, var, s, p, res if use_res else None, relu)
        out = ops.conv3d_bn_act(x.to(dev), pc, relu=relu, residual=res.to(dev) if use_res else None, algo=_lib.ALGO_TSPAN_128x64)
        assert rel_err(out.cpu(), ref) < TIGHT, f"{name} res={use_res} relu={relu}"
        plain = ops.conv3d_bn_act(x.to(dev), pc, relu=relu, residual=res.to(dev) if use_res else None, algo=162)
        assert torch.equal(out, plain)
    # into a channel slice of a wider buffer
    wide = torch.full((bthw[0], cout + 64) + tuple(out.shape[2:]), 9.0, device=dev)
    ops.conv3d_bn_act(x.to(dev), pc, relu=False, residual=res.to(dev), out=wide[:, 64:], algo=_lib.ALGO_TSPAN_128x64)
    assert torch.equal(wide[:, 64:], out) and (wide[:, :64] == 9.0).all()


def test_conv_tspan_rejects_what_it_cannot_run():
    from anomaly_detection_on_video_amd import _lib, ops

    dev = _dev()
    for name in ("l1.conv2", "l2.ds.s2"):  # spatial window / strided
        case = next(c for c in CONV_CASES if c[0] == name)
        x, wt, g, be, mu, var, _res = _conv_case(*case)
        pc = ops.pack_conv(wt.to(dev), g.to(dev), be.to(dev), mu.to(dev), var.to(dev), 1e-5, case[4], case[5], name=name)
        with pytest.raises(_lib.HipExtensionError, match="TSPAN"):
            ops.conv3d_bn_act(x.to(dev), pc, algo=_lib.ALGO_TSPAN_128x64)
    case = ("odd.t", 64, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 3, 5, 5))
    x, wt, g, be, mu, var, _res = _conv_case(*case)
    pc = ops.pack_conv(wt.to(dev), g.to(dev), be.to(dev), mu.to(dev), var.to(dev), 1e-5, case[4], case[5], name="odd.t")
    with pytest.raises(_lib.HipExtensionError, match="even number of frames"):
        ops.conv3d_bn_act(x.to(dev), pc, algo=_lib.ALGO_TSPAN_128x64)


MIXED_CASES = [  # (name, Cin, Cout, (B, T, H, W)): 1x1x1 convs whose 128 x 64 tiles number a little more than one round of 6 x 256 resident workgroups
    ("mixed.even", 64, 128, (1, 1, 8, 12416)),    # M = 776 x 128: 1 552 tiles, 16 past the round -> the last 8 tall m-tile rows cut in two
    ("mixed.ragged", 64, 128, (1, 1, 1, 98596)),  # M = 770 x 128 + 36: 1 542 tiles -> 3 rows cut; the last 64-row tile holds 36 positions
    ("mixed.rows98", 64, 256, (494, 2, 7, 7)),    # 98 positions per sample (layer-4 planes): every sample's rows padded to 100 in the M index space
    ("mixed.below", 64, 128, (1, 1, 4, 512)),     # below one round: the plain 128 x 64 launch
]


@pytest.mark.parametrize("case", MIXED_CASES, ids=[c[0] for c in MIXED_CASES])
def test_conv_mixed_tail_equals_the_plain_tiles_bit_for_bit(case):
    """ADVHIP_ALGO_MIXED_128x64: the last m-tile rows of an unsplit 1x1x1 conv as 64 x 64 tiles, so that the partial last round of
    workgroups is short ones (conv1x1_mixed_tail_kernel).  Same K order per output as the plain 128 x 64 launch: the same bits, with
    and without residual / ReLU; fp64 torch reference within 2e-5."""
    from anomaly_detection_on_video_amd import _lib, ops

    name, cin, cout, bthw = case
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(len(name))
    x = torch.randn((bthw[0], cin) + tuple(bthw[1:]), device=dev, generator=g)
    wt = torch.randn((cout, cin, 1, 1, 1), device=dev, generator=g) * (2.0 / cin) ** 0.5
    gam, bet = torch.rand(cout, device=dev, generator=g) + 0.5, torch.randn(cout, device=dev, generator=g) * 0.2
    mu, var = torch.randn(cout, device=dev, generator=g) * 0.2, torch.rand(cout, device=dev, generator=g) + 0.5
    pc = ops.pack_conv(wt, gam, bet, mu, var, 1e-5, (1, 1, 1), (0, 0, 0), name=name)
    res = torch.randn((bthw[0], cout) + tuple(bthw[1:]), device=dev, generator=g)
    sc = (gam / torch.sqrt(var + 1e-5)).double()
    ref0 = torch.einsum("oc,bcthw->bothw", wt.view(cout, cin).double(), x.double()) * sc.view(1, -1, 1, 1, 1) + (bet.double() - mu.double() * sc).view(1, -1, 1, 1, 1)
    for use_res, relu in ((False, True), (True, True), (True, False)):
        out = ops.conv3d_bn_act(x, pc, relu=relu, residual=res if use_res else None, algo=_lib.ALGO_MIXED_128x64, splits=1)
        plain = ops.conv3d_bn_act(x, pc, relu=relu, residual=res if use_res else None, algo=162, splits=1)
        assert torch.equal(out, plain), f"{name} res={use_res} relu={relu}"
        ref = ref0 + res.double() if use_res else ref0
        ref = ref.clamp_min(0) if relu else ref
        assert rel_err(out.cpu(), ref.cpu()) < TIGHT
    buf = res.clone()  # the residual may be the output buffer (Bottleneck.forward's `out += residual`, src/i3d.py:108-121)
    ops.conv3d_bn_act(x, pc, relu=True, residual=buf, algo=_lib.ALGO_MIXED_128x64, splits=1, out=buf)
    assert torch.equal(buf, ops.conv3d_bn_act(x, pc, relu=True, residual=res, algo=162, splits=1))


def test_conv_mixed_tail_rejects_what_it_cannot_run():
    from anomaly_detection_on_video_amd import _lib, ops

    dev = _dev()
    case = next(c for c in CONV_CASES if c[0] == "l1.conv2")  # a spatial window
    x, wt, g, be, mu, var, _res = _conv_case(*case)
    pc = ops.pack_conv(wt.to(dev), g.to(dev), be.to(dev), mu.to(dev), var.to(dev), 1e-5, case[4], case[5], name="l1.conv2")
    with pytest.raises(_lib.HipExtensionError, match="MIXED"):
        ops.conv3d_bn_act(x.to(dev), pc, algo=_lib.ALGO_MIXED_128x64, splits=1)


@pytest.mark.parametrize("shape,k,s", [
    ((2, 64, 8, 28, 30), (2, 3, 3), (2, 2, 2)),   # maxpool1 (odd output extents)
    ((2, 256, 4, 11, 13), (2, 1, 1), (2, 1, 1)),  # maxpool2
    ((1, 3, 5, 7, 9), (1, 2, 3), (1, 1, 2)),
    ((2, 5, 8, 112, 112), (2, 3, 3), (2, 2, 2)),  # full-width stem pool rows (wave-per-row fast path)
    ((1, 2, 3, 9, 126), (2, 3, 3), (2, 2, 2)),
    ((3, 7, 5, 55, 55), (2, 1, 1), (2, 1, 1)),    # odd HW, odd T (last frame dropped)
    ((1, 2, 7, 3, 5), (3, 1, 1), (2, 1, 1)),
])
def test_maxpool3d_bit_exact(shape, k, s):
    from anomaly_detection_on_video_amd import ops

    x = synth_tensor(f"pool.{shape}", shape, scale=3.0)
    ref = torch.nn.functional.max_pool3d(x, k, s)
    out = ops.maxpool3d(x.to(_dev()), k, s).cpu()
    assert torch.equal(out, ref)  # pure selection: bit exact
    xn = x.clone()
    xn.view(-1)[::97] = float("nan")  # NaN propagation like torch
    assert torch.equal(torch.isnan(ops.maxpool3d(xn.to(_dev()), k, s).cpu()), torch.isnan(torch.nn.functional.max_pool3d(xn, k, s)))


def test_global_avgpool():
    from anomaly_detection_on_video_amd import ops

    x = synth_tensor("avgpool", (3, 2048, 2, 7, 7), scale=20.0).abs()
    ref = torch.nn.functional.adaptive_avg_pool3d(x, 1)
    out = ops.global_avgpool(x.to(_dev())).cpu()
    assert out.shape == ref.shape
    assert rel_err(out, ref) < 1e-6


@pytest.fixture(scope="module")
def model():
    from anomaly_detection_on_video_amd.i3d import I3Res50

    m = I3Res50(use_nl=False)
    m.load_state_dict(synth_i3d_state_dict(), strict=True)
    return m.eval().to(_dev())


def test_fullnet_vs_reference_golden(model):
    g = np.load(os.path.join(GOLDEN, "i3d_fullnet.npz"))
    for seed in (0, 1):
        x = synth_input((2, 3, 16, 224, 224), seed).to(_dev())
        y = model(x)
        assert y.shape == (2, 2048, 1, 1, 1)
        e = rel_err(y.reshape(2, 2048).cpu(), g[f"feat_seed{seed}"])
        assert e < TOL, f"seed {seed}: rel err {e:.3e}"
        assert e < 1e-4, f"fp32-exact path drifted: {e:.3e}"
    x = synth_input((1, 3, 8, 112, 96), 7).to(_dev())
    e = rel_err(model(x).reshape(1, 2048).cpu(), g["feat_small"])
    assert e < TOL and e < 1e-4


def test_fullnet_stage_taps_vs_reference_golden(model):
    g = np.load(os.path.join(GOLDEN, "i3d_fullnet.npz"))
    x = synth_input((2, 3, 16, 224, 224), 0).to(_dev())
    taps = {}
    model.forward_single(x, taps)
    seen = 0
    for name, v in taps.items():
        key = f"stat_{name}"
        if key not in g.files:
            continue
        ref = g[key]
        v = v.cpu()
        flat = v.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, 64).long()
        assert rel_err(flat[idx], ref[3:]) < TOL, name
        assert abs(v.mean().item() - ref[0]) <= 1e-3 * max(1.0, abs(ref[0])), name
        seen += 1
    assert seen >= 16


def test_bottleneck_blocks_vs_reference_golden():
    """The HIP conv chain for one Bottleneck vs the reference's own Bottleneck outputs."""
    from test_oracle_golden import BLOCK_CASES, micro_block_case
    from anomaly_detection_on_video_amd import ops

    dev = _dev()
    for name in BLOCK_CASES:
        c = micro_block_case(name)
        sd, s, tc = c["sd"], c["stride"], c["tc"]

        def pk(conv, bn, stride, pad):
            return ops.pack_conv(sd[f"{conv}.weight"].to(dev), sd[f"{bn}.weight"].to(dev), sd[f"{bn}.bias"].to(dev),
                                 sd[f"{bn}.running_mean"].to(dev), sd[f"{bn}.running_var"].to(dev), 1e-5, stride, pad)

        x = c["x"].to(dev)
        o = ops.conv3d_bn_act(x, pk("conv1", "bn1", (1, 1, 1), (tc, 0, 0)), relu=True)
        o = ops.conv3d_bn_act(o, pk("conv2", "bn2", (1, s, s), (0, 1, 1)), relu=True)
        r = ops.conv3d_bn_act(x, pk("downsample.0", "downsample.1", (1, s, s), (0, 0, 0)), relu=False) if c["has_ds"] else x
        o = ops.conv3d_bn_act(o, pk("conv3", "bn3", (1, 1, 1), (0, 0, 0)), relu=True, residual=r)
        e = rel_err(o.cpu(), c["y"])
        assert e < TOL and e < 1e-4, f"{name}: {e:.3e}"


def test_fullnet_batch_independence_and_determinism(model):
    """Size-independent properties at the benchmark's full shape: the forward is deterministic
    (bit-identical run to run -- split-K partials are reduced in a fixed order, no atomics) and a
    clip's feature does not depend on what else is in the batch beyond fp32 summation order (the
    tile / split-K choice may change with the batch size)."""
    x = synth_input((5, 3, 16, 224, 224), 3).to(_dev())
    y_all = model(x).reshape(5, 2048)
    y_again = model(x).reshape(5, 2048)
    assert torch.equal(y_all, y_again)
    y_one = model(x[3:4].contiguous()).reshape(1, 2048)
    assert rel_err(y_one.cpu(), y_all[3:4].cpu()) < 1e-5
    assert torch.isfinite(y_all).all()


def test_fullnet_split_bf16_vs_reference_golden(monkeypatch):
    """Opt-in arithmetic (ADV_ARITH=bf16x3): all 53 convs on the split-bf16 kernels.  Features agree with the
    reference's fp32 goldens to a few 1e-5 of the feature range -- inside the 1e-3 contract, far from the 2e-7 of the
    default fp32 path, which is why it is not the default."""
    from anomaly_detection_on_video_amd import ops
    from anomaly_detection_on_video_amd.i3d import I3Res50

    monkeypatch.setattr(ops, "ARITH", "bf16x3")
    m = I3Res50(use_nl=False)
    m.load_state_dict(synth_i3d_state_dict(), strict=True)
    m = m.eval().to(_dev())
    g = np.load(os.path.join(GOLDEN, "i3d_fullnet.npz"))
    x = synth_input((2, 3, 16, 224, 224), 0).to(_dev())
    y = m(x).reshape(2, 2048)
    assert all(c.choices and all(a >= 128 for a, _s in c.choices.values()) for c in m.packed_convs())
    e = rel_err(y.cpu(), g["feat_seed0"])
    assert e < 3e-4, f"split-bf16 whole net: rel err {e:.3e}"
    assert e < TOL
    assert torch.equal(y, m(x).reshape(2, 2048))


def test_fullnet_mixed_arithmetic_table():
    """ADV_ARITH=mixed overlays tuned/gfx950_mixed.json (shapes where a split-bf16 kernel measured faster in the
    stream) on the fp32 table.  Checked here without the env var: the overlay's B=32 choices applied to a fresh model
    at batch 2 keys, whole-net features within 3e-4 of the reference goldens."""
    import json
    from anomaly_detection_on_video_amd.i3d import I3Res50

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "anomaly_detection_on_video_amd", "tuned", "gfx950_mixed.json")
    overlay = {tuple(int(v) for v in k.split(",")[:11]): tuple(v) for k, v in json.load(open(path)).items()}
    assert overlay and all(a >= 128 for a, _s in overlay.values())
    m = I3Res50(use_nl=False)
    m.load_state_dict(synth_i3d_state_dict(), strict=True)
    m = m.eval().to(_dev())
    m.prepare()
    x = synth_input((2, 3, 16, 224, 224), 1).to(_dev())
    m(x)  # resolves every conv's choice for these dims
    moved = 0
    for c in m.packed_convs():
        sig = (c.cin, c.cout, *c.kernel, *c.stride, *c.padding)
        if sig in overlay:
            for key in list(c.choices):
                c.choices[key] = (overlay[sig][0], 1)
            moved += 1
    assert moved >= 10
    g = np.load(os.path.join(GOLDEN, "i3d_fullnet.npz"))
    e = rel_err(m(x).reshape(2, 2048).cpu(), g["feat_seed1"])
    assert e < 3e-4, f"mixed arithmetic whole net: rel err {e:.3e}"


def test_batch_split_over_streams(model):
    """The default forward cuts a batch of >= 16 crop-clips into two parts on two HIP streams
    (I3Res50._run_streams).  Same features as the one-stream forward (fp32 summation order may differ
    with the tile choice per part), bit-identical run to run, also on input dims whose gather tables
    did not exist before the fork (they must be built ahead of it), and for an odd batch."""
    from anomaly_detection_on_video_amd.i3d import I3Res50

    assert model._n_streams(32) == 2 and model._n_streams(16) == 2 and model._n_streams(15) == 1
    x = synth_input((17, 3, 16, 112, 112), 5).to(_dev())
    y2 = model(x).reshape(17, 2048)
    assert torch.equal(y2, model(x).reshape(17, 2048))
    keep = model.streams
    try:
        model.streams = 1
        y1 = model(x).reshape(17, 2048)
    finally:
        model.streams = keep
    assert rel_err(y2.cpu(), y1.cpu()) < 1e-5
    fresh = I3Res50(use_nl=False)
    fresh.load_state_dict(synth_i3d_state_dict(), strict=True)
    fresh = fresh.eval().to(_dev())
    xs = synth_input((16, 3, 8, 96, 80), 6).to(_dev())  # dims no other test uses: tables are built inside this call
    ya = fresh(xs).reshape(16, 2048)
    fresh.streams = 1
    yb = fresh(xs).reshape(16, 2048)
    assert rel_err(ya.cpu(), yb.cpu()) < 1e-5


def test_empty_and_ragged_batches(model):
    """Edge cases of the driver loop: an empty batch, and a ragged last batch (fewer clips than the
    tuned batch size) -- extract_features' DataLoader yields both kinds."""
    y = model(torch.empty((0, 3, 16, 224, 224), device=_dev()))
    assert y.shape == (0, 2048, 1, 1, 1)
    x = synth_input((3, 3, 16, 112, 112), 11).to(_dev())
    y3 = model(x).reshape(3, 2048)
    y1 = torch.cat([model(x[i : i + 1].contiguous()).reshape(1, 2048) for i in range(3)])
    assert rel_err(y3.cpu(), y1.cpu()) < 1e-5


def test_product_path_refuses_cpu_and_train_mode():
    from anomaly_detection_on_video_amd import _lib
    from anomaly_detection_on_video_amd.i3d import I3Res50

    m = I3Res50().eval()
    with pytest.raises(_lib.HipExtensionError):
        m(torch.zeros(1, 3, 16, 32, 32))  # CPU tensors: no fallback
    m = m.to(_dev())
    m.train()
    with pytest.raises(_lib.HipExtensionError):
        m(torch.zeros(1, 3, 16, 32, 32, device=_dev()))
