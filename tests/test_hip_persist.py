"""The persistent, wave-specialised kernels (ADVHIP_ALGO_PERSIST_BASE, csrc/conv_igemm.hip: conv1x1_persist_kernel; opt-in, see
profiles/r04_persist_kernel_study.md) -- eight-wave workgroups that stay for the whole launch and walk a share of the output tiles,
four waves multiplying, four fetching operands across tile boundaries and finishing the previous tile -- for the 1x1x1 stride-1
convs of the Bottlenecks (/root/reference/src/i3d.py:85-89, 108-121).  Run with -m gpu on an MI355X.

  * against the CPU oracle (conv + eval BN + residual + ReLU) on the small shapes every other family is checked on;
  * BIT-IDENTICAL to the one-tile-per-workgroup LDS-DMA kernel (same operands, k order and accumulation chain) on shapes
    large enough that every workgroup walks several tiles (ring hand-over between tiles, epilogue beside prefetch), for every
    tile / workgroups-per-CU id, with and without a residual, on channel slices, and on rows that are not a multiple of
    4 positions long (2 x 7 x 7 = 98: the virtually padded M index space);
  * what the family does not take (k > 1, strides, split-K) is an error, not a silent other kernel.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from anomaly_detection_on_video_amd.weights import synth_tensor

pytestmark = pytest.mark.gpu

TIGHT = 2e-5


def _dev():
    return torch.device("cuda:0")


def _ids():
    from anomaly_detection_on_video_amd import _lib

    return list(_lib.PERSIST_ALGOS)


def _pack(name, cin, cout, k=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0)):
    from anomaly_detection_on_video_amd import ops

    dev = _dev()
    fan = cin * k[0] * k[1] * k[2]
    wt = synth_tensor(f"pz.{name}.w", (cout, cin) + tuple(k), scale=float(np.sqrt(6.0 / fan)))
    g = synth_tensor(f"pz.{name}.g", (cout,), scale=0.5, offset=1.0)
    be = synth_tensor(f"pz.{name}.b", (cout,), scale=0.25)
    mu = synth_tensor(f"pz.{name}.m", (cout,), scale=0.25)
    var = synth_tensor(f"pz.{name}.v", (cout,), scale=0.5, offset=1.0)
    return ops.pack_conv(wt.to(dev), g.to(dev), be.to(dev), mu.to(dev), var.to(dev), 1e-5, s, p, name=name), (wt, g, be, mu, var)


SMALL = [("l1.conv3", 64, 256, (2, 4, 13, 11)), ("l2.conv3", 128, 512, (2, 2, 6, 6)), ("l3.conv3", 256, 1024, (3, 2, 3, 3)),
         ("l4.conv3", 512, 2048, (2, 2, 2, 2)), ("l3.conv1.k1", 1024, 256, (2, 2, 3, 3)), ("edge.1pos", 64, 64, (1, 1, 1, 1)),
         ("l4.98", 512, 2048, (3, 2, 7, 7))]


@pytest.mark.parametrize("case", SMALL, ids=[c[0] for c in SMALL])
def test_persistent_conv_vs_oracle(case):
    from anomaly_detection_on_video_amd import ops
    from oracle import i3d_oracle

    name, cin, cout, bthw = case
    dev = _dev()
    pc, (wt, g, be, mu, var) = _pack(name, cin, cout)
    x = synth_tensor(f"pz.{name}.x", (bthw[0], cin) + tuple(bthw[1:]), scale=2.0)
    y0 = i3d_oracle.conv_bn_act(x, wt, g, be, mu, var, (1, 1, 1), (0, 0, 0), None, relu=False)
    res = synth_tensor(f"pz.{name}.r", tuple(y0.shape), scale=1.0)
    for use_res, relu in ((False, True), (True, True), (True, False)):
        ref = i3d_oracle.conv_bn_act(x, wt, g, be, mu, var, (1, 1, 1), (0, 0, 0), res if use_res else None, relu)
        for algo in _ids():
            out = ops.conv3d_bn_act(x.to(dev), pc, relu=relu, residual=res.to(dev) if use_res else None, algo=algo)
            e = rel_err(out.cpu(), ref)
            assert e < TIGHT, f"{name} algo={algo} res={use_res} relu={relu}: rel err {e:.3e}"


# shapes on which a workgroup's share is several tiles (256 CUs x 1..4 workgroups): (name, Cin, Cout, (B, T, H, W))
LARGE = [("l1", 64, 256, (4, 4, 55, 55)), ("l2", 128, 512, (16, 2, 28, 28)), ("l3", 256, 1024, (32, 2, 14, 14)), ("l4", 512, 2048, (32, 2, 7, 7)),
         ("l1.ds", 128, 256, (4, 4, 55, 55)), ("ragged", 64, 128, (3, 1, 37, 28))]


@pytest.mark.parametrize("case", LARGE, ids=[c[0] for c in LARGE])
def test_persistent_conv_is_bit_identical_to_the_one_tile_kernels(case):
    from anomaly_detection_on_video_amd import _lib, ops

    name, cin, cout, bthw = case
    dev = _dev()
    pc, _ = _pack("big." + name, cin, cout)
    x = synth_tensor(f"pz.big.{name}.x", (bthw[0], cin) + tuple(bthw[1:]), scale=2.0).to(dev)
    ref_plain = ops.conv3d_bn_act(x, pc, relu=True, algo=_lib.ALGO_DMA2_BASE + _lib.ALGO_IGEMM_64x64, splits=1)
    res = synth_tensor(f"pz.big.{name}.r", tuple(ref_plain.shape), scale=1.0).to(dev)
    ref_res = ops.conv3d_bn_act(x, pc, relu=True, residual=res, algo=_lib.ALGO_DMA2_BASE + _lib.ALGO_IGEMM_128x64, splits=1)
    ref_fast = ops.conv3d_bn_act(x, pc, relu=True, residual=res, algo=_lib.ALGO_FAST_BASE + _lib.ALGO_IGEMM_64x64, splits=1)  # register-staged family
    assert torch.equal(ref_res, ref_fast)
    for algo in _ids():
        a = ops.conv3d_bn_act(x, pc, relu=True, algo=algo)
        b = ops.conv3d_bn_act(x, pc, relu=True, residual=res, algo=algo)
        assert torch.equal(a, ref_plain), f"{name} algo={algo}: max diff {float((a - ref_plain).abs().max()):.3e}"
        assert torch.equal(b, ref_res), f"{name} algo={algo} (+res): max diff {float((b - ref_res).abs().max()):.3e}"
    # launch after launch on three streams at once, other work in between: every result still the same bits
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    outs = []
    torch.cuda.synchronize()
    for rep in range(4):
        for st, algo in zip(streams, _ids()[rep::2]):
            with torch.cuda.stream(st):
                st.wait_stream(torch.cuda.default_stream(dev))
                ops.conv3d_bn_act(x, pc, relu=False, algo=_lib.ALGO_DMA_BASE + _lib.ALGO_IGEMM_64x64, splits=1)
                outs.append(ops.conv3d_bn_act(x, pc, relu=True, residual=res, algo=algo))
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, ref_res)


def test_persistent_conv_on_channel_slices():
    """x and y as channel slices of wider buffers (layer1.0's conv3 + downsample reads the [x ; h] buffer)."""
    from anomaly_detection_on_video_amd import _lib, ops

    dev = _dev()
    pc, _ = _pack("slice", 128, 256)
    b, t, h, w = 4, 4, 28, 28
    wide_in = torch.full((b, 128 + 64, t, h, w), 7.0, device=dev)
    x = synth_tensor("pz.slice.x", (b, 128, t, h, w), scale=2.0).to(dev)
    wide_in[:, 32:160] = x
    dense = ops.conv3d_bn_act(x, pc, relu=True, algo=_lib.ALGO_DMA2_BASE + _lib.ALGO_IGEMM_128x64, splits=1)
    for algo in _ids():
        wide_out = torch.full((b, 256 + 64, t, h, w), -3.0, device=dev)
        got = ops.conv3d_bn_act(wide_in[:, 32:160], pc, relu=True, out=wide_out[:, 64:], algo=algo)
        assert torch.equal(got, dense)
        assert bool((wide_out[:, :64] == -3.0).all())


def test_persistent_family_rejects_what_it_does_not_take():
    from anomaly_detection_on_video_amd import _lib, ops

    dev = _dev()
    algo = _lib.ALGO_PERSIST_BASE + _lib.ALGO_IGEMM_128x64
    pc3, _ = _pack("rej.k3", 64, 64, k=(3, 1, 1), p=(1, 0, 0))
    x = synth_tensor("pz.rej.x", (2, 64, 4, 6, 6)).to(dev)
    with pytest.raises(_lib.HipExtensionError, match="persistent"):
        ops.conv3d_bn_act(x, pc3, algo=algo)
    pcs, _ = _pack("rej.s2", 64, 64, s=(1, 2, 2))
    with pytest.raises(_lib.HipExtensionError, match="persistent"):
        ops.conv3d_bn_act(x, pcs, algo=algo)
    pc1, _ = _pack("rej.ok", 64, 64)
    with pytest.raises(_lib.HipExtensionError):
        ops.conv3d_bn_act(x, pc1, algo=algo, splits=2)
    pc32, _ = _pack("rej.k32", 32, 64)  # K = 32: two k-tiles per tile, the hand-over schedule needs four
    with pytest.raises(_lib.HipExtensionError, match="K >= 64"):
        ops.conv3d_bn_act(synth_tensor("pz.rej.x32", (2, 32, 4, 6, 6)).to(dev), pc32, algo=algo)
    for bad in (_lib.ALGO_PERSIST_BASE, _lib.ALGO_PERSIST_BASE + 1, _lib.ALGO_PERSIST_BASE + 4, _lib.ALGO_PERSIST_BASE + 5, _lib.ALGO_PERSIST_BASE + 24 + 2,
                _lib.ALGO_PERSIST_BASE + 32 + 2):
        with pytest.raises(_lib.HipExtensionError, match="not instantiated"):
            ops.conv3d_bn_act(x, pc1, algo=bad)
    y = ops.conv3d_bn_act(x, pc1, algo=algo)
    assert torch.isfinite(y).all()
