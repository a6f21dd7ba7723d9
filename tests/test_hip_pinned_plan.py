"""GPU parity of the BENCHMARKED configuration (run with -m gpu on an MI355X).

bench.py times B=32 crop-clips of 16x224x224 per step on the kernel choices of tuned/gfx950.json, three steps in
flight on three HIP stream lanes.  The other whole-net tests run batches whose keys are not in that table (they
exercise the library heuristic), so this file runs the table's own batch sizes (8, 16, 32, 40) end to end:

  * input = the golden 2-clip input (synth_input((2,3,16,224,224), seed), tests/golden/i3d_fullnet.npz holds the
    reference's own I3Res50 output for it) tiled along the batch, so every output row has a reference-made answer;
  * every conv's resolved (algo, splits) must equal its tuned-table entry -- the tuned plan ran, not the heuristic;
  * element-wise tolerance |a-b| <= 1e-3*|b| + 1e-3*rms(b) on top of the norm-wise 1e-4.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO, assert_close_elementwise, rel_err
from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_input, synth_module_state_dict

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _fresh_model():
    from anomaly_detection_on_video_amd.i3d import I3Res50

    m = I3Res50(use_nl=False)
    m.load_state_dict(synth_i3d_state_dict(), strict=True)
    return m.eval().to(_dev())


@pytest.fixture(scope="module")
def model():
    return _fresh_model()


@pytest.fixture(scope="module")
def scorer():
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection

    s = MGFNForVideoAnomalyDetection(MGFNConfig())
    s.load_state_dict(synth_module_state_dict(s))
    return s.eval().to(_dev())


@pytest.fixture(autouse=True)
def _split_k_counters_left_zero():
    """Every launch must leave the self-resetting arrival counters of its stream zero (include/advhip.h,
    advhip_conv3d_epilogue.splitk_counters): a word left non-zero -- an unfinished reduction, a tile index past the block --
    would make the next split-K launch on that stream miss its last arriver.  Checked after every test of this file, on
    every (device, stream) block the process has handed out."""
    yield
    from anomaly_detection_on_video_amd import ops

    torch.cuda.synchronize()
    for key, cnt in ops._SPLITK_COUNTERS.items():
        assert int(cnt.abs().sum()) == 0, f"arrival counters of stream {key} left non-zero"
    for key, ws in ops._ZERO_WORKSPACES.items():
        assert int(ws[:16384].view(torch.int32).abs().sum()) == 0, f"NT-GEMM arrival counters of stream {key} left non-zero"


def _tiled_input(batch: int, seed: int):
    g = np.load(os.path.join(GOLDEN, "i3d_fullnet.npz"))
    x2 = synth_input((2, 3, 16, 224, 224), seed)
    x = x2.repeat(batch // 2, 1, 1, 1, 1).to(_dev())
    ref = torch.from_numpy(g[f"feat_seed{seed}"]).repeat(batch // 2, 1)
    return x, ref


def _assert_tuned_plan(model, batch: int, table=None):
    from anomaly_detection_on_video_amd import tuned

    table = tuned.table() if table is None else table
    seen = 0
    fused_pool = {u.convs[0].name for u in model._plan if u.kind == "stem" and u.pool_unit is not None}
    fused_pool |= {u.convs[2].name for u in model._plan if u.kind == "bottleneck" and u.pool_unit is not None}
    assert fused_pool == ({"conv1", "layer1.2.conv3", "layer4.2.conv3"} if model.fuse_pool else set())
    for pc in model.packed_convs():
        if pc.name in fused_pool:  # conv + pool launches have one tile configuration (128x64x16; bricks / one sample per tile), no table entry
            continue
        keys = [k for k in pc.choices if k[0] == batch]
        assert keys, f"{pc.name}: no choice resolved for batch {batch}"
        for k in keys:
            entry = table.get(pc.key(*k))
            assert entry is not None, f"{pc.name}: {pc.key(*k)} missing from the tuned table (heuristic would run)"
            assert tuple(pc.choices[k]) == tuple(entry), f"{pc.name}: ran {pc.choices[k]}, table says {entry}"
            seen += 1
    assert seen >= 49  # 53 convs, layer1.0's downsample folded into conv3, two convs fused with their max-pool, one with the mean


@pytest.mark.parametrize("batch,streams", [(32, 1), (32, 2), (8, 1), (16, 1), (40, 1)])
def test_tuned_plan_direct_forward_vs_reference_golden(model, batch, streams):
    """model(x) at the table's batch sizes: whole-batch launches (what a pipeline lane runs) and extract_features.py's
    direct call (batch halves on two streams: B=32 runs the table's B=16 entries)."""
    keep = model.streams
    model.streams = streams
    try:
        for seed in (0, 1) if batch == 32 else (batch // 8 % 2,):
            x, ref = _tiled_input(batch, seed)
            y = model(x).reshape(batch, 2048).cpu()
            assert rel_err(y, ref) < 1e-4
            assert_close_elementwise(y, ref, 1e-3, 1e-3)
            # rows of identical clips are identical whatever their position in the batch / stream part / m-tile
            assert torch.equal(y[0::2], y[0:1].expand(batch // 2, -1)) and torch.equal(y[1::2], y[1:2].expand(batch // 2, -1))
        parts = model._n_streams(batch)
        assert parts == streams
        _assert_tuned_plan(model, batch // parts)
    finally:
        model.streams = keep


def test_tuned_plan_three_lane_stream_vs_reference_golden(scorer):
    """bench.py's own loop: ExtractScoreStream.step_async x6 on 3 lanes, whole-batch B=32 launches per lane."""
    from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream

    model = _fresh_model()  # lazily built operands are created inside the stream (first-use ordering is under test)
    stream = ExtractScoreStream(model, scorer, clips_per_video=32, ncrops=10, local_batch=32)
    assert stream.lanes == 3
    x0, ref0 = _tiled_input(32, 0)
    x1, ref1 = _tiled_input(32, 1)
    handles = [stream.step_async(x0 if i % 2 == 0 else x1) for i in range(6)]
    stream.drain()
    outs = [h.result()[0].cpu() for h in handles]
    torch.cuda.synchronize()
    for i, y in enumerate(outs):
        ref = ref0 if i % 2 == 0 else ref1
        assert rel_err(y, ref) < 1e-4, f"step {i}"
        assert_close_elementwise(y, ref, 1e-3, 1e-3)
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[4]) and torch.equal(outs[1], outs[5])
    _assert_tuned_plan(model, 32)
    assert stream.videos_scored == 0  # 6 x 32 = 192 < 320 crop-clips


def test_mixed_arithmetic_overlay_at_benchmark_batch(scorer, monkeypatch):
    """ADV_ARITH=mixed = tuned/gfx950_mixed.json laid over the fp32 table (split-bf16 kernels where they measured
    faster).  Opt-in, tolerance 3e-4.  Also pins the first-use ordering of the lazily packed bf16 weight images:
    the FIRST step on each lane must equal a later step bit for bit (lanes 1 and 2 used to be able to read
    the image before lane 0 had packed it)."""
    from anomaly_detection_on_video_amd import tuned
    from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream

    base = dict(tuned.table())
    with open(os.path.join(REPO, "anomaly_detection_on_video_amd", "tuned", "gfx950_mixed.json")) as f:
        overlay = {k: (int(v[0]), int(v[1])) for k, v in json.load(f).items()}
    assert overlay and all(a >= 128 and a < 160 for a, _ in overlay.values())
    merged = dict(base)
    merged.update(overlay)
    monkeypatch.setattr(tuned, "_TABLE", merged)
    model = _fresh_model()
    stream = ExtractScoreStream(model, scorer, clips_per_video=32, ncrops=10, local_batch=32)
    x0, ref0 = _tiled_input(32, 0)
    handles = [stream.step_async(x0) for _ in range(6)]
    stream.drain()
    outs = [h.result()[0].cpu() for h in handles]
    torch.cuda.synchronize()
    for i in range(1, 6):
        assert torch.equal(outs[0], outs[i]), f"step {i} differs from the first step on lane {i % 3}"
    assert rel_err(outs[0], ref0) < 3e-4
    assert_close_elementwise(outs[0], ref0, 3e-3, 3e-3)
    _assert_tuned_plan(model, 32, merged)
    assert sum(1 for pc in model.packed_convs() if pc.w_split is not None) >= 5
    # the direct forward (batch halves on two streams) on a second fresh model
    model2 = _fresh_model()
    ya = model2(x0).reshape(32, 2048).cpu()
    yb = model2(x0).reshape(32, 2048).cpu()
    assert torch.equal(ya, yb)
    assert rel_err(ya, ref0) < 3e-4


def test_uninstantiated_algo_ids_raise_on_device():
    """The launch switch's default used to return OK without writing y."""
    from anomaly_detection_on_video_amd import _lib, ops
    from anomaly_detection_on_video_amd.weights import synth_tensor

    dev = _dev()
    w = synth_tensor("rej.w", (128, 64, 1, 1, 1)).to(dev)
    one = torch.ones(128, device=dev)
    pc = ops.pack_conv(w, one, one * 0, one * 0, one, 1e-5, (1, 1, 1), (0, 0, 0), name="rej")
    x = synth_tensor("rej.x", (2, 64, 2, 5, 6)).to(dev)
    for algo in (69, 97, 101, 102, 103, 104, 129, 130, 131, 132, 135, 136, 165, 9, 41, 73, 105, 137, 170, 199, 201):  # (169 = the 256 x 64 tile and 200 = the mixed-tail launch exist since round 6)
        with pytest.raises(_lib.HipExtensionError, match="not instantiated"):
            ops.conv3d_bn_act(x, pc, algo=algo)
    y = ops.conv3d_bn_act(x, pc, algo=67)
    assert torch.isfinite(y).all()
    with pytest.raises(ValueError):
        ops.conv3d_bn_act(x, pc, out=torch.empty((2, 128, 2, 5, 5), device=dev))
    with pytest.raises(ValueError):
        ops.conv3d_bn_act(x, pc, out=torch.empty((2, 64, 2, 5, 6), device=dev))
    # an output (and a residual) that is only 4-byte aligned -- a view at storage offset 1: the epilogue must fall back
    # to narrower accesses instead of issuing misaligned 16-byte ones (the C entry point used to trust THW % 4 alone)
    n = y.numel()
    flat = torch.zeros(n + 8, device=dev)
    out = flat[1 : 1 + n].view(y.shape)
    assert out.data_ptr() % 16 == 4
    ops.conv3d_bn_act(x, pc, out=out, algo=67)
    assert torch.equal(out, y) and float(flat[0]) == 0 and float(flat[n + 1 :].abs().max()) == 0
    res = torch.zeros(n + 8, device=dev)[3 : 3 + n].view(y.shape)
    res.copy_(y_norelu(ops, x, pc))
    y2 = ops.conv3d_bn_act(x, pc, residual=res, relu=False, algo=163)
    torch.testing.assert_close(y2, 2 * y_norelu(ops, x, pc), rtol=1e-6, atol=1e-6)


def y_norelu(ops, x, pc):
    return ops.conv3d_bn_act(x, pc, relu=False, algo=67)


# layer-3/4 shapes of the benchmarked batch that the tuned table runs with split-K (few m-tiles, long K)
SPLITK_CASES = [
    ("l4.conv2", 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), (32, 2, 7, 7), 164, 5),
    ("l4.conv1.t3", 2048, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), (32, 2, 7, 7), 164, 5),
    ("l4.conv1.k1", 2048, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (32, 2, 7, 7), 67, 3),
    ("l3.conv2", 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (32, 2, 14, 14), 164, 3),
    ("l3.conv1.t3", 1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (32, 2, 14, 14), 164, 7),
    ("l4.conv2.s9", 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), (32, 2, 7, 7), 163, 9),
]


@pytest.mark.parametrize("case", SPLITK_CASES, ids=[c[0] for c in SPLITK_CASES])
def test_in_kernel_split_k_reduction_under_load(case):
    """The LDS-DMA kernels reduce split-K partial tiles inside the launch: every (tile, slice) workgroup publishes its
    partial tile with write-through stores and draws a ticket; the last arriver sums the slices in slice order.  Checked
    at the benchmark's own shapes (hundreds of tiles, 3-9 slices, every CU busy), on three streams at once with other
    convs in between (uneven load, reducers whose CU has touched the workspace lines before):
      * bit-identical to the two-launch form (slabs + splitk_reduce_kernel, same tile, same slices, same order) of the
        register-staged kernel family, launch after launch -- one stale or torn partial tile would show;
      * within 2e-5 of the unsplit result."""
    from anomaly_detection_on_video_amd import ops
    from anomaly_detection_on_video_amd.weights import synth_tensor

    name, cin, cout, k, s, p, bthw, algo, splits = case
    dev = _dev()
    b, t, h, w = bthw
    fan = cin * k[0] * k[1] * k[2]
    x = synth_tensor(f"sk.{name}.x", (b, cin, t, h, w), scale=2.0).to(dev)
    wt = synth_tensor(f"sk.{name}.w", (cout, cin) + tuple(k), scale=float(np.sqrt(6.0 / fan))).to(dev)
    g = synth_tensor(f"sk.{name}.g", (cout,), scale=0.5, offset=1.0).to(dev)
    be = synth_tensor(f"sk.{name}.b", (cout,), scale=0.25).to(dev)
    one, zero = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    pc = ops.pack_conv(wt, g, be, zero, one, 1e-5, s, p, name=name)
    unsplit = ops.conv3d_bn_act(x, pc, relu=True, algo=algo, splits=1)
    res = synth_tensor(f"sk.{name}.r", tuple(unsplit.shape)).to(dev)
    tile = (algo - 64) % 32 if algo < 128 else (algo - 160)
    two_launch = ops.conv3d_bn_act(x, pc, relu=True, residual=res, algo=32 + tile, splits=splits)  # fast family: slabs + reduce kernel
    ref1 = ops.conv3d_bn_act(x, pc, relu=True, residual=res, algo=algo, splits=1)
    torch.cuda.synchronize()
    assert rel_err(two_launch.cpu(), ref1.cpu()) < 2e-5
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    outs = []
    for rep in range(8):
        for st in streams:
            with torch.cuda.stream(st):
                ops.conv3d_bn_act(x, pc, relu=False, algo=67, splits=1)  # unrelated traffic between the split launches
                outs.append(ops.conv3d_bn_act(x, pc, relu=True, residual=res, algo=algo, splits=splits))
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert torch.equal(o, two_launch), f"launch {i}: max diff {float((o - two_launch).abs().max()):.3e}"


def test_split_k_counters_caller_owned_block_and_memset_form_agree():
    """Two ways to give the in-launch reduction its arrival counters (include/advhip.h, advhip_conv3d_epilogue): the caller's
    zero-initialised block that every launch leaves zero (what `ops` passes: launch after launch of different tile counts on
    one stream, no memset in between), and without one the head of the workspace, cleared by a memset per launch."""
    import ctypes as C

    from anomaly_detection_on_video_amd import _lib, ops
    from anomaly_detection_on_video_amd.weights import synth_tensor

    dev = _dev()
    lib = _lib.load()
    shapes = [(256, 256, (3, 1, 1), (1, 0, 0), (8, 4, 14, 14), 163, 4), (512, 128, (1, 1, 1), (0, 0, 0), (4, 2, 7, 7), 163, 6),
              (256, 256, (3, 1, 1), (1, 0, 0), (8, 4, 14, 14), 164, 3)]
    cnt = ops.splitk_counters(dev)
    for rep in range(2):
        for i, (cin, cout, k, p, bthw, algo, splits) in enumerate(shapes):
            b, t, h, w = bthw
            x = synth_tensor(f"skc.{i}.x", (b, cin, t, h, w), scale=2.0).to(dev)
            wt = synth_tensor(f"skc.{i}.w", (cout, cin) + k, scale=float(np.sqrt(6.0 / (cin * k[0])))).to(dev)
            one, zero = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
            pc = ops.pack_conv(wt, one, zero, zero, one, 0.0, (1, 1, 1), p, name=f"skc{i}")
            got = ops.conv3d_bn_act(x, pc, relu=True, algo=algo, splits=splits)  # caller-owned counters
            assert int(cnt.abs().sum()) == 0, "a launch left its arrival counters non-zero"
            d = pc.desc(b, t, h, w, True, algo, splits)
            need = lib.advhip_conv3d_workspace_bytes(C.byref(d))
            ws = torch.empty((need // 4 + 1,), device=dev, dtype=torch.float32).fill_(float("nan"))  # (garbage where the counters go)
            y = torch.empty_like(got)
            _lib.check(lib.advhip_conv3d_bn_act_f32(C.byref(d), _lib.ptr(x), _lib.ptr(pc.w_packed), _lib.ptr(ops.ensure_ktab(pc, (t, h, w))),
                                                    _lib.ptr(pc.scale), _lib.ptr(pc.shift), None, _lib.ptr(y), _lib.ptr(ws), need, _lib.stream()), "memset form")
            assert torch.equal(y, got)
            # a caller's block that is too small for this launch's tiles is not used: the launch falls back to the workspace
            # head + memset (never indexes past the block), and the small block stays untouched
            small = torch.full((4,), 7, device=dev, dtype=torch.int32)
            ws.fill_(float("nan"))
            y2 = torch.empty_like(got)
            ep = _lib.ConvEpilogue(None, None, None, None, None, _lib.ptr(small), small.numel() * 4)
            _lib.check(lib.advhip_conv3d_bn_act_ex_f32(C.byref(d), _lib.ptr(x), 0, _lib.ptr(pc.w_packed), _lib.ptr(ops.ensure_ktab(pc, (t, h, w))),
                                                       _lib.ptr(pc.scale), _lib.ptr(pc.shift), None, _lib.ptr(y2), 0, C.byref(ep), _lib.ptr(ws), need,
                                                       _lib.stream()), "small counter block")
            assert torch.equal(y2, got) and bool((small == 7).all())
            assert rel_err(got.cpu(), ops.conv3d_bn_act(x, pc, relu=True, algo=algo, splits=1).cpu()) < 2e-5
