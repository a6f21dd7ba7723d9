"""GEMM-shaped layers of the MGFN scorer on the hand-written fp32-MFMA kernels, forward and backward.

Activations are (C, B, T) -- channels outermost, as in models/mgfn/modeling_mgfn.py -- so a 1x1 Conv1d is the GEMM
Y[o, n] = W[o, c] X[c, n] over n = (b, t), and that GEMM is a 1x1x1 convolution of the LDS-DMA implicit-GEMM kernel on a
(1, C, 1, 1, B*T) tensor (a k = 3 Conv1d: kernel (1,1,3) on (1, C, 1, B, T), im2col-free), with bias, GELU, residual, the
channel-LayerNorm fold and the GELU backward in its epilogue (include/advhip.h: advhip_conv3d_bn_act_ex_f32).  Per layer:

    forward   Y  = act(W X + b) (+ R)            conv kernel; every layer's [c][o] operand of a differentiated forward is packed
                                                 by ONE launch (step_packs), inference re-uses cached packs
    backward  dX = W^T dY (* GELU'(z))           conv kernel on the parameter's own [o][c] layout (no packing)
              dW = dY X^T, db = row sums of dY   ONE advhip_gemm_nt launch (both operands position-contiguous: LDS-DMA row
                                                 copies; the n-tile-0 workgroups add up the dY fragments they hold anyway); the
                                                 narrow layers' pairs all together in one grouped launch at the end of the
                                                 backward pass (deferred_param_grads)

Replaces the Conv1d calls of MGFNFeedForward / FocusAttention / FocusBlock / MGFNIntermediate
(/root/reference/src/models/mgfn/modeling_mgfn.py:49-64, 150-216).  CUDA tensors on the current device only.  Shape rules
(`eligible`): channel counts >= MIN_CHANNELS_* (64: every layer of the default architecture), Cout % 64 == 0, Cin % 32 == 0, and with
autograd Cin % 64 == 0 and B*T % 16 == 0.  A layer outside these rules (a non-default architecture) takes the torch expression
of the same arithmetic in modeling_mgfn.py -- DESIGN.md section 1 lists that branch; `ADV_MGFN_STRICT=1` makes it raise instead.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Dict, Optional, Tuple

import torch

from . import _lib, ops
from ._lib import ConvDesc, ConvEpilogue, check, ptr, stream

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
# GELU whose second output is GELU'(pre-activation) instead of the pre-activation, and its consumer: the result times that tensor as is
# (include/advhip.h, advhip_conv3d_desc::relu codes 3 / 4): the fused GELU backward then has no erf / exp in its epilogue and runs on
# every tile's pipelined epilogue (the erf form: the 128 x 128 tile only)
ACT_GELU_D, ACT_MUL = 3, 4
ALGO = _lib.ALGO_DMA2_BASE + _lib.ALGO_IGEMM_128x64  # 128 x 64 x 16 tiles, 2-deep LDS-DMA ring: 139-146 TFLOP/s on these GEMMs

_KTABS: Dict[Tuple, torch.Tensor] = {}
_CONST: Dict[Tuple, torch.Tensor] = {}


# smallest channel count that goes to the HIP GEMMs: 64 = every layer of the default model (stage 0 included: at inference
# 2.49 vs 2.84 ms per video; in a training step at (32,10,32,2049) 17.60 vs 17.94 ms, round 3).  The 2048 -> 64 token conv has a
# form of its own on the same kernels (_TokenTaps: the NT GEMM and one conv launch on the input rows as stored)
MIN_CHANNELS_INFER = int(os.environ.get("ADV_MGFN_HIP_MIN_CHANNELS", "64"))
MIN_CHANNELS_TRAIN = int(os.environ.get("ADV_MGFN_HIP_MIN_CHANNELS_TRAIN", "64"))


# ADV_MGFN_STRICT=1: a GPU activation that misses the HIP kernels' shape rules raises instead of taking the torch expression of the
# same arithmetic (modeling_mgfn.py's `_pointwise_torch` / `_conv_k_torch` / einsum attention / var_mean norms -- the path of
# non-default architectures: odd channel counts, B*T % 16 != 0 with autograd, a tensor on another device than the current one)
STRICT = os.environ.get("ADV_MGFN_STRICT", "0") == "1"


def torch_path(x: torch.Tensor, what: str) -> None:
    """Called where modeling_mgfn.py is about to run a layer on torch ops."""
    if STRICT and x.is_cuda:
        raise _lib.HipExtensionError(f"ADV_MGFN_STRICT=1: {what} on {tuple(x.shape)} would run on torch ops (outside the HIP kernels' shape rules, see mgfn_ops.eligible)")


def _on_current_device(x: torch.Tensor) -> bool:
    """The kernels launch on the CURRENT device's stream: a tensor on another GPU takes the torch ops instead."""
    return x.is_cuda and x.device.index == torch.cuda.current_device()


def eligible(cin: int, cout: int, x: torch.Tensor) -> bool:
    grad = torch.is_grad_enabled()
    floor = MIN_CHANNELS_TRAIN if grad else MIN_CHANNELS_INFER
    if not (_on_current_device(x) and x.dtype == torch.float32 and min(cin, cout) >= floor and cout % 64 == 0 and cin % 32 == 0):
        return False
    if grad:  # the backward GEMMs have their own shape rules: dX = W^T dY is a conv with Cout = cin (a multiple of 64), dW = dY X^T
        #       contracts over the B*T positions in 16-wide k-tiles of 16-byte aligned rows
        npos = x.numel() // max(x.shape[0], 1)
        return cin % 64 == 0 and npos % 16 == 0
    return True


ALGO_SMALL = _lib.ALGO_DMA2_BASE + _lib.ALGO_IGEMM_64x64  # few output tiles: 64 x 64 tiles (+ split-K inside the launch)
# >= 6 tiles of 128 x 128 per CU (the FFN's 1024 -> 4096 layers at 10 240 positions: 2 560): 144 TFLOP/s against the 128 x 64
# tile's 124-141 there (tools/time_gemm_tiles.py); with fewer tiles (Cout = 1024: 640) the 128 x 64 tile's 145 wins over 122
ALGO_WIDE = _lib.ALGO_DMA2_BASE + _lib.ALGO_IGEMM_128x128


def _desc(cin: int, cout: int, k: int, b: int, t: int, act: int) -> ConvDesc:
    # (1, C, 1, b, t) tensor, kernel (1, 1, k), padding (0, 0, k // 2): for k = 1 the caller folds (b, t) into one row
    n = b * t
    tiles128 = -(-n // 128) * (cout // 64)
    algo, splits = ALGO, 1
    if cout % 128 == 0 and -(-n // 128) * (cout // 128) >= 1536:
        algo = ALGO_WIDE
    elif tiles128 < 768:  # not enough 128 x 64 tiles for 256 CUs x 3: smaller tiles, and K slices when K is long
        algo = ALGO_SMALL
        tiles = -(-n // 64) * (cout // 64)
        ktiles = -(-(cin * k) // 16)
        # (a very long K over a handful of tiles -- the token conv's weight gradient: K = all 10 240 positions, 99 tiles -- keeps gaining up
        # to ~4.5 workgroups per CU: 12 slices 76 us, 8 slices 87, tools/time_token_gemms.py)
        target, cap = (1152, 12) if ktiles >= 512 else (768, 8)
        while tiles * splits < target and splits < cap and ktiles // (splits + 1) >= 16:
            splits += 1
    return ConvDesc(1, cin, 1, b, t, cout, 1, 1, k, 1, 1, 1, 0, 0, k // 2, act, algo, splits)


def _ktab(d: ConvDesc, dev) -> torch.Tensor:
    key = (d.Cin, d.H, d.W, d.kw, dev)
    tab = _KTABS.get(key)
    if tab is None:
        lib = _lib.load()
        rows = lib.advhip_conv3d_packed_rows(C.byref(d))
        tab = torch.empty((rows * 6,), device=dev, dtype=torch.int32)
        check(lib.advhip_conv3d_build_ktab(C.byref(d), ptr(tab), stream(dev)), "build_ktab")
        _KTABS[key] = tab
    return tab


def _const(n: int, value: float, dev) -> torch.Tensor:
    key = (n, value, dev)
    if key not in _CONST:
        _CONST[key] = torch.full((n,), value, device=dev, dtype=torch.float32)
    return _CONST[key]


_PACKED: Dict[int, Tuple] = {}
_EPOCH = 0  # part of every cache stamp below


def invalidate_caches() -> None:
    """Forget every cached operand derived from parameters (packed weights, folded LayerNorm operands).  The model calls
    this at the start of every forward that will be differentiated and the runner after every optimizer step: version
    counters alone miss torch's fused optimizers and `.data` updates, and a layer may take the torch path in training but
    the HIP path (and its cache) at inference.  Call it yourself after changing parameters in any other such way."""
    global _EPOCH
    _EPOCH += 1



def pack_kc_cached(param: torch.Tensor, fresh: bool = False) -> torch.Tensor:
    """pack_kc of a parameter; inference re-uses the packed copy call after call while (data_ptr, version) stay put.
    `fresh` (a forward that will be differentiated: the weights are about to change): always re-pack and DROP the cached
    copy -- version counters cannot be relied on there (torch's fused optimizers update parameters without moving them),
    so a later no-grad forward packs again instead of finding a stale copy."""
    key = id(param)
    if fresh:
        _PACKED.pop(key, None)
        live = _STEP_PACKS.get(key)  # (packed with every other weight of this forward in one launch: step_packs)
        if live is not None and live[0]() is param:
            return live[1]
        return pack_kc(param.detach())
    hit = _PACKED.get(key)
    stamp = (param.data_ptr(), param._version, tuple(param.shape), _EPOCH)
    if hit is None or hit[0]() is not param or hit[1] != stamp:  # (ids are re-used once a tensor is gone: check it is the same object)
        if len(_PACKED) > 1024:
            for k in [k for k, v in _PACKED.items() if v[0]() is None]:  # parameters of discarded models
                del _PACKED[k]
        hit = _PACKED[key] = (weakref.ref(param), stamp, pack_kc(param.detach()))
    return hit[2]


# ---- the packed operands of one differentiated forward, all in one launch ------------------------------------------------------
# id(param) -> (weakref, forward operand, input-gradient operand or None): valid between step_packs() and end_step_packs(),
# i.e. inside ONE model forward.  The buffers are per-plan and static (a HIP graph can replay the launch); the next forward
# overwrites them with the same values unless an optimizer step lies in between -- and a backward pass across an optimizer
# step is not a thing autograd allows either.
_STEP_PACKS: Dict[int, Tuple] = {}
_PACK_PLANS: Dict[Tuple, Tuple] = {}


# advhip_pack_item (include/advhip.h), as a numpy record: the item table lives in device memory
PACK_ITEM_FIELDS = [("src", "<u8"), ("dst", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"), ("k", "<i4"), ("mode", "<i4"), ("tile_begin", "<i4"), ("reserved", "<i4")]


def step_packs(convs) -> None:
    """Pack, in ONE launch (advhip_pack_weights_multi_f32), the forward GEMM operand of every Conv1d in `convs` and the
    transposed-conv operand of the k > 1 ones; until end_step_packs() the autograd Functions below take these instead of
    packing layer by layer.  `convs`: nn.Conv1d modules (groups = 1) with contiguous fp32 weights on the current device."""
    import numpy as np

    ws = [c.weight for c in convs]
    if not ws:
        return
    dev = ws[0].device
    key = (dev,) + tuple((id(w), w.data_ptr(), tuple(w.shape)) for w in ws)
    plan = _PACK_PLANS.get(key)
    if plan is None:
        lib = _lib.load()
        item_t = np.dtype(PACK_ITEM_FIELDS)
        rows, bufs, tiles = [], {}, 0
        for w in ws:
            _lib.require_gpu(w)
            cout, cin, k = w.shape
            kc = torch.empty((-(-(cin * k) // 32) * 32, cout), device=dev, dtype=torch.float32)
            dx = torch.empty((-(-(cout * k) // 32) * 32, cin), device=dev, dtype=torch.float32) if k > 1 else None
            bufs[id(w)] = (weakref.ref(w), kc, dx)
            for mode, dst in ((0, kc), (1, dx)):
                if dst is None:
                    continue
                rows.append((w.data_ptr(), dst.data_ptr(), cout, cin, k, mode, tiles, 0))
                tiles += int(lib.advhip_pack_item_tiles(cout, cin, k, mode))
        items = torch.from_numpy(np.array(rows, dtype=item_t).view(np.uint8).copy()).to(dev)
        # evict only plans of parameters that are gone (discarded models): a live plan's buffers may be baked into a captured
        # HIP graph (train_graph.GraphedTrainStep also holds them itself, see hold_plans)
        for k in [k for k, v in _PACK_PLANS.items() if any(ref() is None for ref, _kc, _dx in v[3].values())]:
            del _PACK_PLANS[k]
        plan = _PACK_PLANS[key] = (items, len(rows), tiles, bufs)
    if _PLAN_HOLDERS:
        for held in _PLAN_HOLDERS:
            if not any(h is plan for h in held):
                held.append(plan)
    items, n, tiles, bufs = plan
    check(_lib.load().advhip_pack_weights_multi_f32(ptr(items), n, tiles, stream(ws[0])), "pack_weights_multi")
    _STEP_PACKS.clear()
    _STEP_PACKS.update(bufs)


def end_step_packs() -> None:
    _STEP_PACKS.clear()


_PLAN_HOLDERS: list = []


class hold_plans:
    """`with hold_plans() as held:` -- every pack plan step_packs() uses inside the block is appended to the list `held`
    (strong references to its item table and operand buffers).  A graph capture keeps that list for as long as the graph lives:
    replays write the packed weights into, and read GEMM operands from, exactly those buffers."""

    def __enter__(self):
        self.held: list = []
        _PLAN_HOLDERS.append(self.held)
        return self.held

    def __exit__(self, *exc):
        _PLAN_HOLDERS[:] = [h for h in _PLAN_HOLDERS if h is not self.held]
        return False


def _step_dx(weight: torch.Tensor) -> Optional[torch.Tensor]:
    live = _STEP_PACKS.get(id(weight))
    return live[2] if live is not None and live[0]() is weight else None


def pack_kc(w: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, k) Conv1d weights -> the kernels' [Cin*k (padded to 32)][Cout] operand."""
    cout, cin, k = w.shape
    d = _desc(cin, cout, k, 1, 1, 0)
    lib = _lib.load()
    wp = torch.empty((lib.advhip_conv3d_packed_rows(C.byref(d)), cout), device=w.device, dtype=torch.float32)
    _lib.require_gpu(w, contiguous=False)
    check(lib.advhip_conv3d_pack_weight_f32(C.byref(d), ptr(w.contiguous()), ptr(wp), stream(w)), "pack_weight")
    return wp


def pack_dx(w: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, k) Conv1d weights -> the packed operand [Cout*k (padded to 32)][Cin] of the transposed conv that computes
    the layer's input gradient (include/advhip.h: advhip_conv1d_pack_weight_dx_f32)."""
    cout, cin, k = w.shape
    _lib.require_gpu(w)
    wp = torch.empty((-(-(cout * k) // 32) * 32, cin), device=w.device, dtype=torch.float32)
    check(_lib.load().advhip_conv1d_pack_weight_dx_f32(ptr(w), ptr(wp), cout, cin, k, stream(w)), "pack_weight_dx")
    return wp


def conv_cn(x: torch.Tensor, w_packed: torch.Tensor, cout: int, k: int = 1, shift: Optional[torch.Tensor] = None,
            residual: Optional[torch.Tensor] = None, act: int = ACT_NONE, want_preact: bool = False,
            dact_z: Optional[torch.Tensor] = None, ln: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None):
    """act(conv1d_k(x) + shift) or conv1d_k(x) + shift + residual for x (Cin, B, T) contiguous -> (Cout, B, T) [, pre-activation].
    One launch, on torch's current stream of x's device (which must be the current device)."""
    cin, b, t = x.shape
    _lib.require_gpu(x, w_packed, shift, residual, dact_z, *(ln or ()), contiguous=False)
    if not x.is_contiguous():
        raise _lib.HipExtensionError("conv_cn needs a contiguous (C, B, T) activation")
    if act == ACT_MUL and dact_z is None:
        raise ValueError("conv_cn: ACT_MUL multiplies by the tensor passed as dact_z")
    if act != ACT_NONE and residual is not None:  # (the conv epilogue adds the residual BEFORE the activation, advhip_bgemm_f32 after)
        raise ValueError("conv_cn: an activation together with a residual is not offered (the two GEMM entry points order them differently)")
    d = _desc(cin, cout, k, 1 if k == 1 else b, b * t if k == 1 else t, act)
    dev = x.device
    y = torch.empty((cout, b, t), device=dev, dtype=torch.float32)
    z = torch.empty_like(y) if want_preact else None
    cnt = ops.splitk_counters(dev) if d.splits > 1 else None  # (self-resetting arrival counters: no memset ahead of the launch)
    ep = ConvEpilogue(ptr(z), ptr(dact_z), ptr(ln[0]) if ln else None, ptr(ln[1]) if ln else None, ptr(ln[2]) if ln else None,
                      ptr(cnt), cnt.numel() * 4 if cnt is not None else 0)
    for tns in (residual, dact_z):
        if tns is not None and (tuple(tns.shape) != (cout, b, t) or not tns.is_contiguous()):
            raise ValueError("conv_cn: residual / dact_z must be contiguous (Cout, B, T)")
    lib = _lib.load()
    need = lib.advhip_conv3d_workspace_bytes(C.byref(d)) if d.splits > 1 else 0
    if need < 0:
        check(int(need), "conv_cn workspace")
    ws = ops.workspace(dev, need)
    check(lib.advhip_conv3d_bn_act_ex_f32(C.byref(d), ptr(x), 0, ptr(w_packed), ptr(_ktab(d, dev)), ptr(_const(cout, 1.0, dev)),
                                          ptr(shift if shift is not None else _const(cout, 0.0, dev)), ptr(residual), ptr(y), 0,
                                          C.byref(ep), ptr(ws), need, stream(x)), "conv_cn")
    return (y, z) if want_preact else y


def colsum(partial: torch.Tensor) -> torch.Tensor:
    """(rows, cols) per-block partial sums -> (cols,), rows added in order (include/advhip.h: advhip_colsum_f32)."""
    rows, cols = partial.shape
    out = torch.empty((cols,), device=partial.device, dtype=torch.float32)
    check(_lib.load().advhip_colsum_f32(ptr(partial), ptr(out), rows, cols, stream(partial)), "colsum")
    return out


def colsum_group(items):
    """[(partial (rows, cols), period), ...] -> [column sums (cols,), ...] of all of them in ONE launch (include/advhip.h:
    advhip_colsum_group_f32; period > 0: de-interleaved, see there).  The results are slices of one buffer."""
    if not items:
        return []
    dev = items[0][0].device
    offs, total = [], 0
    for part, _period in items:
        _lib.require_gpu(part)
        offs.append(total)
        total += -(-part.shape[1] // 64) * 64
    res = torch.empty((total,), device=dev, dtype=torch.float32)
    arr = (_lib.ColsumItem * len(items))()
    for it, (part, period), off in zip(arr, items, offs):
        it.src, it.dst, it.rows, it.cols, it.period = part.data_ptr(), res.data_ptr() + off * 4, part.shape[0], part.shape[1], period
    check(_lib.load().advhip_colsum_group_f32(arr, len(items), stream(dev)), "colsum_group")
    return [res[off : off + part.shape[1]] for (part, _p), off in zip(items, offs)]


def chan_stats(x: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(mean, 1 / (std_biased + eps)) over the channels of a contiguous (C, B, T) activation, per position."""
    _lib.require_gpu(x)
    c = x.shape[0]
    n = x.numel() // c
    mu = torch.empty((n,), device=x.device, dtype=torch.float32)
    rs = torch.empty_like(mu)
    check(_lib.load().advhip_chan_stats_f32(ptr(x), ptr(mu), ptr(rs), c, n, C.c_float(eps), stream(x)), "chan_stats")
    return mu, rs


def _unfold3(x: torch.Tensor) -> torch.Tensor:
    """(C, B, T) -> ((c, tap) = 3C, B*T): the rows a k = 3, padding 1 conv contracts with, tap-minor like the weights."""
    c, b, t = x.shape
    if _on_current_device(x) and t % 4 == 0 and x.is_contiguous() and x.dtype == torch.float32:  # one HIP pass (pad + stack: three)
        u = torch.empty((3 * c, b * t), device=x.device, dtype=torch.float32)
        check(_lib.load().advhip_unfold3_f32(ptr(x), ptr(u), c, b, t, stream(x)), "unfold3")
        return u
    xp = torch.nn.functional.pad(x, (1, 1))
    return torch.stack([xp[:, :, j : j + t] for j in range(3)], dim=1).reshape(3 * c, b * t)


# ---- weight / bias gradients beside the dX chain -------------------------------------------------------------------------
# Inside `with overlapped_backward():` the backward of the GEMM-shaped layers issues dW = dY X^T and db = rowsum(dY) on a side
# stream, forked off the stream the dX chain runs on; leaving the context joins it (before the optimizer reads the
# gradients).  dW is a third of the step's FLOPs and none of it is on the critical path of the backward pass; its tails and
# the memory-bound reductions then run beside the next layer's dX GEMM.  Works eagerly and under graph capture (the fork /
# join become graph edges).  Outside the context everything runs on the current stream, as before.
_SIDE: Dict[int, torch.cuda.Stream] = {}
_OVERLAP = {"on": False, "used": set()}


def ensure_side_stream(dev: Optional[int] = None) -> torch.cuda.Stream:
    """The side stream of a device (created on first use -- call this BEFORE a graph capture: streams cannot be created inside one)."""
    dev = torch.cuda.current_device() if dev is None else dev
    side = _SIDE.get(dev)
    if side is None:
        side = _SIDE[dev] = torch.cuda.Stream(device=dev)
    return side


class overlapped_backward:
    def __init__(self, enabled: bool = True):
        self.enabled = enabled

    def __enter__(self):
        self.prev = _OVERLAP["on"]
        _OVERLAP["on"] = bool(self.enabled) and torch.cuda.is_available()
        if _OVERLAP["on"]:
            ensure_side_stream()
        return self

    def __exit__(self, *exc):
        _OVERLAP["on"] = self.prev
        for dev in list(_OVERLAP["used"]):  # join: whatever consumes the gradients next is ordered after the side stream
            torch.cuda.current_stream(dev).wait_stream(_SIDE[dev])
        _OVERLAP["used"].clear()
        return False


class _Fork:
    """`with _Fork(dy, x, ...):` -- the body runs on the side stream after everything issued so far on the current stream;
    the named tensors (inputs produced on the current stream) are kept alive for it."""

    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if t is not None]
        self.ctx = None

    def __enter__(self):
        if not _OVERLAP["on"] or not self.tensors:
            return self
        dev = self.tensors[0].device.index
        side = ensure_side_stream(dev)
        self.main = torch.cuda.current_stream(dev)
        side.wait_stream(self.main)
        for t in self.tensors:
            t.record_stream(side)
        _OVERLAP["used"].add(dev)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def out(self, *tensors):
        """Results produced on the side stream and consumed (after the join) on the main one."""
        if self.ctx is not None:
            for t in tensors:
                if t is not None:
                    t.record_stream(self.main)

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


# ---- weight / bias gradients of the narrow layers, all in one launch at the end of the backward pass ----------------------------
# dW = dY X^T of a 64- or 128-channel layer is a handful of 64 x 64 tiles over K = all positions: ~12 us of latency chain, a reduce
# launch and two launch gaps each, 31 layers per step (~1 ms of a 16-ms step).  Inside `deferred_param_grads()` the backward
# functions below do not compute them: they return None for those parameters, keep (dY, X) alive, and a callback queued on the
# autograd engine runs ONE grouped launch + ONE reduce when the backward pass is over and puts the results where autograd's
# AccumulateGrad would have (param.grad = g, or += g).  Same kernels, slices and summation order as the per-layer launches: the
# same bits.  Opt-in because only `.backward()` sees these gradients -- `torch.autograd.grad` and parameter hooks do not; the
# trainer's step (train_graph.GraphedTrainStep, runner.Trainer) turns it on.
_DEFER = {"on": False, "items": [], "colsums": [], "queued": False}
DEFER_DW = os.environ.get("ADV_MGFN_DEFER_DW", "1") == "1"  # what the trainer's step passes to deferred_param_grads


class deferred_param_grads:
    def __init__(self, enabled: bool = True):
        self.enabled = enabled

    def __enter__(self):
        self.prev = _DEFER["on"]
        _DEFER["on"] = bool(self.enabled)
        return self

    def __exit__(self, *exc):
        _DEFER["on"] = self.prev
        if _DEFER["items"] or _DEFER["colsums"]:  # (a backward pass that never finished: drop what it left)
            _DEFER["items"].clear()
            _DEFER["colsums"].clear()
            _DEFER["queued"] = False
        return False


def _accumulate(p, g) -> None:
    """What autograd's AccumulateGrad does with a leaf's gradient."""
    if p.grad is None:
        p.grad = g
    else:
        p.grad.add_(g)


def _flush_deferred() -> None:
    items, _DEFER["items"] = _DEFER["items"], []
    colsums, _DEFER["colsums"] = _DEFER["colsums"], []
    _DEFER["queued"] = False
    with torch.no_grad():
        by_dev = {}
        for it in colsums:
            by_dev.setdefault(it[0].device, []).append(it)
        for dev, group in by_dev.items():
            with torch.cuda.device(dev):
                sums = colsum_group([(part, period) for part, period, _t in group])
            for (_part, _period, targets), s in zip(group, sums):
                for p, lo, hi in targets:
                    _accumulate(p, s[lo:hi].view(p.shape))
    if not items:
        return
    by_k = {}
    for it in items:  # (one launch per distinct K and device: one, in an MGFN step)
        by_k.setdefault((it[0].shape[1], it[0].device), []).append(it)
    with torch.no_grad():
        for group in by_k.values():
            with torch.cuda.device(group[0][0].device):
                outs = ops.gemm_nt_group([(g2, x2, bp is not None) for g2, x2, _wp, bp in group])
            for (g2, x2, wp, bp), (dw, db) in zip(group, outs):
                _accumulate(wp, dw.view(wp.shape))
                if bp is not None:
                    _accumulate(bp, db)


def _queue_flush() -> None:
    if not _DEFER["queued"]:
        _DEFER["queued"] = True
        torch.autograd.Variable._execution_engine.queue_callback(_flush_deferred)


def _defer_colsum(partial: torch.Tensor, targets, period: int = 0) -> bool:
    """Queue the column sums of a (rows, cols) partial-sum matrix; `targets` = [(parameter, lo, hi), ...]: columns [lo, hi) of the
    (de-interleaved, see advhip_colsum_group_f32) sums are that parameter's gradient.  False = the caller sums now."""
    if not (_DEFER["on"] and targets and all(p is not None for p, _lo, _hi in targets)):
        return False
    _DEFER["colsums"].append((partial, period, targets))
    _queue_flush()
    return True


def _defer_dw(g2: torch.Tensor, x2: torch.Tensor, wparam, bparam) -> bool:
    """Queue dW = g2 x2^T (and db = rowsum(g2) when `bparam` is given) for the grouped launch; False = compute it now."""
    if not (_DEFER["on"] and wparam is not None and ops.gemm_nt_is_small(g2.shape[0], x2.shape[0]) and g2.shape[1] % 256 == 0
            and g2.is_contiguous() and x2.is_contiguous()):
        return False
    _DEFER["items"].append((g2, x2, wparam, bparam))
    _queue_flush()
    return True


# k = 3 layers as plain GEMMs on unfolded operands (K3_AS_GEMM): the im2col-free k = 3 conv gathers 4-byte pieces (its taps shift
# positions inside a row of T, with zero padding at the row ends, so 16-byte pieces are out) and runs 110-116 TFLOP/s; the
# same contraction as a 1x1 GEMM over U = unfold3(x) (3 Cin rows, one 126-MB pass at stage 2) takes the 16-byte operand path at
# ~140 TFLOP/s, and U is what the weight gradient contracts with anyway (kept for the backward pass instead of x).
K3_AS_GEMM = os.environ.get("ADV_MGFN_K3_GEMM", "1") == "1"


# the input gradient of a NARROW k = 3 layer (<= this many channels: stages 0 and 1) straight from the im2col-free transposed conv:
# such a launch is bound by its latency chain, not by the gather's issue rate, so the unfold3(dY) pass in front of the GEMM form
# is one launch (~5.5 us each, six per step) for nothing
NARROW_DX_DIRECT = int(os.environ.get("ADV_MGFN_NARROW_DX_DIRECT", "128"))


class _LinearCN(torch.autograd.Function):
    """y = conv1d_k(x; W) + b (+ residual), k in {1, 3}, no activation."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, fresh):
        cout, cin, k = weight.shape
        shift = bias.detach() if bias is not None else None
        res = residual.detach().contiguous() if residual is not None else None
        unfolded = K3_AS_GEMM and k == 3 and x.shape[2] % 4 == 0 and any(ctx.needs_input_grad)
        if unfolded:  # W viewed (Cout, 3 Cin) times U[(c*3 + j), n] = x[c, n + j - 1]: the packed operand is the k = 3 one
            u = _unfold3(x).view(3 * cin, x.shape[1], x.shape[2])
            y = conv_cn(u, pack_kc_cached(weight, fresh), cout, 1, shift=shift, residual=res)
            ctx.save_for_backward(u, weight)
        else:
            y = conv_cn(x, pack_kc_cached(weight, fresh), cout, k, shift=shift, residual=res)
            ctx.save_for_backward(x, weight)
        ctx.unfolded = unfolded
        ctx.params = (weight, bias)  # (the parameters themselves: deferred_param_grads writes their .grad at the end of the pass)
        ctx.dx_pack = _step_dx(weight) if k > 1 and fresh else None  # (packed with this forward's other operands: step_packs)
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        # y = conv(x) + x (the blocks' `x = scc(x) + x`): dL/dx = conv^T(dy) + dy comes out of ONE launch (dy as the dX GEMM's
        # residual) instead of a GEMM, a pass-through and autograd's add over the whole activation
        ctx.res_is_x = residual is not None and residual.data_ptr() == x.data_ptr() and residual.shape == x.shape and cin == cout
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors  # (x is U = unfold3(x) when ctx.unfolded)
        cout, cin, k = weight.shape
        dy = dy.contiguous()
        dx = dw = db = None
        fold = ctx.res_is_x and ctx.needs_input_grad[0] and ctx.needs_input_grad[3]
        if ctx.needs_input_grad[0]:
            if k == 1:  # dX = W^T dY: the parameter's own (o, c) layout IS the kernels' [K = o][Cout = c] operand
                dx = conv_cn(dy, weight.detach().view(cout, cin), cin, 1, residual=dy if fold else None)
            elif ctx.unfolded and min(cin, cout) > NARROW_DX_DIRECT:  # transposed conv as a GEMM over unfold3(dY): rows (o*3 + j') = dY[o, n + j' - 1] against pack_dx's
                ud = _unfold3(dy).view(3 * cout, dy.shape[1], dy.shape[2])  # [(o*3 + j')][c] = W[o][c][2 - j']
                wdx = ctx.dx_pack if ctx.dx_pack is not None else pack_dx(weight.detach())
                dx = conv_cn(ud, wdx, cin, 1, residual=dy if fold else None)
            else:       # transposed conv: W'[c][o][j] = W[o][c][k-1-j], packed straight from the parameter (one launch)
                wdx = ctx.dx_pack if ctx.dx_pack is not None else pack_dx(weight.detach())
                dx = conv_cn(dy, wdx, cin, k, residual=dy if fold else None)
        with _Fork(dy, x) as fk:
            want_db = ctx.has_bias and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:  # dW = dY X^T, and db = rowsum(dY) out of the same launch
                if k == 1:
                    xk = x.detach().view(cin, -1)
                else:
                    xk = x.view(3 * cin, -1) if ctx.unfolded else _unfold3(x.detach())
                if _defer_dw(dy.view(cout, -1), xk, ctx.params[0], ctx.params[1] if want_db else None):
                    dw = db = None
                else:
                    res = ops.gemm_nt(dy.view(cout, -1), xk, rowsum=want_db)
                    dw, db = res if want_db else (res, None)
                    dw = dw.view(cout, cin, k)
            elif want_db:
                db = dy.sum(dim=(1, 2))
            fk.out(dw, db)
        return dx, dw, db, (None if fold else (dy if ctx.has_res and ctx.needs_input_grad[3] else None)), None


class _FFNCN(torch.autograd.Function):
    """y = W2 GELU(W1 xh + b1) + b2 + x_res: MGFNFeedForward's convs (modeling_mgfn.py:53-64) and the block's residual add
    as two launches forward (GELU and its pre-activation in the first one's epilogue, bias + residual in the second's) and
    two + two backward (GELU' in the epilogue of the GEMM that produces dL/dh)."""

    @staticmethod
    def forward(ctx, xh, x_res, w1, b1, w2, b2, fresh):
        hid, dim = w1.shape[0], w1.shape[1]
        need_z = any(ctx.needs_input_grad)
        # (with a backward pass to come: z holds GELU'(pre-activation), what that pass multiplies by)
        out = conv_cn(xh, pack_kc_cached(w1, fresh), hid, 1, shift=b1.detach(), act=ACT_GELU_D if need_z else ACT_GELU, want_preact=need_z)
        h, z = out if need_z else (out, None)
        y = conv_cn(h, pack_kc_cached(w2, fresh), dim, 1, shift=b2.detach(), residual=x_res.detach().contiguous())
        if need_z:
            ctx.save_for_backward(xh, z, h, w1, w2)
        ctx.params = (w1, b1, w2, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        xh, z, h, w1, w2 = ctx.saved_tensors
        hid, dim = w1.shape[0], w1.shape[1]
        dy = dy.contiguous()
        w1p, b1p, w2p, b2p = ctx.params
        with _Fork(dy, h) as fk:
            dw2, db2 = _dw_db(dy, h, ctx.needs_input_grad[4], ctx.needs_input_grad[5], w2p, b2p)
            fk.out(dw2, db2)
        dz = conv_cn(dy, w2.detach().view(dim, hid), hid, 1, dact_z=z, act=ACT_MUL)  # (W2^T dY) * GELU'(pre-activation), saved by the forward
        with _Fork(dz, xh) as fk:
            dw1, db1 = _dw_db(dz, xh.detach(), ctx.needs_input_grad[2], ctx.needs_input_grad[3], w1p, b1p)
            fk.out(dw1, db1)
        dxh = conv_cn(dz, w1.detach().view(hid, dim), dim, 1) if ctx.needs_input_grad[0] else None
        return dxh, (dy if ctx.needs_input_grad[1] else None), dw1, db1, dw2, db2, None


def _dw_db(g: torch.Tensor, act: torch.Tensor, want_w: bool, want_b: bool, wparam=None, bparam=None):
    """dW = g act^T over the positions (1x1 layer), and db = rowsum(g) out of the same launch -- or (None, None) when the pair was
    queued for the grouped launch at the end of the backward pass (deferred_param_grads; `wparam` / `bparam`: the parameters)."""
    if want_w:
        if _defer_dw(g.view(g.shape[0], -1), act.view(act.shape[0], -1), wparam, bparam if want_b else None):
            return None, None
        res = ops.gemm_nt(g.view(g.shape[0], -1), act.view(act.shape[0], -1), rowsum=want_b)
        w, b = res if want_b else (res, None)
        return w.view(g.shape[0], act.shape[0], 1), b
    return None, (g.sum(dim=(1, 2)) if want_b else None)


class _FFNBlockCN(torch.autograd.Function):
    """y = x + W2 GELU(W1 LN(x) + b1) + b2: a whole `x = ffn(x) + x` step of a block (MGFNFeedForward + the residual add,
    modeling_mgfn.py:49-64, 147, 205) as three launches forward (LayerNorm; GEMM + GELU; GEMM + bias + residual) and, backward,
    two GEMM pairs (dW + db in one launch each, dX with GELU' in its epilogue) and ONE LayerNorm-backward launch that also adds
    the skip connection's gradient and writes dg / db partial sums as one matrix."""

    @staticmethod
    def forward(ctx, x, g, b, eps, w1, b1, w2, b2, fresh):
        _lib.require_gpu(x, g, b, w1, b1, w2, b2, contiguous=False)
        c = x.shape[0]
        n = x.numel() // c
        hid = w1.shape[0]
        xh = torch.empty_like(x)
        mu = torch.empty((n,), device=x.device, dtype=torch.float32)
        rs = torch.empty_like(mu)
        gf, bf = g.detach().reshape(c).contiguous(), b.detach().reshape(c).contiguous()
        ctx.fused = fused_ffn_ok(c, hid, n, w1, w2)
        if ctx.fused:
            # the narrow blocks (64 / 128 channels): LayerNorm, both GEMMs, GELU and the residual add as ONE launch (csrc/ffn_fused.hip);
            # the packed operands are what the backward launch multiplies by (valid until the next forward, like every step pack)
            w1p, w2p = pack_kc_cached(w1, fresh), pack_kc_cached(w2, fresh)
            h = torch.empty((hid,) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
            z, y = torch.empty_like(h), torch.empty_like(x)
            check(_lib.load().advhip_ffn_block_fwd_f32(ptr(x), ptr(gf), ptr(bf), C.c_float(eps), ptr(w1.detach()), ptr(b1.detach()), ptr(w2.detach()),
                                                       ptr(b2.detach()), ptr(xh), ptr(mu), ptr(rs), ptr(h), ptr(z), ptr(y), c, n, stream(x)), "ffn_block_fwd")
            ctx.save_for_backward(x, gf, mu, rs, xh, z, h, w1p, w2p)
            ctx.params = (w1, b1, w2, b2)
            ctx.ln_params = (g, b)
            ctx.eps, ctx.gshape = eps, g.shape
            return y
        check(_lib.load().advhip_chan_layernorm_fwd_f32(ptr(x), ptr(gf), ptr(bf), ptr(xh), ptr(mu), ptr(rs), c, n, C.c_float(eps), stream(x)),
              "chan_layernorm_fwd")
        h, z = conv_cn(xh, pack_kc_cached(w1, fresh), hid, 1, shift=b1.detach(), act=ACT_GELU_D, want_preact=True)  # z = GELU'(pre-activation)
        y = conv_cn(h, pack_kc_cached(w2, fresh), c, 1, shift=b2.detach(), residual=x)
        ctx.save_for_backward(x, gf, mu, rs, xh, z, h, w1, w2)
        ctx.params = (w1, b1, w2, b2)
        ctx.ln_params = (g, b)
        ctx.eps, ctx.gshape = eps, g.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gf, mu, rs, xh, z, h, w1, w2 = ctx.saved_tensors
        hid, dim = z.shape[0], x.shape[0]
        n = x.numel() // dim
        need = ctx.needs_input_grad
        dy = dy.contiguous()
        w1p, b1p, w2p, b2p = ctx.params
        if ctx.fused:  # (w1 / w2 of the saved tensors are the PACKED operands here: [C][4C] and [4C][C])
            lib = _lib.load()
            rows = lib.advhip_ffn_block_partial_rows(n)
            dz, dx = torch.empty_like(z), torch.empty_like(x)
            pgb = torch.empty((rows, 2 * dim), device=x.device, dtype=torch.float32)
            check(lib.advhip_ffn_block_bwd_f32(ptr(dy), ptr(x), ptr(gf), ptr(mu), ptr(rs), C.c_float(ctx.eps), ptr(z), ptr(w2), ptr(w1), ptr(dz), ptr(dx),
                                               ptr(pgb), dim, n, stream(x)), "ffn_block_bwd")
            dw2, db2 = _dw_db(dy, h, need[6], need[7], w2p, b2p)
            dw1, db1 = _dw_db(dz, xh, need[4], need[5], w1p, b1p)
            gp, bp = ctx.ln_params
            if need[1] and need[2] and _defer_colsum(pgb, [(gp, 0, dim), (bp, dim, 2 * dim)]):
                return dx, None, None, None, dw1, db1, dw2, db2, None
            sums = colsum(pgb)
            return dx, sums[:dim].reshape(ctx.gshape), sums[dim:].reshape(ctx.gshape), None, dw1, db1, dw2, db2, None
        dw2, db2 = _dw_db(dy, h, need[6], need[7], w2p, b2p)
        dz = conv_cn(dy, w2.detach().view(dim, hid), hid, 1, dact_z=z, act=ACT_MUL)  # (W2^T dY) * GELU'(pre-activation), saved by the forward
        dw1, db1 = _dw_db(dz, xh, need[4], need[5], w1p, b1p)
        dxh = conv_cn(dz, w1.detach().view(hid, dim), dim, 1)
        lib = _lib.load()
        rows = lib.advhip_chan_layernorm_bwd_partial_rows(n)
        dx = torch.empty_like(x)
        pgb = torch.empty((rows, 2 * dim), device=x.device, dtype=torch.float32)
        check(lib.advhip_chan_layernorm_bwd_add_f32(ptr(dxh), ptr(x), ptr(gf), ptr(mu), ptr(rs), ptr(dy), ptr(dx), ptr(pgb), dim, n,
                                                    C.c_float(ctx.eps), stream(x)), "chan_layernorm_bwd_add")
        gp, bp = ctx.ln_params
        if need[1] and need[2] and _defer_colsum(pgb, [(gp, 0, dim), (bp, dim, 2 * dim)]):
            return dx, None, None, None, dw1, db1, dw2, db2, None
        sums = colsum(pgb)
        return dx, sums[:dim].reshape(ctx.gshape), sums[dim:].reshape(ctx.gshape), None, dw1, db1, dw2, db2, None


# ADV_MGFN_FUSED_FFN=1 (opt-in): the narrow blocks' `x + FFN(LN(x))` step as ONE launch forward and one backward (csrc/ffn_fused.hip) instead of
# three + three.  Measured inside the graph-replayed training step (profiles/r06_studies.md section 7): 64 channels 26.5 + 29.0 us against
# 37 + 29, 128 channels 74 + 99 us against 52 + 46 -- the step as a whole 15.1 ms with it, 14.8 without on the same box.  Off by default.
FUSED_FFN = os.environ.get("ADV_MGFN_FUSED_FFN", "0") == "1"


def fused_ffn_ok(c: int, hid: int, n: int, w1: torch.Tensor, w2: torch.Tensor) -> bool:
    """advhip_ffn_block_fwd/bwd_f32's shape rules: 64 or 128 channels, hidden = 4 x, a multiple of 64 positions, weights as stored."""
    return (FUSED_FFN and c in (64, 128) and hid == 4 * c and n % 64 == 0 and tuple(w1.shape[:2]) == (hid, c) and tuple(w2.shape[:2]) == (c, hid)
            and w1.is_contiguous() and w2.is_contiguous())


PENDING_COUNTERS: list = []  # BatchNorm1d.num_batches_tracked tensors whose += 1 is still owed


def flush_counters(discard: bool = False) -> None:
    """num_batches_tracked += 1 for every BatchNorm layer that ran through _FocusAttnBlockCN since the last call: one
    multi-tensor launch instead of one per layer (MGFNModel.forward calls this after the body; `discard`: a forward that
    raised drops what it queued instead of billing it to a later, unrelated forward)."""
    if PENDING_COUNTERS and not discard:
        torch._foreach_add_(PENDING_COUNTERS, 1)
    PENDING_COUNTERS.clear()


class _FocusAttnBlockCN(torch.autograd.Function):
    """y = x + to_out(rel_pos(to_v(BN(x)))): a whole `x = attention(x) + x` step of a FocusBlock (modeling_mgfn.py:150-180, 203)
    in training mode: BatchNorm with batch statistics (running statistics updated inside the launch), GEMM, depth-wise
    temporal conv, GEMM + bias + residual; backward with dW + db per launch and the skip connection's gradient added inside
    the BatchNorm-backward launch."""

    @staticmethod
    def forward(ctx, x, bn_w, bn_b, wv, wrel, brel, wo, bo, bn, heads, fresh):
        _lib.require_gpu(x, bn_w, bn_b, wv, wrel, brel, wo, bo, contiguous=False)
        c, b_, t = x.shape
        n = b_ * t
        inner = wv.shape[0]
        lib = _lib.load()
        xb = torch.empty_like(x)
        mean = torch.empty((c,), device=x.device, dtype=torch.float32)
        var = torch.empty_like(mean)
        track = bn.track_running_stats and bn.running_mean is not None
        check(lib.advhip_bn_rows_fwd_running_f32(ptr(x), ptr(bn_w.detach()), ptr(bn_b.detach()), ptr(xb), ptr(mean), ptr(var),
                                                 ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                                                 C.c_float(bn.momentum if track else 0.0), c, n, C.c_float(bn.eps), stream(x)), "bn_rows_fwd_running")
        if track:
            PENDING_COUNTERS.append(bn.num_batches_tracked)  # += 1 for all the body's BatchNorm layers in one launch (flush_counters)
        v = conv_cn(xb, pack_kc_cached(wv, fresh), inner, 1)
        k = wrel.shape[-1]
        w2 = wrel.detach().reshape(heads, k).contiguous()
        o = torch.empty_like(v)
        check(lib.advhip_dwconv_t_fwd_f32(ptr(v), ptr(w2), ptr(brel.detach().contiguous()), ptr(o), inner, heads, b_, t, k, stream(x)), "dwconv_t_fwd")
        y = conv_cn(o, pack_kc_cached(wo, fresh), c, 1, shift=bo.detach(), residual=x)
        ctx.save_for_backward(x, bn_w, mean, var, xb, v, w2, o, wv, wo)
        ctx.params = (wv, wo, bo)
        ctx.rel_params = (wrel, brel)
        ctx.eps, ctx.heads, ctx.wrel_shape = bn.eps, heads, wrel.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, bn_w, mean, var, xb, v, w2, o, wv, wo = ctx.saved_tensors
        c, b_, t = x.shape
        inner, heads = wv.shape[0], ctx.heads
        k = w2.shape[1]
        need = ctx.needs_input_grad
        lib = _lib.load()
        dy = dy.contiguous()
        wvp, wop, bop = ctx.params
        dwo, dbo = _dw_db(dy, o, need[6], need[7], wop, bop)
        do = conv_cn(dy, wo.detach().view(c, inner), inner, 1)
        chunks = lib.advhip_dwconv_t_bwd_chunks(inner, b_)
        dv = torch.empty_like(v)
        partial = torch.empty((inner * chunks, k + 1), device=x.device, dtype=torch.float32)
        check(lib.advhip_dwconv_t_bwd_f32(ptr(do), ptr(v), ptr(w2), ptr(dv), ptr(partial), inner, heads, b_, t, k, stream(x)), "dwconv_t_bwd")
        wrelp, brelp = ctx.rel_params
        p2 = partial.view(inner // heads * chunks, heads * (k + 1))  # rows = (c_idx, chunk)
        if need[4] and need[5] and _defer_colsum(p2, [(wrelp, 0, heads * k), (brelp, heads * k, heads * (k + 1))], period=k + 1):
            dwrel = dbrel = None
        else:
            per_head = colsum(p2).view(heads, k + 1)
            dwrel, dbrel = per_head[:, :k].reshape(ctx.wrel_shape), per_head[:, k].contiguous()
        dwv, _ = _dw_db(dv, xb, need[3], False, wvp, None)
        dxb = conv_cn(dv, wv.detach().view(inner, c), c, 1)
        dx = torch.empty_like(x)
        dg = torch.empty((c,), device=x.device, dtype=torch.float32)
        db = torch.empty_like(dg)
        check(lib.advhip_bn_rows_bwd_add_f32(ptr(dxb), ptr(x), ptr(bn_w.detach()), ptr(mean), ptr(var), ptr(dy), ptr(dx), ptr(dg), ptr(db), c,
                                             b_ * t, C.c_float(ctx.eps), stream(x)), "bn_rows_bwd_add")
        return dx, dg, db, dwv, dwrel, dbrel, dwo, dbo, None, None, None


class _GlanceAttnCore(torch.autograd.Function):
    """out = v softmax(scale q^T k)^T per (sequence, head) on (C, B, T) activations: GlanceAttention between its two 1x1 convs
    (modeling_mgfn.py:113-122) as one launch forward, one backward (csrc/mgfn.hip).  T = 32 (every training batch: the runner
    feeds 32-segment videos): the whole (sequence, head) problem in LDS, the softmax kept for the backward pass.  Any other T (the
    validation pass over a whole video, runner.py:42-50): 32-key tiles with an online softmax, the rows' log-sum-exp kept instead
    of the T x T softmax, the backward pass recomputing it tile by tile."""

    @staticmethod
    def forward(ctx, qkv, heads, dim_head, scale):
        _lib.require_gpu(qkv)
        c3, b, t = qkv.shape
        inner = c3 // 3
        lib = _lib.load()
        out = torch.empty((inner, b, t), device=qkv.device, dtype=torch.float32)
        need = any(ctx.needs_input_grad)
        if t == 32:
            p = torch.empty((b, heads, t, t), device=qkv.device, dtype=torch.float32)
            check(lib.advhip_glance_attention_fwd_f32(ptr(qkv), ptr(out), ptr(p), heads, b, t, dim_head, C.c_float(scale), stream(qkv)),
                  "glance_attention_fwd")
            ctx.save_for_backward(qkv, p)
        else:
            lse = torch.empty((b, heads, t), device=qkv.device, dtype=torch.float32) if need else None
            check(lib.advhip_glance_attention_fwd_anyt_f32(ptr(qkv), ptr(out), ptr(lse), heads, b, t, dim_head, C.c_float(scale), stream(qkv)),
                  "glance_attention_fwd_anyt")
            if need:
                ctx.save_for_backward(qkv, out, lse)
        ctx.args = (heads, dim_head, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        heads, dim_head, scale = ctx.args
        qkv = ctx.saved_tensors[0]
        _, b, t = qkv.shape
        dqkv = torch.empty_like(qkv)
        lib = _lib.load()
        if t == 32:
            p = ctx.saved_tensors[1]
            check(lib.advhip_glance_attention_bwd_f32(ptr(dout.contiguous()), ptr(qkv), ptr(p), ptr(dqkv), heads, b, t, dim_head, C.c_float(scale),
                                                      stream(qkv)), "glance_attention_bwd")
        else:
            _qkv, out, lse = ctx.saved_tensors
            check(lib.advhip_glance_attention_bwd_anyt_f32(ptr(dout.contiguous()), ptr(qkv), ptr(out), ptr(lse), ptr(dqkv), heads, b, t, dim_head,
                                                           C.c_float(scale), stream(qkv)), "glance_attention_bwd_anyt")
        return dqkv, None, None, None


class _GlanceAttnBlockCN(torch.autograd.Function):
    """y = x + to_out(core(to_qkv(LN(x)))): a whole `x = attention(x) + x` step of a GlanceBlock (modeling_mgfn.py:107-123, 145) as one
    autograd node -- LayerNorm, GEMM, the attention core, GEMM + bias + residual -- whose LayerNorm-backward launch also adds the skip
    connection's gradient (autograd's own add over the activation, one launch per block, is gone) and queues dg / db with the other
    deferred column sums."""

    @staticmethod
    def forward(ctx, x, g, b, eps, wqkv, wout, bout, heads, dim_head, scale, fresh):
        _lib.require_gpu(x, g, b, wqkv, wout, bout, contiguous=False)
        c = x.shape[0]
        n = x.numel() // c
        lib = _lib.load()
        xh = torch.empty_like(x)
        mu = torch.empty((n,), device=x.device, dtype=torch.float32)
        rs = torch.empty_like(mu)
        gf, bf = g.detach().reshape(c).contiguous(), b.detach().reshape(c).contiguous()
        check(lib.advhip_chan_layernorm_fwd_f32(ptr(x), ptr(gf), ptr(bf), ptr(xh), ptr(mu), ptr(rs), c, n, C.c_float(eps), stream(x)), "chan_layernorm_fwd")
        inner = heads * dim_head
        qkv = conv_cn(xh, pack_kc_cached(wqkv, fresh), 3 * inner, 1)
        _c3, bsz, t = qkv.shape
        core = torch.empty((inner, bsz, t), device=x.device, dtype=torch.float32)
        if t == 32:
            p = torch.empty((bsz, heads, t, t), device=x.device, dtype=torch.float32)
            check(lib.advhip_glance_attention_fwd_f32(ptr(qkv), ptr(core), ptr(p), heads, bsz, t, dim_head, C.c_float(scale), stream(x)), "glance_attention_fwd")
            aux = (p,)
        else:
            lse = torch.empty((bsz, heads, t), device=x.device, dtype=torch.float32)
            check(lib.advhip_glance_attention_fwd_anyt_f32(ptr(qkv), ptr(core), ptr(lse), heads, bsz, t, dim_head, C.c_float(scale), stream(x)),
                  "glance_attention_fwd_anyt")
            aux = (core, lse)
        y = conv_cn(core, pack_kc_cached(wout, fresh), c, 1, shift=bout.detach(), residual=x)
        ctx.save_for_backward(x, gf, mu, rs, xh, qkv, core, wqkv, wout, *aux)
        ctx.params = (wqkv, wout, bout)
        ctx.ln_params = (g, b)
        ctx.args = (eps, heads, dim_head, scale, g.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gf, mu, rs, xh, qkv, core, wqkv, wout = ctx.saved_tensors[:9]
        aux = ctx.saved_tensors[9:]
        eps, heads, dim_head, scale, gshape = ctx.args
        c = x.shape[0]
        n = x.numel() // c
        inner = heads * dim_head
        _c3, bsz, t = qkv.shape
        need = ctx.needs_input_grad
        lib = _lib.load()
        dy = dy.contiguous()
        wqkvp, woutp, boutp = ctx.params
        dwo, dbo = _dw_db(dy, core, need[5], need[6], woutp, boutp)
        dcore = conv_cn(dy, wout.detach().view(c, inner), inner, 1)
        dqkv = torch.empty_like(qkv)
        if t == 32:
            check(lib.advhip_glance_attention_bwd_f32(ptr(dcore), ptr(qkv), ptr(aux[0]), ptr(dqkv), heads, bsz, t, dim_head, C.c_float(scale), stream(x)),
                  "glance_attention_bwd")
        else:
            check(lib.advhip_glance_attention_bwd_anyt_f32(ptr(dcore), ptr(qkv), ptr(aux[0]), ptr(aux[1]), ptr(dqkv), heads, bsz, t, dim_head, C.c_float(scale),
                                                           stream(x)), "glance_attention_bwd_anyt")
        dwq, _ = _dw_db(dqkv, xh, need[4], False, wqkvp, None)
        dxh = conv_cn(dqkv, wqkv.detach().view(3 * inner, c), c, 1)
        rows = lib.advhip_chan_layernorm_bwd_partial_rows(n)
        dx = torch.empty_like(x)
        pgb = torch.empty((rows, 2 * c), device=x.device, dtype=torch.float32)
        check(lib.advhip_chan_layernorm_bwd_add_f32(ptr(dxh), ptr(x), ptr(gf), ptr(mu), ptr(rs), ptr(dy), ptr(dx), ptr(pgb), c, n, C.c_float(eps), stream(x)),
              "chan_layernorm_bwd_add")
        gp, bp = ctx.ln_params
        if need[1] and need[2] and _defer_colsum(pgb, [(gp, 0, c), (bp, c, 2 * c)]):
            return dx, None, None, None, dwq, dwo, dbo, None, None, None, None
        sums = colsum(pgb)
        return dx, sums[:c].reshape(gshape), sums[c:].reshape(gshape), None, dwq, dwo, dbo, None, None, None, None


GLANCE_BLOCK = os.environ.get("ADV_MGFN_GLANCE_BLOCK", "1") == "1"  # (0: LayerNorm, the two GEMMs and the core as separate autograd nodes)


def glance_attention_block_cn(x: torch.Tensor, norm, to_qkv, to_out, heads: int, dim_head: int, scale: float) -> torch.Tensor:
    """x + to_out(attention(to_qkv(LN(x)))) with autograd, as one Function."""
    fresh = _will_train(to_qkv.weight, to_out.weight, to_out.bias)
    return _GlanceAttnBlockCN.apply(x.contiguous(), norm.g, norm.b, norm.eps, to_qkv.weight, to_out.weight, to_out.bias, heads, dim_head, scale, fresh)


def glance_attention_ok(qkv: torch.Tensor, heads: int, dim_head: int) -> bool:
    return fused_ok(qkv) and qkv.is_contiguous() and qkv.shape[2] >= 1 and dim_head == 64 and qkv.shape[0] == 3 * heads * dim_head


def glance_attention_core(qkv: torch.Tensor, heads: int, dim_head: int, scale: float) -> torch.Tensor:
    return _GlanceAttnCore.apply(qkv, heads, dim_head, scale)


class _HeadLnFc(torch.autograd.Function):
    """(xn, scores) = head(y): y (C, B, T) -- the body's layout -- -> xn (B, T, C) = nn.LayerNorm(C) of every position, scores
    (B, T, 1) = sigmoid(nn.Linear(C, 1)(xn)) (modeling_mgfn.py:387-389), one launch forward and one backward: the permute that
    torch has to materialise (42 MB at the training batch, three times per step) happens inside the kernels' LDS tiles."""

    @staticmethod
    def forward(ctx, y, ln_w, ln_b, eps, fc_w, fc_b):
        ctx.set_materialize_grads(False)  # (an unused output's gradient arrives as None, not as a zero-filled tensor)
        _lib.require_gpu(y, ln_w, ln_b, fc_w, fc_b)
        c, b, t = y.shape
        n = b * t
        dev = y.device
        xn = torch.empty((b, t, c), device=dev, dtype=torch.float32)
        mean = torch.empty((n,), device=dev, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        score = torch.empty((b, t, 1), device=dev, dtype=torch.float32)
        check(_lib.load().advhip_head_ln_fc_fwd_f32(ptr(y), ptr(ln_w.detach()), ptr(ln_b.detach()), ptr(fc_w.detach()), ptr(fc_b.detach()), ptr(xn), ptr(mean),
                                                    ptr(rstd), ptr(score), c, n, C.c_float(eps), stream(y)), "head_ln_fc_fwd")
        ctx.save_for_backward(y, ln_w, ln_b, fc_w, mean, rstd, score)
        ctx.head_params = (ln_w, ln_b, fc_w, fc_b)
        return xn, score

    @staticmethod
    def backward(ctx, d_xn, d_score):
        y, ln_w, ln_b, fc_w, mean, rstd, score = ctx.saved_tensors
        c, b, t = y.shape
        n = b * t
        lib = _lib.load()
        rows = lib.advhip_head_ln_fc_partial_rows(n)
        dy = torch.empty_like(y)
        partial = torch.empty((rows, 3 * c + 1), device=y.device, dtype=torch.float32)
        d_xn = None if d_xn is None else d_xn.contiguous()
        d_score = None if d_score is None else d_score.contiguous()
        check(lib.advhip_head_ln_fc_bwd_f32(ptr(d_xn), ptr(d_score), ptr(y), ptr(ln_w.detach()), ptr(ln_b.detach()), ptr(fc_w.detach()), ptr(mean), ptr(rstd),
                                            ptr(score), ptr(dy), ptr(partial), c, n, stream(y)), "head_ln_fc_bwd")
        lw, lb, fw, fb = ctx.head_params
        if all(ctx.needs_input_grad[i] for i in (1, 2, 4, 5)) and _defer_colsum(partial, [(lw, 0, c), (lb, c, 2 * c), (fw, 2 * c, 3 * c), (fb, 3 * c, 3 * c + 1)]):
            return dy, None, None, None, None, None
        sums = colsum(partial)
        return dy, sums[:c], sums[c : 2 * c], None, sums[2 * c : 3 * c].view(1, c), sums[3 * c :]


def head_ok(y: torch.Tensor, ln: torch.nn.LayerNorm, fc: torch.nn.Linear) -> bool:
    return (fused_ok(y) and y.is_contiguous() and ln.elementwise_affine and ln.bias is not None and tuple(ln.normalized_shape) == (y.shape[0],)
            and fc.out_features == 1 and fc.in_features == y.shape[0] and fc.bias is not None)


def head_ln_fc(y: torch.Tensor, ln: torch.nn.LayerNorm, fc: torch.nn.Linear):
    return _HeadLnFc.apply(y, ln.weight, ln.bias, ln.eps, fc.weight, fc.bias)


def ffn_block_cn(x: torch.Tensor, norm, in_conv: torch.nn.Conv1d, out_conv: torch.nn.Conv1d) -> torch.Tensor:
    """x + ffn(LN(x)) with autograd (the whole step as one Function: fewer launches, no separate skip-gradient add)."""
    fresh = _will_train(in_conv.weight, in_conv.bias, out_conv.weight, out_conv.bias)
    if fresh:
        _FOLDED.pop(id(in_conv), None)
    return _FFNBlockCN.apply(x.contiguous(), norm.g, norm.b, norm.eps, in_conv.weight, in_conv.bias, out_conv.weight, out_conv.bias, fresh)


def focus_attention_block_cn(x: torch.Tensor, bn: torch.nn.BatchNorm1d, to_v, rel_pos, to_out, heads: int) -> torch.Tensor:
    """x + to_out(rel_pos(to_v(BN_train(x)))) with autograd, as one Function."""
    fresh = _will_train(to_v.weight, to_out.weight, to_out.bias)
    return _FocusAttnBlockCN.apply(x.contiguous(), bn.weight, bn.bias, to_v.weight, rel_pos.weight, rel_pos.bias, to_out.weight, to_out.bias,
                                   bn, heads, fresh)


def _will_train(*params) -> bool:
    """This forward is being recorded for a backward pass that can reach these parameters."""
    return torch.is_grad_enabled() and any(p is not None and p.requires_grad for p in params)


def linear_cn(x: torch.Tensor, conv: torch.nn.Conv1d, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    return _LinearCN.apply(x.contiguous(), conv.weight, conv.bias, residual, _will_train(conv.weight, conv.bias))


def ffn_cn(xh: torch.Tensor, x_res: torch.Tensor, in_conv: torch.nn.Conv1d, out_conv: torch.nn.Conv1d) -> torch.Tensor:
    fresh = _will_train(in_conv.weight, in_conv.bias, out_conv.weight, out_conv.bias)
    if fresh:
        _FOLDED.pop(id(in_conv), None)  # (the folded-LayerNorm operands of the inference form: same reasoning as pack_kc_cached)
    return _FFNCN.apply(xh.contiguous(), x_res, in_conv.weight, in_conv.bias, out_conv.weight, out_conv.bias, fresh)


_FOLDED: Dict[int, Tuple] = {}


def fold_affine(weight: torch.Tensor, mul: torch.Tensor, add: Optional[torch.Tensor], bias: Optional[torch.Tensor], want_rowsum: bool = False):
    """The per-channel affine map x -> mul x + add folded into the 1x1 layer (weight (O, C[, 1]), bias) behind it: (W diag(mul)
    packed as the kernels' [C][O] operand, bias + W add, row sums of W diag(mul) or None) -- one HIP launch + the pack
    (include/advhip.h: advhip_fold_affine_f32); operand-build time, cached by the callers."""
    o, c = weight.shape[0], weight.shape[1]
    w2 = weight.detach().reshape(o, c)
    _lib.require_gpu(w2, mul, add, bias)
    wf = torch.empty((o, c, 1), device=w2.device, dtype=torch.float32)
    bf = torch.empty((o,), device=w2.device, dtype=torch.float32)
    rs = torch.empty_like(bf) if want_rowsum else None
    check(_lib.load().advhip_fold_affine_f32(ptr(w2), ptr(mul), ptr(add), ptr(bias), ptr(wf), ptr(bf), ptr(rs), o, c, stream(w2)), "fold_affine")
    return pack_kc(wf), bf, rs


def _folded(owner, params, build):
    """Operands derived from `params`, cached on id(owner) until one of them changes (data_ptr / version / the cache epoch)."""
    stamp = tuple((p.data_ptr(), p._version) for p in params) + (_EPOCH,)
    hit = _FOLDED.get(id(owner))
    if hit is None or hit[0]() is not owner or hit[1] != stamp:
        if len(_FOLDED) > 256:
            for k in [k for k, v in _FOLDED.items() if v[0]() is None]:
                del _FOLDED[k]
        hit = _FOLDED[id(owner)] = (weakref.ref(owner), stamp) + tuple(build())
    return hit[2:]


@torch.no_grad()
def ffn_cn_folded_ln(x: torch.Tensor, norm, in_conv: torch.nn.Conv1d, out_conv: torch.nn.Conv1d) -> torch.Tensor:
    """Inference form of `ffn(x) + x` with the channel LayerNorm folded into the first GEMM: W1.LN(x) = (W1.diag(g)) x_raw
    * rs - rowsum(W1.diag(g)) * mu * rs + W1.b, statistics per position from one pass over x (advhip_chan_stats_f32).
    The folded operands are cached until one of the four parameters changes."""
    x = x.contiguous()
    hid, dim = in_conv.weight.shape[0], in_conv.weight.shape[1]
    wg_packed, shift, u = _folded(in_conv, (in_conv.weight, in_conv.bias, norm.g, norm.b),
                                  lambda: fold_affine(in_conv.weight, norm.g.detach().reshape(dim).contiguous(), norm.b.detach().reshape(dim).contiguous(),
                                                      in_conv.bias.detach(), want_rowsum=True))
    mu, rs = chan_stats(x, norm.eps)
    h = conv_cn(x, wg_packed, hid, 1, shift=shift, act=ACT_GELU, ln=(u, mu, rs))
    return conv_cn(h, pack_kc_cached(out_conv.weight), dim, 1, shift=out_conv.bias, residual=x)


def bn_eval_fold_ok(x: torch.Tensor, bn: torch.nn.BatchNorm1d, to_v: torch.nn.Conv1d) -> bool:
    return (not torch.is_grad_enabled() and not bn.training and bn.running_mean is not None and bn.affine and fused_ok(x)
            and to_v.weight.shape[2] == 1 and eligible(to_v.weight.shape[1], to_v.weight.shape[0], x) and _on_current_device(to_v.weight))


@torch.no_grad()
def to_v_folded_bn(x: torch.Tensor, bn: torch.nn.BatchNorm1d, to_v: torch.nn.Conv1d) -> torch.Tensor:
    """to_v(BatchNorm1d_eval(x)) as ONE GEMM launch (modeling_mgfn.py:162, 173-174): eval-mode BN is the affine map x -> s x + t,
    s = gamma / sqrt(running_var + eps), t = beta - running_mean s, so to_v(BN(x)) = (Wv diag(s)) x + Wv t -- the folded operand
    and bias are built once per set of weights (advhip_bn_fold_f32 + advhip_fold_affine_f32) and cached."""
    inner = to_v.weight.shape[0]

    def build():
        scale, shift = ops.bn_fold(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
        wp, bf, _ = fold_affine(to_v.weight, scale, shift, to_v.bias.detach() if to_v.bias is not None else None)
        return wp, bf

    params = (to_v.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var) + ((to_v.bias,) if to_v.bias is not None else ())
    wp, bf = _folded(to_v, params, build)
    return conv_cn(x.contiguous(), wp, inner, 1, shift=bf)


# ---- fused element-wise layers (csrc/mgfn.hip) -------------------------------------------------------------------------
class _ChanLayerNorm(torch.autograd.Function):
    """MGFNLayerNorm over the channels of a (C, B, T) activation: one launch forward, one backward."""

    @staticmethod
    def forward(ctx, x, g, b, eps):
        _lib.require_gpu(x, g, b, contiguous=False)
        c = x.shape[0]
        n = x.numel() // c
        y = torch.empty_like(x)
        mu = torch.empty((n,), device=x.device, dtype=torch.float32)
        rs = torch.empty_like(mu)
        gf, bf = g.detach().reshape(c).contiguous(), b.detach().reshape(c).contiguous()
        check(_lib.load().advhip_chan_layernorm_fwd_f32(ptr(x), ptr(gf), ptr(bf), ptr(y), ptr(mu), ptr(rs), c, n, C.c_float(eps), stream(x)),
              "chan_layernorm_fwd")
        ctx.save_for_backward(x, gf, mu, rs)
        ctx.eps, ctx.gshape = eps, g.shape
        ctx.ln_params = (g, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gf, mu, rs = ctx.saved_tensors
        c = x.shape[0]
        n = x.numel() // c
        dy = dy.contiguous()
        lib = _lib.load()
        rows = lib.advhip_chan_layernorm_bwd_partial_rows(n)
        dx = torch.empty_like(x)
        pg = torch.empty((rows, c), device=x.device, dtype=torch.float32)
        pb = torch.empty_like(pg)
        check(lib.advhip_chan_layernorm_bwd_f32(ptr(dy), ptr(x), ptr(gf), ptr(mu), ptr(rs), ptr(dx), ptr(pg), ptr(pb), c, n,
                                                C.c_float(ctx.eps), stream(x)), "chan_layernorm_bwd")
        gp, bp = ctx.ln_params
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and _defer_colsum(pg, [(gp, 0, c)]) and _defer_colsum(pb, [(bp, 0, c)]):
            return dx, None, None, None
        return dx, colsum(pg).view(ctx.gshape), colsum(pb).view(ctx.gshape), None


class _DWConvT(torch.autograd.Function):
    """FocusAttention.rel_pos on a (C, B, T) activation (channel c -> head c % H): one launch forward, one backward."""

    @staticmethod
    def forward(ctx, v, weight, bias):
        _lib.require_gpu(v, weight, bias, contiguous=False)
        c, b, t = v.shape
        h, _, k = weight.shape
        out = torch.empty_like(v)
        w2 = weight.detach().reshape(h, k).contiguous()
        check(_lib.load().advhip_dwconv_t_fwd_f32(ptr(v), ptr(w2), ptr(bias.detach().contiguous()), ptr(out), c, h, b, t, k, stream(v)), "dwconv_t_fwd")
        ctx.save_for_backward(v, w2)
        ctx.wshape = weight.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        v, w2 = ctx.saved_tensors
        c, b, t = v.shape
        h, k = w2.shape
        dout = dout.contiguous()
        lib = _lib.load()
        chunks = lib.advhip_dwconv_t_bwd_chunks(c, b)
        dv = torch.empty_like(v)
        partial = torch.empty((c * chunks, k + 1), device=v.device, dtype=torch.float32)
        check(lib.advhip_dwconv_t_bwd_f32(ptr(dout), ptr(v), ptr(w2), ptr(dv), ptr(partial), c, h, b, t, k, stream(v)), "dwconv_t_bwd")
        per_head = colsum(partial.view(c // h * chunks, h * (k + 1))).view(h, k + 1)  # rows = (c_idx, chunk), channel = c_idx * H + h_idx
        return dv, per_head[:, :k].reshape(ctx.wshape), per_head[:, k].contiguous()


class _TokenTaps(torch.autograd.Function):
    """z = Wt @ X^T for Wt (k*O, C) -- the k tap matrices of the 2048 -> 64 token conv stacked -- and X the scorer's input rows AS
    STORED (`rows`: (N positions, C + 1) contiguous, the magnitude column rides along; MGFNFeatureAmplifier._tokens_by_taps,
    /root/reference/src/models/mgfn/modeling_mgfn.py:81-93).  Forward: both operands are contraction-contiguous, i.e.
    advhip_gemm_nt_f32 reads them in place (row pitch C + 1).  Backward: dWt[o, c] = sum_n dZ[o, n] X[n, c] contracts over the
    positions, along which X is NOT contiguous -- but X as stored IS the conv kernels' A operand for that product (row n = "channel"
    n of a (N, 1, C + 1) activation, its C + 1 entries the positions): one weight pack of dZ (7.8 MB) and one conv launch with the K
    slices summed inside; column C of the result (the magnitude column's product) is dropped.  The input carries no gradient."""

    @staticmethod
    def forward(ctx, wt, rows):
        c = wt.shape[1]
        ctx.save_for_backward(rows)
        return ops.gemm_nt(wt, rows[:, :c])

    @staticmethod
    def backward(ctx, dz):
        (rows,) = ctx.saved_tensors
        n, c1 = rows.shape
        ko = dz.shape[0]
        wp = pack_kc(dz.contiguous().view(ko, n, 1))                   # [N (padded to 32)][k*O]
        y = conv_cn(rows.view(n, 1, c1), wp, ko, 1)                    # (k*O, 1, C + 1)
        return y[:, 0, : c1 - 1], None


def token_taps_ok(wt: torch.Tensor, rows: torch.Tensor) -> bool:
    return (_on_current_device(rows) and rows.dtype == torch.float32 and wt.dtype == torch.float32 and rows.dim() == 2 and rows.is_contiguous()
            and wt.shape[1] % 16 == 0 and wt.shape[1] + 1 == rows.shape[1] and wt.shape[0] % 64 == 0 and not rows.requires_grad)


def token_taps(wt: torch.Tensor, rows: torch.Tensor) -> torch.Tensor:
    return _TokenTaps.apply(wt.contiguous(), rows)


class _AmpCombine(torch.autograd.Function):
    """tokens = sum_j shift_j(z[j]) + bias + ratio * Conv1d_k3(magnitude): what follows the tap GEMM in MGFNFeatureAmplifier
    (modeling_mgfn.py:81-93) as one launch forward and one backward (include/advhip.h: advhip_amp_combine_*_f32)."""

    @staticmethod
    def forward(ctx, z, bias, mag, wm, bm, ratio):
        _lib.require_gpu(z, bias, wm, bm)
        _lib.require_gpu(mag, contiguous=False)
        k, o, b, t = z.shape
        y = torch.empty((o, b, t), device=z.device, dtype=torch.float32)
        check(_lib.load().advhip_amp_combine_fwd_f32(ptr(z), ptr(bias.detach()), ptr(mag), mag.stride(2), ptr(wm.detach()), ptr(bm.detach()),
                                                     C.c_float(ratio), ptr(y), o, b, t, stream(z)), "amp_combine_fwd")
        ctx.mag, ctx.ratio = mag, ratio
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        o, b, t = dy.shape
        mag = ctx.mag
        dz = torch.empty((3, o, b, t), device=dy.device, dtype=torch.float32)
        d_bias = torch.empty((o,), device=dy.device, dtype=torch.float32)
        d_bm = torch.empty_like(d_bias)
        d_wm = torch.empty((o, 1, 3), device=dy.device, dtype=torch.float32)
        check(_lib.load().advhip_amp_combine_bwd_f32(ptr(dy), ptr(mag), mag.stride(2), C.c_float(ctx.ratio), ptr(dz), ptr(d_bias), ptr(d_wm), ptr(d_bm),
                                                     o, b, t, stream(dy)), "amp_combine_bwd")
        return dz, d_bias, None, d_wm, d_bm, None


def amp_combine_ok(z: torch.Tensor, conv: torch.nn.Conv1d, to_mag: torch.nn.Conv1d, mag: torch.Tensor) -> bool:
    """`mag`: the (1, B, T) magnitude channel as a view of the input (b, t) rows -- element (b, t) at (b * T + t) * stride(2)."""
    return (z.dim() == 4 and z.shape[0] == 3 and _on_current_device(z) and z.dtype == torch.float32 and z.is_contiguous() and conv.bias is not None
            and to_mag.bias is not None and tuple(to_mag.weight.shape) == (z.shape[1], 1, 3) and mag.dim() == 3 and mag.shape[0] == 1
            and mag.stride(1) == mag.shape[2] * mag.stride(2) and not mag.requires_grad and mag.dtype == torch.float32 and _on_current_device(mag))


def amp_combine(z, conv, to_mag, mag, ratio: float) -> torch.Tensor:
    return _AmpCombine.apply(z, conv.bias, mag, to_mag.weight, to_mag.bias, float(ratio))


def fused_ok(x: torch.Tensor) -> bool:
    return _on_current_device(x) and x.dtype == torch.float32 and x.dim() == 3


def chan_layernorm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    return _ChanLayerNorm.apply(x.contiguous(), g, b, eps)


def dwconv_t(v: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    return _DWConvT.apply(v.contiguous(), weight, bias)


class _BNRowsTrain(torch.autograd.Function):
    """Training-mode BatchNorm1d on a (C, B, T) activation: batch statistics, one launch forward, one backward.
    Returns (y, batch mean, biased batch variance); the running-statistics update stays with the caller."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _lib.require_gpu(x, gamma, beta, contiguous=False)
        c = x.shape[0]
        n = x.numel() // c
        y = torch.empty_like(x)
        mean = torch.empty((c,), device=x.device, dtype=torch.float32)
        var = torch.empty_like(mean)
        check(_lib.load().advhip_bn_rows_fwd_f32(ptr(x), ptr(gamma.detach()), ptr(beta.detach()), ptr(y), ptr(mean), ptr(var), c, n,
                                                 C.c_float(eps), stream(x)), "bn_rows_fwd")
        ctx.save_for_backward(x, gamma, mean, var)
        ctx.eps = eps
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, gamma, mean, var = ctx.saved_tensors
        c = x.shape[0]
        n = x.numel() // c
        dx = torch.empty_like(x)
        dg = torch.empty((c,), device=x.device, dtype=torch.float32)
        db = torch.empty_like(dg)
        check(_lib.load().advhip_bn_rows_bwd_f32(ptr(dy.contiguous()), ptr(x), ptr(gamma.detach()), ptr(mean), ptr(var), ptr(dx), ptr(dg),
                                                 ptr(db), c, n, C.c_float(ctx.eps), stream(x)), "bn_rows_bwd")
        return dx, dg, db, None


def bn_rows_train(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float):
    return _BNRowsTrain.apply(x.contiguous(), gamma, beta, eps)
