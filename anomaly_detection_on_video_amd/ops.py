"""Thin Python wrappers over the C ABI (include/advhip.h): shape checks, output allocation
(torch is used only for device memory and streams) and the launch on torch's current stream.

Every function here REQUIRES CUDA tensors and the built HIP extension; none has a fallback.
"""
from __future__ import annotations

import os

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, check, ptr, require_gpu, stream


def _triple(v) -> Tuple[int, int, int]:
    if isinstance(v, int):
        return (v, v, v)
    v = tuple(int(i) for i in v)
    assert len(v) == 3
    return v


def conv_out_dims(in_thw: Sequence[int], k: Sequence[int], s: Sequence[int], p: Sequence[int]) -> Tuple[int, int, int]:
    return tuple((n + 2 * pp - kk) // ss + 1 for n, kk, ss, pp in zip(in_thw, k, s, p))  # type: ignore


@dataclass
class PackedConv:
    """Load-time state of one conv+BN unit: packed weights, folded BN, gather tables per input size."""

    cout: int
    cin: int
    kernel: Tuple[int, int, int]
    stride: Tuple[int, int, int]
    padding: Tuple[int, int, int]
    w_packed: torch.Tensor  # [Kpad, Cout]
    scale: torch.Tensor
    shift: torch.Tensor
    ktabs: Dict[Tuple[int, int, int], torch.Tensor] = field(default_factory=dict)
    ktabs_s2w: Dict[Tuple[int, int, int], torch.Tensor] = field(default_factory=dict)  # column-parity gather tables (stem)
    algo: int = _lib.ALGO_AUTO
    name: str = ""
    splits: int = 0  # 0 = library heuristic
    weight: Optional[torch.Tensor] = None   # the torch-layout fp32 weights (a reference, not a copy)
    w_split: Optional[torch.Tensor] = None  # bf16 hi/lo images for the opt-in split-bf16 kernels, packed on first use
    # (B,T,H,W) -> (algo, splits) resolved from the measured table (tuned.py)
    choices: Dict[Tuple[int, int, int, int], Tuple[int, int]] = field(default_factory=dict)

    def key(self, B: int, T: int, H: int, W: int) -> str:
        return ",".join(str(v) for v in (self.cin, self.cout, *self.kernel, *self.stride, *self.padding, B, T, H, W))

    def desc(self, B: int, T: int, H: int, W: int, relu: bool, algo: Optional[int] = None,
             splits: Optional[int] = None) -> ConvDesc:
        kt, kh, kw = self.kernel
        st, sh, sw = self.stride
        pt, ph, pw = self.padding
        if algo is None and splits is None:
            ch = self.choices.get((B, T, H, W))
            if ch is None:
                from . import tuned

                ch = tuned.lookup(self.key(B, T, H, W), (self.algo, self.splits))
                if ARITH == "bf16x3":  # opt-in: every conv on the split-bf16 kernels (128x64x32 tile fits every Cout of the net)
                    ch = (_lib.ALGO_BF16X3_BASE + 6, 1)
                self.choices[(B, T, H, W)] = ch
            algo, splits = ch
        return ConvDesc(B, self.cin, T, H, W, self.cout, kt, kh, kw, st, sh, sw, pt, ph, pw, int(relu),
                        self.algo if algo is None else algo, self.splits if splits is None else splits)


def bn_fold(gamma, beta, mean, var, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """eval-mode BatchNorm -> (scale, shift) on device."""
    require_gpu(gamma, beta, mean, var)
    lib = _lib.load()
    c = gamma.numel()
    scale = torch.empty(c, device=gamma.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    check(lib.advhip_bn_fold_f32(ptr(gamma), ptr(beta), ptr(mean), ptr(var), C.c_float(eps), c, ptr(scale), ptr(shift), stream()), "bn_fold")
    return scale, shift


def _packed_rows(d: ConvDesc) -> int:
    kpad = _lib.load().advhip_conv3d_packed_rows(C.byref(d))
    if kpad < 0:
        check(kpad, "conv3d_packed_rows")
    return kpad


def _build_ktab(pc: PackedConv, thw: Tuple[int, int, int]) -> torch.Tensor:
    d = pc.desc(1, *thw, relu=False)
    ktab = torch.empty((_packed_rows(d) * 6,), device=pc.w_packed.device, dtype=torch.int32)
    check(_lib.load().advhip_conv3d_build_ktab(C.byref(d), ptr(ktab), stream()), "build_ktab")
    return ktab


def split_weight(pc: PackedConv) -> torch.Tensor:
    """The bf16 hi/lo weight images of `pc` (uint16 [2][Cout][Kpad]) for the ADVHIP_ALGO_BF16X3_* kernels."""
    if pc.w_split is None:
        kt, kh, kw = pc.kernel
        d = pc.desc(1, kt, kh, kw, relu=False, algo=0, splits=1)
        out = torch.empty((2, pc.cout, pc.w_packed.shape[0]), device=pc.weight.device, dtype=torch.int16)
        check(_lib.load().advhip_conv3d_pack_weight_bf16x3(C.byref(d), ptr(pc.weight), ptr(out), stream()), "pack_weight_bf16x3")
        pc.w_split = out
    return pc.w_split


def ensure_ktab(pc: PackedConv, thw: Tuple[int, int, int], batch: Optional[int] = None) -> torch.Tensor:
    """The gather table of `pc` for input dims (T,H,W), built on the current stream on first use.  With `batch`, also
    every other lazily built operand the launch for (batch, T, H, W) will read -- today the bf16 hi/lo weight images
    of the opt-in split-bf16 kernels -- so that a caller about to fork streams has ALL of them behind one event."""
    ktab = pc.ktabs.get(thw)
    if ktab is None:
        ktab = pc.ktabs[thw] = _build_ktab(pc, thw)
    if batch is not None:
        d = pc.desc(batch, *thw, relu=False)
        if _lib.ALGO_BF16X3_BASE <= d.algo < _lib.ALGO_DMA2_BASE:
            split_weight(pc)
    return ktab


def pack_conv(weight: torch.Tensor, gamma, beta, mean, var, eps: float, stride, padding, name: str = "",
              algo: int = _lib.ALGO_AUTO) -> PackedConv:
    """Pack one conv (torch layout (Cout,Cin,kt,kh,kw)) + its eval BatchNorm for the HIP kernels."""
    require_gpu(weight)
    weight = weight.detach().contiguous()
    cout, cin, kt, kh, kw = weight.shape
    scale, shift = bn_fold(gamma.detach(), beta.detach(), mean.detach(), var.detach(), eps)
    pc = PackedConv(cout, cin, (kt, kh, kw), _triple(stride), _triple(padding), weight, scale, shift, name=name, algo=algo)
    d = pc.desc(1, kt, kh, kw, relu=False)
    wp = torch.empty((_packed_rows(d), cout), device=weight.device, dtype=torch.float32)
    check(_lib.load().advhip_conv3d_pack_weight_f32(C.byref(d), ptr(weight), ptr(wp), stream()), "pack_weight")
    pc.w_packed = wp
    pc.weight = weight
    return pc


# ADV_ARITH=bf16x3: opt-in split-bf16 arithmetic for every conv (see include/advhip.h ADVHIP_ALGO_BF16X3_BASE); default: exact fp32
ARITH = os.environ.get("ADV_ARITH", "f32")

_WORKSPACES: Dict[Tuple[torch.device, int], torch.Tensor] = {}


def workspace(dev: torch.device, nbytes: int) -> Optional[torch.Tensor]:
    """Grow-only scratch for the split-K partial slabs (caller-owned, per the C ABI), one per
    (device, launch stream): launches on different streams may run at the same time."""
    if nbytes <= 0:
        return None
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = _WORKSPACES[key] = torch.empty(((nbytes + 3) // 4,), device=dev, dtype=torch.float32)
    return ws


_ZERO_WORKSPACES: Dict[Tuple[torch.device, int], torch.Tensor] = {}


def zeroed_workspace(dev: torch.device, nbytes: int) -> Optional[torch.Tensor]:
    """Grow-only scratch per (device, launch stream) that is ZERO when handed out for the first time -- for kernels that keep
    arrival counters in it and leave them zero (advhip_gemm_nt_reduced_f32): zero-filled once per allocation, not per launch."""
    if nbytes <= 0:
        return None
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _ZERO_WORKSPACES.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = _ZERO_WORKSPACES[key] = torch.zeros(((nbytes + 3) // 4,), device=dev, dtype=torch.float32)
    return ws


_SPLITK_COUNTERS: Dict[Tuple[torch.device, int], torch.Tensor] = {}


def splitk_counters(dev: torch.device) -> torch.Tensor:
    """The arrival-counter block of the in-launch split-K reduction for launches on the current stream of `dev`
    (include/advhip.h: advhip_conv3d_epilogue.splitk_counters): zero-filled ONCE here; every launch leaves it zero."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    cnt = _SPLITK_COUNTERS.get(key)
    if cnt is None:
        cnt = _SPLITK_COUNTERS[key] = torch.zeros((16384,), device=dev, dtype=torch.int32)
    return cnt


def reset_stream_state(dev: Optional[torch.device] = None) -> None:
    """Forget the self-resetting counter block and the zero-initialised workspace of the current stream (of `dev`, default: the
    current device).  Those are zero-filled once and trusted to be left zero by every launch; a launch that did not run to its
    end -- an entry point that reported an error, a graph capture that was abandoned after it recorded the zero-fill -- may
    have broken that promise, so the next call allocates and zero-fills again.  `_lib.check` calls this on every reported error;
    call it yourself after abandoning a capture."""
    if not torch.cuda.is_available():
        return
    d = torch.device("cuda", torch.cuda.current_device()) if dev is None else dev
    key = (d, torch.cuda.current_stream(d).cuda_stream)
    _SPLITK_COUNTERS.pop(key, None)
    _ZERO_WORKSPACES.pop(key, None)


_lib.ON_FAILURE.append(reset_stream_state)


def batch_stride(t: torch.Tensor) -> int:
    """Elements between consecutive samples of an NCDHW tensor that is contiguous per sample: a contiguous tensor,
    or a channel slice `wide[:, a:b]` of one (the kernels take the batch stride; everything inside a sample must be
    dense).  Raises for any other layout."""
    n, c, t_, h, w = t.shape
    inner = (t_ * h * w, h * w, w, 1)
    for dim in range(1, 5):
        if t.shape[dim] > 1 and t.stride(dim) != inner[dim - 1]:
            raise ValueError(f"tensor {tuple(t.shape)} with strides {tuple(t.stride())} is not dense within a sample")
    return int(t.stride(0)) if n > 1 else c * inner[0]


def conv3d_bn_act(x: torch.Tensor, pc: PackedConv, relu: bool = True, residual: Optional[torch.Tensor] = None,
                  algo: Optional[int] = None, out: Optional[torch.Tensor] = None, splits: Optional[int] = None) -> torch.Tensor:
    """act(conv3d(x) * scale + shift (+ residual)) in one fused HIP launch.  `x` and `out` may be channel slices of wider
    NCDHW tensors (see batch_stride); the residual must be contiguous."""
    require_gpu(x, out, contiguous=False)
    require_gpu(residual)
    if x.dim() != 5 or x.shape[1] != pc.cin:
        raise ValueError(f"{pc.name}: expected (B,{pc.cin},T,H,W), got {tuple(x.shape)}")
    B, _, T, H, W = x.shape
    to, ho, wo = conv_out_dims((T, H, W), pc.kernel, pc.stride, pc.padding)
    if min(to, ho, wo) <= 0:
        raise ValueError(f"{pc.name}: input {tuple(x.shape)} smaller than the kernel")
    y = out if out is not None else torch.empty((B, pc.cout, to, ho, wo), device=x.device, dtype=torch.float32)
    if tuple(y.shape) != (B, pc.cout, to, ho, wo) or y.dtype != torch.float32 or y.device != x.device:
        raise ValueError(f"{pc.name}: out {tuple(y.shape)} {y.dtype} on {y.device} != {(B, pc.cout, to, ho, wo)} float32 on {x.device}")
    if residual is not None and residual.shape != y.shape:
        raise ValueError(f"{pc.name}: residual {tuple(residual.shape)} != output {tuple(y.shape)}")
    ktab = ensure_ktab(pc, (T, H, W))
    if algo is not None and splits is None:
        splits = 1  # an explicitly pinned tile runs unsplit unless the caller also pins the split
    d = pc.desc(B, T, H, W, relu, algo, splits)
    lib = _lib.load()
    need = lib.advhip_conv3d_workspace_bytes(C.byref(d))
    if need < 0:
        check(int(need), f"conv3d_workspace_bytes[{pc.name}]")
    ws = workspace(x.device, need)
    w = pc.w_packed
    if _lib.ALGO_BF16X3_BASE <= d.algo < _lib.ALGO_DMA2_BASE:
        w = split_weight(pc)
    xbs, ybs = batch_stride(x), batch_stride(y)  # (x and / or y may be a channel slice of a wider tensor)
    ep = None
    if need > 0:  # split-K: this stream's self-resetting arrival counters instead of a memset ahead of every launch
        cnt = splitk_counters(x.device)
        ep = C.byref(_lib.ConvEpilogue(None, None, None, None, None, ptr(cnt), cnt.numel() * 4))
    check(lib.advhip_conv3d_bn_act_ex_f32(C.byref(d), ptr(x), xbs, ptr(w), ptr(ktab), ptr(pc.scale), ptr(pc.shift), ptr(residual), ptr(y), ybs,
                                          ep, ptr(ws), need, stream()), f"conv3d[{pc.name}]")
    return y


def conv_pool_out_dims(in_thw: Sequence[int], pc: "PackedConv", pool_kernel, pool_stride) -> Tuple[int, int, int]:
    """(T, H, W) of conv `pc` on in_thw followed by a floor-mode, padding-0 max-pool."""
    return conv_out_dims(conv_out_dims(in_thw, pc.kernel, pc.stride, pc.padding), _triple(pool_kernel), _triple(pool_stride), (0, 0, 0))


# The fused stem can gather 16-byte pieces from column-parity planes of its input (advhip.h: advhip_conv3d_s2w_*): the conv
# launch is 6.5 % faster (2.25 vs 2.41 ms at B = 32), the planes cost a pass over the input (0.145 ms for an fp32 NCDHW
# tensor).  For resized uint8 frames the TenCrop / normalise pass writes the planes itself, so that path uses them by default
# (U8_STEM_FORM "planes"); for an fp32 NCDHW input the extra pass cancels the gain in isolation and adds 0.7 GB of HBM
# traffic per step; in the three-lane stream it buys +0.4 % on the 20-step line and +0.7 % sustained (round 5, four alternating
# runs on one box: 3 796 -> 3 812 and 3 846 -> 3 872 clips/s; round 3 on its kernels: +0.3 %), bit-identical features: ON by
# default since round 5 (ADV_STEM_S2W=0: the 4-byte gather straight from the NCDHW tensor).
STEM_S2W = os.environ.get("ADV_STEM_S2W", "1") == "1"


def s2w_ok(pc: PackedConv, W: int) -> bool:
    """Stride 2, odd kernel <= 9 with padding kw // 2 along w, W even and W / 2 a multiple of 4 (advhip.h: advhip_conv3d_s2w_*)."""
    kw, sw, pw = pc.kernel[2], pc.stride[2], pc.padding[2]
    return sw == 2 and kw % 2 == 1 and pw == kw // 2 and pw <= 4 and W % 8 == 0 and W >= 8 and pc.kernel[0] <= 10 and pc.kernel[1] <= 10


def split_w(x: torch.Tensor) -> torch.Tensor:
    """(B, C, T, H, W) contiguous -> (B, C, T, H, 2, W/2 + 4): the even and odd columns of every row as two zero-padded planes
    (one HIP pass; the operand of the stem's 16-byte gather, include/advhip.h: advhip_split_w_f32)."""
    require_gpu(x)
    B, Cc, T, H, W = x.shape
    lib = _lib.load()
    wp = lib.advhip_split_w_plane_floats(W)
    if wp < 0:
        raise ValueError(f"split_w: W = {W} must be even")
    xs = torch.empty((B, Cc, T, H, 2, wp), device=x.device, dtype=torch.float32)
    check(lib.advhip_split_w_f32(ptr(x), ptr(xs), B * Cc * T * H, W, stream(x)), "split_w")
    return xs


def ensure_ktab_s2w(pc: PackedConv, thw: Tuple[int, int, int]) -> torch.Tensor:
    tab = pc.ktabs_s2w.get(thw)
    if tab is None:
        d = pc.desc(1, *thw, relu=True, algo=0, splits=1)
        tab = torch.empty((_packed_rows(d) * 2,), device=pc.w_packed.device, dtype=torch.int32)
        check(_lib.load().advhip_conv3d_s2w_build_ktab(C.byref(d), ptr(tab), stream()), "build_ktab_s2w")
        pc.ktabs_s2w[thw] = tab
    return tab


def conv3d_s2w_bn_relu_maxpool233(xs: torch.Tensor, pc: PackedConv, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The fused stem on column-parity planes xs (B, Cin, T, H, 2, W/2 + 4) of its input (split_w / tencrop_planes_u8)."""
    require_gpu(xs)
    require_gpu(out, contiguous=False)
    if xs.dim() != 6 or xs.shape[1] != pc.cin or xs.shape[4] != 2:
        raise ValueError(f"{pc.name}: expected planes (B,{pc.cin},T,H,2,W/2+4), got {tuple(xs.shape)}")
    B, _, T, H, _, wp = xs.shape
    W = 2 * (wp - 4)
    if not s2w_ok(pc, W):
        raise ValueError(f"{pc.name}: the column-parity gather does not apply to W = {W} with k/s/p {pc.kernel[2]}/{pc.stride[2]}/{pc.padding[2]}")
    d = pc.desc(B, T, H, W, True, 0, 1)
    lib = _lib.load()
    tp, hp, wpo = C.c_int32(), C.c_int32(), C.c_int32()
    check(lib.advhip_conv3d_pool_out_dims(C.byref(d), 2, 3, 3, 2, 2, 2, C.byref(tp), C.byref(hp), C.byref(wpo)), "conv3d_pool_out_dims")
    shape = (B, pc.cout, tp.value, hp.value, wpo.value)
    if min(shape) <= 0:
        raise ValueError(f"{pc.name}: input too small for conv + (2,3,3) pooling")
    y = out if out is not None else torch.empty(shape, device=xs.device, dtype=torch.float32)
    if tuple(y.shape) != shape or y.dtype != torch.float32 or y.device != xs.device:
        raise ValueError(f"{pc.name}: out {tuple(y.shape)} != {shape}")
    need = lib.advhip_conv3d_relu_maxpool233_workspace_bytes(C.byref(d))
    if need < 0:
        check(int(need), f"conv3d_relu_maxpool233_workspace_bytes[{pc.name}]")
    ws = workspace(xs.device, need)
    check(lib.advhip_conv3d_s2w_bn_relu_maxpool233_f32(C.byref(d), ptr(xs), 0, ptr(pc.w_packed), ptr(ensure_ktab_s2w(pc, (T, H, W))), ptr(pc.scale),
                                                       ptr(pc.shift), ptr(y), batch_stride(y), ptr(ws), need, stream(xs)), f"conv3d s2w+pool233[{pc.name}]")
    return y


def tencrop_planes_u8(frames: torch.Tensor, first: int, count: int, frames_per_clip: int = 16, crop: int = 224, mean: float = 114.75,
                      std: float = 57.375) -> torch.Tensor:
    """Resized uint8 frames (F, H, W, C) -> column-parity planes (count, C, frames_per_clip, crop, 2, crop/2 + 4) of crop-clips
    [first, first + count) (row = clip * 10 + crop): TenCrop, float, normalise, LoopPad and the layout permutes of
    TenCropVideoFrameDataset / _extract (src/dataset.py:175-195, src/gtransforms.py, extract_features.py:83) in one HIP pass,
    written as the operand of the stem's 16-byte gather.  Values = mil_ops.tencrop_normalize_u8's."""
    frames = frames.contiguous()
    require_gpu(frames)
    if frames.dtype != torch.uint8 or frames.dim() != 4:
        raise ValueError(f"expected uint8 (F,H,W,C), got {frames.dtype} {tuple(frames.shape)}")
    f, h, w, c = frames.shape
    n = -(-f // frames_per_clip) * 10
    if h < crop or w < crop or crop % 2 or first < 0 or count <= 0 or first + count > n:
        raise ValueError(f"tencrop_planes_u8: crop-clips [{first},{first + count}) of {n}, frames {h}x{w}, crop {crop}")
    xs = torch.empty((count, c, frames_per_clip, crop, 2, crop // 2 + 4), device=frames.device, dtype=torch.float32)
    check(_lib.load().advhip_tencrop_normalize_planes_u8(ptr(frames), ptr(xs), f, h, w, c, frames_per_clip, crop, first, count, C.c_float(mean),
                                                         C.c_float(std), stream(frames)), "tencrop_normalize_planes_u8")
    return xs


def conv3d_bn_relu_maxpool233(x: torch.Tensor, pc: PackedConv, out: Optional[torch.Tensor] = None, s2w: Optional[bool] = None) -> torch.Tensor:
    """maxpool3d(relu(conv3d(x) * scale + shift), (2,3,3), (2,2,2)) without the un-pooled activation ever reaching HBM
    (the stem of I3Res50, src/i3d.py:303-306).  Bit-identical to conv3d_bn_act(relu=True) + maxpool3d.  `s2w` (default: where
    the geometry allows): gather 16-byte pieces from column-parity planes of x (one extra pass over x, 4x fewer gather
    instructions in the conv; same results bit for bit)."""
    require_gpu(x, out, contiguous=False)
    if x.dim() != 5 or x.shape[1] != pc.cin:
        raise ValueError(f"{pc.name}: expected (B,{pc.cin},T,H,W), got {tuple(x.shape)}")
    B, _, T, H, W = x.shape
    d = pc.desc(B, T, H, W, True, 0, 1)
    lib = _lib.load()
    tp, hp, wp = C.c_int32(), C.c_int32(), C.c_int32()
    check(lib.advhip_conv3d_pool_out_dims(C.byref(d), 2, 3, 3, 2, 2, 2, C.byref(tp), C.byref(hp), C.byref(wp)), "conv3d_pool_out_dims")
    shape = (B, pc.cout, tp.value, hp.value, wp.value)
    if min(shape) <= 0:
        raise ValueError(f"{pc.name}: input {tuple(x.shape)} too small for conv + (2,3,3) pooling")
    y = out if out is not None else torch.empty(shape, device=x.device, dtype=torch.float32)
    if tuple(y.shape) != shape or y.dtype != torch.float32 or y.device != x.device:
        raise ValueError(f"{pc.name}: out {tuple(y.shape)} != {shape}")
    ktab = ensure_ktab(pc, (T, H, W))
    need = lib.advhip_conv3d_relu_maxpool233_workspace_bytes(C.byref(d))
    if need < 0:
        check(int(need), f"conv3d_relu_maxpool233_workspace_bytes[{pc.name}]")
    ws = workspace(x.device, need)
    if s2w is None:
        s2w = STEM_S2W
    if s2w and s2w_ok(pc, W) and x.is_contiguous():
        return conv3d_s2w_bn_relu_maxpool233(split_w(x), pc, out=y)
    xbs, ybs = batch_stride(x), batch_stride(y)
    check(lib.advhip_conv3d_bn_relu_maxpool233_f32(C.byref(d), ptr(x), xbs, ptr(pc.w_packed), ptr(ktab), ptr(pc.scale), ptr(pc.shift),
                                                   ptr(y), ybs, ptr(ws), need, stream()), f"conv3d+pool233[{pc.name}]")
    return y


PIXEL_MEAN, PIXEL_STD = 114.75, 57.375  # GroupNormalize constants, src/dataset.py:180-181


# how forward_frames feeds the stem from resized uint8 frames: "planes" (default where the geometry allows): one TenCrop /
# normalise pass writing column-parity planes + the 16-byte-gather stem (features equal the fp32 pipeline's bit for bit);
# "taps": the stem kernel reads whole pixels itself (one 4-byte gather per tap, no fp32 tensor at all); "bytes": one byte
# gather per (channel, tap)
U8_STEM_FORM = os.environ.get("ADV_U8_STEM", "planes")


def readable_bytes(t: torch.Tensor) -> int:
    """Bytes of the tensor's allocation from its first element on."""
    return t.untyped_storage().nbytes() - t.storage_offset() * t.element_size()


def with_slack(frames: torch.Tensor, slack: int = 4) -> torch.Tensor:
    """`frames` in an allocation that extends `slack` bytes past its last element (a copy only if it does not already)."""
    if readable_bytes(frames) >= frames.numel() + slack:
        return frames
    buf = torch.empty((frames.numel() + slack,), device=frames.device, dtype=torch.uint8)
    buf[: frames.numel()].copy_(frames.reshape(-1))
    return buf[: frames.numel()].view(frames.shape)


def ensure_u8_taps_tables(pc: "PackedConv", frame_hw: Tuple[int, int], clip_thw: Tuple[int, int, int], mean: float = PIXEL_MEAN):
    key = ("taps", tuple(frame_hw), tuple(clip_thw), float(mean))
    cache = pc.__dict__.setdefault("_u8_tables", {})
    tabs = cache.get(key)
    if tabs is None:
        d = pc.desc(1, *clip_thw, True, 0, 1)
        lib = _lib.load()
        nk, nf, nw = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib.advhip_conv3d_u8_taps_table_sizes(C.byref(d), C.byref(nk), C.byref(nf), C.byref(nw)), "conv3d_u8_taps_table_sizes")
        dev = pc.w_packed.device
        ktab = torch.empty((nk.value,), device=dev, dtype=torch.int32)
        corr = torch.empty((nf.value,), device=dev, dtype=torch.float32)
        wt = torch.empty((nw.value,), device=dev, dtype=torch.float32)
        check(lib.advhip_conv3d_u8_taps_build_tables(C.byref(d), frame_hw[0], frame_hw[1], ptr(pc.w_packed), C.c_float(mean), ptr(ktab),
                                                     ptr(corr), ptr(wt), stream(dev)), f"conv3d_u8_taps_build_tables[{pc.name}]")
        tabs = cache[key] = (ktab, corr, wt)
    return tabs


def ensure_u8_tables(pc: "PackedConv", frame_hw: Tuple[int, int], clip_thw: Tuple[int, int, int], mean: float = PIXEL_MEAN):
    """Gather / border tables of the uint8-frame stem for frames of (FH, FW) and clips of (T, crop, crop); cached on the conv
    (built on the current stream: callers that fork streams build them first, like the other lazily built tables)."""
    key = (tuple(frame_hw), tuple(clip_thw), float(mean))
    cache = pc.__dict__.setdefault("_u8_tables", {})
    tabs = cache.get(key)
    if tabs is None:
        d = pc.desc(1, *clip_thw, True, 0, 1)
        lib = _lib.load()
        nk, nf = C.c_int64(), C.c_int64()
        check(lib.advhip_conv3d_u8_table_sizes(C.byref(d), C.byref(nk), C.byref(nf)), "conv3d_u8_table_sizes")
        dev = pc.w_packed.device
        ktab = torch.empty((nk.value,), device=dev, dtype=torch.int32)
        corr = torch.empty((nf.value,), device=dev, dtype=torch.float32)
        check(lib.advhip_conv3d_u8_build_tables(C.byref(d), frame_hw[0], frame_hw[1], ptr(pc.w_packed), C.c_float(mean), ptr(ktab),
                                                ptr(corr), stream(dev)), f"conv3d_u8_build_tables[{pc.name}]")
        tabs = cache[key] = (ktab, corr)
    return tabs


def conv3d_u8_tencrop_bn_relu_maxpool233(frames: torch.Tensor, pc: "PackedConv", first: int, count: int, frames_per_clip: int = 16,
                                         crop: int = 224, out: Optional[torch.Tensor] = None, mean: float = PIXEL_MEAN,
                                         std: float = PIXEL_STD) -> torch.Tensor:
    """The stem (conv1 + bn1 + relu + maxpool1, src/i3d.py:303-306) of crop-clips [first, first + count) of a video given as
    resized uint8 frames (F, FH, FW, 3): row = clip * 10 + crop (TenCrop order).  TenCrop, float conversion and
    (x - mean) / std happen in the conv's load stage (src/gtransforms.py:29-38,57-73, extract_features.py:83-89)."""
    require_gpu(frames)
    require_gpu(out, contiguous=False)
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != pc.cin:
        raise ValueError(f"{pc.name}: expected uint8 (F,H,W,{pc.cin}) frames, got {frames.dtype} {tuple(frames.shape)}")
    F, FH, FW, _ = frames.shape
    if F % frames_per_clip or FH < crop or FW < crop:
        raise ValueError(f"{pc.name}: {F} frames of {FH}x{FW} are not whole {frames_per_clip}-frame clips of at least {crop}x{crop}")
    if count <= 0 or first < 0 or first + count > F // frames_per_clip * 10:
        raise ValueError(f"{pc.name}: crop-clips [{first}, {first + count}) outside the video's {F // frames_per_clip * 10}")
    d = pc.desc(count, frames_per_clip, crop, crop, True, 0, 1)
    lib = _lib.load()
    tp, hp, wp = C.c_int32(), C.c_int32(), C.c_int32()
    check(lib.advhip_conv3d_pool_out_dims(C.byref(d), 2, 3, 3, 2, 2, 2, C.byref(tp), C.byref(hp), C.byref(wp)), "conv3d_pool_out_dims")
    shape = (count, pc.cout, tp.value, hp.value, wp.value)
    if min(shape) <= 0:
        raise ValueError(f"{pc.name}: clips of ({frames_per_clip},{crop},{crop}) too small for conv + (2,3,3) pooling")
    y = out if out is not None else torch.empty(shape, device=frames.device, dtype=torch.float32)
    if tuple(y.shape) != shape or y.dtype != torch.float32 or y.device != frames.device:
        raise ValueError(f"{pc.name}: out {tuple(y.shape)} != {shape}")
    need = lib.advhip_conv3d_relu_maxpool233_workspace_bytes(C.byref(d))
    if need < 0:
        check(int(need), f"conv3d_relu_maxpool233_workspace_bytes[{pc.name}]")
    ws = workspace(frames.device, need)
    if U8_STEM_FORM in ("taps", "planes") and pc.cin == 3 and pc.cout == 64:
        frames = with_slack(frames)
        ktab, corr, wt = ensure_u8_taps_tables(pc, (FH, FW), (frames_per_clip, crop, crop), mean)
        check(lib.advhip_conv3d_u8_taps_tencrop_bn_relu_maxpool233_f32(C.byref(d), ptr(frames), F, FH, FW, readable_bytes(frames), first, ptr(wt),
                                                                       ptr(ktab), ptr(corr), ptr(pc.scale), ptr(pc.shift), C.c_float(std),
                                                                       ptr(y), batch_stride(y), ptr(ws), need, stream()),
              f"conv3d_u8_taps+pool233[{pc.name}]")
        return y
    ktab, corr = ensure_u8_tables(pc, (FH, FW), (frames_per_clip, crop, crop), mean)
    check(lib.advhip_conv3d_u8_tencrop_bn_relu_maxpool233_f32(C.byref(d), ptr(frames), F, FH, FW, first, ptr(pc.w_packed), ptr(ktab),
                                                              ptr(corr), ptr(pc.scale), ptr(pc.shift),
                                                              C.c_float(std), ptr(y), batch_stride(y), ptr(ws), need, stream()),
          f"conv3d_u8+pool233[{pc.name}]")
    return y


def conv3d_bn_act_maxpool211(x: torch.Tensor, pc: PackedConv, relu: bool = True, residual: Optional[torch.Tensor] = None,
                             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """maxpool3d(act(conv3d(x) * scale + shift (+ residual)), (2,1,1), (2,1,1)) in one launch, for a 1x1x1 stride-1 conv
    (layer1's last conv3 + maxpool2, src/i3d.py:111-121, 309).  `residual`: un-pooled shape, contiguous."""
    require_gpu(x, out, contiguous=False)
    require_gpu(residual)
    if x.dim() != 5 or x.shape[1] != pc.cin:
        raise ValueError(f"{pc.name}: expected (B,{pc.cin},T,H,W), got {tuple(x.shape)}")
    B, _, T, H, W = x.shape
    if T < 2:
        raise ValueError(f"{pc.name}: T={T} smaller than the temporal pooling window")
    shape = (B, pc.cout, T // 2, H, W)
    y = out if out is not None else torch.empty(shape, device=x.device, dtype=torch.float32)
    if tuple(y.shape) != shape or y.dtype != torch.float32 or y.device != x.device:
        raise ValueError(f"{pc.name}: out {tuple(y.shape)} != {shape}")
    if residual is not None and tuple(residual.shape) != (B, pc.cout, T, H, W):
        raise ValueError(f"{pc.name}: residual {tuple(residual.shape)} != un-pooled output {(B, pc.cout, T, H, W)}")
    ktab = ensure_ktab(pc, (T, H, W))
    d = pc.desc(B, T, H, W, relu, 0, 1)
    check(_lib.load().advhip_conv3d_bn_act_maxpool211_f32(C.byref(d), ptr(x), batch_stride(x), ptr(pc.w_packed), ptr(ktab), ptr(pc.scale),
                                                          ptr(pc.shift), ptr(residual), ptr(y), batch_stride(y), stream()),
          f"conv3d+pool211[{pc.name}]")
    return y


def maxpool3d(x: torch.Tensor, kernel, stride, out: Optional[torch.Tensor] = None, padding=(0, 0, 0)) -> torch.Tensor:
    """`out`: optional destination, contiguous or a channel slice of a wider NCDHW tensor (un-padded pooling only)."""
    require_gpu(x)
    require_gpu(out, contiguous=False)
    B, Cc, T, H, W = x.shape
    k, s = _triple(kernel), _triple(stride)
    p = _triple(padding)
    if any(p):
        if out is not None:
            raise ValueError("maxpool3d: `out` is not supported together with padding")
        to, ho, wo = conv_out_dims((T, H, W), k, s, p)
        y = torch.empty((B, Cc, to, ho, wo), device=x.device, dtype=torch.float32)
        check(_lib.load().advhip_maxpool3d_padded_f32(ptr(x), ptr(y), B, Cc, T, H, W, *k, *s, *p, stream()), "maxpool3d_padded")
        return y
    to, ho, wo = conv_out_dims((T, H, W), k, s, (0, 0, 0))
    y = out if out is not None else torch.empty((B, Cc, to, ho, wo), device=x.device, dtype=torch.float32)
    if tuple(y.shape) != (B, Cc, to, ho, wo):
        raise ValueError(f"maxpool3d: out {tuple(y.shape)} != {(B, Cc, to, ho, wo)}")
    ybs = batch_stride(y)
    check(_lib.load().advhip_maxpool3d_strided_f32(ptr(x), ptr(y), 0 if ybs == Cc * to * ho * wo else ybs, B, Cc, T, H, W, *k, *s, stream()),
          "maxpool3d")
    return y


FUSE_AVGPOOL = os.environ.get("ADV_I3D_FUSE_AVGPOOL", "1") == "1"


def avgpool_fusable(pc: PackedConv, thw: Tuple[int, int, int]) -> bool:
    """conv3d_bn_act_avgpool takes this conv on (T,H,W) inputs (include/advhip.h: advhip_conv3d_epilogue.avgpool_out)."""
    return (pc.kernel == (1, 1, 1) and pc.stride == (1, 1, 1) and pc.padding == (0, 0, 0) and thw[0] * thw[1] * thw[2] <= 128
            and pc.cout % 64 == 0 and pc.cin % 32 == 0)


def conv3d_bn_act_avgpool(x: torch.Tensor, pc: PackedConv, relu: bool = True, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """global_avgpool(act(conv3d(x) * scale + shift (+ residual))) -> (B, Cout, 1, 1, 1) in ONE launch: the conv's own output is
    never written.  Bit for bit global_avgpool(conv3d_bn_act(...)) (src/i3d.py:111-121 + 314)."""
    require_gpu(x, residual)
    B, _, T, H, W = x.shape
    if x.shape[1] != pc.cin or not avgpool_fusable(pc, (T, H, W)):
        raise ValueError(f"{pc.name}: conv + mean in one launch needs a 1x1x1 stride-1 conv on <= 128 positions, got {tuple(x.shape)}")
    if residual is not None and tuple(residual.shape) != (B, pc.cout, T, H, W):
        raise ValueError(f"{pc.name}: residual {tuple(residual.shape)} != output {(B, pc.cout, T, H, W)}")
    y = torch.empty((B, pc.cout, 1, 1, 1), device=x.device, dtype=torch.float32)
    d = pc.desc(B, T, H, W, relu, _lib.ALGO_DMA2_BASE + _lib.ALGO_IGEMM_128x64, 1)
    ep = _lib.ConvEpilogue(None, None, None, None, None, None, 0, ptr(y))
    check(_lib.load().advhip_conv3d_bn_act_ex_f32(C.byref(d), ptr(x), 0, ptr(pc.w_packed), ptr(ensure_ktab(pc, (T, H, W))), ptr(pc.scale),
                                                  ptr(pc.shift), ptr(residual), None, 0, C.byref(ep), None, 0, stream()), f"conv3d+mean[{pc.name}]")
    return y


def global_avgpool(x: torch.Tensor) -> torch.Tensor:
    """(B,C,T,H,W) -> (B,C,1,1,1)"""
    require_gpu(x)
    B, Cc = x.shape[:2]
    n = x[0, 0].numel()
    y = torch.empty((B, Cc, 1, 1, 1), device=x.device, dtype=torch.float32)
    check(_lib.load().advhip_global_avgpool_f32(ptr(x), ptr(y), B * Cc, n, stream()), "global_avgpool")
    return y


def bgemm(a: torch.Tensor, b: torch.Tensor, *, alpha: float = 1.0, bias_m: Optional[torch.Tensor] = None,
          bias_n: Optional[torch.Tensor] = None, act: int = 0, residual: Optional[torch.Tensor] = None, beta: float = 1.0,
          ln: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i] = epi(alpha * a[i] @ b[i]) on the fp32 MFMA (include/advhip.h: advhip_bgemm_f32).  a: (batch, M, K), b: (batch, K, N)
    -- any views whose last two dims have one unit stride each (transposes and channel slices included; batch stride 0 =
    broadcast); 2-D operands are one batch.  `ln` = (u[M], mu[batch*N], rs[batch*N]): the LayerNorm fold of the header.
    `act`: 0 none, 1 ReLU, 2 GELU(erf).  `residual`: like out."""
    if a.dim() == 2:
        a = a.unsqueeze(0)
    if b.dim() == 2:
        b = b.unsqueeze(0)
    require_gpu(a, b, out, residual, contiguous=False)
    require_gpu(bias_m, bias_n)
    batch = max(a.shape[0], b.shape[0])
    if a.dim() != 3 or b.dim() != 3 or a.shape[2] != b.shape[1] or a.shape[0] not in (1, batch) or b.shape[0] not in (1, batch):
        raise ValueError(f"bgemm: incompatible operands {tuple(a.shape)} x {tuple(b.shape)}")
    if a.dtype != torch.float32 or b.dtype != torch.float32:
        raise _lib.HipExtensionError("bgemm computes in fp32")
    M, K, N = a.shape[1], a.shape[2], b.shape[2]
    y = out if out is not None else torch.empty((batch, M, N), device=a.device, dtype=torch.float32)
    if y.dim() == 2:
        y = y.unsqueeze(0)
    if tuple(y.shape) != (batch, M, N) or y.dtype != torch.float32:
        raise ValueError(f"bgemm: out {tuple(y.shape)} != {(batch, M, N)}")
    if residual is not None:
        if residual.dim() == 2:
            residual = residual.unsqueeze(0)
        if tuple(residual.shape) != (batch, M, N) or residual.stride() != y.stride():
            raise ValueError("bgemm: residual must have out's shape and strides")

    def strides(t, nb):
        sb = t.stride(0) if t.shape[0] == nb and nb > 1 else 0
        return sb, t.stride(1), t.stride(2)

    sa, sb_, sc = strides(a, batch), strides(b, batch), strides(y, batch)
    if M > 1 and K > 1 and 1 not in sa[1:] or N > 1 and K > 1 and 1 not in sb_[1:]:
        raise ValueError(f"bgemm: each operand needs a unit stride in its last two dims (got {a.stride()}, {b.stride()})")
    # a size-1 dim's stride is arbitrary: give the kernel a consistent unit stride
    sa = (sa[0], 1 if M == 1 and sa[2] != 1 else sa[1], 1 if K == 1 and sa[1] != 1 else sa[2])
    sb_ = (sb_[0], 1 if K == 1 and sb_[2] != 1 else sb_[1], 1 if N == 1 and sb_[1] != 1 else sb_[2])
    d = _lib.GemmDesc(M, N, K, batch, *sa, *sb_, *sc, float(alpha), float(beta), int(act), ptr(bias_m), ptr(bias_n),
                      ptr(ln[0]) if ln else None, ptr(ln[1]) if ln else None, ptr(ln[2]) if ln else None, ptr(residual))
    if ln is not None:
        require_gpu(*ln)
        if ln[0].numel() != M or ln[1].numel() != batch * N or ln[2].numel() != batch * N:
            raise ValueError("bgemm: LayerNorm fold vectors must be u[M], mu[batch*N], rs[batch*N]")
    check(_lib.load().advhip_bgemm_f32(C.byref(d), ptr(a), ptr(b), ptr(y), stream()), "bgemm")
    return y if out is None or out.dim() == 3 else out


def softmax_rows(x: torch.Tensor, scale: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(x * scale) over the last (contiguous) dim, one wavefront per row; out may be x."""
    require_gpu(x, out)
    y = out if out is not None else torch.empty_like(x)
    n = x.shape[-1]
    check(_lib.load().advhip_softmax_rows_f32(ptr(x), ptr(y), x.numel() // n, n, C.c_float(scale), stream()), "softmax_rows")
    return y


# (M, N) -> (tile id, K slices) of advhip_gemm_nt_rowsum_f32 at K = 10 240, measured on one MI355X (tools/tune_gemm_nt.py,
# profiles/r03_gemm_nt_tune.txt): the weight-gradient shapes of the MGFN scorer at its training batch.  Other shapes: heuristic.
GEMM_NT_TUNED: Dict[Tuple[int, int, int], Tuple[int, int]] = {
    (4096, 1024, 10240): (2, 2), (1024, 4096, 10240): (3, 3), (1024, 1024, 10240): (1, 4), (1024, 3072, 10240): (3, 4),
}


def gemm_nt_choice(M: int, N: int, K: int) -> Tuple[int, int]:
    hit = GEMM_NT_TUNED.get((M, N, K))
    if hit is not None:
        return hit
    tiles = -(-M // 64) * -(-N // 64)
    if tiles < 64:  # few output tiles (the 64- / 128-channel layers): 16 K slices to slabs measured best for every such shape
        return 1, max(1, min(NT_SMALL_SPLITS, K // 256))  # (profiles/r03_gemm_nt_tune.txt: 32-35 us vs 36-58 for 40)
    return 0, max(1, min(64, 1024 // max(tiles, 1), K // 256))


NT_SMALL_SPLITS = 16


def gemm_nt_is_small(M: int, N: int) -> bool:
    """gemm_nt takes the slab + one-reduce form for this output (a handful of 64 x 64 tiles): what gemm_nt_group batches."""
    return -(-M // 64) * -(-N // 64) < 64


def gemm_nt_group(items):
    """[(a (M,K), b (N,K), rowsum: bool), ...] -> [(a @ b^T, row sums of a or None), ...]: every small product of the list in ONE
    launch (K slices to one slab matrix) + ONE reduce launch (include/advhip.h: advhip_gemm_nt_group_slabs_f32) -- bit for bit what
    gemm_nt returns for each of them alone.  All items contract over the same K."""
    if not items:
        return []
    K = items[0][0].shape[1]
    dev = items[0][0].device
    offs, total = [], 0
    for a, b, rs in items:
        require_gpu(a, b, contiguous=False)
        if a.dim() != 2 or b.dim() != 2 or a.shape[1] != K or b.shape[1] != K or a.stride(1) != 1 or b.stride(1) != 1:
            raise ValueError(f"gemm_nt_group: need (M,K) and (N,K) operands over one K = {K} with unit inner stride, got {tuple(a.shape)} / {tuple(b.shape)}")
        offs.append(total)
        total += -(-(a.shape[0] * b.shape[0] + (a.shape[0] if rs else 0)) // 64) * 64
    splits = max(1, min(NT_SMALL_SPLITS, K // 256))
    slabs = torch.empty((splits, total), device=dev, dtype=torch.float32)
    res = torch.empty((total,), device=dev, dtype=torch.float32)
    arr = (_lib.NtItem * len(items))()
    base = slabs.data_ptr()
    for it, (a, b, rs), off in zip(arr, items, offs):
        M, N = a.shape[0], b.shape[0]
        it.A, it.B, it.C = a.data_ptr(), b.data_ptr(), base + off * 4
        it.rowsum = base + (off + M * N) * 4 if rs else None
        it.M, it.N, it.lda, it.ldb = M, N, a.stride(0), b.stride(0)
    lib = _lib.load()
    check(lib.advhip_gemm_nt_group_slabs_f32(arr, len(items), K, splits, total, stream(dev)), "gemm_nt_group")
    check(lib.advhip_sum_slabs_f32(ptr(slabs), ptr(res), total, splits, total, stream(dev)), "sum_slabs")
    out = []
    for (a, b, rs), off in zip(items, offs):
        M, N = a.shape[0], b.shape[0]
        out.append((res[off : off + M * N].view(M, N), res[off + M * N : off + M * N + M] if rs else None))
    return out


def gemm_nt(a: torch.Tensor, b: torch.Tensor, splits: int = 0, rowsum: bool = False, tile: int = 0,
            reduce_in_kernel: Optional[bool] = None):
    """a (M, K) @ b (N, K)^T -> (M, N) for two k-contiguous operands (row pitch = stride(0)); K % 16 == 0.  The weight
    gradient of a GEMM-shaped layer with (channel, position) activations (include/advhip.h: advhip_gemm_nt_rowsum_f32).
    `splits`: K slices (0 = the measured / heuristic choice), summed inside the launch by the last-arriving workgroup of
    every tile (slice order: run-to-run bit-identical).  `rowsum`: also return the row sums of `a` (M,) -- the bias gradient
    of the same layer, out of the same launch."""
    require_gpu(a, b, contiguous=False)
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1] or a.stride(1) != 1 or b.stride(1) != 1:
        raise ValueError(f"gemm_nt: need (M,K) and (N,K) with unit inner stride, got {tuple(a.shape)} {a.stride()} / {tuple(b.shape)} {b.stride()}")
    M, K = a.shape
    N = b.shape[0]
    if splits <= 0:
        t, splits = gemm_nt_choice(M, N, K)
        tile = tile or t
    lib = _lib.load()
    if splits > 1 and reduce_in_kernel is None:
        reduce_in_kernel = -(-M // 64) * -(-N // 64) >= 64
    if splits > 1 and not reduce_in_kernel:
        # few output tiles: slices to slabs, product and row sums reduced by ONE further launch (the in-kernel form would
        # leave the whole sum to the last arriver of each of a handful of tiles)
        n = M * N + (M if rowsum else 0)
        slabs = torch.empty((splits, n), device=a.device, dtype=torch.float32)
        res = torch.empty((n,), device=a.device, dtype=torch.float32)
        check(lib.advhip_gemm_nt_slabs_f32(ptr(a), ptr(b), ptr(slabs), M, N, K, a.stride(0), b.stride(0), splits, tile, int(rowsum), stream(a)),
              "gemm_nt_slabs")
        check(lib.advhip_sum_slabs_f32(ptr(slabs), ptr(res), n, splits, n, stream(a)), "sum_slabs")
        out = res[: M * N].view(M, N)
        return (out, res[M * N :]) if rowsum else out
    out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    rs = torch.empty((M,), device=a.device, dtype=torch.float32) if rowsum else None
    need = lib.advhip_gemm_nt_workspace_bytes(M, N, splits, tile) if splits > 1 else 0
    ws = zeroed_workspace(a.device, need)  # (arrival counters: zero on entry, left zero by the launch)
    check(lib.advhip_gemm_nt_reduced_f32(ptr(a), ptr(b), ptr(out), ptr(rs), M, N, K, a.stride(0), b.stride(0), N, splits, tile, ptr(ws), need,
                                         stream(a)), "gemm_nt")
    return (out, rs) if rowsum else out
