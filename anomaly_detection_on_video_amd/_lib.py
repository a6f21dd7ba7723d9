"""ctypes binding of libadvhip.so (include/advhip.h).

The shared library is built in-tree by `anomaly_detection_on_video_amd.build` (hipcc,
--offload-arch=gfx950) and is the ONLY compute path of this package for the ops it covers:
there is no eager / CPU fallback.  If the library is missing, or a tensor is not on a GPU, the
callers raise -- loudly -- instead of silently computing somewhere else.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libadvhip.so")

ALGO_AUTO = 0
ALGO_IGEMM_128x128 = 1
ALGO_IGEMM_128x64 = 2
ALGO_IGEMM_64x64 = 3
ALGO_IGEMM_64x128 = 4
ALGO_IGEMM_128x128x32 = 5
ALGO_IGEMM_128x64x32 = 6
ALGO_IGEMM_64x64x32 = 7
ALGO_IGEMM_64x128x32 = 8
ALGO_IGEMM_256x64 = 9  # 256 x 64 x 16, four waves stacked along M; the 2-deep LDS-DMA family only
ALGO_FAST_BASE = 32  # + tile id: scalar-offset / tap-mask gather (<= 32 taps)
IGEMM_ALGOS = (1, 2, 3, 4, 5, 6, 7, 8)
FAST_ALGOS = tuple(ALGO_FAST_BASE + a for a in IGEMM_ALGOS)
ALGO_DMA_BASE = 64  # + tile id: fast gather + LDS-DMA staging into a 3-deep ring (no 128x128x32)
DMA_ALGOS = tuple(ALGO_DMA_BASE + a for a in (1, 2, 3, 4, 6, 7, 8))
ALGO_DMA4_BASE = 96  # + tile id 2..4: 4-deep ring
DMA4_ALGOS = tuple(ALGO_DMA4_BASE + a for a in (2, 3, 4))
ALGO_BF16X3_BASE = 128  # + tile id 5 (128x128x32) / 6 (128x64x32): opt-in split-bf16 arithmetic (3 bf16 MFMAs per fragment)
BF16X3_ALGOS = tuple(ALGO_BF16X3_BASE + a for a in (5, 6))
ALGO_DMA2_BASE = 160  # + tile id: LDS-DMA kernel, 2-deep ring (less LDS, more resident workgroups)
DMA2_ALGOS = tuple(ALGO_DMA2_BASE + a for a in (1, 2, 3, 4, 6, 7, 8, 9))
ALGO_TSPAN_128x64 = 192  # (kt,1,1) convs: 128x64 tile of 2 or 4 frames x flattened spatial positions
ALGO_MIXED_128x64 = 200  # unsplit 1x1x1 stride-1 convs: 128x64 tiles + 64x64 tiles for the last, partial round of workgroups
ALGO_PERSIST_BASE = 224  # + tile id (2: 128x64, 3: 64x64) + 8 * (workgroups per CU - 1), 1..3: persistent wave-specialised kernel (opt-in), unsplit 1x1x1 stride-1 convs, K >= 64
PERSIST_ALGOS = tuple(ALGO_PERSIST_BASE + t + 8 * (w - 1) for w in (1, 2, 3) for t in (2, 3))


def algo_tile(algo: int):
    """(BM, BN, BK) of an implicit-GEMM algorithm id."""
    if algo in (ALGO_TSPAN_128x64, ALGO_MIXED_128x64):
        return (128, 64, 16)
    if ALGO_PERSIST_BASE <= algo < ALGO_PERSIST_BASE + 32:
        return (128 if (algo - ALGO_PERSIST_BASE) & 7 == 2 else 64, 64, 16)
    if algo == ALGO_DMA2_BASE + ALGO_IGEMM_256x64:
        return (256, 64, 16)
    if algo >= ALGO_DMA2_BASE:
        algo -= ALGO_DMA2_BASE
    if algo >= ALGO_BF16X3_BASE:
        algo -= ALGO_BF16X3_BASE
    if algo >= ALGO_DMA4_BASE:
        algo -= ALGO_DMA4_BASE
    if algo >= ALGO_DMA_BASE:
        algo -= ALGO_DMA_BASE
    if algo >= ALGO_FAST_BASE:
        algo -= ALGO_FAST_BASE
    t = (algo - 1) & 3
    return (128 if t in (0, 1) else 64, 128 if t in (0, 3) else 64, 32 if algo >= 5 else 16)


class HipExtensionError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "B", "Cin", "T", "H", "W", "Cout", "kt", "kh", "kw", "st", "sh", "sw", "pt", "ph", "pw",
        "relu", "algo", "splits")]


_P = C.c_void_p
_I = C.c_int32
_L = C.c_int64


class ConvEpilogue(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("y_preact", "dact_z", "ln_u", "ln_mu", "ln_rs", "splitk_counters")] + [("splitk_counter_bytes", C.c_int64)]
                + [("avgpool_out", C.c_void_p)])


class ColsumItem(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int32), ("period", C.c_int32)]


class AdamItem(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("param", "grad", "exp_avg", "exp_avg_sq", "step")] + [("n", C.c_int64)]


class NtItem(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("A", "B", "C", "rowsum")] + [("M", C.c_int32), ("N", C.c_int32), ("lda", C.c_int64), ("ldb", C.c_int64)]


class GemmDesc(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("M", "N", "K", "batch")]
                + [(n, C.c_int64) for n in ("sAb", "sAm", "sAk", "sBb", "sBk", "sBn", "sCb", "sCm", "sCn")]
                + [("alpha", C.c_float), ("beta", C.c_float), ("act", C.c_int32)]
                + [(n, C.c_void_p) for n in ("bias_m", "bias_n", "ln_u", "ln_mu", "ln_rs", "residual")])


# name -> (restype, argtypes); every symbol include/advhip.h declares
SIGNATURES = {
    "advhip_abi_version": (C.c_int, []),
    "advhip_last_error": (C.c_char_p, []),
    "advhip_target_arch": (C.c_char_p, []),
    "advhip_conv3d_out_dims": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "advhip_conv3d_packed_rows": (C.c_int, [C.POINTER(ConvDesc)]),
    "advhip_conv3d_pack_weight_f32": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P]),
    "advhip_conv3d_pack_weight_bf16x3": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P]),
    "advhip_conv3d_build_ktab": (C.c_int, [C.POINTER(ConvDesc), _P, _P]),
    "advhip_bn_fold_f32": (C.c_int, [_P, _P, _P, _P, C.c_float, _I, _P, _P, _P]),
    "advhip_conv3d_workspace_bytes": (_L, [C.POINTER(ConvDesc)]),
    "advhip_conv3d_bn_act_f32": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "advhip_conv3d_bn_act_strided_f32": (C.c_int, [C.POINTER(ConvDesc), _P, _L, _P, _P, _P, _P, _P, _P, _L, _P, _L, _P]),
    "advhip_conv3d_pool_out_dims": (C.c_int, [C.POINTER(ConvDesc)] + [_I] * 6 + [C.POINTER(_I)] * 3),
    "advhip_conv3d_relu_maxpool233_workspace_bytes": (_L, [C.POINTER(ConvDesc)]),
    "advhip_conv3d_bn_relu_maxpool233_f32": (C.c_int, [C.POINTER(ConvDesc), _P, _L, _P, _P, _P, _P, _P, _L, _P, _L, _P]),
    "advhip_split_w_plane_floats": (_I, [_I]),
    "advhip_split_w_f32": (C.c_int, [_P, _P, _L, _I, _P]),
    "advhip_conv3d_s2w_build_ktab": (C.c_int, [C.POINTER(ConvDesc), _P, _P]),
    "advhip_conv3d_s2w_bn_relu_maxpool233_f32": (C.c_int, [C.POINTER(ConvDesc), _P, _L, _P, _P, _P, _P, _P, _L, _P, _L, _P]),
    "advhip_conv3d_u8_table_sizes": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "advhip_conv3d_u8_build_tables": (C.c_int, [C.POINTER(ConvDesc), _I, _I, _P, C.c_float, _P, _P, _P]),
    "advhip_conv3d_u8_taps_table_sizes": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "advhip_conv3d_u8_taps_build_tables": (C.c_int, [C.POINTER(ConvDesc), _I, _I, _P, C.c_float, _P, _P, _P, _P]),
    "advhip_conv3d_u8_taps_tencrop_bn_relu_maxpool233_f32": (C.c_int, [C.POINTER(ConvDesc), _P, _L, _I, _I, _L, _L, _P, _P, _P, _P, _P,
                                                                       C.c_float, _P, _L, _P, _L, _P]),
    "advhip_conv3d_u8_tencrop_bn_relu_maxpool233_f32": (C.c_int, [C.POINTER(ConvDesc), _P, _L, _I, _I, _L, _P, _P, _P, _P, _P,
                                                                  C.c_float, _P, _L, _P, _L, _P]),
    "advhip_conv3d_bn_act_maxpool211_f32": (C.c_int, [C.POINTER(ConvDesc), _P, _L, _P, _P, _P, _P, _P, _P, _L, _P]),
    "advhip_bgemm_f32": (C.c_int, [C.POINTER(GemmDesc), _P, _P, _P, _P]),
    "advhip_gemm_nt_f32": (C.c_int, [_P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _L, _P]),
    "advhip_gemm_nt_rowsum_f32": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _L, _I, _P]),
    "advhip_gemm_nt_workspace_bytes": (_L, [_I, _I, _I, _I]),
    "advhip_gemm_nt_reduced_f32": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _I, _P, _L, _P]),
    "advhip_gemm_nt_slabs_f32": (C.c_int, [_P, _P, _P, _I, _I, _I, _L, _L, _I, _I, _I, _P]),
    "advhip_sum_slabs_f32": (C.c_int, [_P, _P, _L, _I, _L, _P]),
    "advhip_gemm_nt_group_slabs_f32": (C.c_int, [C.POINTER(NtItem), _I, _I, _I, _L, _P]),
    "advhip_softmax_rows_f32": (C.c_int, [_P, _P, _L, _I, C.c_float, _P]),
    "advhip_conv3d_bn_act_ex_f32": (C.c_int, [C.POINTER(ConvDesc), _P, _L, _P, _P, _P, _P, _P, _P, _L, C.POINTER(ConvEpilogue), _P, _L, _P]),
    "advhip_chan_stats_f32": (C.c_int, [_P, _P, _P, _I, _L, C.c_float, _P]),
    "advhip_maxpool3d_f32": (C.c_int, [_P, _P] + [_I] * 11 + [_P]),
    "advhip_maxpool3d_padded_f32": (C.c_int, [_P, _P] + [_I] * 14 + [_P]),
    "advhip_maxpool3d_strided_f32": (C.c_int, [_P, _P, _L] + [_I] * 11 + [_P]),
    "advhip_global_avgpool_f32": (C.c_int, [_P, _P, _L, _I, _P]),
    "advhip_chan_layernorm_fwd_f32": (C.c_int, [_P] * 6 + [_I, _L, C.c_float, _P]),
    "advhip_chan_layernorm_bwd_partial_rows": (_L, [_L]),
    "advhip_chan_layernorm_bwd_f32": (C.c_int, [_P] * 8 + [_I, _L, C.c_float, _P]),
    "advhip_bn_rows_fwd_f32": (C.c_int, [_P] * 6 + [_I, _L, C.c_float, _P]),
    "advhip_bn_rows_bwd_f32": (C.c_int, [_P] * 8 + [_I, _L, C.c_float, _P]),
    "advhip_bn_rows_fwd_running_f32": (C.c_int, [_P] * 8 + [C.c_float, _I, _L, C.c_float, _P]),
    "advhip_bn_rows_bwd_add_f32": (C.c_int, [_P] * 9 + [_I, _L, C.c_float, _P]),
    "advhip_chan_layernorm_bwd_add_f32": (C.c_int, [_P] * 8 + [_I, _L, C.c_float, _P]),
    "advhip_conv1d_pack_weight_dx_f32": (C.c_int, [_P, _P, _I, _I, _I, _P]),
    "advhip_amp_combine_fwd_f32": (C.c_int, [_P, _P, _P, _L, _P, _P, C.c_float, _P, _I, _L, _I, _P]),
    "advhip_amp_combine_bwd_f32": (C.c_int, [_P, _P, _L, C.c_float, _P, _P, _P, _P, _I, _L, _I, _P]),
    "advhip_colsum_f32": (C.c_int, [_P, _P, _L, _I, _P]),
    "advhip_colsum_group_f32": (C.c_int, [C.POINTER(ColsumItem), _I, _P]),
    "advhip_adam_multi_f32": (C.c_int, [C.POINTER(AdamItem), _I, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _P]),
    "advhip_pack_item_tiles": (C.c_int64, [_I, _I, _I, _I]),
    "advhip_pack_weights_multi_f32": (C.c_int, [_P, _I, _I, _P]),
    "advhip_unfold3_f32": (C.c_int, [_P, _P, _I, _L, _I, _P]),
    "advhip_dwconv_t_fwd_f32": (C.c_int, [_P] * 4 + [_I, _I, _L, _I, _I, _P]),
    "advhip_dwconv_t_bwd_chunks": (_I, [_I, _L]),
    "advhip_dwconv_t_bwd_f32": (C.c_int, [_P] * 5 + [_I, _I, _L, _I, _I, _P]),
    "advhip_glance_attention_fwd_f32": (C.c_int, [_P, _P, _P, _I, _L, _I, _I, C.c_float, _P]),
    "advhip_glance_attention_bwd_f32": (C.c_int, [_P, _P, _P, _P, _I, _L, _I, _I, C.c_float, _P]),
    "advhip_ffn_block_partial_rows": (_L, [_L]),
    "advhip_ffn_block_fwd_f32": (C.c_int, [_P, _P, _P, C.c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _P]),
    "advhip_ffn_block_bwd_f32": (C.c_int, [_P, _P, _P, _P, _P, C.c_float, _P, _P, _P, _P, _P, _P, _I, _L, _P]),
    "advhip_glance_attention_fwd_anyt_f32": (C.c_int, [_P, _P, _P, _I, _L, _I, _I, C.c_float, _P]),
    "advhip_glance_attention_bwd_anyt_f32": (C.c_int, [_P, _P, _P, _P, _P, _I, _L, _I, _I, C.c_float, _P]),
    "advhip_fold_affine_f32": (C.c_int, [_P] * 7 + [_I, _I, _P]),
    "advhip_head_ln_fc_partial_rows": (_L, [_L]),
    "advhip_head_ln_fc_fwd_f32": (C.c_int, [_P] * 9 + [_I, _L, C.c_float, _P]),
    "advhip_head_ln_fc_bwd_f32": (C.c_int, [_P] * 11 + [_I, _L, _P]),
    "advhip_mil_magnitude_f32": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "advhip_mil_topk_select_f32": (C.c_int, [_P] * 7 + [_I] * 5 + [_P]),
    "advhip_mil_topk_select_bwd_f32": (C.c_int, [_P] * 5 + [_I] * 5 + [_P]),
    "advhip_mil_magnitude_bwd_f32": (C.c_int, [_P] * 5 + [_I] * 4 + [_P]),
    "advhip_mgfn_loss_ws_floats": (_L, [_I, _I, _I]),
    "advhip_mgfn_loss_fwd_f32": (C.c_int, [_P] * 9 + [_I] * 5 + [_P]),
    "advhip_mgfn_loss_bwd_f32": (C.c_int, [_P] * 14 + [_I] * 5 + [_P]),
    "advhip_segment_features_f32": (C.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "advhip_add_magnitude_f32": (C.c_int, [_P, _P, _L, _I, _P]),
    "advhip_tencrop_normalize_planes_u8": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _L, _L, C.c_float, C.c_float, _P]),
    "advhip_tencrop_normalize_u8": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, C.c_float, C.c_float, _P]),
    "advhip_normalize_permute_u8": (C.c_int, [_P, _P, _L, _I, _I, _I, _I, C.c_float, C.c_float, _P]),
}

_lib: Optional[C.CDLL] = None


def load(path: Optional[str] = None) -> C.CDLL:
    """dlopen the C-ABI library and bind every declared symbol (raises if one is missing)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise HipExtensionError(
            f"{p} not found: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or "
            "`python -m anomaly_detection_on_video_amd.build`). There is no CPU fallback."
        )
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise HipExtensionError(f"{p} does not export {name}; rebuild the extension") from e
        fn.restype = res
        fn.argtypes = args
    if lib.advhip_abi_version() != 2:
        raise HipExtensionError(f"{p}: ABI version {lib.advhip_abi_version()} != 2; rebuild the extension")
    if path is None:
        _lib = lib
    return lib


ON_FAILURE: list = []  # callables run when an entry point reports an error (ops drops per-stream state a failed launch may have left dirty)


def check(rc: int, what: str = "advhip") -> None:
    if rc != 0:
        msg = load().advhip_last_error().decode("utf-8", "replace")
        for hook in ON_FAILURE:
            try:
                hook()
            except Exception:  # pragma: no cover - a clean-up hook must not mask the error being reported
                pass
        raise HipExtensionError(f"{what} failed (code {rc}): {msg}")


def require_gpu(*tensors: torch.Tensor, contiguous: bool = True) -> None:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if t.is_cuda:
            if dev is None:
                dev = t.device
                if dev.index is not None and dev.index != torch.cuda.current_device():
                    # kernels launch on the CURRENT device: a tensor elsewhere would be read through peer access (or
                    # fault) on a stream that orders nothing on its own device
                    raise HipExtensionError(
                        f"tensor on {dev} but the current device is cuda:{torch.cuda.current_device()}: "
                        "call torch.cuda.set_device (one process per GPU)")
            elif t.device != dev:
                raise HipExtensionError(f"tensors on different devices: {dev} and {t.device}")
        if not t.is_cuda:
            raise HipExtensionError(
                "this op runs only as a HIP kernel on an AMD GPU; got a tensor on "
                f"'{t.device}'. Move the module and its inputs to cuda (there is no CPU fallback)."
            )
        if t.dtype not in (torch.float32, torch.int64, torch.int32, torch.uint8):
            raise HipExtensionError(f"unsupported dtype {t.dtype}; the kernels compute in fp32")
        if contiguous and not t.is_contiguous():
            raise HipExtensionError("HIP kernels need contiguous tensors")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream(dev=None) -> int:
    """Raw hipStream_t of torch's current stream on `dev` (a tensor, a device, or None = the current device)."""
    if isinstance(dev, torch.Tensor):
        dev = dev.device
    return torch.cuda.current_stream(dev).cuda_stream
