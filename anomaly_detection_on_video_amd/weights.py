"""Deterministic synthetic weights for the I3D-ResNet50 backbone and the MGFN scorer.

There is no network in the build container or on the GPU box, so the pretrained
checkpoints the reference downloads (`src/i3d.py:354`, repo `jinmang2/test_video_fe`)
are unreachable.  Parity fixtures and benchmarks therefore use weights produced by a
pure function of (tensor name, element index): both sides of a comparison regenerate
bit-identical tensors without shipping 109 MB of floats and without the reference.

The generator is a counter-based hash (splitmix64 finaliser) evaluated with numpy
uint64 arithmetic, so it does not depend on any library's RNG stream.

BatchNorm statistics are deliberately NOT the identity (gamma in [0.5,1.5], non-zero
beta / running_mean, running_var in [0.5,1.5]) so that BN-folding bugs are visible
(SURVEY.md section 8(c)).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def hash_uniform(name: str, numel: int, salt: int = 0) -> np.ndarray:
    """`numel` float64 values in [-1, 1), a pure function of (name, salt, index)."""
    seed = np.uint64(zlib.crc32(name.encode("utf-8")) + (int(salt) << 32))
    with np.errstate(over="ignore"):
        idx = np.arange(numel, dtype=np.uint64)
        h = _splitmix64(idx * np.uint64(0x2545F4914F6CDD1D) + _splitmix64(np.array([seed], dtype=np.uint64))[0])
    # top 53 bits -> [0,1)
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return 2.0 * u - 1.0


def synth_tensor(name: str, shape: Tuple[int, ...], scale: float = 1.0, offset: float = 0.0, salt: int = 0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    v = hash_uniform(name, n, salt) * scale + offset
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def synth_input(shape: Tuple[int, ...], seed: int = 0, name: str = "input") -> torch.Tensor:
    """Synthetic normalised clip tensor: roughly the range of (pixel-114.75)/57.375."""
    return synth_tensor(f"{name}/{seed}", shape, scale=2.0, salt=seed)


# --------------------------------------------------------------------------- I3D
def nonlocal_positions(use_nl: bool):
    """Blocks that carry a NonLocalBlock when I3Res50(use_nl=True): nonlocal_mod = 2 on layer2 and layer3, i.e. every
    odd block index there (src/i3d.py:219, 221-240, 296)."""
    return {("layer2", 1), ("layer2", 3), ("layer3", 1), ("layer3", 3), ("layer3", 5)} if use_nl else set()


def i3d_state_dict_spec(use_nl: bool = False, in_channels: int = 3) -> Iterable[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) for every entry of the reference I3Res50 state dict.

    Topology restated from `src/i3d.py:198-300` (layers [3,4,6,3]; temporal kernels
    L1 [1,1,1], L2 [1,0,1,0], L3 [1,0,1,0,1,0], L4 [0,1,0]; downsample on block 0).
    """

    def bn(prefix, c):
        yield f"{prefix}.weight", (c,), "bn_gamma"
        yield f"{prefix}.bias", (c,), "bn_beta"
        yield f"{prefix}.running_mean", (c,), "bn_mean"
        yield f"{prefix}.running_var", (c,), "bn_var"
        yield f"{prefix}.num_batches_tracked", (), "bn_count"

    yield "conv1.weight", (64, in_channels, 5, 7, 7), "conv"  # (3: the reference's RGB stem; 2: a flow stream's, BASELINE config 5)
    yield from bn("bn1", 64)
    inplanes = 64
    cfg = [
        ("layer1", 64, [1, 1, 1]),
        ("layer2", 128, [1, 0, 1, 0]),
        ("layer3", 256, [1, 0, 1, 0, 1, 0]),
        ("layer4", 512, [0, 1, 0]),
    ]
    for lname, planes, temp in cfg:
        for i, tc in enumerate(temp):
            p = f"{lname}.{i}"
            yield f"{p}.conv1.weight", (planes, inplanes, 1 + 2 * tc, 1, 1), "conv"
            yield from bn(f"{p}.bn1", planes)
            yield f"{p}.conv2.weight", (planes, planes, 1, 3, 3), "conv"
            yield from bn(f"{p}.bn2", planes)
            yield f"{p}.conv3.weight", (planes * 4, planes, 1, 1, 1), "conv"
            yield from bn(f"{p}.bn3", planes * 4)
            if i == 0:
                yield f"{p}.downsample.0.weight", (planes * 4, inplanes, 1, 1, 1), "conv"
                yield from bn(f"{p}.downsample.1", planes * 4)
            if (lname, i) in nonlocal_positions(use_nl):  # NonLocalBlock(outplanes, outplanes, outplanes // 2), src/i3d.py:93-96
                outp, inner = planes * 4, planes * 2
                for conv, (co, ci) in (("theta", (inner, outp)), ("phi", (inner, outp)), ("g", (inner, outp)), ("out", (outp, inner))):
                    yield f"{p}.nl.{conv}.weight", (co, ci, 1, 1, 1), "conv"
                    yield f"{p}.nl.{conv}.bias", (co,), "conv_bias"
                yield from bn(f"{p}.nl.bn", outp)
            inplanes = planes * 4


def synth_i3d_state_dict(salt: int = 0, use_nl: bool = False, in_channels: int = 3) -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    for key, shape, kind in i3d_state_dict_spec(use_nl, in_channels):
        if kind == "conv":
            fan_in = int(np.prod(shape[1:]))
            # uniform[-a,a] has variance a^2/3; aim at var = 2/fan_in (ReLU-preserving)
            a = float(np.sqrt(3.0 * 2.0 / fan_in))
            sd[key] = synth_tensor(key, shape, scale=a, salt=salt)
        elif kind == "conv_bias":
            sd[key] = synth_tensor(key, shape, scale=0.1, salt=salt)
        elif kind == "bn_gamma":
            # the last BN of every residual branch gets a smaller gain so that 16 stacked
            # residual additions do not blow the activations up
            small = key.endswith("bn3.weight") or ".downsample.1." in key or ".nl.bn." in key
            centre = 0.5 if small else 1.0
            sd[key] = synth_tensor(key, shape, scale=0.5 * centre, offset=centre, salt=salt)
        elif kind == "bn_beta":
            sd[key] = synth_tensor(key, shape, scale=0.25, salt=salt)
        elif kind == "bn_mean":
            sd[key] = synth_tensor(key, shape, scale=0.25, salt=salt)
        elif kind == "bn_var":
            sd[key] = synth_tensor(key, shape, scale=0.5, offset=1.0, salt=salt)
        elif kind == "bn_count":
            sd[key] = torch.tensor(0, dtype=torch.long)
        else:  # pragma: no cover
            raise AssertionError(kind)
    return sd


NONLOCAL_CASES = ["nl512", "nl1024", "nl512odd"]


def synth_nonlocal_case(name: str):
    """(dim, inner, state dict, x) of a stand-alone NonLocalBlock(dim, dim, inner) parity case (src/i3d.py:124-195): the
    two block sizes of I3Res50(use_nl=True) on small clips, one with odd spatial extents (the (1,2,2) pool floors)."""
    dim, inner, (b, t, h, w) = {"nl512": (512, 256, (2, 2, 6, 8)), "nl1024": (1024, 512, (1, 2, 4, 6)),
                                "nl512odd": (512, 256, (1, 3, 7, 5))}[name]
    sd: Dict[str, torch.Tensor] = {}
    for conv, (co, ci) in (("theta", (inner, dim)), ("phi", (inner, dim)), ("g", (inner, dim)), ("out", (dim, inner))):
        sd[f"{conv}.weight"] = synth_tensor(f"{name}.{conv}.w", (co, ci, 1, 1, 1), scale=float(np.sqrt(3.0 / ci)))
        sd[f"{conv}.bias"] = synth_tensor(f"{name}.{conv}.b", (co,), scale=0.1)
    sd["bn.weight"] = synth_tensor(f"{name}.bn.g", (dim,), scale=0.25, offset=0.5)
    sd["bn.bias"] = synth_tensor(f"{name}.bn.b", (dim,), scale=0.25)
    sd["bn.running_mean"] = synth_tensor(f"{name}.bn.m", (dim,), scale=0.25)
    sd["bn.running_var"] = synth_tensor(f"{name}.bn.v", (dim,), scale=0.5, offset=1.0)
    sd["bn.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    x = synth_tensor(f"{name}.x", (b, dim, t, h, w), scale=2.0)
    return dim, inner, sd, x


# --------------------------------------------------------------------------- generic
def synth_module_state_dict(module: torch.nn.Module, salt: int = 0, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Deterministic values for an arbitrary module tree (used for the MGFN scorer).

    * >=2-D float tensors: uniform with variance gain/fan_in (fan_in = prod(shape[1:]))
    * 1-D `weight` / `g`: centred at 1; other 1-D floats: small, centred at 0
    * `running_var`: in [0.5, 1.5]; integer buffers: zero
    """
    out: Dict[str, torch.Tensor] = {}
    for key, ref in module.state_dict().items():
        shape = tuple(ref.shape)
        if not ref.is_floating_point():
            out[key] = torch.zeros_like(ref)
            continue
        leaf = key.rsplit(".", 1)[-1]
        squeezed = [s for s in shape if s != 1]
        if leaf == "running_var":
            out[key] = synth_tensor(key, shape, scale=0.5, offset=1.0, salt=salt)
        elif len(squeezed) >= 2 and leaf not in ("g", "b"):
            fan_in = int(np.prod(shape[1:]))
            a = float(np.sqrt(3.0 * gain / fan_in))
            out[key] = synth_tensor(key, shape, scale=a, salt=salt)
        elif leaf in ("weight", "g"):
            out[key] = synth_tensor(key, shape, scale=0.25, offset=1.0, salt=salt)
        else:
            out[key] = synth_tensor(key, shape, scale=0.1, salt=salt)
    return out
