"""torch.optim.Adam whose step is ONE HIP launch per 80 parameter tensors (include/advhip.h: advhip_adam_multi_f32).

The reference trains with `torch.optim.Adam(params, lr, weight_decay)` (/root/reference/src/runner.py:53-59).  torch's fused,
capturable form of it costs seven launches per step on the scorer's 130 parameters (0.28 ms of a 15.7-ms step); here the update
of all of them is two launches plus one add on the step counters.  Same update rule, same state layout -- `state[p]` holds
`step` (a device-side fp32 scalar, as torch's capturable Adam keeps it), `exp_avg`, `exp_avg_sq` -- so `state_dict()` /
`load_state_dict()` interchange with torch.optim.Adam, checkpoints of the reference included.  CUDA fp32 parameters only;
amsgrad / maximize / differentiable are not offered (the reference uses none of them).
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check


class HipAdam(torch.optim.Adam):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if torch.is_tensor(lr) or any(torch.is_tensor(b) for b in betas):
            raise ValueError("HipAdam takes lr / betas as Python numbers (a tensor would be read back on the host every step: a sync inside a graph capture)")
        # capturable=True: the step counters live on the device, which is what lets a whole training step be one HIP graph
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, foreach=False, capturable=True)
        self._steps = None  # one flat fp32 tensor; state[p]["step"] is a 0-dim view of it

    def _bind_steps(self, params, live):
        """Every parameter's `step` as a view of ONE flat tensor (slot i = parameter i), so that `+= 1` for all of them is one
        launch; re-bound whenever the state was replaced from outside (load_state_dict hands every parameter a tensor of its own).
        State is created lazily, for the parameters in `live` (those with a gradient) only -- as torch.optim.Adam does, so that the
        state_dict of a model with frozen parameters has the same entries."""
        flat = self._steps
        ok = flat is not None and flat.numel() == len(params)
        if ok:
            base = flat.data_ptr()
            live_ids = {id(p) for p in live}
            for i, p in enumerate(params):
                st = self.state.get(p)
                if st:
                    ok = ok and "step" in st and torch.is_tensor(st["step"]) and st["step"].data_ptr() == base + 4 * i
                elif id(p) in live_ids:
                    ok = False
        if ok:
            return flat
        dev = params[0].device
        vals = [float(self.state[p]["step"]) if p in self.state and "step" in self.state[p] else 0.0 for p in params]
        flat = torch.tensor(vals, device=dev, dtype=torch.float32)
        live_ids = {id(p) for p in live}
        for i, p in enumerate(params):
            if p not in self.state and id(p) not in live_ids:
                continue
            st = self.state[p]
            st["step"] = flat[i]
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        self._steps = flat
        return flat

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        every = [p for g in self.param_groups for p in g["params"]]
        if not every:
            return loss
        for p in every:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise _lib.HipExtensionError("HipAdam updates contiguous fp32 parameters on the GPU (use torch.optim.Adam for anything else)")
        # (the counters of ALL parameters: a parameter without a gradient this step keeps its count, as in torch)
        capturing = torch.cuda.is_current_stream_capturing()
        with_grad = [p for p in every if p.grad is not None]
        if not capturing:
            self._bind_steps(every, with_grad)
        elif self._steps is None or any(p not in self.state for p in with_grad):
            raise _lib.HipExtensionError("HipAdam: a graph capture needs one eager step first (the step counters and moments are created with host -> device "
                                         "copies; train_graph.GraphedTrainStep runs that step)")
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize") or group.get("differentiable"):
                raise _lib.HipExtensionError("HipAdam: amsgrad / maximize / differentiable are not offered")
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            if len(live) == len(every):  # (every slot of the flat tensor is some parameter's live counter)
                self._steps.add_(1.0)
            else:
                torch._foreach_add_([self.state[p]["step"] for p in live], 1.0)
            arr = (_lib.AdamItem * len(live))()
            keep = []
            for it, p in zip(arr, live):
                g = p.grad
                if g.is_sparse:
                    raise _lib.HipExtensionError("HipAdam does not take sparse gradients")
                if not g.is_contiguous() or g.dtype != torch.float32:
                    g = g.contiguous().float()
                    keep.append(g)
                st = self.state[p]
                it.param, it.grad, it.exp_avg, it.exp_avg_sq, it.step, it.n = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                                                                  st["step"].data_ptr(), p.numel())
            lr = group["lr"]
            if torch.is_tensor(lr):
                raise _lib.HipExtensionError("HipAdam: a tensor learning rate is not offered (float(lr) would synchronise, and cannot be captured)")
            b1, b2 = group["betas"]
            check(lib.advhip_adam_multi_f32(arr, len(live), float(lr), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                            _lib.stream(live[0])), "adam_multi")
        return loss
