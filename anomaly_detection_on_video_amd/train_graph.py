"""One MGFN training step -- zero_grad -> forward -> four losses -> backward -> Adam -- replayed as ONE HIP graph.

The eager step issues ~215 launches from Python autograd (the hand-written GEMM / norm / attention / MIL / loss / Adam kernels
through ctypes; round 2: ~600, stage 0 and the attention einsums still on torch ops then): on a slow host the step is bound by
that issue loop, not by the GPU (round 2: 22.0 ms driver-timed against 17.5-18.7 ms on a faster host, same kernels).  The runner feeds a fixed
`(2B, 10, 32, 2049)` shape (/root/reference/src/runner.py:29-39, configs/data: batch_size 16), so after a few eager steps
the step is captured once (torch.cuda.CUDAGraph: every launch of this package goes to torch's current stream, nothing
inside allocates or synchronises outside torch's allocator) and replayed with the batch copied into static buffers:

  * same kernels, same order, same arithmetic as the eager step -- the graph only removes the host from the loop;
  * the weight re-pack of every differentiated forward (mgfn_ops.pack_kc_cached(fresh=True)) is captured with it, so a
    replay always packs the live parameters; `mgfn_ops.invalidate_caches()` runs after every replay for the inference
    caches, as after an eager optimizer step;
  * dropout masks of the MIL head come from torch's graph-safe Philox state: a different mask every replay;
  * `overlap=True` (opt-in) issues weight and bias gradients on a side stream inside the captured graph
    (mgfn_ops.overlapped_backward: forked off the `dX` chain, joined before the optimizer).  Measured on one MI355X at
    (32,10,32,2049): 19.3-19.4 ms against 18.6 ms serial -- two MFMA-bound GEMMs side by side run slower than one after the
    other here (unlike the extraction stream, whose lanes pair MFMA-bound with memory-bound launches), so it stays off.

Adam must be created with `capturable=True` (its step counters live on the device).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

from . import mgfn_ops


class GraphedTrainStep:
    """step(video (2B,10,T,C+1), abnormal_labels (B,), normal_labels (B,)) -> loss (a static device tensor, overwritten by
    the next call).  The first `eager_steps` calls run eagerly (they are real training steps and warm every lazily built
    operand); the next call captures and replays; a call with other shapes falls back to the eager step."""

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, eager_steps: int = 3,
                 clip_grad_norm: Optional[float] = None, overlap: bool = False,
                 after_step: Optional[Callable[[], None]] = None):
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("GraphedTrainStep needs an optimizer created with capturable=True (device-side step counters)")
        self.model, self.optimizer = model, optimizer
        # at least one eager step before any capture: it builds what a capture must not (pack plans, gather tables, constants and
        # HipAdam's flat step-counter tensor are created with blocking host -> device copies)
        self.eager_left = max(int(eager_steps), 1)
        self.clip, self.overlap = clip_grad_norm, overlap
        self.after_step = after_step
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.static: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None
        self.loss: Optional[torch.Tensor] = None
        self.replays = 0
        self.captures = 0
        self.held_plans: list = []          # mgfn_ops pack plans whose buffers the captured graph reads and writes
        self._stamp: Optional[Tuple] = None  # what the graph baked in besides shapes: parameter addresses, optimizer scalars
        if overlap and torch.cuda.is_available():
            mgfn_ops.ensure_side_stream()  # (streams cannot be created inside a capture)
        # The stream the step is captured on -- and on which the LAST eager step before a capture runs: the kernels' per-stream scratch
        # (split-K counter blocks and workspaces that are zero-filled once when allocated and left zero by every launch,
        # ops.zeroed_workspace / ops.splitk_counters) then exists at its final size before the capture starts.  Allocated inside a
        # capture, every such zero-fill becomes a node of the graph and runs again on every replay (12 fills, ~120 MB, per step).
        self._cap_stream: Optional[torch.cuda.Stream] = torch.cuda.Stream() if torch.cuda.is_available() else None

    def _step(self, video, al, nl) -> torch.Tensor:
        self.optimizer.zero_grad(set_to_none=True)
        # (the narrow layers' weight / bias gradients: one grouped launch at the end of the backward pass, ADV_MGFN_DEFER_DW=0: per layer)
        with mgfn_ops.overlapped_backward(self.overlap), mgfn_ops.deferred_param_grads(mgfn_ops.DEFER_DW):
            loss = self.model(video=video, abnormal_labels=al, normal_labels=nl).loss
            loss.backward()
        if self.clip:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)
        self.optimizer.step()
        return loss

    MAX_CAPTURES = 4  # hyper-parameters that keep changing (a per-step LR schedule): after this many captures the step stays eager

    def _baked(self) -> Tuple:
        """Everything a captured graph holds by value or by address: the parameters' storage (a `.data =` / `.to()` moves it)
        and the optimizer's scalar hyper-parameters (the fused capturable Adam takes lr / betas / eps / weight_decay as kernel
        arguments) and the clip value.  A replay with a different stamp would train with stale numbers or touch freed memory:
        __call__ compares and captures again."""
        groups = tuple((float(g["lr"]) if not torch.is_tensor(g["lr"]) else ("t", g["lr"].data_ptr()), tuple(g.get("betas", ())), g.get("eps"),
                        g.get("weight_decay"), g.get("amsgrad"), g.get("maximize"), tuple(p.data_ptr() for p in g["params"]))
                       for g in self.optimizer.param_groups)
        state = self.optimizer.state  # (optimizer.load_state_dict() replaces these tensors: a replay would update orphaned moments)
        moments = tuple(tuple(state[p][k].data_ptr() if torch.is_tensor(state[p].get(k)) else None for k in ("exp_avg", "exp_avg_sq", "step"))
                        for g in self.optimizer.param_groups for p in g["params"] if p in state)
        return (groups, self.clip, tuple(p.data_ptr() for p in self.model.parameters()), moments)

    def inputs(self) -> Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """The captured graph's own input buffers (video, abnormal_labels, normal_labels), or None while there is no graph.  A feeder
        that writes the next batch straight into them -- runner.Trainer copies the host batch there -- and then calls
        `step(*step.inputs())` saves the device-to-device staging copy of the batch (84 MB at the runner's batch shape)."""
        return self.static if self.graph is not None else None

    def _finish(self) -> None:
        mgfn_ops.invalidate_caches()  # (fused / captured optimizers do not move version counters)
        if self.after_step is not None:
            self.after_step()

    def __call__(self, video: torch.Tensor, abnormal_labels: torch.Tensor, normal_labels: torch.Tensor) -> torch.Tensor:
        key = (tuple(video.shape), tuple(abnormal_labels.shape), tuple(normal_labels.shape))
        if self.graph is not None and key == self._key and self._baked() != self._stamp:
            # a parameter moved or an optimizer scalar changed since the capture: the graph is stale -> capture again below
            # (or, after MAX_CAPTURES of them, keep to the eager step, which reads everything live)
            # ONE live eager step first: a moved parameter means new pack plans / tables, a replaced optimizer state new step-counter
            # bindings -- all built with host -> device copies that are illegal inside a capture
            self.graph, self.static, self.loss, self.held_plans = None, None, None, []
            self.eager_left = 1 << 62 if self.captures >= self.MAX_CAPTURES else 1
        if self.graph is not None and key == self._key:
            for dst, src in zip(self.static, (video, abnormal_labels, normal_labels)):
                if src.data_ptr() != dst.data_ptr():  # (a caller that filled `inputs()` in place has nothing to copy)
                    dst.copy_(src)
            self.graph.replay()
            self.replays += 1
            self._finish()
            return self.loss
        if self.eager_left > 0 or self.graph is not None or not self.model.training:
            if self.model.training:  # only a TRAINING step is the warm step a capture needs (an eval call must not use it up)
                self.eager_left -= 1
            if self.eager_left == 0 and self.graph is None and self.model.training and self._cap_stream is not None:
                # the last eager step before a capture: on the capture stream (see __init__), fenced by device-wide synchronisations
                torch.cuda.synchronize()
                with torch.cuda.stream(self._cap_stream):
                    loss = self._step(video, abnormal_labels, normal_labels).detach()
                torch.cuda.synchronize()
            else:
                loss = self._step(video, abnormal_labels, normal_labels).detach()
            self._finish()
            return loss
        # capture (records, does not run), then replay on this batch
        self._key = key
        self.static = tuple(t.detach().clone() for t in (video, abnormal_labels, normal_labels))
        torch.cuda.synchronize()
        self.optimizer.zero_grad(set_to_none=True)  # the captured backward allocates the gradients in the graph's pool
        graph = torch.cuda.CUDAGraph()
        with mgfn_ops.hold_plans() as held, torch.cuda.graph(graph, stream=self._cap_stream):
            loss = self._step(*self.static)
        self.graph, self.loss = graph, loss.detach()
        self.held_plans, self._stamp = held, self._baked()
        self.captures += 1
        self.graph.replay()
        self.replays += 1
        self._finish()
        return self.loss
