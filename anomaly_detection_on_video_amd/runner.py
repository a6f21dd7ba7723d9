"""Training / validation orchestration for the MIL scorer.

`VideoAnomalyDetectionRunner` keeps the reference LightningModule's constructor and hook names
(`/root/reference/src/runner.py:18-140`): training_step / validation_step / configure_optimizers /
on_validation_epoch_end / setup / train_dataloader / val_dataloader.  Lightning, wandb and
matplotlib are not in the target image, so the hooks are driven by the plain `Trainer` below
(automatic optimisation: zero_grad -> training_step -> backward -> Adam step; validation every
epoch; JSONL logging; last / top-k checkpoints), which the shipped `configs/trainer/default.yaml`
instantiates instead of `lightning.pytorch.Trainer`.
"""
from __future__ import annotations

import json
import os
import time
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import metrics, mgfn_ops
from .dataset import build_feature_dataset


class VideoAnomalyDetectionRunner:
    def __init__(self, model: torch.nn.Module, optimizer, data):
        self.model = model
        self.hparams = type("HParams", (), {"optimizer": optimizer, "data": data})()
        self.validation_step_outputs: List[Dict[str, np.ndarray]] = []
        self.logged: Dict[str, float] = {}
        self.device = torch.device("cpu")

    # -- logging hook the Trainer reads back
    def log(self, name: str, value, **_kw) -> None:
        self.logged[name] = float(value.detach()) if torch.is_tensor(value) else float(value)

    def to(self, device):
        self.device = torch.device(device)
        self.model.to(self.device)
        return self

    # runner.py:29-39 -- normal batch first, abnormal second
    @staticmethod
    def training_batch(batch) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(video (2B,10,T,C+1), abnormal_labels, normal_labels): the model inputs training_step assembles."""
        ninputs, ainputs = batch
        return torch.cat((ninputs["feature"], ainputs["feature"]), dim=0), ainputs["anomaly"], ninputs["anomaly"]

    def training_step(self, batch, batch_idx) -> torch.Tensor:
        inputs, abnormal, normal = self.training_batch(batch)
        outputs = self.model(video=inputs, abnormal_labels=abnormal, normal_labels=normal)
        self.log("train_loss", outputs.loss)
        return outputs.loss

    # runner.py:42-50 -- one video per step: (1, T, 10, 2049) -> (1, 10, T, 2049)
    def validation_step(self, batch, batch_idx) -> None:
        features = batch["feature"].permute(0, 2, 1, 3).contiguous()
        outputs = self.model(video=features)
        self.validation_step_outputs.append({
            "preds": outputs.scores.squeeze(0).squeeze(-1).detach().cpu().numpy(),
            "labels": batch["label"].squeeze(0).cpu().numpy(),
        })

    # runner.py:53-59 -- Adam with L2-in-gradient weight decay, no scheduler
    def configure_optimizers(self) -> List[torch.optim.Optimizer]:
        opt = self.hparams.optimizer
        params = list(self.model.parameters())
        # same update rule as the reference's torch.optim.Adam; on GPU parameters torch's fused implementation (a few launches
        # instead of ~25 multi-tensor ones: 0.73 -> 0.3 ms of the 19-ms step)
        # and with its step counters on the device (capturable): the whole step can then be replayed as one HIP graph
        fused = bool(params) and all(p.is_cuda for p in params)
        hip_ok = fused and all(p.dtype == torch.float32 and p.is_contiguous() for p in params)  # (bf16 / channels_last parameters: torch's Adam)
        if hip_ok and os.environ.get("ADV_HIP_ADAM", "1") == "1":
            # ... and here the whole update as one launch per 80 tensors (optim.HipAdam: torch.optim.Adam's rule and state layout)
            from .optim import HipAdam

            return [HipAdam(params, lr=float(opt["learning_rate"]), weight_decay=float(opt["weight_decay"]))]
        return [torch.optim.Adam(params, lr=float(opt["learning_rate"]), weight_decay=float(opt["weight_decay"]), fused=fused,
                                 capturable=fused)]

    # runner.py:62-90 (metrics; the matplotlib/wandb chart is out of scope)
    def on_validation_epoch_end(self) -> Dict[str, float]:
        outs = self.validation_step_outputs
        rec_auc, pr_auc = metrics.frame_level_auc([o["preds"] for o in outs], [o["labels"] for o in outs],
                                                  int(self.hparams.data["frames_per_clip"]))
        self.log("valid/rec_auc", rec_auc)
        self.log("valid/pr_auc", pr_auc)
        self.validation_step_outputs = []
        return {"valid/rec_auc": rec_auc, "valid/pr_auc": pr_auc}

    # runner.py:93-105
    def setup(self, stage: str = "fit") -> None:
        d = self.hparams.data
        kw = dict(revision=d.get("revision", "main"), cache_dir=d.get("cache_dir"), dynamic_load=bool(d.get("dynamic_load", False)))
        local = d.get("local_path")
        self.train_dataset = build_feature_dataset(mode="train", local_path=local, filename="train.zip" if local else None, **kw)
        self.valid_dataset = build_feature_dataset(mode="test", local_path=local, filename="test.zip" if local else None, **kw)

    # runner.py:108-124 -- two loaders zipped, shuffle=False, drop_last=True
    def train_dataloader(self) -> Tuple[DataLoader, DataLoader]:
        d = self.hparams.data
        mk = lambda ds: DataLoader(ds, batch_size=int(d["batch_size"]), shuffle=False, drop_last=True, num_workers=int(d.get("num_workers", 0)))
        return mk(self.train_dataset["normal"]), mk(self.train_dataset["abnormal"])

    def val_dataloader(self) -> DataLoader:
        return DataLoader(self.valid_dataset, batch_size=1, shuffle=False)

    def on_load_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
        pass

    def on_save_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
        pass


def _to_device(batch, device):
    if torch.is_tensor(batch):
        return batch.to(device=device, dtype=torch.float32 if batch.is_floating_point() else None, non_blocking=True)
    if isinstance(batch, dict):
        return {k: _to_device(v, device) for k, v in batch.items()}
    if isinstance(batch, (tuple, list)):
        return type(batch)(_to_device(v, device) for v in batch)
    return batch


class JSONLLogger:
    """Append one JSON object per log call (stands in for the wandb logger)."""

    def __init__(self, path: str = "logs/train.jsonl", **_ignored):
        self.path = path
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)

    def log_metrics(self, m: Dict[str, float], step: int) -> None:
        with open(self.path, "a") as f:
            f.write(json.dumps({"step": step, "time": time.time(), **m}) + "\n")


class LearningRateMonitor:
    def __init__(self, logging_interval: str = "step"):
        self.logging_interval = logging_interval

    def on_step(self, trainer, optimizer) -> Dict[str, float]:
        return {"lr-Adam": optimizer.param_groups[0]["lr"]}


LIGHTNING_LAYOUT_VERSION = "2.0.0"  # the checkpoint layout written below (no migration needed by lightning >= 2.0)


def checkpoint_state(runner: "VideoAnomalyDetectionRunner", optimizer, epoch: int, global_step: int, metrics_: Dict[str, float]) -> Dict[str, Any]:
    """A checkpoint in the key layout lightning.pytorch.ModelCheckpoint writes for the reference's LightningModule
    (`self.model = model`, runner.py:21-24): `state_dict` with the `model.` prefix, `optimizer_states`, `epoch`,
    `global_step`, `hyper_parameters`, plus the keys Lightning's loader reads before anything else
    (`pytorch-lightning_version` for its migration step, `loops`, `callbacks`).  Reference -> here is tested
    (tests/test_hip_train.py); here -> Lightning is by construction only: Lightning is not installed in this image, the key set
    is checked by tests/test_capi_and_host.py::test_checkpoint_carries_the_keys_lightning_reads."""
    return {
        "pytorch-lightning_version": LIGHTNING_LAYOUT_VERSION, "loops": {}, "callbacks": {},
        "epoch": epoch, "global_step": global_step,
        "state_dict": {"model." + k: v for k, v in runner.model.state_dict().items()},
        "optimizer_states": [optimizer.state_dict()], "lr_schedulers": [],
        "hyper_parameters": {"optimizer": dict(runner.hparams.optimizer), "data": dict(runner.hparams.data)},
        "metrics": dict(metrics_),
    }


def load_checkpoint(path: str, runner: "VideoAnomalyDetectionRunner", optimizer=None, strict: bool = True) -> Dict[str, Any]:
    """Load a checkpoint written by this trainer, by the reference's Lightning trainer (`state_dict` keys prefixed
    `model.`) or by this package's round-1 layout (`model` / `optimizer`) into the runner (and the optimizer)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if "state_dict" in ckpt:
        sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in ckpt["state_dict"].items()}
        opt_state = (ckpt.get("optimizer_states") or [None])[0]
    elif "model" in ckpt:
        sd, opt_state = ckpt["model"], ckpt.get("optimizer")
    else:  # a bare state dict
        sd, opt_state = ckpt, None
    runner.model.load_state_dict(sd, strict=strict)
    if optimizer is not None and opt_state is not None:
        optimizer.load_state_dict(opt_state)
    runner.on_load_checkpoint(ckpt)
    return ckpt


class ModelCheckpoint:
    """dirpath/last.ckpt every `every_n_epochs` epochs (+ top-k by `monitor`) in Lightning's checkpoint layout
    (`checkpoint_state`).  The reference's config monitors `rec_auc` while the runner logs `valid/rec_auc`
    (SURVEY.md S6): both spellings work."""

    def __init__(self, dirpath: str = "checkpoints", save_last: bool = True, save_top_k: int = 10, every_n_epochs: int = 10,
                 monitor: str = "rec_auc", mode: str = "max", verbose: bool = False):
        self.dirpath, self.save_last, self.save_top_k = dirpath, save_last, save_top_k
        self.every_n_epochs, self.monitor, self.mode, self.verbose = every_n_epochs, monitor, mode, verbose
        self.best: List[Tuple[float, str]] = []

    def on_epoch_end(self, trainer, runner, optimizer, epoch: int, metrics_: Dict[str, float]) -> None:
        if (epoch + 1) % max(1, self.every_n_epochs):
            return
        os.makedirs(self.dirpath, exist_ok=True)
        state = checkpoint_state(runner, optimizer, epoch, trainer.global_step, metrics_)
        runner.on_save_checkpoint(state)
        if self.save_last:
            torch.save(state, os.path.join(self.dirpath, "last.ckpt"))
        val = metrics_.get(self.monitor, metrics_.get("valid/" + self.monitor))
        if val is None or self.save_top_k == 0:
            return
        path = os.path.join(self.dirpath, f"epoch={epoch}-{self.monitor.replace('/', '_')}={val:.4f}.ckpt")
        torch.save(state, path)
        self.best.append((val if self.mode == "max" else -val, path))
        self.best.sort(reverse=True)
        for _, stale in self.best[self.save_top_k :]:
            if os.path.exists(stale):
                os.remove(stale)
        self.best = self.best[: self.save_top_k]


class Trainer:
    """Plain single-device fit loop with Lightning's keyword names (configs/trainer/default.yaml)."""

    def __init__(self, accelerator: str = "auto", gradient_clip_val: Optional[float] = None, max_steps: int = -1,
                 max_epochs: int = 1000, log_every_n_steps: Optional[int] = None, precision: str = "32-true",
                 logger=None, callbacks=None, check_val_every_n_epoch: int = 1, **_ignored):
        if str(precision) not in ("32-true", "32"):
            raise ValueError("the MI355X path computes in fp32 (precision: 32-true)")
        self.gradient_clip_val, self.max_steps, self.max_epochs = gradient_clip_val, max_steps, max_epochs
        self.log_every_n_steps = log_every_n_steps or 1
        self.loggers = list(logger or [])
        self.callbacks = list(callbacks or [])
        self.check_val_every_n_epoch = check_val_every_n_epoch
        if accelerator not in ("auto", "gpu", "cuda"):
            raise ValueError(f"accelerator={accelerator!r}: the scorer's MIL head and losses are HIP kernels (GPU only)")
        if not torch.cuda.is_available():
            raise RuntimeError("no AMD GPU visible: Trainer.fit needs one (there is no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.global_step = 0
        self.history: List[Dict[str, float]] = []

    def _log(self, m: Dict[str, float]) -> None:
        self.history.append({"step": self.global_step, **m})
        for lg in self.loggers:
            lg.log_metrics(m, self.global_step)

    @staticmethod
    def _max_size_cycle(*loaders):
        """Lightning's default way to combine the two train loaders (CombinedLoader mode "max_size_cycle"): as many
        steps as the LONGER loader has, the shorter one restarts -- not zip(), which would stop at the shorter."""
        n = max(len(ld) for ld in loaders)
        its = [iter(ld) for ld in loaders]
        for _ in range(n):
            batch = []
            for i, ld in enumerate(loaders):
                try:
                    batch.append(next(its[i]))
                except StopIteration:
                    its[i] = iter(ld)
                    batch.append(next(its[i]))
            yield tuple(batch)

    def fit(self, model: VideoAnomalyDetectionRunner, ckpt_path: Optional[str] = None) -> None:
        """`ckpt_path`: resume from a checkpoint (this trainer's or the reference's Lightning layout): weights,
        optimizer state, epoch and global step."""
        runner = model.to(self.device)
        runner.setup("fit")
        (optimizer,) = runner.configure_optimizers()
        first_epoch = 0
        if ckpt_path:
            ckpt = load_checkpoint(ckpt_path, runner, optimizer)
            first_epoch = int(ckpt.get("epoch", -1)) + 1
            self.global_step = int(ckpt.get("global_step", 0))
        # the step as one HIP graph once the (fixed) batch shape has been seen a few times (train_graph.py); ADV_TRAIN_GRAPH=0:
        # always the eager loop
        graphed = None
        # (only for the stock training_step: a subclass that overrides it -- its own loss, its own logging -- keeps the eager loop,
        # which calls it; the graphed step calls runner.model directly)
        if (os.environ.get("ADV_TRAIN_GRAPH", "1") == "1" and hasattr(runner, "training_batch")
                and type(runner).training_step is VideoAnomalyDetectionRunner.training_step
                and all(g.get("capturable", False) for g in optimizer.param_groups)):
            from .train_graph import GraphedTrainStep

            graphed = GraphedTrainStep(runner.model, optimizer, eager_steps=3, clip_grad_norm=self.gradient_clip_val)
        self.graphed_step = graphed
        for epoch in range(first_epoch, self.max_epochs):
            runner.model.train()
            nloader, aloader = runner.train_dataloader()
            for batch_idx, batch in enumerate(self._max_size_cycle(nloader, aloader)):
                if 0 <= self.max_steps <= self.global_step:
                    break
                if graphed is not None and self._feed_graph_inputs(graphed, batch):
                    # (the host batch went straight into the captured step's input buffers: no torch.cat, no staging copy)
                    runner.log("train_loss", graphed(*graphed.inputs()))
                elif graphed is not None:
                    runner.log("train_loss", graphed(*runner.training_batch(_to_device(batch, self.device))))
                else:
                    batch = _to_device(batch, self.device)
                    optimizer.zero_grad(set_to_none=True)
                    loss = runner.training_step(batch, batch_idx)
                    loss.backward()
                    if self.gradient_clip_val:
                        torch.nn.utils.clip_grad_norm_(runner.model.parameters(), self.gradient_clip_val)
                    optimizer.step()
                    mgfn_ops.invalidate_caches()  # (fused optimizers do not move version counters)
                self.global_step += 1
                if self.global_step % self.log_every_n_steps == 0:
                    m = {"train_loss": runner.logged["train_loss"], "epoch": epoch}
                    for cb in self.callbacks:
                        if hasattr(cb, "on_step"):
                            m.update(cb.on_step(self, optimizer))
                    self._log(m)
            epoch_metrics: Dict[str, float] = {}
            if (epoch + 1) % self.check_val_every_n_epoch == 0:
                epoch_metrics = self.validate(runner)
                self._log({**epoch_metrics, "epoch": epoch})
            for cb in self.callbacks:
                if hasattr(cb, "on_epoch_end"):
                    cb.on_epoch_end(self, runner, optimizer, epoch, epoch_metrics)
            if 0 <= self.max_steps <= self.global_step:
                break

    @staticmethod
    def _feed_graph_inputs(graphed, batch) -> bool:
        """Copy a (normal, abnormal) loader batch into the captured step's input buffers: normal features into rows [0, B) of the
        video buffer, abnormal ones into [B, 2B) (src/runner.py:29-39's torch.cat order), the two label vectors into theirs.
        False (nothing written) while there is no graph or when the batch has another shape."""
        static = graphed.inputs()
        if static is None:
            return False
        try:
            ninputs, ainputs = batch
            nf, af, al, nl = ninputs["feature"], ainputs["feature"], ainputs["anomaly"], ninputs["anomaly"]
        except (TypeError, KeyError, ValueError):
            return False
        video, s_al, s_nl = static
        b = nf.shape[0]
        if (af.shape != nf.shape or (2 * b,) + tuple(nf.shape[1:]) != tuple(video.shape) or tuple(al.shape) != tuple(s_al.shape)
                or tuple(nl.shape) != tuple(s_nl.shape)):
            return False
        if any(src.dtype != dst.dtype or (src.is_cuda and src.device != dst.device)
               for src, dst in ((nf, video), (af, video), (al, s_al), (nl, s_nl))):
            return False  # copy_ would cast silently; the general path runs such a batch as given
        # these copies are issued on the CURRENT stream, the one graph.replay() is issued on right after (GraphedTrainStep.__call__):
        # stream order is what makes the replay see them
        video[:b].copy_(nf, non_blocking=True)
        video[b:].copy_(af, non_blocking=True)
        s_al.copy_(al, non_blocking=True)
        s_nl.copy_(nl, non_blocking=True)
        return True

    @torch.no_grad()
    def validate(self, runner: VideoAnomalyDetectionRunner) -> Dict[str, float]:
        runner.model.eval()
        for i, batch in enumerate(runner.val_dataloader()):
            runner.validation_step(_to_device(batch, self.device), i)
        return runner.on_validation_epoch_end()
