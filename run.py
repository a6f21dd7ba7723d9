#!/usr/bin/env python3
"""Train the MIL scorer: `python run.py [runner=mgfn] [data=synthetic] [trainer.cls.max_epochs=2] ...`

Same flow as the reference entry point (/root/reference/run.py:15-35): instantiate the model config,
locate the model and runner classes, build loggers / callbacks / trainer from the config, fit.
Hydra is replaced by the small composer in anomaly_detection_on_video_amd.config.
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from anomaly_detection_on_video_amd.config import _locate, compose, instantiate  # noqa: E402


def main(argv=None):
    args = compose(os.path.join(ROOT, "configs"), "default", list(sys.argv[1:] if argv is None else argv))
    if args.runner.get("model_class") is None:
        raise SystemExit("runner.model_class is null: pick a runner with a model, e.g. `python run.py runner=mgfn`")
    config = instantiate(args.runner.model_config)
    model = _locate(args.runner.model_class)(config)
    runner = _locate(args.runner.cls)(model=model, optimizer=args.runner.optimizer, data=args.data)

    loggers = []
    for name, logger in (args.trainer.get("logger") or {}).items():
        if name == "wandb":
            import wandb

            wandb.login(key=args.get("wandb_key"))
        loggers.append(instantiate(logger))
    callbacks = [instantiate(cb) for cb in (args.trainer.get("callbacks") or {}).values()]
    trainer = instantiate(args.trainer.cls, logger=loggers, callbacks=callbacks)
    trainer.fit(model=runner, ckpt_path=args.get("ckpt_path"))  # ckpt_path=<file>: resume (Lightning checkpoint layout)
    return trainer


if __name__ == "__main__":
    main()
