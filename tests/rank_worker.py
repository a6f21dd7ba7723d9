#!/usr/bin/env python3
"""One rank of the sharded extract -> score stream on the REAL kernels (BASELINE config 3; the reference's
per-clip loop: /root/reference/extract_features.py:85-100).  Started by torch.distributed.run, one process per rank:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 --master-port P \
        tests/rank_worker.py --backend nccl|gloo [--share-gpu] --out DIR

`--backend nccl`: rank r on cuda:r, the all-gather is RCCL.  `--backend gloo --share-gpu`: every rank on cuda:0 and the
collective over gloo -- the one-GPU rehearsal of the same sharding / ordering / ownership logic.  Writes DIR/r<rank>.pt:
the gathered rows of every global batch and the (video, scores) pairs this rank scored.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CLIPS_PER_VIDEO, NCROPS, LOCAL_BATCH, STEPS = 3, 2, 2, 3  # W videos of 3 clips x 2 crops in 3 global batches of 2W crop-clips


def models(dev):
    from anomaly_detection_on_video_amd.i3d import I3Res50
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection
    from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_module_state_dict

    bb = I3Res50()
    bb.load_state_dict(synth_i3d_state_dict())
    sc = MGFNForVideoAnomalyDetection(MGFNConfig())
    sc.load_state_dict(synth_module_state_dict(sc))
    return bb.eval().to(dev), sc.eval().to(dev)


def clips(world):
    from anomaly_detection_on_video_amd.weights import synth_tensor

    return synth_tensor("dist.gpu.x", (STEPS * LOCAL_BATCH * world, 3, 16, 48, 48), scale=2.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from anomaly_detection_on_video_amd import dist as adist
    from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream

    rank, local_rank, world = adist.env_world()
    if args.share_gpu:
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if args.backend == "nccl":
        os.environ["LOCAL_RANK"] = str(local_rank)
    adist.init_process_group(args.backend)
    bb, sc = models(dev)
    stream = ExtractScoreStream(bb, sc, clips_per_video=CLIPS_PER_VIDEO, ncrops=NCROPS, local_batch=LOCAL_BATCH, world=world, rank=rank)
    x = clips(world)
    gb = LOCAL_BATCH * world
    handles = []
    for g in range(STEPS):
        lo = g * gb + rank * LOCAL_BATCH
        handles.append(stream.step_async(x[lo : lo + LOCAL_BATCH].to(dev)))
    stream.drain()
    torch.cuda.synchronize()
    res = [h.result() for h in handles]
    torch.save({"gathered": [g.cpu() for g, _s in res], "scored": [(v, s.cpu()) for _g, sl in res for v, s in sl],
                "backend": torch.distributed.get_backend(), "world": torch.distributed.get_world_size(), "device": str(dev)},
               os.path.join(args.out, f"r{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
