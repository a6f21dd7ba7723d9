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

# "small": W videos of 3 clips x 2 crops in 3 global batches of 2W crop-clips of 16 x 48 x 48 (ordering / ownership logic);
# "bench": the benchmarked shape -- 32 crop-clips of 16 x 224 x 224 per rank and step (the tuned B = 32 plan, its split-K
# launches on per-stream counter blocks, the stream's lanes), 2 steps, videos of 4 clips x 10 crops so that some complete
# "ragged": BASELINE config 3's variable-length stream in miniature -- videos of 5, 3, 7, 4 clips x 2 crops (never fewer than k = 3: torch.topk raises in the reference too) (cyclic), global batches that end one
# video and begin the next, every video scored with T = its own clip count by rank v % W
CONFIGS = {
    "small": dict(clips_per_video=3, ncrops=2, local_batch=2, steps=3, hw=48),
    "ragged": dict(clips_per_video=[5, 3, 7, 4], ncrops=2, local_batch=2, steps=9, hw=48),
    "bench": dict(clips_per_video=4, ncrops=10, local_batch=32, steps=2, hw=224),
}
CLIPS_PER_VIDEO, NCROPS, LOCAL_BATCH, STEPS = 3, 2, 2, 3  # (the "small" configuration, by its old names)


def models(dev):
    from anomaly_detection_on_video_amd.i3d import I3Res50
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection
    from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_module_state_dict

    bb = I3Res50()
    bb.load_state_dict(synth_i3d_state_dict())
    sc = MGFNForVideoAnomalyDetection(MGFNConfig())
    sc.load_state_dict(synth_module_state_dict(sc))
    return bb.eval().to(dev), sc.eval().to(dev)


def clips(world, shape="small"):
    from anomaly_detection_on_video_amd.weights import synth_input, synth_tensor

    cfg = CONFIGS[shape]
    n = cfg["steps"] * cfg["local_batch"] * world
    if shape in ("small", "ragged"):
        return synth_tensor("dist.gpu.x" if shape == "small" else "dist.gpu.ragged", (n, 3, 16, cfg["hw"], cfg["hw"]), scale=2.0)
    # four distinct full-size clips (hashing 100+ of them takes minutes), dealt so that neighbouring rows, ranks and steps differ
    base = torch.cat([synth_input((2, 3, 16, cfg["hw"], cfg["hw"]), seed) for seed in (0, 1)])
    idx = torch.tensor([(i * 7 + i // cfg["local_batch"]) % 4 for i in range(n)])
    return base[idx]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--out", required=True)
    ap.add_argument("--shape", default="small", choices=sorted(CONFIGS))
    args = ap.parse_args()
    cfg = CONFIGS[args.shape]
    lb, steps = cfg["local_batch"], cfg["steps"]
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
    stream = ExtractScoreStream(bb, sc, clips_per_video=cfg["clips_per_video"], ncrops=cfg["ncrops"], local_batch=lb, world=world, rank=rank)
    x = clips(world, args.shape)
    gb = lb * world
    handles = []
    for g in range(steps):
        lo = g * gb + rank * lb
        handles.append(stream.step_async(x[lo : lo + lb].to(dev)))
    stream.drain()
    torch.cuda.synchronize()
    res = [h.result() for h in handles]
    torch.save({"gathered": [g.cpu() for g, _s in res], "scored": [(v, s.cpu()) for _g, sl in res for v, s in sl],
                "backend": torch.distributed.get_backend(), "world": torch.distributed.get_world_size(), "device": str(dev), "lanes": stream.lanes},
               os.path.join(args.out, f"r{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
