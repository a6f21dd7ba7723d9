#!/usr/bin/env python3
"""BASELINE config 3 on real kernels: the extract -> score stream sharded over two ranks must give, bit for bit, what the
same crop-clip blocks give in one process (SURVEY.md 8(d) cfg 3).  Both ranks share the one GPU of the box and the
collective runs over gloo (the production backend is RCCL; ordering, sharding and ownership logic are the same).

A manual check, not a pytest test: it has to start its rank processes BEFORE this process touches the GPU (on the GPU pool
a process that has initialised the GPU must not exec, and pytest's conftest has already done so).

    python tools/check_two_rank_stream.py        # prints OK or raises
"""
import os
import socket
import sys
import tempfile

import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _models():
    from anomaly_detection_on_video_amd.i3d import I3Res50
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection
    from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_module_state_dict

    bb = I3Res50()
    bb.load_state_dict(synth_i3d_state_dict())
    sc = MGFNForVideoAnomalyDetection(MGFNConfig())
    sc.load_state_dict(synth_module_state_dict(sc))
    return bb.eval().to("cuda:0"), sc.eval().to("cuda:0")


def _clips():
    from anomaly_detection_on_video_amd.weights import synth_tensor

    # 3 global batches of 4 crop-clips = 2 videos of 3 clips x 2 crops
    return synth_tensor("dist.gpu.x", (12, 3, 16, 48, 48), scale=2.0)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from anomaly_detection_on_video_amd import dist as adist
    from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream

    adist.init_process_group("gloo")
    bb, sc = _models()
    stream = ExtractScoreStream(bb, sc, clips_per_video=3, ncrops=2, local_batch=2, world=world, rank=rank)
    x = _clips()
    handles = []
    for g in range(3):
        lo = g * 4 + rank * 2
        handles.append(stream.step_async(x[lo : lo + 2].to("cuda:0")))
    stream.drain()
    torch.cuda.synchronize()
    res = [h.result() for h in handles]
    torch.save({"gathered": [g.cpu() for g, _s in res], "scored": [(v, s.cpu()) for _g, sl in res for v, s in sl]}, os.path.join(out, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


def main():
    tmp_path = tempfile.mkdtemp()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)  # before this process initialises the GPU
    from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream

    r0 = torch.load(os.path.join(tmp_path, "r0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "r1.pt"))
    # single process, same blocks of 2 crop-clips per launch, stream order
    bb, sc = _models()
    x = _clips()
    single = ExtractScoreStream(bb, sc, clips_per_video=3, ncrops=2, local_batch=2, world=1, rank=0)
    rows, scores = [], {}
    for i in range(0, 12, 2):
        g, sl = single.step(x[i : i + 2].to("cuda:0"))
        rows.append(g.cpu())
        for v, s in sl:
            scores[v] = s.cpu()
    rows = torch.cat(rows)
    for g in range(3):
        assert torch.equal(r0["gathered"][g], rows[4 * g : 4 * g + 4])   # rank-major all-gather == stream order
        assert torch.equal(r1["gathered"][g], rows[4 * g : 4 * g + 4])   # every rank holds every row
    assert [v for v, _ in r0["scored"]] == [0] and [v for v, _ in r1["scored"]] == [1]  # video v -> rank v % 2
    for v, s in r0["scored"] + r1["scored"]:
        assert torch.equal(s, scores[v])
    print("OK: 2-rank stream == single process, bit for bit (3 global batches, 2 videos)")


if __name__ == "__main__":
    main()
