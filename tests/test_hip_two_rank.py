"""The sharded extract -> score stream on the real kernels with a real process group (BASELINE config 3):
W fresh rank processes (tests/rank_worker.py, one per rank, started through the GPU-free launcher so that this process --
which has already initialised the GPU -- execs nothing) must give, bit for bit, what the same crop-clip blocks give in one
process: every rank holds every row of every global batch in stream order (the reference's (n_clips, 10, 2048) order,
/root/reference/extract_features.py:93-100) and video v is scored by rank v % W.

  * gloo, both ranks on cuda:0: runs on the one-GPU box (ordering, sharding, ring and ownership logic; gloo collective);
  * nccl (= RCCL), one GPU per rank, world = min(device_count, 8): skipped where fewer than 2 GPUs are visible -- it runs
    the moment the suite is started on a multi-GPU node.
"""
import os
import socket
import sys

import pytest
import torch

from conftest import REPO, launch_fresh
import rank_worker

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(world, backend, share_gpu, out_dir):
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), os.path.join(REPO, "tests", "rank_worker.py"), "--backend", backend, "--out", out_dir]
    if share_gpu:
        argv.append("--share-gpu")
    rep = launch_fresh(argv, env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"}, unset=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"), timeout=900)
    assert rep["rc"] == 0, rep["stderr"][-4000:] + rep["stdout"][-2000:]
    return [torch.load(os.path.join(out_dir, f"r{r}.pt")) for r in range(world)]


def _single_process(world):
    from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream

    bb, sc = rank_worker.models("cuda:0")
    x = rank_worker.clips(world)
    lb = rank_worker.LOCAL_BATCH
    single = ExtractScoreStream(bb, sc, clips_per_video=rank_worker.CLIPS_PER_VIDEO, ncrops=rank_worker.NCROPS, local_batch=lb, world=1, rank=0)
    rows, scores = [], {}
    for i in range(0, x.shape[0], lb):  # the same blocks of LOCAL_BATCH crop-clips per launch, stream order
        g, sl = single.step(x[i : i + lb].to("cuda:0"))
        rows.append(g.cpu())
        for v, s in sl:
            scores[v] = s.cpu()
    return torch.cat(rows), scores


def _check(ranks, world, backend):
    rows, scores = _single_process(world)
    gb = rank_worker.LOCAL_BATCH * world
    assert len(scores) == world  # W videos in the stream
    seen = []
    for r, rec in enumerate(ranks):
        assert rec["backend"] == backend and rec["world"] == world
        for g in range(rank_worker.STEPS):
            assert torch.equal(rec["gathered"][g], rows[gb * g : gb * (g + 1)]), (r, g)  # rank-major all-gather == stream order, on every rank
        assert [v for v, _ in rec["scored"]] == [v for v in sorted(scores) if v % world == r]  # video v -> rank v % W
        for v, s in rec["scored"]:
            assert torch.equal(s, scores[v]), (r, v)
            seen.append(v)
    assert sorted(seen) == sorted(scores)


def test_two_ranks_sharing_the_gpu_over_gloo_equal_one_process(tmp_path):
    ranks = _run_ranks(2, "gloo", True, str(tmp_path))
    assert all(rec["device"] == "cuda:0" for rec in ranks)
    _check(ranks, 2, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs: one rank per GPU over RCCL")
def test_one_rank_per_gpu_over_rccl_equals_one_process(tmp_path):
    world = min(torch.cuda.device_count(), 8)
    ranks = _run_ranks(world, "nccl", False, str(tmp_path))
    assert sorted(rec["device"] for rec in ranks) == sorted(f"cuda:{r}" for r in range(world))
    _check(ranks, world, "nccl")
