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


def _run_ranks(world, backend, share_gpu, out_dir, shape="small"):
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), os.path.join(REPO, "tests", "rank_worker.py"), "--backend", backend, "--out", out_dir, "--shape", shape]
    if share_gpu:
        argv.append("--share-gpu")
    rep = launch_fresh(argv, env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"}, unset=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"), timeout=900)
    assert rep["rc"] == 0, rep["stderr"][-4000:] + rep["stdout"][-2000:]
    return [torch.load(os.path.join(out_dir, f"r{r}.pt")) for r in range(world)]


def _single_process(world, shape="small"):
    from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream

    cfg = rank_worker.CONFIGS[shape]
    bb, sc = rank_worker.models("cuda:0")
    x = rank_worker.clips(world, shape)
    lb = cfg["local_batch"]
    single = ExtractScoreStream(bb, sc, clips_per_video=cfg["clips_per_video"], ncrops=cfg["ncrops"], local_batch=lb, world=1, rank=0)
    rows, scores = [], {}
    # the same blocks of LOCAL_BATCH crop-clips per launch, stream order, through the same entry point as the ranks (step_async:
    # whole-batch launches on the stream's lanes -- the synchronous step() spreads a batch of >= 16 over two streams of half
    # batches, i.e. other table entries with other K slices: equal within tolerance, not bit for bit)
    handles = [single.step_async(x[i : i + lb].to("cuda:0")) for i in range(0, x.shape[0], lb)]
    single.drain()
    torch.cuda.synchronize()
    for h in handles:
        g, sl = h.result()
        rows.append(g.cpu())
        for v, s in sl:
            scores[v] = s.cpu()
    return torch.cat(rows), scores


def _check(ranks, world, backend, shape="small"):
    cfg = rank_worker.CONFIGS[shape]
    rows, scores = _single_process(world, shape)
    gb = cfg["local_batch"] * world
    cl = cfg["clips_per_video"] if isinstance(cfg["clips_per_video"], list) else [cfg["clips_per_video"]]
    done, end = 0, 0
    while end + cl[done % len(cl)] * cfg["ncrops"] <= cfg["steps"] * gb:
        end += cl[done % len(cl)] * cfg["ncrops"]
        done += 1
    assert len(scores) == done and done >= 2  # the videos that complete in the stream
    for v, s in scores.items():
        assert s.shape == (cl[v % len(cl)],)  # T = the video's own clip count
    seen = []
    for r, rec in enumerate(ranks):
        assert rec["backend"] == backend and rec["world"] == world
        for g in range(cfg["steps"]):
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


@pytest.mark.parametrize("world", [2, 4])
def test_variable_length_stream_ranks_sharing_the_gpu_equal_one_process(tmp_path, world):
    """BASELINE config 3's stream shape (videos of different lengths) on 2 and on 4 rank processes sharing cuda:0 over gloo: every rank
    holds every gathered row in stream order, video v (T = its own clip count) is scored once, by rank v % W, with the bits one
    process gives.  World 4 is the widest rehearsal one card allows (first-contact insurance for the 8-rank run)."""
    ranks = _run_ranks(world, "gloo", True, str(tmp_path), shape="ragged")
    _check(ranks, world, "gloo", shape="ragged")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs: one rank per GPU over RCCL")
def test_variable_length_stream_over_rccl_equals_one_process(tmp_path):
    world = min(torch.cuda.device_count(), 8)
    ranks = _run_ranks(world, "nccl", False, str(tmp_path), shape="ragged")
    _check(ranks, world, "nccl", shape="ragged")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs: one rank per GPU over RCCL")
def test_one_rank_per_gpu_over_rccl_equals_one_process(tmp_path):
    world = min(torch.cuda.device_count(), 8)
    ranks = _run_ranks(world, "nccl", False, str(tmp_path))
    assert sorted(rec["device"] for rec in ranks) == sorted(f"cuda:{r}" for r in range(world))
    _check(ranks, world, "nccl")


def test_two_ranks_at_the_benchmarked_shape_equal_one_process(tmp_path):
    """Two rank processes at once on the tuned B = 32 / 16 x 224 x 224 plan (what `bench.py --gpus N` runs per rank: split-K launches
    on per-stream counter blocks, three lanes per rank), sharing cuda:0 over gloo: same bits as one process."""
    ranks = _run_ranks(2, "gloo", True, str(tmp_path), shape="bench")
    assert all(rec["lanes"] == 3 for rec in ranks)
    _check(ranks, 2, "gloo", shape="bench")


def _bench_two_ranks(extra_env):
    import json

    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    env.update(extra_env)
    rep = launch_fresh([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--sustain-s", "0"],
                       env=env, unset=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR") + (() if extra_env else ("ADV_BENCH_SHARE_GPU",)), timeout=900)
    assert rep["rc"] == 0, rep["stderr"][-4000:] + rep["stdout"][-2000:]
    lines = [ln for ln in rep["stdout"].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, rep["stdout"][-2000:]  # exactly ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak" and out["unit"] == "clips/s"
    cfg = out["config"]
    assert cfg["world_size_observed"] == 2 and cfg["global_batch"] == 64 and cfg["parallelism"] == "dp2"
    assert cfg["rank_clips_per_s_min"] > 0 and cfg["rank_clips_per_s_max"] >= cfg["rank_clips_per_s_min"]
    assert out["value"] > 0 and abs(out["value"] - 2 * cfg["rank_clips_per_s_min"]) < 1e-6 * out["value"] + 0.02  # whole-job rate = all ranks' clips / the slowest rank's time
    assert 0 < out["roofline"]["frac"] < 1 and out["roofline"]["unit"] == "TFLOP/s"
    assert "cpu_baseline" not in out and "mgfn_train_step" not in out  # rank 0 at N = 1 only
    return out


def test_bench_two_rank_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` the way the driver starts it (no torchrun around it), as the one-GPU rehearsal: bench.py is
    its own launcher, both ranks run the real B = 32 stream on cuda:0, the gather goes over gloo, rank 0 prints the one line."""
    out = _bench_two_ranks({"ADV_BENCH_SHARE_GPU": "1"})
    assert out["config"]["backend"] == "gloo" and "REHEARSAL" in out["data"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs: one rank per GPU over RCCL")
def test_bench_two_ranks_over_rccl():
    out = _bench_two_ranks({})
    assert out["config"]["backend"] == "nccl" and out["data"] == "synthetic"
