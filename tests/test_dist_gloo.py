"""world_size-2 (and 3, ragged) tests of the sharded extraction path on CPU with the gloo backend:
the all-gather must reproduce the single-process row order exactly (extract_features.py:93-100)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _feature_fn(units: torch.Tensor) -> torch.Tensor:
    # stand-in for the backbone: a deterministic per-unit row (each unit independent of the others)
    flat = units.reshape(units.shape[0], -1)
    return torch.stack([flat.sum(1), flat.mean(1), flat[:, 0], flat[:, -1] * 3], dim=1)


def _worker(rank, world, port, n_units, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from anomaly_detection_on_video_amd import dist as adist

    r, _lr, w = adist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(7)
    units = torch.randn((n_units, 3, 2, 4), generator=g)
    rows = adist.sharded_map_rows(_feature_fn, units)
    ok = torch.equal(rows, _feature_fn(units))
    # plain all_gather_rows: rank-major concatenation
    local = torch.full((2, 3), float(rank))
    allr = adist.all_gather_rows(local)
    ok = ok and torch.equal(allr, torch.arange(world, dtype=torch.float32).repeat_interleave(2).unsqueeze(1).expand(-1, 3))
    # extract_clip_batch(sharded=True) with a CPU stand-in model: (B, ncrops, T, 3, H, W) -> (B, ncrops, 2048)
    from anomaly_detection_on_video_amd import extract

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x):
            return x.reshape(x.shape[0], -1).sum(1, keepdim=True).expand(-1, 2048).reshape(-1, 2048, 1, 1, 1)

    clips = torch.randn((3, 10, 2, 3, 4, 4), generator=g)
    a = extract.extract_clip_batch(M(), clips, max_crop_clips=4, sharded=True)
    dist.barrier()
    b = extract.extract_clip_batch(M(), clips, max_crop_clips=7, sharded=False)
    ok = ok and a.shape == (3, 10, 2048) and torch.equal(a, b)
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_units", [(2, 32), (2, 7), (3, 10), (2, 1)])
def test_sharded_rows_match_single_process(world, n_units):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_units, out), nprocs=world, join=True)
    assert dict(out) == {r: True for r in range(world)}


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` as the driver calls it (no torchrun around it, WORLD_SIZE unset): bench.py starts N
    rank processes itself before touching the GPU and relays rank 0's single JSON line; a failing rank makes the
    launcher exit non-zero.  --dry-run keeps the ranks off the GPU (gloo rendezvous only)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["dry_run"] is True
    if not torch.cuda.is_available():  # the real path needs a GPU: every rank exits non-zero, so must the launcher
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode != 0 and not p.stdout.strip()
        assert "needs an AMD GPU" in p.stderr


def _stream_worker(rank, world, port, clips, crops, local_batch, steps, out):
    """One rank of a variable-length extract -> score stream on CPU: a stand-in backbone (rows = f(stream position)), the REAL
    all-gather (gloo), ring, video bookkeeping and ownership rule."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from anomaly_detection_on_video_amd import dist as adist
    from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream

    adist.init_process_group("gloo")

    class BB(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x):  # (local_batch, 1) stream positions -> (local_batch, 4) rows
            return torch.cat([x, x * 2, x + 0.5, -x], dim=1)

    class S(ExtractScoreStream):
        def score_video(self, feats):
            self.videos_scored += 1
            return feats[:, :, 0].clone()

    st = S(BB(), None, clips_per_video=clips, ncrops=crops, local_batch=local_batch, world=world, rank=rank, feat_dim=4)
    gb = local_batch * world
    mine = {}
    for k in range(steps):
        pos = k * gb + rank * local_batch  # rank r takes rows [r B, (r + 1) B) of every global batch
        g, scored = st.step(torch.arange(pos, pos + local_batch, dtype=torch.float32).unsqueeze(1))
        assert torch.equal(g[:, 0], torch.arange(k * gb, (k + 1) * gb, dtype=torch.float32))
        for v, ids in scored:
            mine[v] = ids.reshape(-1).tolist()
    out[rank] = mine
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_variable_length_stream_over_gloo(world):
    """BASELINE config 3's shape on CPU ranks: videos of different lengths, every rank sees every gathered row in stream order,
    video v is scored exactly once, by rank v % world, from its own rows."""
    clips, crops, local_batch, steps = [7, 3, 12, 5, 9], 2, 4, 25
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_stream_worker, args=(world, port, clips, crops, local_batch, steps, out), nprocs=world, join=True)
    got = {}
    for r in range(world):
        for v, ids in out[r].items():
            assert v % world == r and v not in got
            got[v] = ids
    total, v, s0 = steps * local_batch * world, 0, 0
    while s0 + clips[v % 5] * crops <= total:
        n = clips[v % 5] * crops
        assert got[v] == [float(i) for i in range(s0, s0 + n)]
        s0 += n
        v += 1
    assert len(got) == v and v >= 10


def test_bench_launches_eight_ranks_dry():
    """First-contact insurance for the 8-GPU run nobody can rehearse here: `python bench.py --gpus 8 --dry-run` -- bench.py as its
    own launcher, eight rank processes, gloo rendezvous on 127.0.0.1, one JSON line relayed from rank 0."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 20 and out["warmup"] == 5 and out["dry_run"] is True
    # the N > 1 line says what environment its ranks talked under (bench.launch_ranks sets both for the processes it starts)
    assert out["config"]["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0" and out["config"]["env"].get("NCCL_DEBUG") == "WARN"
