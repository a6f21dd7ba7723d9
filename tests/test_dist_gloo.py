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
