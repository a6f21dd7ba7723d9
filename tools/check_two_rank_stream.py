#!/usr/bin/env python3
"""BASELINE config 3 on real kernels, by hand: the extract -> score stream sharded over W ranks must give, bit for bit, what
the same crop-clip blocks give in one process.  The same check runs under pytest (tests/test_hip_two_rank.py); this tool
runs it outside pytest:

    python tools/check_two_rank_stream.py                  # 2 ranks sharing cuda:0, gloo
    python tools/check_two_rank_stream.py --backend nccl   # one rank per visible GPU (RCCL)
    python tools/check_two_rank_stream.py --world 6 --shape ragged   # the widest share-GPU rehearsal a one-GPU box allows (at most 6
                                                                     # processes may hold the card; this parent touches it only after the ranks exit)
"""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--world", type=int, default=0)
    ap.add_argument("--shape", default="small", help="a key of tests/rank_worker.py CONFIGS: small, ragged (videos of different lengths), bench")
    args = ap.parse_args()
    import torch

    world = args.world or (2 if args.backend == "gloo" else min(torch.cuda.device_count(), 8))  # (device_count does not initialise the GPU)
    out = tempfile.mkdtemp()
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(ROOT, "tests", "rank_worker.py"), "--backend", args.backend, "--out", out, "--shape", args.shape]
    if args.backend == "gloo":
        argv.append("--share-gpu")
    subprocess.run(argv, check=True, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))  # before this process initialises the GPU
    import test_hip_two_rank as t

    ranks = [torch.load(os.path.join(out, f"r{r}.pt")) for r in range(world)]
    t._check(ranks, world, args.backend, shape=args.shape)
    owners = {r: [v for v, _ in rec["scored"]] for r, rec in enumerate(ranks)}
    print(f"OK: {world}-rank stream ({args.backend}, shape {args.shape}) == single process, bit for bit; videos scored per rank: {owners}")


if __name__ == "__main__":
    main()
