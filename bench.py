#!/usr/bin/env python3
"""Headline benchmark: 16-frame 224x224 RGB crop-clips/s through I3D extract -> MIL score.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
every rank runs the I3D-ResNet50 backbone (53 fused fp32-MFMA conv launches + pools) on its 32
crop-clips, the 2048-d rows are all-gathered (RCCL; no-op at N=1), and every video (32 clips x 10
crops = 320 crop-clips) whose last crop-clip arrived is scored by its owner rank with the MGFN
scorer (eval).  Weak scaling: per-GPU work is fixed.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# launches of the conv stack per forward: the reference's 53 Conv3d as 52 conv launches (layer1.0's conv3 + downsample are one GEMM over
# [x ; h]), the stem's column-parity planes pass in front (ADV_STEM_S2W=1) and the pooled stem's merge pass behind it
CONV_LAUNCHES = "52 conv launches for the 53 Conv3d + the stem's planes pass + its pool-merge pass"
GFLOP_PER_CLIP = 32.829145088  # 2 x 16.414572544 GMAC, 53 bias-free Conv3d (oracle.conv_macs)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak


def metric_name() -> str:
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "16-frame 224² RGB clips/sec (I3D extract→MIL score), 1/2/4/8 GPU + %MFMA-peak"


TRAFFIC_PROFILE = "profiles/traffic.json"  # written by tools/summarize_prof.py from the PMC passes of THIS command


def measured_traffic(batch: int):
    """(HBM bytes per launch set, source) from the committed PMC passes of this same command
    (`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, separate passes; FETCH_SIZE doubled per the
    gfx950 correction).  Counters cannot be read from inside the process, so this is the profiled value for the
    benchmarked batch -- with the profile file, the commit it was taken at and whether the kernels have changed
    since (then it is stale and says so) -- else null."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_PROFILE)) as f:
            t = json.load(f)
        if batch != int(t.get("batch", 32)):
            return None, None
        src = {"file": TRAFFIC_PROFILE, "commit": t.get("commit"), "kernels_sha16": t.get("kernels_sha16"),
               "stale": t.get("kernels_sha16") != kernels_sha16()}
        return float(t["conv"]["bytes_corrected"]), src
    except Exception:
        return None, None


def kernels_sha16() -> str:
    """Hash of the conv-stack kernel sources + tuned table + plan: a traffic profile is only as fresh as these."""
    import hashlib

    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "anomaly_detection_on_video_amd")
    # what decides the conv stack's HBM traffic: its kernels, the tile table and the launch plan
    for path in [os.path.join(pkg, "csrc", "conv_igemm.hip"), os.path.join(pkg, "csrc", "pool.hip"), os.path.join(pkg, "csrc", "common.h"),
                 os.path.join(pkg, "tuned", "gfx950.json"), os.path.join(pkg, "i3d.py"), os.path.join(pkg, "ops.py"), os.path.join(pkg, "pipeline.py")]:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def ucf_crime_clip_counts(n: int = 64, seed: int = 2024):
    """Clip counts of a UCF-Crime-shaped stream (SURVEY.md 8(d) cfg 3): n_clips ~ U[50, 500] per video, seeded; the stream walks
    the list cyclically.  Every video is `n_clips x 10 crops` crop-clips (/root/reference/extract_features.py:93-100)."""
    import random

    rng = random.Random(seed)
    return [rng.randint(50, 500) for _ in range(n)]


def stream_start(clips, ncrops: int, global_batch: int, warmup: int, steps: int) -> int:
    """Where the synthetic stream is picked up (a multiple of the global batch): such that a video's last crop-clip arrives in the
    middle of the K timed steps whatever N, K and W are -- the timed region always holds at least one whole-video MIL scoring
    pass (T = that video's clip count), i.e. never less than the stream's amortised scoring share (one per ~86 steps at N = 1)."""
    mid = (warmup + max(steps // 2, 1)) * global_batch
    end, v = 0, 0
    while True:
        end += clips[v % len(clips)] * ncrops
        if end >= mid:
            return (end - mid) // global_batch * global_batch
        v += 1


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N rank processes (one per GPU) through
    torch.distributed.run and relay rank 0's JSON line.  Called BEFORE this process has touched the GPU (a process
    that has initialised HIP must never exec / be replaced; the children are fresh processes)."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("NCCL_DEBUG", "WARN")               # RCCL's own account of a failed communicator setup goes to the ranks' stderr
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if proc.returncode != 0 or len(lines) != 1:
        # a failed first contact must be attributable: the ranks' output, RCCL's warnings and the environment it ran under
        sys.stderr.write(proc.stdout[-4000:])
        sys.stderr.write("\n---- ranks' stderr (tail) ----\n" + proc.stderr[-6000:])
        sys.stderr.write(f"\nbench.py: {n}-rank launch failed (exit {proc.returncode}, {len(lines)} result lines); env: {json.dumps(comm_env(env))}\n")
        return proc.returncode or 1
    sys.stderr.write(proc.stderr[-2000:])
    print(lines[0], flush=True)
    return 0


def comm_env(env=None) -> dict:
    """The environment variables that decide how the ranks talk to each other (recorded in the N > 1 line and in a failed launch's tail)."""
    env = os.environ if env is None else env
    keys = ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "NCCL_SOCKET_IFNAME", "NCCL_IB_DISABLE", "NCCL_P2P_DISABLE", "RCCL_MSCCL_ENABLE", "HIP_VISIBLE_DEVICES",
            "ROCR_VISIBLE_DEVICES", "MASTER_ADDR", "MASTER_PORT", "ADV_BENCH_SHARE_GPU", "ADV_PIPELINE_LANES")
    return {k: env[k] for k in keys if k in env}


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or platform.machine()


def cpu_baseline(budget_s: float = 12.0):
    """The oracle (CPU restatement of the reference's path, pinned by tests/golden) timed on this
    host's cores on a bounded sample: config 1 of BASELINE.json (8 crop-clips per I3D forward) plus
    one 32-clip x 10-crop video through the MGFN oracle, amortised per crop-clip."""
    from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_input, synth_module_state_dict
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection
    from oracle import i3d_oracle, mgfn_oracle

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # a 1-GPU box exposes the whole host's logical CPUs but grants this job a 16-core share;
    # oversubscribing (256 threads) is 30x slower than 16.  ADV_CPU_THREADS overrides.
    cores = max(1, min(cores, int(os.environ.get("ADV_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    sd = synth_i3d_state_dict()
    x = synth_input((8, 3, 16, 224, 224), 0)
    i3d_oracle.i3d_forward(x, sd)  # warm-up
    reps, t0 = 0, time.perf_counter()
    times = []
    while reps < 3 or (time.perf_counter() - t0 < budget_s and reps < 40):
        t = time.perf_counter()
        feats = i3d_oracle.i3d_forward(x, sd)
        times.append(time.perf_counter() - t)
        reps += 1
    times.sort()
    t_clip = times[len(times) // 2] / 8.0
    msd = synth_module_state_dict(MGFNForVideoAnomalyDetection(MGFNConfig()))
    vid = torch.randn(1, 10, 32, 2048).abs()
    vid = torch.cat([vid, vid.norm(dim=3, keepdim=True)], dim=3)
    with torch.no_grad():
        mgfn_oracle.mgfn_forward(vid, msd)
        t = time.perf_counter()
        for _ in range(3):
            mgfn_oracle.mgfn_forward(vid, msd)
        t_video = (time.perf_counter() - t) / 3
    per_clip = t_clip + t_video / 320.0
    return {
        "value": round(1.0 / per_clip, 3), "unit": "clips/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
        "sample": f"{reps} x I3D oracle forward of 8 crop-clips (3x16x224x224, median) + MGFN oracle eval of one 32x10 video / 320",
        "i3d_ms_per_8_clips": round(times[len(times) // 2] * 1e3, 2), "mgfn_ms_per_video": round(t_video * 1e3, 2),
    }


def dry_run(args, clips, rank: int, world: int):
    """`--dry-run`: the N-rank control flow of the real run on CPU tensors -- the same ExtractScoreStream (ring, variable-length
    video bookkeeping, owner rule), the same all-gather per step (gloo), the same barrier / max-over-ranks timing and result line --
    with a stand-in backbone (rows = the crop-clips' stream positions) and a stand-in scorer.  Rank 0 returns the line (others {})
    after checking that every gathered row arrived in stream order and every completed video was scored once, by its owner."""
    from anomaly_detection_on_video_amd import dist as adist
    from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream

    class Rows(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x):
            return x.expand(-1, 8)

    class Stream(ExtractScoreStream):
        def score_video(self, feats):
            self.videos_scored += 1
            return feats[:, 0, 0].clone()

    st = Stream(Rows(), None, clips_per_video=clips, ncrops=10, local_batch=args.batch, world=world, rank=rank, feat_dim=8)
    gb = args.batch * world
    pos0 = stream_start(clips, 10, gb, args.warmup, args.steps)
    st.seek(pos0)
    scored = []

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def run(k0, n):
        for k in range(k0, k0 + n):
            pos = pos0 + k * gb + rank * args.batch
            g, sc = st.step(torch.arange(pos, pos + args.batch, dtype=torch.float32).unsqueeze(1))
            assert torch.equal(g[:, 0], torch.arange(pos0 + k * gb, pos0 + (k + 1) * gb, dtype=torch.float32)), "gathered rows out of stream order"
            scored.extend(v for v, _s in sc)

    run(0, args.warmup)
    barrier()
    before = len(scored)
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.zeros(world, dtype=torch.float64)
    t[rank] = elapsed
    n_scored = torch.tensor([float(len(scored))])
    if world > 1:
        torch.distributed.all_reduce(t)
        torch.distributed.all_reduce(n_scored)
    assert all(v % world == rank for v in scored) and len(set(scored)) == len(scored)
    end = pos0 + (args.warmup + args.steps) * gb  # every video that ends inside the stream fed so far was scored by exactly one rank
    s0, v, done = 0, 0, 0
    while s0 + clips[v % len(clips)] * 10 <= end:
        s0 += clips[v % len(clips)] * 10
        done += s0 > pos0  # (seek: a video that ends at or before the start counts as done)
        v += 1
    assert int(n_scored.item()) == done, (int(n_scored.item()), done)
    if rank != 0:
        return {}
    slowest = float(t.max())
    return {"metric": metric_name(), "dry_run": True, "value": round(gb * args.steps / slowest, 2), "unit": "clips/s (CPU stand-in backbone: NOT a measurement)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": "weak",
            "config": {"local_batch": args.batch, "global_batch": gb, "clips_per_video": "U[50,500] seeded", "stream_start": pos0,
                       "videos_scored_all_ranks": int(n_scored.item()), "videos_scored_rank0_timed": len(scored) - before,
                       "backend": torch.distributed.get_backend() if world > 1 else "none (single process)",
                       "world_size_observed": torch.distributed.get_world_size() if world > 1 else 1,
                       **({"env": comm_env()} if world > 1 else {})}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=32, help="crop-clips per GPU per step (BASELINE config 2: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--h2d", action="store_true", help="the full set of PCIe-inclusive legs (batch 40, two-pass and uint8-crop variants); "
                    "the two headline legs (resized uint8 frames, fp32 crops) run by default at N=1")
    ap.add_argument("--no-pcie", action="store_true", help="skip the default PCIe-inclusive legs")
    ap.add_argument("--sustain-s", type=float, default=3.0, help="seconds of back-to-back steps after the K timed ones (the `sustained` record: DVFS shows here)")
    ap.add_argument("--no-mgfn-train", action="store_true", help="skip the MGFN training-step record (config 4)")
    ap.add_argument("--stream-start", type=int, default=-1, help="crop-clip position the synthetic stream is picked up at (default: such that a video ends "
                    "in the middle of the timed steps); 0 in the PMC passes of tools/profile_bench.sh: no video completes in their few steps, so the "
                    "conv-kernel counters are the backbone's alone (the scorer's GEMMs run on the same kernels)")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous only (gloo, no GPU): checks the rank launch + result relay")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:  # plain `python bench.py --gpus N`: be our own launcher
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    from anomaly_detection_on_video_amd import dist as adist
    from anomaly_detection_on_video_amd.i3d import I3Res50
    from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection
    from anomaly_detection_on_video_amd.pipeline import ExtractScoreStream
    from anomaly_detection_on_video_amd.weights import synth_i3d_state_dict, synth_module_state_dict

    from anomaly_detection_on_video_amd import ops as aops

    rank, local_rank, world = adist.env_world()
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    if aops.ARITH != "f32":  # the line below prices the step against the fp32 MFMA peak and says dtype f32: refuse anything else
        raise SystemExit(f"bench.py: ADV_ARITH={aops.ARITH!r} switches conv kernels to split-bf16 arithmetic; the reported line is "
                         "the exact-fp32 path (dtype f32, 157.3 TFLOP/s roofline) -- unset ADV_ARITH")
    clips = ucf_crime_clip_counts()
    if args.dry_run:  # no GPU call anywhere on this path: the rank launch, the rendezvous, the stream's bookkeeping, the result line
        adist.init_process_group("gloo")
        line = dry_run(args, clips, rank, world)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the hot path has no CPU fallback")
    # ADV_BENCH_SHARE_GPU=1 (rehearsal only): every rank uses cuda:0 and the collective runs over gloo,
    # so the N>1 code path can be exercised on a one-GPU box; never used for reported numbers
    rehearsal = os.environ.get("ADV_BENCH_SHARE_GPU") == "1"
    if rehearsal:
        local_rank = 0
    adist.init_process_group("gloo" if rehearsal else "nccl")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    backbone = I3Res50()
    backbone.load_state_dict(synth_i3d_state_dict())
    backbone = backbone.eval().to(dev)
    scorer = MGFNForVideoAnomalyDetection(MGFNConfig())
    scorer.load_state_dict(synth_module_state_dict(scorer))
    scorer = scorer.eval().to(dev)
    # the synthetic UCF-Crime-shaped stream (SURVEY 8(d) cfg 3): n_clips ~ U[50, 500] per video, 10 crops per clip; every video is
    # scored with T = its own clip count when its last crop-clip has arrived
    stream = ExtractScoreStream(backbone, scorer, clips_per_video=clips, ncrops=10, local_batch=args.batch, world=world, rank=rank)
    pos0 = stream_start(clips, 10, args.batch * world, args.warmup, args.steps) if args.stream_start < 0 else args.stream_start
    stream.seek(pos0)
    stream.ring.uniform_(0.0, 3.0)  # (the rows of the picked-up video that lie before the stream's start: plausible features, not zeros)

    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn((args.batch, 3, 16, 224, 224), device=dev, generator=gen)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # untimed: first-use costs (MIOpen/rocBLAS kernel selection of the scorer's torch ops, weight
    # packing) must not land in the timed region whatever W is
    stream.score_video(torch.rand(32, 10, 2048, device=dev))
    stream.videos_scored = 0
    if world > 1:  # RCCL communicator setup (seconds) must not land in the timed region even with --warmup 0
        adist.all_gather_rows(torch.zeros((args.batch, 2048), device=dev))
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        stream.step_async(x)
    stream.drain()
    barrier()
    events = []
    orig_forward = backbone.forward
    backbone.forward = lambda b: backbone.forward_single(b, events=events)
    videos_before = stream.videos_scored
    log_before = len(stream.scored_log)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stream.step_async(x)
    stream.drain()
    barrier()
    elapsed = time.perf_counter() - t0
    videos_timed = stream.videos_scored - videos_before  # (the counter keeps running through the sustained / PCIe legs below)
    clips_timed = [n for _v, n in stream.scored_log[log_before:]]
    backbone.forward = orig_forward
    rank_elapsed = [elapsed]
    backend_observed, world_observed = "none (single process)", 1
    if world > 1:
        t = torch.zeros(world, device=dev, dtype=torch.float64)
        t[rank] = elapsed
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)  # every rank's own clock around the same K steps
        rank_elapsed = [float(v) for v in t.tolist()]
        elapsed = max(rank_elapsed)
        backend_observed, world_observed = torch.distributed.get_backend(), torch.distributed.get_world_size()

    # backbone duration per step from the HIP events recorded on the launch stream(s).
    #  - one stream, one lane: [start, (pool start, end) x3, end] per step -> conv stack = span minus the pool launches;
    #  - default (consecutive steps alternate between two lanes and/or a batch is split over streams): steps overlap,
    #    so the denominator is (first step's start .. last step's end) / K -- pools, gathers and scoring included,
    #    a lower bound on the conv stack's own rate.
    n_streams = 1 if stream.lanes > 1 else backbone._n_streams(args.batch)  # lanes run whole-batch launches
    overlapped = n_streams > 1 or stream.lanes > 1
    per = len(events) // max(args.steps, 1)
    if overlapped:
        conv_ms_avg = events[0].elapsed_time(events[-1]) / args.steps
    else:
        conv_ms = []
        for s in range(args.steps):
            ev = events[s * per : (s + 1) * per]
            total = ev[0].elapsed_time(ev[-1])
            pools = sum(ev[i].elapsed_time(ev[i + 1]) for i in range(1, per - 1, 2))
            conv_ms.append(total - pools)
        conv_ms_avg = sum(conv_ms) / len(conv_ms)
    achieved = args.batch * GFLOP_PER_CLIP / conv_ms_avg  # GFLOP/ms == TFLOP/s

    # sustained: >= sustain_s seconds of back-to-back steps right after the timed ones (short runs hold a higher
    # clock than the chip sustains; this record shows the difference instead of hiding it).  Same step count on
    # every rank (derived from the max-reduced time), same barrier / synchronize bracket.
    sustained = None
    if args.sustain_s > 0:
        n_sus = max(args.steps, int(args.sustain_s / (elapsed / args.steps)) + 1)
        barrier()
        t1 = time.perf_counter()
        for _ in range(n_sus):
            stream.step_async(x)
        stream.drain()
        barrier()
        sus = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([sus], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            sus = float(t.item())
        sus_clips = args.batch * world * n_sus / sus
        sustained = {"steps": n_sus, "seconds": round(sus, 3), "clips_per_s": round(sus_clips, 2),
                     "ms_per_step": round(sus / n_sus * 1e3, 4),
                     "frac_of_mfma_peak_whole_step": round(sus_clips / world * GFLOP_PER_CLIP / 1e3 / PEAK_F32_MFMA_TFLOPS, 4)}

    # BASELINE config 4: one MGFN training step (32,10,32,2049): forward + 4 losses + backward + Adam, rank 0 at N=1 --
    # the step the runner's Trainer executes: captured once as a HIP graph (train_graph.GraphedTrainStep), replayed per batch
    mgfn_train = None
    if world == 1 and not args.no_mgfn_train:
        from anomaly_detection_on_video_amd.train_graph import GraphedTrainStep

        scorer.train()
        # (the runner's optimizer on GPU parameters, runner.configure_optimizers: torch.optim.Adam's rule as one HIP launch per 80
        # tensors, device-side step counters; ADV_HIP_ADAM=0: torch's fused capturable Adam)
        if os.environ.get("ADV_HIP_ADAM", "1") == "1":
            from anomaly_detection_on_video_amd.optim import HipAdam

            opt = HipAdam(scorer.parameters(), lr=1e-3, weight_decay=5e-4)
        else:
            opt = torch.optim.Adam(scorer.parameters(), lr=1e-3, weight_decay=5e-4, fused=True, capturable=True)
        vb = torch.rand(32, 10, 32, 2048, device=dev, generator=gen) * 3
        vb = torch.cat([vb, vb.norm(dim=3, keepdim=True)], dim=3)
        al, nl = torch.ones(16, device=dev), torch.zeros(16, device=dev)
        step = GraphedTrainStep(scorer, opt, eager_steps=3)

        def timed_steps(n):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                # (once the graph exists the batch sits in its own input buffers, as runner.Trainer feeds it -- host batch copied
                # straight there --: no device-to-device staging copy inside the step)
                step(*(step.inputs() or (vb, al, nl)))
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / n * 1e3

        timed_steps(3)                # the eager steps (warm every lazily built operand)
        eager_ms = None
        if os.environ.get("ADV_TRAIN_GRAPH", "1") == "1":
            step.eager_left = 10
            eager_ms = timed_steps(10)    # the same step issued launch by launch from Python
            timed_steps(2)                # capture + first replays
            ms = timed_steps(20)
            assert step.graph is not None and step.replays >= 22
        else:
            step.eager_left = 1 << 30
            ms = timed_steps(10)
        tflop = 3 * 2 * 293.3e9 / 1e12  # forward 293.3 GMAC, backward = 2 x forward
        mgfn_train = {"workload": "run.py MIL scorer + losses, fwd+bwd+Adam, (32,10,32,2049) fp32, 1 GPU", "ms_per_step": round(ms, 3),
                      "input": "resident in HBM, in the captured step's input buffers (GraphedTrainStep.inputs(); runner.Trainer copies host batches straight into them)",
                      "mode": "one HIP graph replay per step (train_graph.GraphedTrainStep, what runner.Trainer runs)" if step.graph is not None else "eager",
                      "optimizer": type(opt).__name__,
                      "eager_ms_per_step": None if eager_ms is None else round(eager_ms, 3),
                      "tflop_per_step": round(tflop, 3), "achieved_tflops": round(tflop / ms * 1e3, 2),
                      "frac_of_f32_mfma_peak": round(tflop / ms * 1e3 / PEAK_F32_MFMA_TFLOPS, 4)}
        del step, opt
        scorer.eval()
        scorer.zero_grad(set_to_none=True)

    h2d = h2d_u8 = h2d_frames = None
    pcie = None
    if world == 1 and not args.no_pcie and not args.h2d:
        # SURVEY 8(d) cfg 1-2: the end-to-end rate INCLUDING the host -> device copy, labelled separately (never `value`).
        # Pinned host buffers every step; the copy runs on its own stream ahead of the step that consumes it; 3 + K steps each.
        from anomaly_detection_on_video_amd.pipeline import FrameCrops

        from anomaly_detection_on_video_amd.pipeline import HostFeeder

        def pcie_leg(st, host, prep, batch, wrap=None):
            """`prep`: a prepare callable run on the step's lane (resident input: None); `wrap`: instead, feed `host` through a
            HostFeeder (H2D on a copy stream of its own, four device buffers deep) and wrap the device buffer for the step."""
            feeder = HostFeeder(dev) if wrap is not None else None

            def one():
                if feeder is None:
                    st.step_async(host, prepare=prep)
                else:
                    p = feeder.feed(host, wrap)
                    feeder.done(p, st.step_async(host, prepare=p))

            for _ in range(3):
                one()
            st.drain()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                one()
            st.drain()
            torch.cuda.synchronize()
            return batch * args.steps / (time.perf_counter() - t1)

        # (a) what a decoder + GroupResize(256) hand over: 4 clips = 64 resized uint8 frames per step -> 40 crop-clips through
        #     I3Res50.forward_frames (one TenCrop + float + normalise pass writing column-parity planes, then the stem with
        #     16-byte gather pieces; ADV_U8_STEM=taps: the stem kernel reads the pixels itself); beside it the resident rate
        #     of that same 40-crop-clip batch
        fr = torch.randint(0, 256, (64, 256, 341, 3), dtype=torch.uint8).pin_memory()
        stream40 = ExtractScoreStream(backbone, scorer, clips_per_video=32, ncrops=10, local_batch=40, world=1, rank=0)
        x40 = torch.randn((40, 3, 16, 224, 224), device=dev, generator=gen)
        res40 = pcie_leg(stream40, x40, None, 40)
        frames40 = pcie_leg(stream40, fr, None, 40, wrap=lambda d: FrameCrops(d, 0, 40))
        del x40, stream40
        # (b) the reference's own hand-over: fp32 crop-clips from the host (extract_features.py:83-88), batch 32
        xh = x.cpu().pin_memory()
        fp32_32 = pcie_leg(stream, xh, None, args.batch, wrap=lambda d: d)
        pcie = {
            "note": "pinned host buffers every step, H2D on a copy stream of its own four device buffers deep (pipeline.HostFeeder); never the headline `value`",
            "resized_frames_u8": {"clips_per_s": round(frames40, 2), "crop_clips_per_step": 40, "h2d_bytes_per_step": fr.numel(),
                                  "resident_same_batch_clips_per_s": round(res40, 2), "ratio_to_resident": round(frames40 / res40, 4),
                                  "input": "64 uint8 frames 256x341x3 (4 clips) -> I3Res50.forward_frames",
                                  "u8_stem_form": aops.U8_STEM_FORM},
            "fp32_crops": {"clips_per_s": round(fp32_32, 2), "crop_clips_per_step": args.batch, "h2d_bytes_per_step": xh.numel() * 4,
                           "ratio_to_resident": None, "input": "fp32 (32,3,16,224,224) host tensor, the reference's hand-over"},
        }
        del xh, fr
    if args.h2d and world == 1:  # single-process extra; with N > 1 a rank-0-only step would leave the collective hanging
        # host buffers every step: the copy (and the uint8 pre-processing) is issued on the step's lane, so PCIe
        # transfers overlap the other lanes' compute
        from anomaly_detection_on_video_amd import mil_ops

        def timed(host, prep, st=None, batch=None):
            st, batch = st or stream, batch or args.batch
            for _ in range(3):
                st.step_async(host, prepare=prep)
            st.drain()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                st.step_async(host, prepare=prep)
            st.drain()
            torch.cuda.synchronize()
            return batch * args.steps / (time.perf_counter() - t1)

        # resized uint8 frames (what a decoder + GroupResize(256) hand over) + TenCrop / normalise / LoopPad / permutes on
        # the device: 4 clips (64 frames of 256 x 341 x 3) are shipped per step and the first 32 of their 40 crop-clips run
        # (the stream's batch), so the PCIe side is overstated by 25 % -- conservative
        fr = torch.randint(0, 256, (64, 256, 341, 3), dtype=torch.uint8).pin_memory()
        from anomaly_detection_on_video_amd.pipeline import FrameCrops

        # the stem kernel reads the uint8 pixels itself (TenCrop + float + normalise in its load stage, I3Res50.forward_frames)
        h2d_frames = timed(fr, lambda h: FrameCrops(h.to(dev, non_blocking=True), 0, args.batch))
        h2d_frames_2pass = timed(fr, lambda h: mil_ops.tencrop_normalize_u8(h.to(dev, non_blocking=True))[: args.batch])
        # like with like: 4 whole clips per step = 40 crop-clips, against the resident rate of that same batch
        stream40 = ExtractScoreStream(backbone, scorer, clips_per_video=32, ncrops=10, local_batch=40, world=1, rank=0)
        x40 = torch.randn((40, 3, 16, 224, 224), device=dev, generator=gen)
        res40 = timed(x40, None, stream40, 40)
        frames40 = timed(fr, lambda h: FrameCrops(h.to(dev, non_blocking=True), 0, 40), stream40, 40)
        frames40_2pass = timed(fr, lambda h: mil_ops.tencrop_normalize_u8(h.to(dev, non_blocking=True)), stream40, 40)
        del x40
        h2d = timed(x.cpu().pin_memory(), lambda h: h.to(dev, non_blocking=True))
        # uint8 pixels over PCIe + on-device normalise/permute (4x fewer bytes)
        xu = torch.randint(0, 256, (args.batch, 16, 3, 224, 224), dtype=torch.uint8).pin_memory()
        h2d_u8 = timed(xu, lambda h: mil_ops.normalize_permute_u8(h.to(dev, non_blocking=True)))

    if rank == 0:
        traffic, traffic_src = measured_traffic(args.batch)
        total_clips = args.batch * world * args.steps
        out = {
            "metric": metric_name(),
            "value": round(total_clips / elapsed, 2),
            "unit": "clips/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: ranks share one GPU, gloo)",
            "config": {
                "workload": "I3D-RGB feature extraction, batch=32 clips, 1xMI355X (HIP conv3d + fused BN/ReLU) -> MGFN MIL score per video (T = its clip count) x 10 crops"
                if world == 1 else
                f"I3D-RGB extraction sharded over {world}xMI355X, RCCL all-gather of 2048-d features, synthetic UCF-Crime-shape stream -> MGFN MIL score",
                "clip": "3x16x224x224 fp32", "local_batch": args.batch, "global_batch": args.batch * world,
                "clips_per_video": "n_clips ~ U[50,500] per video, seeded (SURVEY 8(d) cfg 3); stream picked up at crop-clip %d" % pos0,
                "ncrops": 10, "videos_scored_rank0": videos_timed, "clips_of_videos_scored_rank0": clips_timed,
                "weights": "deterministic synthetic (no network)", "parallelism": f"dp{world}",
                "backend": backend_observed, "world_size_observed": world_observed,  # what torch.distributed reports, not what was asked for
                "rank_clips_per_s_min": round(args.batch * args.steps / max(rank_elapsed), 2),
                "rank_clips_per_s_max": round(args.batch * args.steps / min(rank_elapsed), 2),
                "arith": aops.ARITH,
                # which form of the stem ran on this fp32 NCDHW input: "planes" = one split_w pass writes column-parity planes, the stem
                # gathers 16-byte pieces from them (ADV_STEM_S2W=1, the default); "ncdhw" = 4-byte gather straight from the input
                "stem_form": "planes" if aops.STEM_S2W else "ncdhw",
                **({"env": comm_env()} if world > 1 else {}),
            },
            "roofline": {
                "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                # HBM-side GB/s of the conv stack: the profiled bytes per launch set over THIS run's measured time per launch set
                "hbm_gbps": None if traffic is None else round(traffic / (conv_ms_avg * 1e-3) / 1e9, 1), "hbm_peak_gbps": 8000.0,
                "traffic_unit": "bytes per launch set (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC passes in profiles/)",
                "traffic_source": traffic_src,
                "kernel": f"conv3d fp32-MFMA stack ({CONV_LAUNCHES} per step on rank 0)" if not overlapped else
                f"conv3d fp32-MFMA stack ({CONV_LAUNCHES} per step and stream, {n_streams} stream(s) per step on rank 0; steps alternate between {stream.lanes} HIP stream lanes, "
                f"batch split over {n_streams} stream(s) per step; time = first start .. last end over the K steps / K, pools + scoring included)",
                "flop_per_launch_set": args.batch * GFLOP_PER_CLIP * 1e9, "avg_ms_per_launch_set": round(conv_ms_avg, 4),
            },
        }
        if sustained is not None:
            out["sustained"] = sustained
        if mgfn_train is not None:
            out["mgfn_train_step"] = mgfn_train
        if pcie is not None:
            pcie["fp32_crops"]["ratio_to_resident"] = round(pcie["fp32_crops"]["clips_per_s"] / (total_clips / elapsed), 4)
            out["pcie_inclusive"] = pcie
        if h2d is not None:
            out["pcie_inclusive_clips_per_s"] = round(h2d, 2)
            out["pcie_inclusive_uint8_clips_per_s"] = round(h2d_u8, 2)
            out["pcie_inclusive_resized_frames_u8_clips_per_s"] = round(h2d_frames, 2)
            out["batch40"] = {"note": "4 whole TenCrop'd clips per step (40 crop-clips): resident vs resized uint8 frames over PCIe + TenCrop on the device",
                              "resident_clips_per_s": round(res40, 2), "pcie_inclusive_resized_frames_u8_clips_per_s": round(frames40, 2),
                              "ratio": round(frames40 / res40, 4),
                              "two_pass_clips_per_s": round(frames40_2pass, 2),  # TenCrop + normalise as its own HIP pass (round-2 mid form)
                              "two_pass_b32_clips_per_s": round(h2d_frames_2pass, 2)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
