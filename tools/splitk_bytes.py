#!/usr/bin/env python3
"""Where the layer-4 launches' FETCH_SIZE comes from: XCD replication vs split-K partial tiles.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT/fetch -- python3 tools/splitk_bytes.py run
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d OUT/write -- python3 tools/splitk_bytes.py run
    python3 tools/splitk_bytes.py time > OUT/time.json          (un-profiled timings of the same launches)
    python3 tools/splitk_bytes.py report OUT                    (the table)

`run` issues, for each of the four layer-4 shapes the review names, the tuned (algo, splits) and the same tile with 1 .. 6 K slices,
REPS launches each in a fixed order; `report` walks the counter rows in dispatch order.  FETCH_SIZE at splits = 1 is what the
eight private L2s cost (every XCD fetches the weights and its share of the activations); the growth with the slice count is the
partial tiles (each slice publishes M x N fp32, the last arriver reads them back)."""
import csv
import glob
import json
import os
import sys

REPS = 6
SHAPES = [  # name, key (Cin,Cout,kt,kh,kw,st,sh,sw,pt,ph,pw,B,T,H,W), residual
    ("layer4.0.conv2", "512,512,1,3,3,1,2,2,0,1,1,32,2,14,14", False),
    ("layer4.1.conv1", "2048,512,3,1,1,1,1,1,1,0,0,32,2,7,7", False),
    ("layer4.1.conv2", "512,512,1,3,3,1,1,1,0,1,1,32,2,7,7", False),
    ("layer4.0.downsample", "1024,2048,1,1,1,1,2,2,0,0,0,32,2,14,14", False),
]
SPLITS = (1, 2, 3, 4, 6)


def configs():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tuned = json.load(open(os.path.join(root, "anomaly_detection_on_video_amd", "tuned", "gfx950.json")))
    out = []
    for name, key, res in SHAPES:
        algo, splits = tuned[key]
        out.append((name, key, res, algo, splits, "tuned"))
        for s in SPLITS:
            if s != splits:
                out.append((name, key, res, algo, s, ""))
    return out


def run(timed):
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from anomaly_detection_on_video_amd import ops

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    res_out = []
    for name, key, res, algo, splits, tag in configs():
        cin, cout, kt, kh, kw, st, sh, sw, pt, ph, pw, B, T, H, W = (int(v) for v in key.split(","))
        w = torch.randn((cout, cin, kt, kh, kw), device=dev, generator=g) * (2.0 / (cin * kt * kh * kw)) ** 0.5
        ones = torch.ones(cout, device=dev)
        pc = ops.pack_conv(w, ones, ones * 0.1, ones * 0.05, ones, 1e-5, (st, sh, sw), (pt, ph, pw), name="one")
        x = torch.relu(torch.randn((B, cin, T, H, W), device=dev, generator=g))
        y = ops.conv3d_bn_act(x, pc, relu=True, algo=algo, splits=splits)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 40 if timed else REPS
        s.record()
        for _ in range(n):
            ops.conv3d_bn_act(x, pc, relu=True, algo=algo, splits=splits, out=y)
        e.record()
        torch.cuda.synchronize()
        res_out.append({"name": name, "algo": algo, "splits": splits, "tag": tag, "us": s.elapsed_time(e) / n * 1e3,
                        "gflop": 2.0 * y.numel() * cin * kt * kh * kw / 1e9})
    if timed:
        print(json.dumps(res_out))


def counter_rows(d, counter):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "conv3d_igemm" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]


def report(out):
    fetch, write = counter_rows(out + "/fetch", "FETCH_SIZE"), counter_rows(out + "/write", "WRITE_SIZE")
    times = {(t["name"], t["splits"]): t for t in json.load(open(out + "/time.json"))}
    cfgs = configs()
    per = REPS + 1
    assert len(fetch) == per * len(cfgs) and len(write) == per * len(cfgs), (len(fetch), len(write), per * len(cfgs))
    print("| conv | algo | K slices | us | TFLOP/s | 2 x FETCH_SIZE MB | WRITE_SIZE MB | partial tiles written MB (slices x M x N x 4) |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    for i, (name, key, res, algo, splits, tag) in enumerate(cfgs):
        f = fetch[i * per + 2 : (i + 1) * per]
        w = write[i * per + 2 : (i + 1) * per]
        # FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB on gfx950 per the guide's table: FETCH doubled (64-B requests counted as 32)
        fm = 2 * sum(f) / len(f) * 1024 / 1e6
        wm = sum(w) / len(w) * 1024 / 1e6
        cin, cout, kt, kh, kw, st, sh, sw, pt, ph, pw, B, T, H, W = (int(v) for v in key.split(","))
        To, Ho, Wo = (T + 2 * pt - kt) // st + 1, (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        part = splits * B * To * Ho * Wo * cout * 4 / 1e6 if splits > 1 else 0.0
        t = times[(name, splits)]
        print(f"| {name} {tag} | {algo} | {splits} | {t['us']:.1f} | {t['gflop'] / t['us'] * 1e3:.1f} | {fm:.0f} | {wm:.0f} | {part:.0f} |")


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "run":
        run(False)
    elif mode == "time":
        run(True)
    else:
        report(sys.argv[2])
