#!/usr/bin/env python3
"""Per-conv HBM traffic from the rocprofv3 PMC passes of bench.py (FETCH_SIZE / WRITE_SIZE, one value per dispatch)
next to the compulsory bytes of the layer (input once + weights + residual + output), B = 32.

    python tools/traffic_by_layer.py gpurun_out/final2/fetch/runc gpurun_out/final2/write/runc [stream parts per forward, default 2]
"""
import csv
import glob
import sys


def layers(B=32):
    out = [("conv1", 3, 64, (5, 7, 7), (2, 2, 2), (16, 224, 224), False)]
    T, H = 4, 55
    inpl = 64
    cfg = [("layer1", 64, 3, 1, [1, 1, 1]), ("layer2", 128, 4, 2, [1, 0, 1, 0]), ("layer3", 256, 6, 2, [1, 0, 1, 0, 1, 0]), ("layer4", 512, 3, 2, [0, 1, 0])]
    for name, planes, n, stride, tc in cfg:
        if name == "layer2":
            T = 2
        for b in range(n):
            s = stride if b == 0 else 1
            kt = 1 + 2 * tc[b]
            out.append((f"{name}.{b}.conv1", inpl, planes, (kt, 1, 1), (1, 1, 1), (T, H, H), False))
            out.append((f"{name}.{b}.conv2", planes, planes, (1, 3, 3), (1, s, s), (T, H, H), False))
            Ho = (H + 2 - 3) // s + 1
            if b == 0 and s == 1:  # layer1.0: downsample folded into conv3 (one conv over [x ; h], no residual)
                out.append((f"{name}.{b}.conv3+downsample", inpl + planes, planes * 4, (1, 1, 1), (1, 1, 1), (T, Ho, Ho), False))
            else:
                if b == 0:
                    out.append((f"{name}.{b}.downsample", inpl, planes * 4, (1, 1, 1), (1, s, s), (T, H, H), False))
                out.append((f"{name}.{b}.conv3", planes, planes * 4, (1, 1, 1), (1, 1, 1), (T, Ho, Ho), True))
            H = Ho
            inpl = planes * 4
    res = []
    for name, cin, cout, k, s, (t, h, w), resid in out:
        pad = (k[0] // 2, k[1] // 2, k[2] // 2) if name != "conv1" else (2, 3, 3)
        if k == (1, 1, 1):
            pad = (0, 0, 0)
        to = (t + 2 * pad[0] - k[0]) // s[0] + 1
        ho = (h + 2 * pad[1] - k[1]) // s[1] + 1
        wo = (w + 2 * pad[2] - k[2]) // s[2] + 1
        x = B * cin * t * h * w * 4
        y = B * cout * to * ho * wo * 4
        wt = cout * cin * k[0] * k[1] * k[2] * 4
        macs = B * cout * to * ho * wo * cin * k[0] * k[1] * k[2]
        wbytes = y
        if name == "conv1":  # fused with maxpool1: writes the per-brick partial maxima (27 per brick and channel), not the activation
            tp, hp, wp = to // 2, (ho - 3) // 2 + 1, (wo - 3) // 2 + 1
            wbytes = B * tp * ((2 * hp + 1 + 3) // 4) * ((2 * wp + 1 + 15) // 16) * cout * 27 * 4
        if name == "layer1.2.conv3":  # fused with maxpool2: writes the temporally pooled output
            wbytes = y // 2
        res.append(dict(name=name, read=x + wt + (y if resid else 0), write=wbytes, macs=macs))
    return res


def per_dispatch(d):
    rows = list(csv.DictReader(open(glob.glob(d + "/*_counter_collection.csv")[0])))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def last_forward(rows, parts=2):
    """Per conv of the plan: [counter KB, ns, kernel] summed over the stream parts of the last forward (dispatch ids follow
    launch order: part 0's whole chain, then part 1's)."""
    # (a forward ends in global_avgpool, or in the conv whose epilogue takes the mean: EPI_AVG = 6)
    ends = [i for i, r in enumerate(rows) if "global_avgpool" in r["Kernel_Name"] or "conv3d_igemm_dma_kernel<128, 64, 16, false, 2, 6," in r["Kernel_Name"]]
    total = None
    for p in range(parts):
        e = len(ends) - parts + p
        one = one_chain(rows[ends[e - 1] + 1 : ends[e] + 1])
        if total is None:
            total = one
        else:
            for t, o in zip(total, one):
                t[0] += o[0]
                t[1] += o[1]
                if o[2] != t[2]:
                    t[2] += " | " + o[2]
    return total


def one_chain(seg):
    convs = []
    for r in seg:
        n = r["Kernel_Name"]
        if "conv3d_igemm" in n:
            convs.append([float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), n.split("<")[1].split(">")[0]])
        elif "splitk_reduce" in n:
            convs[-1][0] += float(r["Counter_Value"])
            convs[-1][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            convs[-1][2] += " +reduce"
        elif "split_w_kernel" in n:  # the stem's column-parity planes pass (ADV_STEM_S2W, default since round 5): billed to conv1
            pending = [float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])]
            convs.append([pending[0], pending[1], "split_w + "])
            convs[-1].append("merge-next")
    out = []
    for c in convs:  # fold a split_w entry into the conv launch that follows it
        if out and len(out[-1]) == 4:
            head = out.pop()
            c = [c[0] + head[0], c[1] + head[1], head[2] + c[2]]
        out.append(c)
    return out


def main():
    parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    f = last_forward(per_dispatch(sys.argv[1]), parts)
    w = last_forward(per_dispatch(sys.argv[2]), parts)
    L = layers()
    assert len(f) == len(L) == len(w), (len(f), len(w), len(L))
    print("| conv | kernel | compulsory read MB | FETCH_SIZE x2 MB | ratio | compulsory write MB | WRITE_SIZE MB | us | TFLOP/s |\n|---|---|---:|---:|---:|---:|---:|---:|---:|")
    tr = tf = tw = tww = 0
    for l, (fk, fd, kn), (wk, wd, _) in zip(L, f, w):
        fb, wb = 2 * fk * 1024, wk * 1024
        print("| %s | %s | %.0f | %.0f | %.2f | %.0f | %.0f | %.0f | %.1f |" % (l["name"], kn, l["read"] / 1e6, fb / 1e6, fb / l["read"], l["write"] / 1e6, wb / 1e6,
                                                                     fd / 1e3, 2 * l["macs"] / fd / 1e3))
        tr += l["read"]; tf += fb; tw += l["write"]; tww += wb
    print("| total | | %.0f | %.0f | %.2f | %.0f | %.0f | | |" % (tr / 1e6, tf / 1e6, tf / tr, tw / 1e6, tww / 1e6))


if __name__ == "__main__":
    main()
