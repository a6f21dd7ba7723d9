#!/bin/bash
# Where the waves' cycles go IN THE STREAM (three lanes, kernels sharing the chip), not per kernel alone: SQ wave-cycle buckets and the L2 hit rate
# summed over the conv kernels of a short bench.py run, one rocprofv3 --pmc pass per counter group.   tools/stream_counters.sh OUTDIR
set -e -o pipefail
OUT=$(realpath -m "$1"); ROOT=$(pwd); mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --stream-start 0 --no-cpu-baseline --no-mgfn-train --no-pcie --sustain-s 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/sq1" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/sq1.log"
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/sq2" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/sq2.log"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/tcc" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/tcc.log"
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3d_igemm" in r["Kernel_Name"] or "split_w" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot):
    print(f"{k:24s} {tot[k]:.4e}")
wc = tot.get("SQ_WAVE_CYCLES", 0)
if wc:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        print(f"{k} / SQ_WAVE_CYCLES = {tot[k] / wc:.3f}")
if tot.get("TCC_HIT_sum"):
    print(f"L2 hit rate = {tot['TCC_HIT_sum'] / (tot['TCC_HIT_sum'] + tot['TCC_MISS_sum']):.3f}")
PY
