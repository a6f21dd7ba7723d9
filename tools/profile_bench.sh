#!/bin/bash
# The rocprofv3 passes behind profiles/rNN_*: kernel trace + stats of bench.py, then FETCH_SIZE, WRITE_SIZE and the MFMA
# counters, each in its own pass (counters never together with tracing domains other than the kernel trace).
#   tools/profile_bench.sh gpurun_out/r02prof          (run on the GPU box, from the repo root)
set -e -o pipefail
OUT=$(realpath -m "$1")
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace/runc" -- python3 "$ROOT/bench.py" --steps 12 --warmup 3 --no-cpu-baseline --no-mgfn-train --no-pcie --sustain-s 0 > "$OUT/bench.json" 2> "$OUT/trace.log"
echo "trace done" >> "$OUT/progress.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch/runc" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --stream-start 0 --no-cpu-baseline --no-mgfn-train --no-pcie --sustain-s 0 > /dev/null 2> "$OUT/fetch.log"
echo "fetch done" >> "$OUT/progress.txt"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write/runc" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --stream-start 0 --no-cpu-baseline --no-mgfn-train --no-pcie --sustain-s 0 > /dev/null 2> "$OUT/write.log"
echo "write done" >> "$OUT/progress.txt"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d "$OUT/mfma/runc" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --stream-start 0 --no-cpu-baseline --no-mgfn-train --no-pcie --sustain-s 0 > /dev/null 2> "$OUT/mfma.log"
echo "mfma done" >> "$OUT/progress.txt"
cd "$ROOT"
find "$OUT" -name "*.csv" | head -20
