#!/bin/bash
# round 6, GPU run h: the fused narrow-FFN launches: parity, timing alone, the training step with and without them
O=gpurun_out/r6h; mkdir -p $O
python -m pytest tests/test_hip_mgfn.py -m gpu -x -q -k "fused_narrow_ffn" > $O/tests_ffn.log 2>&1; echo rc=$? >> $O/tests_ffn.log; tail -15 $O/tests_ffn.log
# (a stand-alone timing tool ran here in the first build of this script; it crashed inside a bare graph capture of backward() and was removed -- the durations come from tools/runs/r6i.sh)
python -m pytest tests/test_hip_mgfn_bench_shape.py tests/test_hip_train.py tests/test_hip_strict.py -m gpu -x -q > $O/tests_train.log 2>&1; echo rc=$? >> $O/tests_train.log; tail -5 $O/tests_train.log
for v in 1 0 1 0; do ADV_MGFN_FUSED_FFN=$v python tools/prof_mgfn_train.py 30 graph 2>/dev/null | tail -1 | sed "s/^/fused=$v /" >> $O/train_step_ms.txt; done; cat $O/train_step_ms.txt
