#!/bin/bash
# round 6, GPU run l: the whole-video scoring pass eager vs one HIP-graph replay (what a graph per length would buy)
O=gpurun_out/r6l; mkdir -p $O
for T in 290 2048 8192; do timeout -k 10 200 python tools/prof_mgfn_eval.py $T 10 graph 2>/dev/null | grep eval >> $O/eval_graph_ms.txt; done; cat $O/eval_graph_ms.txt
