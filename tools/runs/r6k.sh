#!/bin/bash
# round 6, GPU run k (study, not a product change): what B = 64 crop-clips per launch would buy -- the conv table tuned at B = 64, then the
# three-lane stream at --batch 64 against --batch 32 on the same box
O=gpurun_out/r6k; mkdir -p $O
timeout -k 10 600 python tools/tune_convs.py --batch 64 --reps 4 --algos 161,162,163,164,166,167,168,67,68,99,192 --out $O/tuned_b64.json > $O/tune_b64.txt 2>&1; tail -3 $O/tune_b64.txt
for r in 1 2; do
  python bench.py --batch 32 --no-cpu-baseline --no-pcie --no-mgfn-train --sustain-s 2 > $O/b32_$r.json 2> $O/b32_$r.err
  ADV_TUNED_OVERLAY=$O/tuned_b64.json python bench.py --batch 64 --no-cpu-baseline --no-pcie --no-mgfn-train --sustain-s 2 > $O/b64_$r.json 2> $O/b64_$r.err
  python -c "
import json
for n in ('b32','b64'):
    d=json.loads(open('$O/%s_$r.json'%n).read().strip().splitlines()[-1]); print(n, $r, d['value'], d['roofline']['frac'], d['sustained']['clips_per_s'])" | tee -a $O/summary.txt
done
