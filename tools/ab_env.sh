#!/bin/bash
# A/B of environment settings on one box: tools/ab_env.sh OUT "name1:VAR=val" "name2:VAR=val VAR2=val" ...   ("name:" = no variables)
# BENCH_ARGS: extra bench.py arguments
OUT=$1; shift
mkdir -p $OUT
for round in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; vars=${spec#*:}
  env $vars python bench.py --no-cpu-baseline --no-pcie --no-mgfn-train --sustain-s 2 $BENCH_ARGS > $OUT/bench_${name}_$round.json 2> $OUT/bench_${name}_$round.err || exit 1
  python -c "import json; d=json.load(open('$OUT/bench_${name}_$round.json')); print('$name', $round, d['value'], d['roofline']['frac'], d['sustained']['clips_per_s'])" | tee -a $OUT/summary.txt
done
done
