#!/bin/bash
# A/B of library variants on one box: tools/ab_lib_variants.sh OUT name1=path1.so name2=path2.so ...  ("base" = the in-tree library)
OUT=$1; shift
mkdir -p $OUT
SO=anomaly_detection_on_video_amd/csrc/libadvhip.so
cp $SO /tmp/base.so
for round in 1 2; do
for spec in "$@"; do
  name=${spec%%=*}; path=${spec#*=}
  if [ "$path" = "base" ]; then cp /tmp/base.so $SO; else cp $path $SO; fi
  python bench.py --no-cpu-baseline --no-pcie --sustain-s 2 > $OUT/bench_${name}_$round.json 2> $OUT/bench_${name}_$round.err || exit 1
  python -c "import json; d=json.load(open('$OUT/bench_${name}_$round.json')); print('$name', $round, d['value'], d['roofline']['frac'], d['sustained']['clips_per_s'], d['mgfn_train_step']['ms_per_step'])" | tee -a $OUT/summary.txt
done
done
cp /tmp/base.so $SO
