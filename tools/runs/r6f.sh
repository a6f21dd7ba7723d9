#!/bin/bash
# round 6, GPU run f: the profile set of the final state (profiles/r06_final_*), as tools/profile_bench.sh + tools/summarize_prof.py produce it
O=gpurun_out/r6f; mkdir -p $O
bash tools/profile_bench.sh $O/prof > $O/profile_bench.log 2>&1 || { tail -20 $O/profile_bench.log; exit 1; }
python tools/summarize_prof.py --trace $O/prof/trace/runc --bench $O/prof/bench.json --fetch $O/prof/fetch/runc --write $O/prof/write/runc --mfma $O/prof/mfma/runc --out $O/r06_final --parts 1 > $O/summarize.log 2>&1 || { tail -20 $O/summarize.log; exit 1; }
cp profiles/traffic.json $O/traffic.json
python tools/traffic_by_layer.py $O/prof/fetch/runc $O/prof/write/runc 1 > $O/r06_final_traffic_by_layer.md 2> $O/traffic_by_layer.err
f=$(find $O/prof/trace -name "*kernel_stats.csv" | head -1); cp $f $O/r06_final_kernel_stats.csv
cp $O/prof/bench.json $O/r06_final_bench_profiled.json
python tools/non_advhip_kernels.py $O/prof/trace/runc --after-first-step > $O/r06_final_non_advhip_kernels.txt 2>&1
python tools/concurrency_profile.py $O/prof/trace/runc > $O/r06_final_concurrency.txt 2>&1
rm -rf $O/prof/trace $O/prof/fetch $O/prof/write $O/prof/mfma
ls -la $O; head -12 $O/r06_final_summary.md
python bench.py --steps 20 --warmup 5 > $O/r06_final_bench.json 2> $O/bench.err; tail -c 600 $O/r06_final_bench.json
