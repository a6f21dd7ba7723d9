#!/bin/bash
# round 6, GPU run f: the rocprofv3 passes of the final state (tools/profile_bench.sh); the raw CSVs come back under gpurun_out/r6f/prof and
# are summarised in the build container (tools/summarize_prof.py tags profiles/traffic.json with the commit), then the default bench line
O=gpurun_out/r6f; rm -rf $O; mkdir -p $O
bash tools/profile_bench.sh $O/prof > $O/profile_bench.log 2>&1 || { tail -20 $O/profile_bench.log; exit 1; }
find $O/prof -name "*agent_info.csv" -delete; du -sh $O/prof
python bench.py --steps 20 --warmup 5 > $O/r06_final_bench.json 2> $O/bench.err; tail -c 400 $O/r06_final_bench.json
