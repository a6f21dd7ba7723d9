#!/bin/bash
# round 6, GPU run i: kernel durations of the fused narrow-FFN launches inside one graph-replayed training step
O=gpurun_out/r6i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/prof_mgfn_train.py 10 graph serial > $O/log.txt 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -1 "$f" | cut -c1-120; grep -i "ffn_block" "$f" | cut -c1-260; cp "$f" $O/kernel_stats.csv; fi
python tools/mgfn_replay_launches.py $O/prof > $O/launches.txt 2>&1; tail -8 $O/launches.txt
rm -rf $O/prof
