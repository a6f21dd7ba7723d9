#!/bin/bash
# round 6, GPU run c: tuned MFMA attention, 6-rank share-GPU rehearsal
O=gpurun_out/r6c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_mgfn.py -m gpu -x -q -k "glance" > $O/tests_glance.log 2>&1; echo rc=$? >> $O/tests_glance.log; tail -3 $O/tests_glance.log
for T in 290 2048 8192; do python tools/prof_mgfn_eval.py $T 5 2>/dev/null | sed 's/^/mfma2  /' >> $O/eval_ms.txt; done; cat $O/eval_ms.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eval8192 -- python3 tools/prof_mgfn_eval.py 8192 3 > $O/prof.log 2>&1
f=$(find $O/prof_eval8192 -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200 > $O/eval8192_kernel_stats.txt; cat $O/eval8192_kernel_stats.txt; rm -rf $O/prof_eval8192
python tools/check_two_rank_stream.py --world 6 --shape ragged > $O/six_rank_rehearsal.txt 2>&1; echo rc=$? >> $O/six_rank_rehearsal.txt; tail -4 $O/six_rank_rehearsal.txt
