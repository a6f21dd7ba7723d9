#!/bin/bash
# round 6, GPU run d: full GPU suite on the refactored LDS-DMA kernel body, mixed-tail timing + in-stream A/B, 5-rank rehearsal
O=gpurun_out/r6d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo rc=$? >> $O/tests_gpu.log; tail -4 $O/tests_gpu.log
python tools/time_tile_tail.py 162 2>/dev/null > $O/tile_tail_162.txt
python tools/time_tile_tail.py 200 2>/dev/null > $O/tile_tail_200.txt
paste -d'|' <(cut -c1-95 $O/tile_tail_162.txt) <(cut -c58-95 $O/tile_tail_200.txt)
bash tools/ab_env.sh $O/ab_mixed "plain:" "mixed:ADV_TUNED_OVERLAY=tools/overlays/layer3_conv3_mixed.json"
python tools/check_two_rank_stream.py --world 5 --shape ragged > $O/five_rank_rehearsal.txt 2>&1; echo rc=$? >> $O/five_rank_rehearsal.txt; tail -3 $O/five_rank_rehearsal.txt
