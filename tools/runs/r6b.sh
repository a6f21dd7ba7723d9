#!/bin/bash
# round 6, GPU run b: MFMA attention tests + scoring-pass A/B, 256x64 tile PMC, layer-4 split-K in-stream A/B
# (ADVHIP_GLANCE_MFMA_MIN_T: an environment switch of the STUDY build this script ran on; the committed library has the threshold as a constant)
O=gpurun_out/r6b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_mgfn.py -m gpu -x -q -k "glance" > $O/tests_glance.log 2>&1; echo rc=$? >> $O/tests_glance.log; tail -3 $O/tests_glance.log
python -m pytest tests/test_hip_strict.py -m gpu -x -q > $O/tests_strict.log 2>&1; echo rc=$? >> $O/tests_strict.log; tail -3 $O/tests_strict.log
for T in 290 2048 8192; do
  ADVHIP_GLANCE_MFMA_MIN_T=100000000 python tools/prof_mgfn_eval.py $T 5 2>/dev/null | sed 's/^/vector /' >> $O/eval_ms.txt
  python tools/prof_mgfn_eval.py $T 5 2>/dev/null | sed 's/^/mfma   /' >> $O/eval_ms.txt
done
cat $O/eval_ms.txt
ADVHIP_GLANCE_MFMA_MIN_T=100000000 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eval8192_vector -- python3 tools/prof_mgfn_eval.py 8192 3 > $O/prof_vector.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eval8192_mfma -- python3 tools/prof_mgfn_eval.py 8192 3 > $O/prof_mfma.log 2>&1
for d in vector mfma; do f=$(find $O/prof_eval8192_$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -14 $f | cut -c1-160; done > $O/eval8192_kernel_stats.txt
cat $O/eval8192_kernel_stats.txt
# 256x64 vs 128x64: matrix-pipe busy cycles of the launches alone
for key in 64,64,1,3,3,1,1,1,0,1,1,32,4,55,55 256,64,3,1,1,1,1,1,1,0,0,32,4,55,55; do for algo in 162 169; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d $O/pmc_${algo}_${key%%,1,*} -- python3 tools/run_one_conv.py --key $key --algo $algo --splits 1 --reps 12 --relu-input > $O/pmc_${algo}.log 2>&1
  echo "== key $key algo $algo" >> $O/tile256_pmc.txt; python tools/pmc_by_kernel.py $O/pmc_${algo}_${key%%,1,*} conv3d_igemm >> $O/tile256_pmc.txt 2>&1
done; done
cat $O/tile256_pmc.txt
rm -rf $O/pmc_1* $O/prof_eval8192_*
# layer-4 K slices in the three-lane stream: tuned 3 vs 4 vs 6
bash tools/ab_env.sh $O/ab_l4 "s3:" "s4:ADV_TUNED_OVERLAY=tools/overlays/layer4_splits4.json" "s6:ADV_TUNED_OVERLAY=tools/overlays/layer4_splits6.json"
