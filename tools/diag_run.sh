#!/bin/bash
# Timing experiments behind profiles/*_pmc_notes.md: which part of the memory side costs what.
# Run on the GPU box after tools/diag_build.sh 4 36 8 16 24.
cd "$(dirname "$0")/.."
for key_algo in "256,64,3,1,1,1,1,1,1,0,0,32,4,55,55 67 1" "3,64,5,7,7,2,2,2,2,3,3,32,16,224,224 67 1" "128,128,1,3,3,1,1,1,0,1,1,32,2,28,28 67 2" "256,256,1,3,3,1,1,1,0,1,1,32,2,14,14 67 3" "512,128,1,1,1,1,1,1,0,0,0,32,2,28,28 67 1"; do
  set -- $key_algo
  echo "=== $1 algo $2 splits $3"
  run() { printf "%-34s" "$1"; shift; env "$@" python tools/run_one_conv.py --key $K --algo $A --splits $S --reps 30 --relu-input | sed 's/.*: //'; }
  K=$1; A=$2; S=$3
  run "(warm-up)" X=1
  run "product" X=1
  run "A loads out of range (no traffic)" ADVHIP_DEBUG_ZERO_RECORDS=1
  run "B loads out of range" ADVHIP_DEBUG_ZERO_RECORDS=2
  run "A+B out of range" ADVHIP_DEBUG_ZERO_RECORDS=3
  run "A as one 16-byte DMA per wave" ADVHIP_LIBRARY=tools/_build/libadvhip_diag4.so
  run "... in a channels-last pattern" ADVHIP_LIBRARY=tools/_build/libadvhip_diag36.so
  run "no A DMA issued" ADVHIP_LIBRARY=tools/_build/libadvhip_diag8.so
  run "no B DMA issued" ADVHIP_LIBRARY=tools/_build/libadvhip_diag16.so
  run "no DMA issued" ADVHIP_LIBRARY=tools/_build/libadvhip_diag24.so
done
