#!/bin/bash
# Diagnostic builds of the conv kernels (timing experiments; results are wrong on purpose):
#   tools/diag_build.sh 4 8 16 24   ->  tools/_build/libadvhip_diag<bits>.so, used via ADVHIP_LIBRARY=...
# bits: see ADVHIP_DIAG in csrc/conv_igemm.hip.
set -e
cd "$(dirname "$0")/.."
C=anomaly_detection_on_video_amd/csrc
mkdir -p tools/_build
for bits in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -DADVHIP_DIAG=$bits -c $C/conv_igemm.hip -o tools/_build/conv_igemm_diag$bits.o &
done
wait
for bits in "$@"; do
  objs=$(ls $C/build/*.o | grep -v conv_igemm)
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_build/libadvhip_diag$bits.so tools/_build/conv_igemm_diag$bits.o $objs
  echo tools/_build/libadvhip_diag$bits.so
done
