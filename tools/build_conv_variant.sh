#!/bin/bash
# tools/build_conv_variant.sh NAME "-DFLAG=..."  -> tools/_build/libadvhip_NAME.so (conv_igemm.hip rebuilt with the flags, other objects as built)
set -e
mkdir -p tools/_build
CS=anomaly_detection_on_video_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC $2 -c $CS/conv_igemm.hip -o tools/_build/conv_$1.o
OBJS=$(ls $CS/build/*.o | grep -v conv_igemm.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_build/libadvhip_$1.so $OBJS tools/_build/conv_$1.o
rm -f tools/_build/conv_$1.o
echo built tools/_build/libadvhip_$1.so
