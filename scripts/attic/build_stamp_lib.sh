#!/bin/bash
# Diagnostic library with in-kernel s_memtime stamps of the bf16 convolution kernels (never shipped, never loaded by default):
#   bash scripts/build_stamp_lib.sh && GPEMSR_LIB_PATH=gpemsr_amd/lib/libgpemsr_stamp.so python3 scripts/stamp_probe16.py
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/gpemsr_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DGP16_STAMP -I$R/include -I$R/gpemsr_amd/csrc -c $R/gpemsr_amd/csrc/conv_bf16.hip -o $L/conv_bf16_stamp.o
OBJS=$(ls $L/*.o | grep -v "conv_bf16.o\|_stamp.o\|vgg_mask_stamp")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libgpemsr_stamp.so $OBJS $L/conv_bf16_stamp.o
echo built $L/libgpemsr_stamp.so
