#!/bin/bash
# Diagnostic libraries of the F(4x4,3x3) Winograd kernel with single phases compiled OUT (never shipped, never loaded by default; results are
# wrong on purpose):   bash scripts/build_wino_probe_lib.sh 3 4 7 24
#   for m in 0 3 4 7 24; do GPEMSR_LIB_PATH=gpemsr_amd/lib/libgpemsr_w4skip$m.so python3 scripts/wino4_probe.py; done
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/gpemsr_amd/lib
OBJS=$(ls $L/*.o | grep -v "conv_wino4.o\|_stamp.o\|_wprobe.o\|_w4skip")
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DW4_SKIP=$m -I$R/include -I$R/gpemsr_amd/csrc -c $R/gpemsr_amd/csrc/conv_wino4.hip -o $L/conv_wino4_w4skip$m.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libgpemsr_w4skip$m.so $OBJS $L/conv_wino4_w4skip$m.o
  echo built $L/libgpemsr_w4skip$m.so
done
