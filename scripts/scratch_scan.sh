#!/bin/bash
# CPU-side check (no GPU needed): every kernel of every .hip file whose compiled code uses scratch memory (register spills or
# dynamically indexed private arrays).  Round 2 found 164-356 bytes per lane of scratch in five kernels this way -- one of them in
# EVERY fp32 convolution launch (+150 GB of HBM traffic per step).  usage: bash scripts/scratch_scan.sh [file.hip ...]
R=$(cd "$(dirname "$0")/.." && pwd)
FILES=${@:-$R/gpemsr_amd/csrc/*.hip}
for f in $FILES; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I$R/include -I$R/gpemsr_amd/csrc -c $f -o /dev/null \
      -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|ScratchSize| VGPRs:" | sed 's/.*remark: //; s/\[-Rpass.*//' | paste - - - \
      | grep -v "ScratchSize \[bytes/lane\]: 0" | sed "s|^|$(basename $f): |"
done
