#!/bin/bash
# GPU box: HBM-traffic + MFMA-busy counters of the default bench command, separate --pmc passes (no trace domains besides
# --kernel-trace).  usage: bash scripts/pmc_round.sh <tag>
set -u
TAG=${1:-r01c}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG}_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pass_$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pass_$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
python3 $R/scripts/pmc_family.py $OUT $R/gpurun_out/${TAG}_fp32_pmc_summary.json "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, one pass each) of \`python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras\` (2 forward passes of 16 tiles), fp32 path, HEAD of round 1. FETCH_SIZE/WRITE_SIZE are KiB; FETCH doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM)."
rm -rf $OUT/pass_*/
