#!/bin/bash
# GPU box: HBM-traffic + MFMA-busy counters of `bench.py --precision <p> --steps 1 --warmup 1` (3 forward passes of 16 tiles),
# separate --pmc passes with --kernel-trace only (gpurun refuses --pmc beside other trace domains), then the per-kernel time
# statistics of the same command.   usage: bash scripts/pmc_round2.sh <tag> <fp32|bf16>
set -u
TAG=${1:-r02}
PREC=${2:-fp32}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG}_${PREC}_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 500 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pass_$i -- python3 $R/bench.py --precision $PREC --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pass_$i.log 2>&1
  rc=$?
  echo "pass $i ($set) rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 $OUT/pass_$i.log; exit $rc; fi
done
# (bench.py runs warm-up + the official pass without events + one profiled pass: 3 forward passes with --steps 1 --warmup 1)
python3 $R/scripts/pmc_family2.py $OUT $R/gpurun_out/${TAG}_${PREC}_pmc_summary.json $PREC 3 "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, one pass each, --kernel-trace only) of \`python3 bench.py --precision $PREC --steps 1 --warmup 1 --no-cpu-baseline --no-extras\` (3 forward passes of 16 tiles: warm-up, official, profiled). FETCH_SIZE/WRITE_SIZE are KiB; FETCH doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM)." || exit 1
rm -rf $OUT/pass_*/
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --precision $PREC --no-cpu-baseline --no-extras --layer-report $R/gpurun_out/${TAG}_${PREC}_layers.tsv > $R/gpurun_out/${TAG}_${PREC}_bench_under_rocprof.json 2> $OUT/kt.log
rc=$?
echo "kernel stats rc=$rc"
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_${PREC}_kernel_stats.csv \;
rm -rf $OUT/kt
exit $rc
