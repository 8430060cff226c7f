#!/bin/bash
# GPU box: the judged artifacts of round 4 (copied into profiles/ afterwards).  usage: bash scripts/profile_round4.sh [tag]
set -u
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
# 1. the default bench line (official fp32 value + extras incl. the bf16 leg, volume mode, cpu baseline)
python3 bench.py --layer-report $OUT/${TAG}_default_layers.tsv > $OUT/${TAG}_default_bench.json 2> $OUT/bench.err
tail -c 300 $OUT/${TAG}_default_bench.json; echo
# 2. counters + kernel statistics of both legs (writes gpurun_out/${TAG}_<prec>_*)
bash scripts/pmc_round2.sh $TAG bf16 > $OUT/pmc_bf16.log 2>&1; tail -2 $OUT/pmc_bf16.log
bash scripts/pmc_round2.sh $TAG fp32 > $OUT/pmc_fp32.log 2>&1; tail -2 $OUT/pmc_fp32.log
# 3. x16 configuration (BASELINE configs[3] geometry, one GPU's share: 8 windows of 64 x 64 -> 1024 x 1024), both precisions
python3 bench.py --scale 16 --lr 64 --tiles 8 --no-cpu-baseline --no-extras > $OUT/${TAG}_x16_fp32_bench.json 2>> $OUT/bench.err
python3 bench.py --scale 16 --lr 64 --tiles 8 --precision bf16 --no-cpu-baseline --no-extras > $OUT/${TAG}_x16_bf16_bench.json 2>> $OUT/bench.err
# 4. the stage-3 training step (BASELINE configs[4] geometry)
python3 bench.py --mode train --layer-report $OUT/${TAG}_train_layers.tsv > $OUT/${TAG}_train_bench.json 2>> $OUT/bench.err
# 5. stage-2 / stage-1 training steps and the kernel statistics of the stage-1 step
python3 bench.py --mode train2 > $OUT/${TAG}_train2_bench.json 2>> $OUT/bench.err
python3 bench.py --mode train1 --layer-report $OUT/${TAG}_train1_layers.tsv > $OUT/${TAG}_train1_bench.json 2>> $OUT/bench.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train1_prof -o train1 -- python3 $R/bench.py --mode train1 --no-profile --no-extras --no-cpu-baseline > $OUT/${TAG}_train1_bench_under_rocprof.json 2>> $OUT/bench.err )
cp $(find $OUT/train1_prof -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_train1_kernel_stats.csv 2>/dev/null
rm -rf $OUT/train1_prof
ls -la $OUT $R/gpurun_out | head -60
