#!/bin/bash
# GPU box: the judged artifacts of round 5 (copied into profiles/ afterwards).  usage: bash scripts/profile_round5.sh [tag]
set -u
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
# 1. the default bench line: official fp32 value + bf16 leg + x16 legs + training leg + volume mode + cpu baseline
python3 bench.py --layer-report $OUT/${TAG}_default_layers.tsv > $OUT/${TAG}_default_bench.json 2> $OUT/bench.err
tail -c 300 $OUT/${TAG}_default_bench.json; echo
# 2. counters + kernel statistics of both legs (writes gpurun_out/${TAG}_<prec>_*)
bash scripts/pmc_round2.sh $TAG bf16 > $OUT/pmc_bf16.log 2>&1; tail -2 $OUT/pmc_bf16.log
bash scripts/pmc_round2.sh $TAG fp32 > $OUT/pmc_fp32.log 2>&1; tail -2 $OUT/pmc_fp32.log
# 3. the training steps as their own lines (stage 3 with extras; stage 2; stage 1)
python3 bench.py --mode train --layer-report $OUT/${TAG}_train_layers.tsv > $OUT/${TAG}_train_bench.json 2>> $OUT/bench.err
python3 bench.py --mode train2 > $OUT/${TAG}_train2_bench.json 2>> $OUT/bench.err
python3 bench.py --mode train1 > $OUT/${TAG}_train1_bench.json 2>> $OUT/bench.err
ls -la $OUT $R/gpurun_out | head -40
