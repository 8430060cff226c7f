#!/bin/bash
# GPU box: the judged artifacts of round 6 (copied into profiles/ afterwards).  usage: bash scripts/profile_round6.sh [tag]
set -u
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
# 1. counters + kernel statistics of both legs (writes gpurun_out/${TAG}_<prec>_*)
bash scripts/pmc_round2.sh $TAG bf16 > $OUT/pmc_bf16.log 2>&1; tail -2 $OUT/pmc_bf16.log
bash scripts/pmc_round2.sh $TAG fp32 > $OUT/pmc_fp32.log 2>&1; tail -2 $OUT/pmc_fp32.log
# (bench.py reads the counter summary of THIS tree from profiles/: copy it there before the default line is printed)
cp $R/gpurun_out/${TAG}_fp32_pmc_summary.json $R/gpurun_out/${TAG}_bf16_pmc_summary.json $R/profiles/ 2>/dev/null
# 2. the default bench line: official fp32 value + bf16 leg + x16 legs + training leg + volume mode + cpu baseline
# (the driver's exact command; the ONE stdout line is the compact one, the full document goes to --detail)
python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/${TAG}_default_bench_detail.json --layer-report $OUT/${TAG}_default_layers.tsv > $OUT/${TAG}_default_bench_line.json 2> $OUT/bench.err
wc -c $OUT/${TAG}_default_bench_line.json; tail -c 300 $OUT/${TAG}_default_bench_line.json; echo
# 3. the training steps as their own lines (stage 3 with extras; stage 2; stage 1)
python3 bench.py --mode train --layer-report $OUT/${TAG}_train_layers.tsv > $OUT/${TAG}_train_bench.json 2>> $OUT/bench.err
python3 bench.py --mode train2 > $OUT/${TAG}_train2_bench.json 2>> $OUT/bench.err
python3 bench.py --mode train1 > $OUT/${TAG}_train1_bench.json 2>> $OUT/bench.err
ls -la $OUT $R/gpurun_out | head -40
