#!/bin/bash
# GPU box: refresh the judged artifacts of a round.  usage: bash scripts/profile_round.sh <tag e.g. r01_final>
set -u
TAG=${1:-r01_final}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the default bench line (official number) + per-layer table
python3 $R/bench.py --layer-report $OUT/${TAG}_fp32_layers.tsv > $OUT/${TAG}_fp32_bench.json 2> $OUT/bench.err
tail -c 600 $OUT/${TAG}_fp32_bench.json
# 2. rocprofv3 kernel trace + stats of the same command (no cpu baseline / extras: kernel averages only)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --no-cpu-baseline --no-extras > $OUT/${TAG}_fp32_bench_under_rocprof.json 2> $OUT/rocprof.err
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_fp32_kernel_stats.csv \;
# 3. same for the bf16x3 mode
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt3 -- python3 $R/bench.py --precision bf16x3 --no-cpu-baseline --no-extras --layer-report $OUT/${TAG}_bf16x3_layers.tsv > $OUT/${TAG}_bf16x3_bench_under_rocprof.json 2>> $OUT/rocprof.err
find $OUT/kt3 -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_bf16x3_kernel_stats.csv \;
# 4. x16 configuration (BASELINE configs[3] geometry, one GPU's share: 8 windows of 64x64 -> 1024x1024)
python3 $R/bench.py --scale 16 --lr 64 --tiles 8 --no-cpu-baseline --no-extras > $OUT/${TAG}_x16_bench.json 2>> $OUT/bench.err
rm -rf $OUT/kt $OUT/kt3
ls -la $OUT
