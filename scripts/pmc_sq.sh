#!/bin/bash
# GPU box: per-kernel SQ counters (matrix pipe busy, LDS activity / bank conflicts, wait cycles) of one bf16 forward.
# usage: bash scripts/pmc_sq.sh <tag>
set -u
TAG=${1:-sq}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG}_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters.txt
wc -l $OUT/sq_counters.txt
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pass_$i -- python3 $R/bench.py --precision bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pass_$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
python3 - <<PY
import collections, csv, glob
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gpemsr::", "")
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({c for v in per.values() for c in v})
with open("$R/gpurun_out/${TAG}_sq_counters.tsv", "w") as o:
    o.write("kernel\t" + "\t".join(names) + "\n")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", kv[1].get("SQ_WAVE_CYCLES", 0))):
        o.write(k[:90] + "\t" + "\t".join("%.4g" % v.get(c, 0) for c in names) + "\n")
print("wrote", "$R/gpurun_out/${TAG}_sq_counters.tsv")
PY
rm -rf $OUT/pass_*/
