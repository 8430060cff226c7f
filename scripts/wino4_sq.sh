#!/bin/bash
# GPU box: SQ counters of the F(4x4) Winograd kernel on the probe's wide layer shapes (LDS bank conflicts, LDS / VALU / VMEM instruction
# cycles, wait cycles; separate --pmc passes, --kernel-trace only).   usage: bash scripts/wino4_sq.sh
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/wino4_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pass_$i -- python3 $R/scripts/wino4_probe.py > $OUT/pass_$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
python3 - <<PY
import collections, csv, glob
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_wino" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gpemsr::", "") + " grid " + r["Grid_Size"]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({c for v in per.values() for c in v})
for k, v in sorted(per.items()):
    print(k)
    for c in names:
        print(f"    {c:28s} {v.get(c, 0):.4g}")
PY
rm -rf $OUT/pass_*/
