#!/bin/bash
# GPU box: SQ / LDS counters of the conv_bf16 kernels on two shapes (separate --pmc passes, kernel-trace only)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for shape in "256 256 3 80 128 128 0" "64 64 3 80 128 128 0"; do
  tag=$(echo $shape | tr ' ' '_')
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/${tag}_p$i -- python3 $R/scripts/pmc_conv16.py $shape > $OUT/${tag}_p$i.log 2>&1
    echo "$tag pass $i rc=$?"
  done
done
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc16"
for d in sorted(set(p.rsplit("_p", 1)[0] for p in glob.glob(root + "/*_p[0-9]"))):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(d + "_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv" in r["Kernel_Name"] and "cast" not in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(os.path.basename(d))
    for k in sorted(acc): print("   %-28s %16.0f  (%d dispatches)" % (k, acc[k] / max(n[k], 1), n[k]))
PY
rm -rf $OUT/*_p[0-9]
