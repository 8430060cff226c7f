#!/bin/bash
# GPU box: PMC counters of the conv kernels on single micro-bench shapes.  usage: bash scripts/pmc_shapes.sh <precision> shape1 shape2 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PREC=$1; shift
for shp in "$@"; do
  for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_shapes/$PREC/$shp -- python3 $R/scripts/conv_microbench.py 3 $shp $PREC > /dev/null 2>&1
  done
  echo "== $PREC $shp"; python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_shapes/$PREC/$shp $R/gpurun_out/pmc_shapes/${PREC}_$shp.json | grep -E "conv_mfma|conv_split"
  python3 - <<PY
import csv,glob
ds=[]
for f in glob.glob("$R/gpurun_out/pmc_shapes/$PREC/$shp/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_mfma" in r["Kernel_Name"] or "conv_split" in r["Kernel_Name"]:
            ds.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("dur_us median", sorted(ds)[len(ds)//2]/1e3, "n", len(ds))
PY
done
