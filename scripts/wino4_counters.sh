#!/bin/bash
# GPU box: L2-miss bytes (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE, KiB) of the F(4x4) Winograd kernel on the probe's four wide
# layer shapes, per launch-order grouping GPEMSR_WINO4_CGROUP (separate --pmc passes, --kernel-trace only).   usage: bash scripts/wino4_counters.sh
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/wino4_counters
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for g in 1 2 4 8; do
  for set in FETCH_SIZE WRITE_SIZE; do
    export GPEMSR_WINO4_CGROUP=$g
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/g${g}_$set -- python3 $R/scripts/wino4_probe.py > $OUT/g${g}_$set.log 2>&1 || { tail -3 $OUT/g${g}_$set.log; exit 1; }
  done
  python3 - $OUT $g <<'PY'
import csv, glob, sys
out, g = sys.argv[1], sys.argv[2]
tot = {}
for cs in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/g{g}_{cs}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if "conv_wino4" in r["Kernel_Name"] and r["Counter_Name"] == cs:
            key = (r["Grid_Size"], cs)
            tot.setdefault(key, []).append(float(r["Counter_Value"]))
for grid in sorted({k[0] for k in tot}):
    fe = tot.get((grid, "FETCH_SIZE"), [0]); wr = tot.get((grid, "WRITE_SIZE"), [0])
    print(f"cgroup {g}: grid {grid:>9s}: {len(fe)} launches, fetch {2 * sum(fe) / len(fe) * 1024 / 1e9:7.3f} GB, write {sum(wr) / len(wr) * 1024 / 1e9:6.3f} GB per launch", flush=True)
PY
  rm -rf $OUT/g${g}_FETCH_SIZE $OUT/g${g}_WRITE_SIZE
done
