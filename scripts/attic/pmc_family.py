#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes (one directory per counter set) of one bench.py command into the conv-family summary that
bench.py reads for `roofline.traffic`:  python3 scripts/pmc_family.py <dir with pass_*/ subdirs> <out.json> "<note>"
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md HBM section)."""
import collections, csv, glob, json, sys

root, out_path, note = sys.argv[1], sys.argv[2], sys.argv[3]
per = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
fam = collections.defaultdict(float)
n_fam = 0
per_kernel = {}
for k, c in per.items():
    n = max(len(disp[(k, name)]) for name in c)
    rd = 2.0 * 1024.0 * c.get("FETCH_SIZE", 0.0)
    wr = 1024.0 * c.get("WRITE_SIZE", 0.0)
    per_kernel[k[:90]] = {"dispatches": n, "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr,
                          "mfma_busy_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), "gui_active": c.get("GRBM_GUI_ACTIVE", 0.0)}
    if "conv_mfma_kernel" in k:
        n_fam += n
        fam["rd"] += rd; fam["wr"] += wr; fam["mfma"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); fam["gui"] += c.get("GRBM_GUI_ACTIVE", 0.0)
summary = {"note": note,
           "conv_mfma_family": {"dispatches": n_fam, "hbm_read_bytes_corrected": fam["rd"], "hbm_write_bytes": fam["wr"],
                                "hbm_bytes_per_launch": (fam["rd"] + fam["wr"]) / max(n_fam, 1),
                                # SQ_VALU_MFMA_BUSY_CYCLES sums over 4 SIMDs x 256 CUs; GRBM_GUI_ACTIVE over 8 XCDs
                                "mfma_busy_cycles_per_simd": fam["mfma"] / 1024.0, "gpu_active_cycles_per_xcd": fam["gui"] / 8.0,
                                "mfma_util": (fam["mfma"] / 1024.0) / max(fam["gui"] / 8.0, 1.0)},
           "per_kernel": per_kernel}
json.dump(summary, open(out_path, "w"), indent=1)
print(json.dumps(summary["conv_mfma_family"]))
