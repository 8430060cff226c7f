#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes (one directory per counter set) of ONE bench.py command into the summary bench.py reads for
`roofline.traffic` (profiles/r05_<precision>_pmc_summary.json):
    python3 scripts/pmc_family2.py <dir with pass_*/> <out.json> <precision> <forward passes in the profiled command> "<note>"
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md, HBM section: 128-B requests
tallied at 64 B).  Infinity-cache hits are counted as traffic by these counters, so the figures are an upper bound of DRAM bytes."""
import collections, csv, glob, json, sys

root, out_path, precision, passes, note = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
FAMILY = {"fp32": ("conv_mfma_kernel", "conv_wino", "conv7_wino"),
          "bf16": ("conv_bf16_kernel", "conv64_resident", "convt64_resident", "conv7_c32_cout16", "conv7_c8_cout32", "vgg_mask", "flash_attn512", "gemm_direct_bf16", "dcn_fused16")}[precision]
per = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gpemsr::", "")
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
fam = collections.defaultdict(float)
n_fam = 0
tot_rd = tot_wr = 0.0
per_kernel = {}
for k, c in per.items():
    n = max(len(disp[(k, name)]) for name in c)
    rd = 2.0 * 1024.0 * c.get("FETCH_SIZE", 0.0)
    wr = 1024.0 * c.get("WRITE_SIZE", 0.0)
    tot_rd += rd; tot_wr += wr
    per_kernel[k[:100]] = {"dispatches_per_step": n / passes, "hbm_read_GB_per_step": rd / passes / 1e9, "hbm_write_GB_per_step": wr / passes / 1e9,
                           "mfma_busy_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), "gui_active": c.get("GRBM_GUI_ACTIVE", 0.0)}
    if any(f in k for f in FAMILY):
        n_fam += n
        fam["rd"] += rd; fam["wr"] += wr; fam["mfma"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); fam["gui"] += c.get("GRBM_GUI_ACTIVE", 0.0)
per_kernel = dict(sorted(per_kernel.items(), key=lambda kv: -(kv[1]["hbm_read_GB_per_step"] + kv[1]["hbm_write_GB_per_step"])))
summary = {
    "note": note, "precision": precision, "forward_passes_profiled": passes,
    "hbm_bytes_per_step_all_kernels": (tot_rd + tot_wr) / passes,
    "hbm_read_bytes_per_step_all_kernels": tot_rd / passes, "hbm_write_bytes_per_step_all_kernels": tot_wr / passes,
    "dominant_family": {"kernels": list(FAMILY), "launches_per_step": round(n_fam / passes),
                        "hbm_bytes_per_launch": (fam["rd"] + fam["wr"]) / max(n_fam, 1),
                        "hbm_bytes_per_step": (fam["rd"] + fam["wr"]) / passes,
                        # SQ_VALU_MFMA_BUSY_CYCLES sums over 4 SIMDs x 256 CUs; GRBM_GUI_ACTIVE over 8 XCDs
                        "mfma_busy_cycles_per_simd": fam["mfma"] / 1024.0, "gpu_active_cycles_per_xcd": fam["gui"] / 8.0,
                        "mfma_util": (fam["mfma"] / 1024.0) / max(fam["gui"] / 8.0, 1.0)},
    "per_kernel": per_kernel}
json.dump(summary, open(out_path, "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("hbm_bytes_per_step_all_kernels", "dominant_family")}))
