#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 kernel_stats.csv: python3 scripts/kstats_top.py <dir or csv> [rows]"""
import csv, glob, os, sys
src = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = src if os.path.isfile(src) else glob.glob(src + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.1f  (%d kernels)" % (tot / 1e6, len(rows)))
for r in rows[:n]:
    name = r["Name"].replace("void ", "").replace("gpemsr::", "")
    print("%-100s %6s %9.1f ms %6s%%  avg %8.1f us" % (name[:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, r["Percentage"], float(r["AverageNs"]) / 1e3))
