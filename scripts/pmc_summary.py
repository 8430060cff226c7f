"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name (sum, dispatches)."""
import csv, glob, sys, collections, json
d = sys.argv[1]
out = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
res = {k: dict(dispatches=len(cnt[k]), **{c: v for c, v in out[k].items()}) for k in out}
json.dump(res, open(sys.argv[2], "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["dispatches"])[:12]:
    print(k, v)
