#!/usr/bin/env python3
"""Sum the counters of a rocprofv3 --pmc --output-format csv run per kernel: pmc_by_kernel.py <dir> [name-filter]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?").split("(")[0][:70]
        if flt and flt not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen: seen.add(key); n[k] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", sum(acc[k].values()))):
    print(f"{k}  dispatches={n[k]}  " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(acc[k].items())))
