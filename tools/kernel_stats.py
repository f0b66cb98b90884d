#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats --output-format csv run: kernel_stats.py <dir> [n]"""
import csv, glob, os, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if not f: sys.exit("no kernel_stats.csv under " + d)
print("kernel,calls,total_ms,avg_us,percent,min_us,max_us")
for r in list(csv.DictReader(open(f[0])))[:n]:
    name = r["Name"].replace("void ", "").split("(")[0].replace("<true>", "<wind>").replace("<false>", "<water>")
    print(f"{name},{r['Calls']},{int(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.2f},{r['Percentage']},{float(r['MinNs'])/1e3:.2f},{float(r['MaxNs'])/1e3:.2f}")
