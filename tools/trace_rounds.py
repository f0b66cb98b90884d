#!/usr/bin/env python3
"""Per-launch view of a rocprofv3 --kernel-trace CSV: the sequence of speculative rounds of the last tick(s).
usage: trace_rounds.py <kernel_trace.csv> [ntail]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ntail = int(sys.argv[2]) if len(sys.argv) > 2 else 400
t0 = None
for r in rows[-ntail:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    print(f"{(s - t0) / 1e6:10.3f} ms  +{(e - s) / 1e3:10.1f} us  {name}  grid={r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
