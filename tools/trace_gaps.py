#!/usr/bin/env python3
"""Where does the time BETWEEN kernels go? From a rocprofv3 --kernel-trace CSV (…kernel_trace.csv: one row per dispatch with start / end
timestamps): busy time, idle time, and the idle time attributed to (previous kernel -> next kernel) pairs.
  trace_gaps.py <dir-or-csv> [--skip-ms N] [--last-ms M] [--last-ticks K]      (skip everything before N ms after the first dispatch: the warm-up)"""
import csv, glob, os, sys
from collections import defaultdict

def short(n):
    return n.replace("void ", "").split("(")[0].replace("smx::", "").replace("<true>", "<wind>").replace("<false>", "<water>")

def main():
    src = sys.argv[1]
    skip_ms = float(sys.argv[sys.argv.index("--skip-ms") + 1]) if "--skip-ms" in sys.argv else 0.0
    if os.path.isdir(src):
        f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
        if not f: sys.exit("no kernel_trace.csv under " + src)
        src = f[0]
    rows = []
    for r in csv.DictReader(open(src)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    t0 = rows[0][0] + int(skip_ms * 1e6)
    rows = [r for r in rows if r[0] >= t0]
    if "--last-ticks" in sys.argv:                            # whole ticks: a tick ends with its k_map_frequency launch
        k = int(sys.argv[sys.argv.index("--last-ticks") + 1])
        ends = [r[1] for r in rows if r[2].startswith("k_map_frequency")]
        if len(ends) > k:
            rows = [r for r in rows if ends[-k - 1] < r[0] and r[1] <= ends[-1]]
    if "--last-ms" in sys.argv:
        t1 = rows[-1][1] - int(float(sys.argv[sys.argv.index("--last-ms") + 1]) * 1e6)
        rows = [r for r in rows if r[0] >= t1]
    busy = 0; idle = 0; end = rows[0][0]
    pair = defaultdict(lambda: [0, 0]); kern = defaultdict(lambda: [0, 0])
    hist = defaultdict(int)
    prev = None
    for s, e, n in rows:
        kern[n][0] += 1; kern[n][1] += e - s
        if s > end:
            g = s - end
            idle += g
            if prev is not None:
                pair[(prev, n)][0] += 1; pair[(prev, n)][1] += g
            b = 0
            while (1 << b) * 1000 < g: b += 1
            hist[b] += 1
        busy += max(0, e - max(s, end))
        if e > end: end = e; prev = n
    span = end - rows[0][0]
    print(f"dispatches {len(rows)}  span {span / 1e6:.1f} ms  busy {busy / 1e6:.1f} ms ({100.0 * busy / span:.1f} %)  idle {idle / 1e6:.1f} ms ({100.0 * idle / span:.1f} %)")
    print("gap histogram (gaps > 0):  " + "  ".join(f"<{1 << b}us:{c}" for b, c in sorted(hist.items())))
    print("idle by (previous kernel -> next kernel), top 25:")
    for (a, b), (c, t) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {t / 1e6:9.2f} ms  {c:7d} gaps  avg {t / c / 1e3:8.2f} us   {a} -> {b}")
    print("kernels by time:")
    for n, (c, t) in sorted(kern.items(), key=lambda kv: -kv[1][1])[:20]:
        print(f"  {t / 1e6:9.2f} ms  {c:7d} calls  avg {t / c / 1e3:8.2f} us   {n}")

if __name__ == "__main__":
    main()
