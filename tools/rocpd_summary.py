#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) result into the per-kernel stats table we commit under profiles/."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print("| kernel | calls | total (us) | avg (us) | % |")
print("|---|---:|---:|---:|---:|")
for n, c, t, a, p in rows:
    print(f"| `{n}` | {c} | {t:.1f} | {a:.1f} | {p:.4f} |")
