#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/gpu_r02_profiles.sh into small files (run on the GPU box, results merged back)."""
import collections, csv, glob, json, os, sys

out = sys.argv[1]

def find(d, pat):
    g = glob.glob(os.path.join(out, d, "**", pat), recursive=True)
    return g[0] if g else None

def short(name):
    n = name.replace("void ", "").split("(")[0]
    return n.replace("<true>", "<wind>").replace("<false>", "<water>")

def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    if not path: return agg, {}
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in n.items()}

stats = find("trace", "*kernel_stats.csv")
rows = list(csv.DictReader(open(stats))) if stats else []
with open(os.path.join(out, "kernel_stats.csv"), "w") as f:
    w = csv.writer(f); w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent", "min_us", "max_us"])
    for r in rows[:24]:
        w.writerow([short(r["Name"]), r["Calls"], f"{int(r['TotalDurationNs'])/1e6:.3f}", f"{float(r['AverageNs'])/1e3:.2f}", r["Percentage"],
                    f"{float(r['MinNs'])/1e3:.2f}", f"{float(r['MaxNs'])/1e3:.2f}"])
fa, fn = counters(find("fetch", "*counter_collection.csv"))
wa, wn = counters(find("write", "*counter_collection.csv"))
sa, sn = counters(find("sq", "*counter_collection.csv"))
pmc = {"workload": "4096/64000/16000/batched", "command": "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lines",
       "unit_note": "FETCH_SIZE / WRITE_SIZE as reported by rocprofv3 on gfx950 (KiB), x 1024; separate passes. MI355X_MICROARCH.md (HBM): wide "
                    "coalesced streams read 2x the reported FETCH_SIZE; narrow scattered accesses (this path: 32-byte records) are uncalibrated, "
                    "the raw value is kept. Infinity-Cache hits are counted.",
       "kernels": {}}
for k in sorted(fa, key=lambda k: -fa[k]["FETCH_SIZE"]):
    f = fa[k]["FETCH_SIZE"] / max(fn[k], 1); wv = wa.get(k, {}).get("WRITE_SIZE", 0.0) / max(wn.get(k, 1), 1)
    e = {"launches": fn[k], "fetch_kib_per_launch": f, "write_kib_per_launch": wv, "hbm_bytes_per_launch": (f + wv) * 1024}
    if k in sa:
        v = sa[k]; wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
        e["sq"] = {"launches": sn[k], "valu_per_launch": v.get("SQ_INSTS_VALU", 0) / sn[k], "salu_per_launch": v.get("SQ_INSTS_SALU", 0) / sn[k],
                   "vmem_rd_per_launch": v.get("SQ_INSTS_VMEM_RD", 0) / sn[k], "vmem_wr_per_launch": v.get("SQ_INSTS_VMEM_WR", 0) / sn[k],
                   "wait_any_frac": v.get("SQ_WAIT_ANY", 0) / wc, "active_inst_frac": v.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                   "wait_inst_frac": v.get("SQ_WAIT_INST_ANY", 0) / wc}
    pmc["kernels"][k] = e
json.dump(pmc, open(os.path.join(out, "pmc.json"), "w"), indent=1)
print("kernels:", list(pmc["kernels"])[:8])
