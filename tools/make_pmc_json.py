#!/usr/bin/env python3
"""Condense separate rocprofv3 --pmc passes (tools/gpu_run.sh pmc: FETCH_SIZE, WRITE_SIZE, an SQ set) into the per-kernel JSON
bench.py reads as `roofline.traffic` (profiles/rNN_pmc.json):   make_pmc_json.py <run-dir> <workload-string> "<command>" """
import collections, csv, glob, json, os, sys

out, workload, command = sys.argv[1], sys.argv[2], sys.argv[3]


def short(name):
    n = name.replace("void ", "").split("(")[0]
    return n.replace("<true>", "<wind>").replace("<false>", "<water>")


def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for path in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in n.items()}


fa, fn = counters("pmc_FETCH_SIZE")
wa, wn = counters("pmc_WRITE_SIZE")
sa, sn = counters("pmc_SQ")
pmc = {"workload": workload, "command": command,
       "unit_note": "FETCH_SIZE / WRITE_SIZE as reported by rocprofv3 on gfx950 (KiB) x 1024, separate passes. MI355X_MICROARCH.md (HBM): a wide "
                    "coalesced stream reads 2x the reported FETCH_SIZE; narrow scattered accesses (this path: 32-byte records) and WRITE_SIZE are "
                    "uncalibrated -- the raw value is kept, so hbm_bytes_per_launch is a LOWER bound by at most 2x on the read side. "
                    "Infinity-Cache hits are counted (the whole working set of a phase fits the 256 MiB cache only for the particle records).",
       "kernels": {}}
for k in sorted(fa, key=lambda k: -fa[k]["FETCH_SIZE"]):
    f = fa[k]["FETCH_SIZE"] / max(fn[k], 1); wv = wa.get(k, {}).get("WRITE_SIZE", 0.0) / max(wn.get(k, 1), 1)
    e = {"launches": fn[k], "fetch_kib_per_launch": f, "write_kib_per_launch": wv, "hbm_bytes_per_launch": (f + wv) * 1024}
    if k in sa:
        v = sa[k]; wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
        e["sq"] = {"launches": sn[k], "valu_per_launch": v.get("SQ_INSTS_VALU", 0) / sn[k], "salu_per_launch": v.get("SQ_INSTS_SALU", 0) / sn[k],
                   "vmem_per_launch": v.get("SQ_INSTS_VMEM", 0) / sn[k], "waves_per_launch": v.get("SQ_WAVES", 0) / sn[k],
                   "wait_any_frac": v.get("SQ_WAIT_ANY", 0) / wc, "active_inst_frac": v.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                   "wait_inst_frac": v.get("SQ_WAIT_INST_ANY", 0) / wc}
    pmc["kernels"][k] = e
json.dump(pmc, open(os.path.join(out, "pmc.json"), "w"), indent=1)
for k, e in list(pmc["kernels"].items())[:12]:
    print(f"{k}: launches {e['launches']} fetch {e['fetch_kib_per_launch']:.1f} KiB write {e['write_kib_per_launch']:.1f} KiB per launch")
