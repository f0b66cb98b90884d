#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of one round (gpurun_out/...) into the files we commit under profiles/."""
import collections, csv, json, sys

def kernel_stats(path):
    rows = list(csv.DictReader(open(path)))
    out = ["| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for r in rows[:16]:
        out.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    return "\n".join(out), rows

def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in n.items()}

if __name__ == "__main__":
    stats_csv, fetch_csv, write_csv, sq_csv, bench_json, out_md, out_json = sys.argv[1:8]
    table, rows = kernel_stats(stats_csv)
    fa, fn = counters(fetch_csv)
    wa, wn = counters(write_csv)
    sa, sn = counters(sq_csv)
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    names = {"k_spec_exec<true>": "k_spec_exec<wind>", "k_spec_exec<false>": "k_spec_exec<water>",
             "k_spec_exec_coop<true>": "k_spec_exec_coop<wind>", "k_spec_exec_coop<false>": "k_spec_exec_coop<water>",
             "k_spec_scout_coop<true>": "k_spec_scout_coop<wind>", "k_spec_scout_coop<false>": "k_spec_scout_coop<water>"}
    pmc = {"workload_size": 4096, "unit_note": "FETCH_SIZE/WRITE_SIZE are KiB as reported by rocprofv3 on gfx950; narrow scattered accesses are "
           "uncalibrated (MI355X_MICROARCH.md §HBM says wide streams read 2x the reported FETCH_SIZE) -- bytes below use the raw value x 1024",
           "kernels": {}}
    lines = ["| kernel | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM bytes/launch (raw) |", "|---|---:|---:|---:|---:|"]
    for k in sorted(fa, key=lambda k: -fa[k]["FETCH_SIZE"]):
        f = fa[k]["FETCH_SIZE"] / max(fn[k], 1)
        w = wa.get(k, {}).get("WRITE_SIZE", 0.0) / max(wn.get(k, 1), 1)
        kk = names.get(k, k)
        pmc["kernels"][kk] = {"launches": fn[k], "fetch_kib_per_launch": f, "write_kib_per_launch": w, "hbm_bytes_per_launch": (f + w) * 1024}
        lines.append(f"| `{kk}` | {fn[k]} | {f:.1f} | {w:.1f} | {(f + w) * 1024:.3e} |")
    sq = ["| kernel | launches | VALU | SALU | VMEM rd | VMEM wr | LDS | wave quad-cycles | wait % | active % |", "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for k, v in sa.items():
        if "spec_exec" in k or "scout" in k or "serial" in k or "boundary" in k:
            wc = max(v["SQ_WAVE_CYCLES"], 1)
            sq.append(f"| `{names.get(k, k)}` | {sn[k]} | {v['SQ_INSTS_VALU']:.3e} | {v['SQ_INSTS_SALU']:.3e} | {v['SQ_INSTS_VMEM_RD']:.3e} | {v['SQ_INSTS_VMEM_WR']:.3e} | "
                      f"{v['SQ_INSTS_LDS']:.3e} | {wc:.3e} | {100 * v['SQ_WAIT_ANY'] / wc:.0f} | {100 * v['SQ_ACTIVE_INST_ANY'] / wc:.0f} |")
    json.dump(pmc, open(out_json, "w"), indent=1)
    with open(out_md, "w") as f:
        f.write("# Round 1 — measurements on MI355X (1 GPU)\n\n")
        f.write("Workload: BASELINE.json configs[3] — 4096² layermap, `rockgravelpebblessand.soil`, SEED 0, 1000 water + 250 wind particles per tick, "
                "speculative exact engine; `python bench.py` (5 timed ticks after 2 warm-up ticks).\n\n")
        f.write("## bench.py line\n\n```json\n" + json.dumps(bench, indent=1) + "\n```\n\n")
        cb = bench.get("cpu_baseline") or {}
        if cb:
            f.write(f"GPU tick {bench['ms_per_tick']:.1f} ms vs reference CPU tick {cb['ms_per_tick']:.1f} ms on one core of the box's {cb.get('host_cpu','')} "
                    f"(water {cb['ms_water']:.0f} + grid pass {cb['ms_grid']:.0f} + wind {cb['ms_wind']:.0f} ms): "
                    f"{cb['ms_per_tick'] / bench['ms_per_tick']:.1f}x per tick, bit-identical state.\n\n")
        f.write("## rocprofv3 --kernel-trace --stats (same command, `--no-cpu-baseline`)\n\n" + table + "\n\n")
        f.write("## HBM traffic (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes, 3 timed + 2 warm-up ticks)\n\n" + "\n".join(lines) + "\n\n" + pmc["unit_note"] + "\n\n")
        f.write("## SQ counters (`--pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY`)\n\n" + "\n".join(sq) + "\n")
