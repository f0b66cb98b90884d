#!/usr/bin/env python3
"""Refresh sections 1-4 of profiles/r02_summary.md from profiles/r02_bench.json, r02_kernel_stats.csv and r02_pmc.json
(copied from gpurun_out/r02p after tools/gpu_r02_profiles.sh); the other sections are prose."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
b = json.load(open(os.path.join(P, "r02_bench.json"))); pmc = json.load(open(os.path.join(P, "r02_pmc.json")))
ks = open(os.path.join(P, "r02_kernel_stats.csv")).read().strip()
s = open(os.path.join(P, "r02_summary.md")).read()
ph, st, cpu, rf = b["phase_ms_per_tick"], b["steps_per_tick"], b["cpu_baseline"], b["roofline"]
rows = ""
for l in b["lines"]:
    q = l["phase_ms_per_tick"]; rp = l.get("roofline_particles") or {}
    fr = " / ".join(f"{rp[k]['frac']:.1e}" if rp.get(k) else "-" for k in ("water", "wind"))
    rows += f"| {l['name']} | {l['engine']} | {l['value']/1e6:.2f} | {l['ms_per_tick']:.1f} | {q['water']:.1f} | {q['grid']:.2f} | {q['wind']:.1f} | {l['steps_per_tick']['nested_particles']:.1f} | {fr} |\n"
ac = cpu.get("all_cores") or {}
lb = b.get("lbm_wind") or {}
sec1 = f"""## 1. Default bench line (`python bench.py`, {b['steps']} timed ticks after {b['warmup']} warm-up ticks) — `r02_bench.json`

Headline: batched engine, 4096² `rockgravelpebblessand.soil`, SEED 0, area-scaled counts 64 000 water + 16 000 wind particles per tick: **{b['value']/1e6:.1f} M particle-steps/s, {b['ms_per_step']:.0f} ms/tick** (water {ph['water']:.0f} + grid {ph['grid']:.0f} + wind {ph['wind']:.0f} ms; {st['water']/1e6:.1f} M water + {st['wind']/1e6:.1f} M wind steps, {st['nested_particles']:.0f} nested particles and {st['floods']:.0f} floods per tick; nothing dropped: `work_dropped` = {json.dumps(b['work_dropped'])}). State digest after the timed region: `{json.dumps(b['digest'])}`.

| line | engine | M steps/s | ms/tick | water | grid | wind | nested/tick | particle kernels: fraction of 8 TB/s (water / wind) |
|---|---|---:|---:|---:|---:|---:|---:|---|
{rows}
The last three lines keep the headline's particle density on larger maps: a phase is a latency chain of about constant length, so the rate grows with the area (`r02_batched_schedules.md`, area scaling).

CPU (reference's own loop, `oracle/_ref/soil_ref_lean`): {cpu['sample']}: **{cpu['value']/1e6:.2f} M top-level steps/s**, {cpu['ms_per_tick']:.0f} ms/tick (water {cpu['ms_water']:.0f}, grid pass {cpu['ms_grid']:.0f}, wind {cpu['ms_wind']:.0f} ms). GPU `value_top_level` = {b['value_top_level']/1e6:.1f} M/s.
All cores: {ac.get('cores')} independent `soil_ref_lean` processes, one pinned core each → {ac.get('value', 0)/1e6:.1f} M steps/s in sum (the reference has no threads).

LBM wind (`lbm_wind` object of the same line): {lb.get('lattice')} lattice, {lb.get('us_per_step', 0):.0f} µs per step = {lb.get('cell_updates_per_s', 0)/1e9:.1f} G cell updates/s = {lb.get('roofline', {}).get('achieved', 0)/1e3:.2f} TB/s = **{lb.get('roofline', {}).get('frac', 0):.2f} of the HBM peak** (`r02_lbm_profile.md`: measured traffic = algorithmic bytes).

"""
a, e = s.index("## 1. Default bench line"), s.index("## 2. Roofline object")
s = s[:a] + sec1 + s[e:]
pw = rf["phases"]
sec2 = f"""## 2. Roofline object of the same line

Dominant kernel `k_batch_epoch<water>`: {rf['launches']} launches, {1e3*rf['avg_launch_ms']:.1f} µs average (HIP events around nothing but these launches), {rf['algorithmic_bytes_per_launch']/1e3:.0f} KB algorithmic bytes per launch (192 B × the water steps of the launch) → {rf['achieved']:.2f} GB/s = **{rf['frac']:.1e} of 8 TB/s**. Wind kernel: {pw['particles_wind']['achieved']:.1f} GB/s ({pw['particles_wind']['frac']:.1e}). Streaming kernels: `k_map_frequency` {pw['map_frequency']['achieved']/1e3:.2f} TB/s ({pw['map_frequency']['frac']:.2f} of peak), `k_grid_classify8` {pw['grid_classify']['achieved']:.0f} GB/s ({pw['grid_classify']['avg_launch_us']:.0f} µs for 18.9 MB; the scalar kernel of round 1: 74 µs).

"""
a, e = s.index("## 2. Roofline object"), s.index("## 3. rocprofv3")
s = s[:a] + sec2 + s[e:]
a = s.index("```\nkernel,calls"); e = s.index("```", a + 4)
s = s[:a] + "```\n" + ks + "\n" + s[e:]
wk = [l for l in ks.splitlines() if l.startswith("k_batch_epoch<water>")][0].split(",")
s = re.sub(r"The HIP-event average of section 2 \(.*?spread\.", f"The HIP-event average of section 2 ({1e3*rf['avg_launch_ms']:.1f} µs over the {b['steps']+b['warmup']} ticks of the default run, which include the longer first ticks) and rocprof's ({float(wk[3]):.1f} µs over 3 ticks) agree to within the tick-to-tick spread.", s, flags=re.S)
rowsp = ""
for k, v in pmc["kernels"].items():
    sq = v.get("sq", {})
    rowsp += f"| `{k}` | {v['launches']} | {v['fetch_kib_per_launch']:.0f} | {v['write_kib_per_launch']:.0f} | {v['hbm_bytes_per_launch']:.3e} | {sq.get('valu_per_launch',0):.3e} | {sq.get('vmem_rd_per_launch',0):.0f} | {100*sq.get('wait_any_frac',0):.0f} | {100*sq.get('active_inst_frac',0):.0f} |\n"
a, e = s.index("| kernel | launches | FETCH KiB/launch"), s.index("FETCH_SIZE / WRITE_SIZE as reported")
hdr = "| kernel | launches | FETCH KiB/launch | WRITE KiB/launch | bytes/launch (raw) | VALU/launch | VMEM rd/launch | wait % | active % |\n|---|---:|---:|---:|---:|---:|---:|---:|---:|\n"
s = s[:a] + hdr + rowsp + "\n" + s[e:]
w = pmc["kernels"]["k_batch_epoch<water>"]
s = re.sub(r"Reading: the water epoch kernel moves .*? algorithmic \(\d+×\)\.", f"Reading: the water epoch kernel moves {w['hbm_bytes_per_launch']/1e6:.1f} MB per launch against {rf['algorithmic_bytes_per_launch']/1e6:.2f} MB algorithmic ({w['hbm_bytes_per_launch']/rf['algorithmic_bytes_per_launch']:.0f}×).", s, flags=re.S)
open(os.path.join(P, "r02_summary.md"), "w").write(s)
print("profiles/r02_summary.md refreshed")
