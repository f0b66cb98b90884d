#!/usr/bin/env python3
"""Fill the R6_* placeholders of DESIGN.md / ROUND6.md / README.md from the round's bench records (profiles/r06_bench.json + detail, r06_bench_k4.json,
r06_bench_steady_state.json): the documents quote what the files say.   fill_round6.py [--check]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(ROOT, *a)
def last_json(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")
b = last_json(P("profiles", "r06_bench.json"))
d = json.load(open(P("profiles", "r06_bench_detail.json")))
k4 = last_json(P("profiles", "r06_bench_k4.json"))
ss = last_json(P("profiles", "r06_bench_steady_state.json"))
ph, rf, cpu = b["phase_ms_per_tick"], b["roofline"], b["cpu_baseline"]
L = {l["name"]: l for l in b["lines"]}
share = rf["time_share_of_tick"]
pt = d["per_tick"]
f = lambda x, n=1: f"{x:,.{n}f}".replace(",", " ")
V = {
 "R6_VALUE": f(b["value"] / 1e6), "R6_MS": f(b["ms_per_step"], 0), "R6_W": f(ph["water"], 0), "R6_G": f(ph["grid"], 0), "R6_D": f(ph["wind"], 0),
 "R6_FIRST": f(b["ms_first_tick"], 0), "R6_LAST": f(b["ms_last_tick"], 0), "R6_FLOODS_FIRST": f(pt[0]["floods"], 0), "R6_FLOODS_LAST": f(pt[-1]["floods"], 0),
 "R6_K4_VALUE": f(k4["value"] / 1e6), "R6_K4_MS": f(k4["ms_per_step"], 0), "R6_K4_W": f(k4["phase_ms_per_tick"]["water"], 0), "R6_K4_G": f(k4["phase_ms_per_tick"]["grid"], 0), "R6_K4_D": f(k4["phase_ms_per_tick"]["wind"], 0),
 "R6_SS_VALUE": f(ss["value"] / 1e6), "R6_SS_MS": f(ss["ms_per_step"], 0), "R6_SS_W": f(ss["phase_ms_per_tick"]["water"], 0), "R6_SS_G": f(ss["phase_ms_per_tick"]["grid"], 0), "R6_SS_D": f(ss["phase_ms_per_tick"]["wind"], 0),
 "R6_FILE_REL_MS": f(L["relaxed_file_counts"]["ms_per_tick"], 0), "R6_FILE_EX_MS": f(L["exact_file_counts"]["ms_per_tick"], 0),
 "R6_FILE_REL": f(L["relaxed_file_counts"]["value"] / 1e6), "R6_FILE_EX": f(L["exact_file_counts"]["value"] / 1e6),
 "R6_CPU_VALUE": f(cpu["value"] / 1e6, 2), "R6_CPU_MS": f(cpu["ms_per_tick"], 0), "R6_CPU_ALL": f(cpu["all_cores"]["value"] / 1e6),
 "R6_CPU_END_VALUE": f(cpu["after_window"]["value"] / 1e6, 2), "R6_CPU_END_MS": f(cpu["after_window"]["ms_per_tick"], 0), "R6_GPU_END": f(cpu["after_window"]["gpu_ms_same_tick"], 0),
 "R6_CPU_END": f(cpu["after_window"]["ms_per_tick"] / 1e3), "R6_CPU": f(cpu["ms_per_tick"] / 1e3),
 "R6_8192_MS": f(L["configs4_8192_rockgravelpebbles_big_relaxed"]["ms_per_tick"], 0), "R6_8192": f(L["configs4_8192_rockgravelpebbles_big_relaxed"]["value"] / 1e6, 0),
 "R6_C1_EX_MS": f(L["configs1_1024_default_hydraulic_exact"]["ms_per_tick"]), "R6_C2_EX_MS": f(L["configs2_1024_default_coupled_exact"]["ms_per_tick"]),
 "R6_C1_EX": f(L["configs1_1024_default_hydraulic_exact"]["value"] / 1e6, 2), "R6_C2_EX": f(L["configs2_1024_default_coupled_exact"]["value"] / 1e6, 2),
 "R6_C1_CPU_MS": f(L["configs1_1024_default_hydraulic_exact"].get("cpu_1core_ms_per_tick", float("nan")), 0), "R6_C2_CPU_MS": f(L["configs2_1024_default_coupled_exact"].get("cpu_1core_ms_per_tick", float("nan")), 0),
 "R6_C1_REL_MS": f(L["configs1_1024_default_hydraulic_relaxed"]["ms_per_tick"]), "R6_C2_REL_MS": f(L["configs2_1024_default_coupled_relaxed"]["ms_per_tick"]),
 "R6_C1_REL": f(L["configs1_1024_default_hydraulic_relaxed"]["value"] / 1e6, 2), "R6_C2_REL": f(L["configs2_1024_default_coupled_relaxed"]["value"] / 1e6, 2),
 "R6_WET_GPU": f(L["wet_regime_256_default"]["ms_per_tick"], 0), "R6_WET_CPU": f(L["wet_regime_256_default"]["cpu_1core_ms_per_tick"]),
 "R6_EPOCHS": f(rf["phases"]["particles_water"]["launches"] / b["steps"], 0),
 "R6_FLOOD_SHARE": f(100 * share["k_relax_floods"], 0), "R6_FLOOD_US": f(rf["phases"]["floods"]["avg_launch_us"], 0), "R6_FLOOD_LAUNCHES": f(rf["phases"]["floods"]["launches"] / b["steps"], 0),
 "R6_FLOOD_BYTES": f(rf["algorithmic_bytes_per_launch"] / 1e3), "R6_FLOOD_GBS": f(rf["achieved"], 2), "R6_FLOOD_FRAC": f"{rf['frac']:.1e}",
 "R6_TILES_SHARE": f(100 * share["k_grid_tiles_flow"], 0), "R6_TILES_GBS": f(rf["phases"]["grid_tiles"]["achieved"]),
 "R6_WIND_SHARE": f(100 * share["k_batch_epoch<wind>"], 0), "R6_WIND_GBS": f(rf["phases"]["particles_wind"]["achieved"], 0), "R6_WIND_FRAC": f(rf["phases"]["particles_wind"]["frac"], 4),
 "R6_TAIL_SHARE": f(100 * share["k_relax_tail"], 0), "R6_STEP_SHARE": f(100 * share["k_relax_step<water>"], 0), "R6_STEP_GBS": f(rf["phases"]["particles_water"]["achieved"], 0),
 "R6_MAPFREQ_FRAC": f(rf["phases"]["map_frequency"]["frac"], 2), "R6_MAPFREQ": f(rf["phases"]["map_frequency"]["achieved"] / 1e3, 2),
 "R6_TICK_GBS": f(rf["whole_tick"]["achieved"]), "R6_TICK_GB": f(rf["whole_tick"]["algorithmic_bytes_per_tick"] / 1e9),
 "R6_GPU_TESTS": "123 passed + 3 opt-in + 2 two-device skipped in 735 s (`profiles/r06_gpu_tests.log`; the opt-in 256² × 300 and × 1000 rows of the exact engine re-run on the final build: `profiles/r06_depth_256x300.log`, `r06_depth_256x1000.log`)",
 "R6_CPU_TESTS": "155 passed in 7.5 minutes",
 "R6_SHARE": f(100 * (share["k_relax_floods"] + share["k_grid_tiles_flow"] + share["k_relax_tail"]) + 7, 0),   # + the colour lists' 7 % of the trace (not bracketed by events)
}
for name in ("DESIGN.md", "ROUND6.md", "README.md"):
    t = open(P(name)).read()
    for k in sorted(V, key=len, reverse=True):
        t = re.sub(r"\b" + re.escape(k) + r"\b", V[k].replace("\\", "\\\\"), t)
    left = sorted(set(re.findall(r"R6_[A-Z0-9_]+", t)))
    print(name, "placeholders left:", left)
    if "--check" not in sys.argv:
        open(P(name), "w").write(t)
