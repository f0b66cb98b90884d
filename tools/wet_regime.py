#!/usr/bin/env python3
"""Wet-regime timing: the tick with standing water, phase by phase, next to the reference's own
CPU loop on the same box. usage: wet_regime.py [--size 256] [--soil default.soil] [--warm 300] [--ticks 50]
[--engine spec|serial|batched|relaxed] [--nwater N] [--cpu 1]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from soilmachine_amd import capi
from soilmachine_amd.machine import SoilMachine
from soilmachine_amd.soilfile import loadsoil

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256); ap.add_argument("--soil", default="default.soil")
ap.add_argument("--warm", type=int, default=300); ap.add_argument("--ticks", type=int, default=50)
ap.add_argument("--engine", default="spec"); ap.add_argument("--nwater", type=int, default=None)
ap.add_argument("--nwind", type=int, default=0); ap.add_argument("--cpu", type=int, default=1)
ap.add_argument("--chunk", type=int, default=50)
a = ap.parse_args()
cfg = loadsoil(os.path.join(ROOT, "soilmachine_amd", "soils", a.soil))
eng = {"spec": capi.ENGINE_SPECULATIVE, "serial": capi.ENGINE_SERIAL, "batched": getattr(capi, "ENGINE_BATCHED", 2), "relaxed": capi.ENGINE_RELAXED}[a.engine]
nw = cfg.NWATER if a.nwater is None else a.nwater
sm = SoilMachine(cfg, a.size, seed=0, nwater=nw, nwind=a.nwind, dowind=a.nwind > 0, engine=eng,
                 pool=max(10_000_000, 4 * a.size * a.size))
m = sm.map
done = 0
rows = []
while done < a.warm + a.ticks:
    k = min(a.chunk, a.warm + a.ticks - done)
    if done < a.warm: k = min(k, a.warm - done)
    c0 = m.counters(); m.timing_reset(); t0 = time.perf_counter()
    sm.tick(k, sync=True)
    dt = time.perf_counter() - t0; tm = m.timing(); c1 = m.counters()
    done += k
    row = dict(ticks_done=done, ms_tick=1e3 * dt / k, ms_water=tm["ms_water"] / k, ms_grid=tm["ms_grid"] / k, ms_wind=tm["ms_wind"] / k,
               nested=(c1["nested_particles"] - c0["nested_particles"]) / k, active=(c1["grid_active_cells"] - c0["grid_active_cells"]) / k,
               steps_all=(c1["steps_water_all"] - c0["steps_water_all"]) / k, timed=done > a.warm,
               floods=(c1["floods"] - c0["floods"]) / k, flood_nested_steps=(c1["flood_nested_steps"] - c0["flood_nested_steps"]) / k,
               grid_nested_steps=(c1["grid_nested_steps"] - c0["grid_nested_steps"]) / k,
               per_tick={kk: (vv / k) for kk, vv in tm.items() if kk.startswith(("ms_kernel", "launches_kernel", "epochs_kernel", "launches_floods"))})
    rows.append(row); print(json.dumps(row), flush=True)
out = {"gpu": rows, "size": a.size, "soil": a.soil, "engine": a.engine, "nwater": nw, "digest": m.digest()}
if a.cpu:
    from oracle_lib import run_ref
    t0 = time.time()
    j = run_ref(os.path.join(ROOT, "soilmachine_amd", "soils", a.soil), seed=0, size=a.size, ticks=a.warm + a.ticks, nwater=nw,
                nwind=a.nwind, wind=a.nwind > 0, lean=True, pool=max(10_000_000, 4 * a.size * a.size))
    T = a.warm + a.ticks
    out["cpu_lean_whole_run"] = dict(ms_tick=1e3 * (j["t_water"] + j["t_grid"] + j["t_wind"] + j["t_freq"]) / T, ms_water=1e3 * j["t_water"] / T,
                                     ms_grid=1e3 * j["t_grid"] / T, wall_s=time.time() - t0, nsec=j["nsec"], typehash=j["typehash"])
    if a.warm > 0:                                            # the same WINDOW the GPU line times: (run to warm+ticks) - (run to warm)
        j0 = run_ref(os.path.join(ROOT, "soilmachine_amd", "soils", a.soil), seed=0, size=a.size, ticks=a.warm, nwater=nw,
                     nwind=a.nwind, wind=a.nwind > 0, lean=True, pool=max(10_000_000, 4 * a.size * a.size))
        d = lambda k: 1e3 * (j[k] - j0[k]) / a.ticks
        out["cpu_lean_timed_window"] = dict(ms_tick=d("t_water") + d("t_grid") + d("t_wind") + d("t_freq"), ms_water=d("t_water"), ms_grid=d("t_grid"),
                                            ticks=[a.warm, a.warm + a.ticks])
print("RESULT " + json.dumps(out))
