#!/usr/bin/env python3
"""Tier-P2 parity of the throughput engine against THE REFERENCE ITSELF, at the benchmarked workload.

oracle/_ref/soil_ref_lean -- the reference's own headers compiled headless (oracle/Makefile) -- runs on the box's host cores:
  * SEED 0: the run the device is compared with (same terrain, same rand() stream at the start of tick 1);
  * NCTL control streams: the same terrain, srand(other) once it stands (ref_driver --rand-seed) -- what the reference's own
    chaos does to every figure;
all in parallel (the reference has no threads: one process per stream), while the device runs the same ticks with the engine under
test. Every figure of the engine and of every control is taken relative to the SEED-0 reference run;
z = (engine - mean(controls)) / std(controls, ddof=1); height RMSE is one-sided. CHECKER ONLY: nothing here is on the product path.

  p2_reference.py [--size 4096] [--ticks 10] [--nctl 8] [--engine relaxed] [--soil rockgravelpebblessand.soil] [--out profiles/...json]
Counts are area-scaled (250 * (N/256)^2 water, NWIND/NWATER of the soil file).
"""
from __future__ import annotations

import argparse, json, os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

FIGS = ("steps_water_top", "steps_wind", "nested", "sections", "standing", "water_volume", "sumh", "height_rmse")


def _ref_cmd(soil, size, nwater, nwind, ticks, pool, heights_out, rand_seed=None, core=None):
    exe = os.path.join(ROOT, "oracle", "_ref", "soil_ref_lean")
    cmd = [exe, "--soil", os.path.join(ROOT, "soilmachine_amd", "soils", soil), "--seed", "0", "--size", str(size), "--ticks", str(ticks),
           "--nwater", str(nwater), "--nwind", str(nwind), "--wind", "1" if nwind > 0 else "0", "--pool", str(pool), "--heights-out", heights_out]
    if rand_seed is not None:
        cmd += ["--rand-seed", str(rand_seed)]
    if core is not None:
        cmd = ["taskset", "-c", str(core)] + cmd
    return cmd


def _parse(out):
    for line in out.splitlines():
        if line.startswith("JSON "):
            return json.loads(line[5:])
    return None


def zscores(eng, ref, ctl):
    out = {}
    for k in FIGS:
        c = np.array([x[k] for x in ctl], float)
        mu, sd = float(c.mean()), (float(c.std(ddof=1)) if len(c) > 1 else 0.0)
        e = float(eng[k])
        z = (e - mu) / sd if sd > 0 else (0.0 if e == mu else float("inf"))
        one = k == "height_rmse"
        rel = (lambda v, r=ref[k]: v / r - 1.0) if (not one and ref[k]) else (lambda v: v)
        out[k] = {"engine": e, "reference_seed0": ref[k], "control_mean": mu, "control_std": sd, "control_min": float(c.min()), "control_max": float(c.max()),
                  "z": z, "one_sided": one, "outlier": (z > 2.0) if one else (abs(z) > 2.0),
                  "engine_rel": rel(e), "control_rel_min": rel(float(c.min())), "control_rel_max": rel(float(c.max()))}
    return out


def run(soil="rockgravelpebblessand.soil", size=4096, ticks=10, nctl=8, engine="relaxed", device=0, verbose=True, nwater=None, nwind=None, water_batches=None):
    """Returns the report dict, or {"skipped": reason}."""
    from soilmachine_amd import capi
    from soilmachine_amd.machine import SoilMachine
    from soilmachine_amd.soilfile import loadsoil
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "soil_ref_lean")):
        return {"skipped": "oracle/_ref/soil_ref_lean is not built (oracle/Makefile needs the reference tree)"}
    cfg = loadsoil(os.path.join(ROOT, "soilmachine_amd", "soils", soil))
    area = (size / 256.0) ** 2
    nwater = int(250 * area) if nwater is None else nwater
    nwind = int(250 * area * cfg.NWIND / max(cfg.NWATER, 1)) if nwind is None else nwind
    pool = max(10_000_000, 3 * size * size)
    ncpu = os.cpu_count() or 1
    try:
        mem_gb = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2**30
    except (ValueError, OSError):
        mem_gb = 64.0
    per_proc_gb = (pool * 56 + size * size * 60) / 2**30 + 0.5
    fit = int(max(1, (0.7 * mem_gb) // per_proc_gb))
    nproc = min(nctl + 1, fit, max(ncpu - 1, 1))
    if nproc < 3:
        return {"skipped": f"host too small for a control distribution ({ncpu} cores, {mem_gb:.0f} GB: {nproc} streams fit)"}
    nctl = nproc - 1
    t0 = time.time()
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        procs = []
        for k in range(nproc):
            hp = os.path.join(td, f"h{k}.f64")
            procs.append((hp, subprocess.Popen(_ref_cmd(soil, size, nwater, nwind, ticks, pool, hp, None if k == 0 else 1000 + 7919 * k, core=(k + 1) % ncpu),
                                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)))
        # the device, meanwhile
        ENG = {"relaxed": capi.ENGINE_RELAXED, "batched": capi.ENGINE_BATCHED, "spec": capi.ENGINE_SPECULATIVE}
        sm = SoilMachine(cfg, size, seed=0, nwater=nwater, nwind=nwind, dowind=nwind > 0, pool=pool, device=device, engine=ENG[engine])
        tg = time.time()
        if water_batches:                                         # the tick's water particles as k consecutive generations (smx_set_water_generations; default: the engine's own 4)
            sm.map.set_water_generations(water_batches)
        sm.tick(ticks, sync=True)
        t_gpu = time.time() - tg
        m = sm.map
        c = m.counters()
        hg = m.heights()
        s = m.snapshot()
        end = np.cumsum(s.count.astype(np.int64)); nz = s.count > 0
        top = end[nz] - 1
        air = s.type[top] == 0
        eng = {"steps_water_top": c["steps_water_top"], "steps_wind": c["steps_wind"], "nested": c["nested_particles"], "sections": int(s.nsec),
               "standing": int(air.sum()), "water_volume": float(s.size[top][air].sum()), "sumh": float(hg.sum()), "floods": c["floods"],
               "steps_water_all": c["steps_water_all"]}
        del s
        m.close()
        if verbose:
            print(f"[p2_reference] device ({engine}) done in {t_gpu:.1f} s; waiting for {nproc} reference streams ...", file=sys.stderr, flush=True)
        js = []
        for hp, p in procs:
            j = _parse(p.communicate()[0])
            if j is None:
                return {"skipped": "a reference process produced no result (out of memory?)"}
            js.append(j)
        h0 = np.fromfile(procs[0][0], dtype=np.float64)
        def fig(j, hp):
            h = h0 if hp == procs[0][0] else np.fromfile(hp, dtype=np.float64)
            d = h - h0
            return {"steps_water_top": j["steps_water_top"], "steps_wind": j["steps_wind"], "nested": j["nested"], "sections": j["nsec"], "standing": j["standing"],
                    "water_volume": j["water_volume"], "sumh": float(h.sum()), "height_rmse": float(np.sqrt((d * d).mean()))}
        ref = fig(js[0], procs[0][0])
        ctl = [fig(j, hp) for j, (hp, _) in zip(js[1:], procs[1:])]
        d = hg - h0
        eng["height_rmse"] = float(np.sqrt((d * d).mean()))
        eng["height_max_abs"] = float(np.abs(d).max())
        hstd = float(h0.std())
    z = zscores(eng, ref, ctl)
    rep = {"what": __doc__.split("\n\n")[0], "workload": f"{size}x{size}, soil/{soil}, SEED 0, nwater={nwater}, nwind={nwind}, ticks 1-{ticks} from the initial terrain",
           "engine": engine, "reference": "oracle/_ref/soil_ref_lean (the reference's own headers, g++ -O2, one process per stream, pinned cores)",
           "controls": nctl, "figures": z, "outliers": [k for k in FIGS if z[k]["outlier"]], "height_std": hstd,
           "engine_extra": {k: eng[k] for k in ("floods", "steps_water_all", "height_max_abs")},
           "reference_seed0_s_per_tick": (js[0]["t_water"] + js[0]["t_grid"] + js[0]["t_wind"] + js[0]["t_freq"]) / ticks,
           "device_s": t_gpu, "wall_s": time.time() - t0, "host_cores": ncpu}
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--ticks", type=int, default=10)
    ap.add_argument("--nctl", type=int, default=8)
    ap.add_argument("--engine", default="relaxed")
    ap.add_argument("--soil", default="rockgravelpebblessand.soil")
    ap.add_argument("--out", default=None)
    ap.add_argument("--water-batches", type=int, default=int(os.environ.get("P2_WATER_BATCHES", "0")), help="0 = the engine's default (4)")
    a = ap.parse_args()
    rep = run(a.soil, a.size, a.ticks, a.nctl, a.engine, water_batches=a.water_batches)
    rep["water_generations"] = a.water_batches or 4
    if "figures" in rep:
        for k in FIGS:
            f = rep["figures"][k]
            print(f"{k:16s} engine {f['engine']:.6g}  ref0 {f['reference_seed0']:.6g}  controls {f['control_min']:.6g} .. {f['control_max']:.6g}  z = {f['z']:+.2f}{'  OUTLIER' if f['outlier'] else ''}", file=sys.stderr)
    s = json.dumps(rep, indent=1)
    if a.out:
        open(a.out, "w").write(s)
    print(s)


if __name__ == "__main__":
    main()
