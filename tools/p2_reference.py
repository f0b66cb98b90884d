#!/usr/bin/env python3
"""Tier-P2 parity of the throughput engine against THE REFERENCE ITSELF, at the benchmarked workload.

oracle/_ref/soil_ref_lean -- the reference's own headers compiled headless (oracle/Makefile) -- runs on the box's host cores:
  * SEED 0: the run the device is compared with (same terrain, same rand() stream at the start of tick 1);
  * NCTL control streams: the same terrain, srand(other) once it stands (ref_driver --rand-seed) -- what the reference's own
    chaos does to every figure;
all in parallel (the reference has no threads: one process per stream), ONCE, while the device runs the same ticks with every engine
VARIANT under test (water generations, column-strip schedule, other device rand() streams). Per variant and figure:
  z      = (engine - mean(controls)) / std(controls, ddof=1)          the figure rounds 4-5 reported (controls = every stream but SEED 0)
  z_all  = the same against ALL reference streams, SEED 0 included    (the null the engine is supposed to be a draw from)
  rank   = how many reference streams lie below the engine's value    (distribution-free: 0 or N = outside every reference run)
and for the INSTRUMENT itself (round 6): the leave-one-out z of every reference stream against the others
(`calibration`): how often does the reference itself leave |z| < 2 / < 3 on each figure?  A figure whose own streams do that often is
heavy-tailed at this sample size, and its z is quoted with that caveat. Height RMSE is one-sided and relative to the SEED-0 run.
CHECKER ONLY: nothing here is on the product path.

  p2_reference.py [--size 4096] [--ticks 10] [--nctl 32] [--variants k4,k8,k4s2,k4r1,k4r2] [--soil ...] [--out profiles/...json]
A variant is k<generations>[e<stagger>][s<strips>][r<device rand stream>]: k8 = eight water generations; k8e140 = the eight generations born 140 epochs apart inside one phase
(smx_set_water_stagger); k4s2 = the two-strip schedule in one context
(smx_set_batch_strips -- what `bench.py --gpus 2` runs on every rank); k4r1 = the device under srand(1000 + 7919 * 101) instead of SEED's.
Counts are area-scaled (250 * (N/256)^2 water, NWIND/NWATER of the soil file).
"""
from __future__ import annotations

import argparse, json, os, re, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

FIGS = ("steps_water_top", "steps_wind", "nested", "sections", "standing", "water_volume", "sumh", "height_rmse")
ONE_SIDED = ("height_rmse",)


def gate(zs: dict) -> dict:
    """The rule of tests/test_gpu_relaxed.py and bench.py's parity_gate: nothing at |z| >= 3, at most one figure in [2, 3).
    (Eight t-distributed scores of one run against <= 32 controls put a perfect schedule beyond 2 on one of them every third run.)"""
    big = [k for k, z in zs.items() if (z >= 3.0 if k in ONE_SIDED else abs(z) >= 3.0)]
    mid = [k for k, z in zs.items() if k not in big and (z >= 2.0 if k in ONE_SIDED else abs(z) >= 2.0)]
    return {"pass": not big and len(mid) <= 1, "beyond_3": big, "between_2_and_3": mid}


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


def _z(e, c, one):
    c = np.asarray(c, float)
    mu, sd = float(c.mean()), (float(c.std(ddof=1)) if len(c) > 1 else 0.0)
    z = (e - mu) / sd if sd > 0 else (0.0 if e == mu else float("inf"))
    return z, mu, sd


def zscores(eng, ref, ctl):
    """eng: the engine's figures; ref: the SEED-0 reference run's; ctl: the control streams' (list of dicts)."""
    out = {}
    for k in FIGS:
        one = k in ONE_SIDED
        c = np.array([x[k] for x in ctl], float)
        e = float(eng[k])
        z, mu, sd = _z(e, c, one)
        call = c if one else np.append(c, float(ref[k]))          # (the SEED-0 run's RMSE against itself is 0 by construction: not a sample)
        z_all, mu_all, sd_all = _z(e, call, one)
        rel = (lambda v, r=ref[k]: v / r - 1.0) if (not one and ref[k]) else (lambda v: v)
        out[k] = {"engine": e, "reference_seed0": ref[k], "control_mean": mu, "control_std": sd, "control_min": float(c.min()), "control_max": float(c.max()),
                  "z": z, "z_all": z_all, "rank": int((call < e).sum()), "of": int(len(call)), "one_sided": one,
                  "outlier": (z > 2.0) if one else (abs(z) > 2.0),
                  "engine_rel": rel(e), "control_rel_min": rel(float(c.min())), "control_rel_max": rel(float(c.max()))}
    return out


def calibration(ref, ctl):
    """Leave-one-out z of every reference stream (SEED 0 first) against the others, per figure: what the instrument says about the
    reference itself. height_rmse is relative to the SEED-0 run, so the SEED-0 stream has no score there."""
    allr = [ref] + list(ctl)
    out = {}
    for k in FIGS:
        one = k in ONE_SIDED
        pool = list(ctl) if one else allr
        v = np.array([x[k] for x in pool], float)
        zs = []
        for i in range(len(v)):
            rest = np.delete(v, i)
            sd = rest.std(ddof=1)
            zs.append(float((v[i] - rest.mean()) / sd) if sd > 0 else 0.0)
        a = np.abs(np.array(zs))
        out[k] = {"loo_z": [round(z, 3) for z in zs], "streams": len(zs), "beyond_2": int((a > 2).sum()), "beyond_3": int((a > 3).sum()), "max_abs": float(a.max()),
                  "seed0_loo_z": (None if one else round(zs[0], 3)),
                  "mean": float(v.mean()), "std": float(v.std(ddof=1)), "skew": float(((v - v.mean()) ** 3).mean() / max(v.std() ** 3, 1e-300))}
    # how often does a reference stream pass the gate that is asked of the engine? (leave-one-out, all figures of one stream together)
    n = len(allr)
    passed = 0
    for i in range(n):
        zs = {}
        for k in FIGS:
            if k in ONE_SIDED:
                if i == 0:
                    continue
                zs[k] = out[k]["loo_z"][i - 1]
            else:
                zs[k] = out[k]["loo_z"][i]
        passed += 1 if gate(zs)["pass"] else 0
    out["_streams_passing_the_gate"] = {"passed": passed, "of": n}
    return out


def parse_variant(name: str) -> dict:
    m = re.fullmatch(r"k(\d+)(?:e(\d+))?(?:s(\d+))?(?:r(\d+))?", name)
    if not m:
        raise ValueError(f"variant {name!r}: expected k<generations>[e<stagger epochs>][s<strips>][r<stream>]")
    return {"name": name, "generations": int(m.group(1)), "stagger": int(m.group(2) or 0), "strips": int(m.group(3) or 1), "stream": int(m.group(4) or 0)}


def device_run(cfg, size, ticks, nwater, nwind, pool, device, engine, var):
    from soilmachine_amd import capi
    from soilmachine_amd.machine import SoilMachine
    ENG = {"relaxed": capi.ENGINE_RELAXED, "batched": capi.ENGINE_BATCHED, "spec": capi.ENGINE_SPECULATIVE}
    sm = SoilMachine(cfg, size, seed=0, nwater=nwater, nwind=nwind, dowind=nwind > 0, pool=pool, device=device, engine=ENG[engine])
    m = sm.map
    if var["stream"]:
        m._chk(m.L.smx_srand(m.h, 1000 + 7919 * (100 + var["stream"])))      # another rand() stream on the same terrain (never one of the controls')
    if var["generations"]:
        m.set_water_generations(var["generations"])
    if var.get("stagger") is not None:
        m.set_water_stagger(var["stagger"])                         # (0: consecutive generations, whatever the library's default)
    if var["strips"] > 1:
        m.set_batch_strips(var["strips"])
    tg = time.time()
    sm.tick(ticks, sync=True)
    t_gpu = time.time() - tg
    c = m.counters()
    hg = m.heights()
    s = m.snapshot()
    end = np.cumsum(s.count.astype(np.int64)); nz = s.count > 0
    top = end[nz] - 1
    air = s.type[top] == 0
    eng = {"steps_water_top": c["steps_water_top"], "steps_wind": c["steps_wind"], "nested": c["nested_particles"], "sections": int(s.nsec),
           "standing": int(air.sum()), "water_volume": float(s.size[top][air].sum()), "sumh": float(hg.sum()), "floods": c["floods"],
           "steps_water_all": c["steps_water_all"], "device_s": t_gpu}
    del s
    m.close()
    return eng, hg


def run(soil="rockgravelpebblessand.soil", size=4096, ticks=10, nctl=8, engine="relaxed", device=0, verbose=True, nwater=None, nwind=None, water_batches=None,
        variants=None):
    """Returns the report dict, or {"skipped": reason}. Without `variants` the report has round 5's shape (one engine configuration: `figures`,
    `outliers`) plus `calibration` and `gate`; with variants every configuration's block sits under `variants[name]`."""
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
    single = variants is None
    if single:
        variants = [{"name": "default", "generations": int(water_batches or 0), "stagger": None, "strips": 1, "stream": 0}]
    t0 = time.time()
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        procs = []
        for k in range(nproc):
            hp = os.path.join(td, f"h{k}.f64")
            procs.append((hp, subprocess.Popen(_ref_cmd(soil, size, nwater, nwind, ticks, pool, hp, None if k == 0 else 1000 + 7919 * k, core=(k + 1) % ncpu),
                                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)))
        # the device, meanwhile
        engs = []
        for var in variants:
            e, hg = device_run(cfg, size, ticks, nwater, nwind, pool, device, engine, var)
            engs.append((var, e, hg))
            if verbose:
                print(f"[p2_reference] device ({engine}, {var['name']}) done in {e['device_s']:.1f} s", file=sys.stderr, flush=True)
        if verbose:
            print(f"[p2_reference] waiting for {nproc} reference streams ...", file=sys.stderr, flush=True)
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
        hstd = float(h0.std())
        blocks = {}
        for var, eng, hg in engs:
            d = hg - h0
            eng["height_rmse"] = float(np.sqrt((d * d).mean()))
            eng["height_max_abs"] = float(np.abs(d).max())
            z = zscores(eng, ref, ctl)
            blocks[var["name"]] = {"variant": var, "figures": z, "outliers": [k for k in FIGS if z[k]["outlier"]],
                                   "gate": gate({k: z[k]["z"] for k in FIGS}), "gate_all_streams": gate({k: z[k]["z_all"] for k in FIGS}),
                                   "engine_extra": {k: eng[k] for k in ("floods", "steps_water_all", "height_max_abs", "device_s")}}
        del engs
    rep = {"what": __doc__.split("\n\n")[0], "workload": f"{size}x{size}, soil/{soil}, SEED 0, nwater={nwater}, nwind={nwind}, ticks 1-{ticks} from the initial terrain",
           "engine": engine, "reference": "oracle/_ref/soil_ref_lean (the reference's own headers, g++ -O2, one process per stream, pinned cores)",
           "controls": nctl, "control_seeds": "srand(1000 + 7919 k), k = 1..controls, once the terrain stands (ref_driver --rand-seed); SEED 0 = srand(0)",
           "calibration": calibration(ref, ctl), "height_std": hstd,
           "reference_seed0_s_per_tick": (js[0]["t_water"] + js[0]["t_grid"] + js[0]["t_wind"] + js[0]["t_freq"]) / ticks,
           "wall_s": time.time() - t0, "host_cores": ncpu}
    if single:
        b = blocks["default"]
        rep.update({"figures": b["figures"], "outliers": b["outliers"], "gate": b["gate"], "gate_all_streams": b["gate_all_streams"],
                    "engine_extra": b["engine_extra"], "device_s": b["engine_extra"]["device_s"]})
    else:
        rep["variants"] = blocks
        # device streams of one configuration as an ENSEMBLE against the reference's streams (Welch's t on the means): variants that differ
        # only in the device's rand() stream are grouped
        groups = {}
        for name, b in blocks.items():
            v = b["variant"]
            groups.setdefault((v["generations"], v.get("stagger") or 0, v["strips"]), []).append(name)
        ens = {}
        allr = [ref] + ctl
        for (g, e, s), names in groups.items():
            if len(names) < 2:
                continue
            row = {}
            for k in FIGS:
                if k in ONE_SIDED:
                    continue
                a = np.array([blocks[n]["figures"][k]["engine"] for n in names], float)
                r = np.array([x[k] for x in allr], float)
                se = np.sqrt(a.var(ddof=1) / len(a) + r.var(ddof=1) / len(r))
                row[k] = {"engine_mean": float(a.mean()), "engine_std": float(a.std(ddof=1)), "reference_mean": float(r.mean()), "reference_std": float(r.std(ddof=1)),
                          "rel": float(a.mean() / r.mean() - 1.0) if r.mean() else 0.0, "welch_t": float((a.mean() - r.mean()) / se) if se > 0 else 0.0}
            ens[f"k{g}" + (f"e{e}" if e else "") + (f"s{s}" if s > 1 else "")] = {"device_streams": names, "reference_streams": len(allr), "figures": row}
        rep["ensembles"] = ens
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--ticks", type=int, default=10)
    ap.add_argument("--nctl", type=int, default=32)
    ap.add_argument("--engine", default="relaxed")
    ap.add_argument("--soil", default="rockgravelpebblessand.soil")
    ap.add_argument("--out", default=None)
    ap.add_argument("--variants", default="", help="comma list of k<generations>[s<strips>][r<stream>]; empty = one run with the engine's defaults")
    ap.add_argument("--water-batches", type=int, default=int(os.environ.get("P2_WATER_BATCHES", "0")), help="(single run) 0 = the engine's default")
    a = ap.parse_args()
    variants = [parse_variant(v) for v in a.variants.split(",") if v] or None
    rep = run(a.soil, a.size, a.ticks, a.nctl, a.engine, water_batches=a.water_batches, variants=variants)
    def show(name, b):
        print(f"--- {name}: gate {'PASS' if b['gate']['pass'] else 'FAIL'} {b['gate']}", file=sys.stderr)
        for k in FIGS:
            f = b["figures"][k]
            print(f"{k:16s} engine {f['engine']:.6g}  ref0 {f['reference_seed0']:.6g}  controls {f['control_min']:.6g} .. {f['control_max']:.6g}  z = {f['z']:+.2f} (all streams {f['z_all']:+.2f}, rank {f['rank']}/{f['of']}){'  OUTLIER' if f['outlier'] else ''}", file=sys.stderr)
    if "variants" in rep:
        for name, b in rep["variants"].items():
            show(name, b)
    elif "figures" in rep:
        show("default", rep)
    if "calibration" in rep:
        cal = rep["calibration"]
        print("--- the instrument on the reference itself (leave-one-out):", file=sys.stderr)
        for k in FIGS:
            c = cal[k]
            print(f"{k:16s} streams {c['streams']}  |z|>2: {c['beyond_2']}  |z|>3: {c['beyond_3']}  max {c['max_abs']:.2f}  SEED-0 {c['seed0_loo_z']}  skew {c['skew']:+.2f}", file=sys.stderr)
        print(f"reference streams passing the gate: {cal['_streams_passing_the_gate']}", file=sys.stderr)
    s = json.dumps(rep, indent=1)
    if a.out:
        open(a.out, "w").write(s)
    print(s)


if __name__ == "__main__":
    main()
