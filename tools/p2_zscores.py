#!/usr/bin/env python3
"""Tier-P2 parity of a throughput schedule as z-scores against the reference's OWN rand()-stream distribution, on the CPU.

For every case: the reference-order oracle (oracle/soil_oracle.cpp, pinned bit for bit to oracle/_ref) runs SEED 0 and NCTL control
streams (the same terrain, srand(other) once it stands); the throughput schedule runs as the device headers on host threads
(tests/hostsim: bit-identical to the device). Every figure of the schedule and of every control is taken relative to the SEED-0 run;
z = (schedule - mean(controls)) / std(controls, ddof=1). Height RMSE is one-sided (a schedule closer to SEED 0 than the controls are
is not an outlier). VERDICT r3 item 1: "each statistic as a z-score against the control distribution ... >= 8 control streams".

  p2_zscores.py [relaxed|batched] [--nctl 8] [--size 256] [--out profiles/r04_p2_zscores_cpu.json] [case ...]
"""
import argparse, json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from common import load_cfg
from hostsim_lib import HostSim
from oracle_lib import Oracle

CASES = {"rgps": ("rockgravelpebblessand.soil", 40, 250, 250), "rocksand": ("rocksand.soil", 40, 250, 100), "painted": ("painted.soil", 40, 250, 100),
         "sand": ("sand.soil", 40, 250, 250), "rgpbig": ("rockgravelpebbles_big.soil", 20, 1000, 0), "default_wet": ("default.soil", 150, 250, 0)}
FIGS = ("steps_water", "steps_top", "nested", "floods", "sections", "standing", "sumh", "height_rmse")


def standing(s):
    end = np.cumsum(s.count.astype(np.int64)); top = end[s.count > 0] - 1
    return int((s.type[top] == 0).sum())


def figures(snap, ctr, h0):
    h = snap.heights(); d = h - h0
    return {"steps_water": ctr["steps_water_all"], "steps_top": ctr["steps_water_top"], "nested": ctr["nested_particles"], "floods": ctr["floods"], "sections": int(snap.nsec),
            "standing": standing(snap), "sumh": float(h.sum()), "height_rmse": float(np.sqrt((d * d).mean()))}


def zscores(eng, ref, ctl):
    out = {}
    for k in FIGS:
        c = np.array([x[k] for x in ctl], float)
        mu, sd = float(c.mean()), float(c.std(ddof=1)) if len(c) > 1 else 0.0
        e = float(eng[k])
        z = (e - mu) / sd if sd > 0 else (0.0 if e == mu else float("inf"))
        rel = (lambda v: v / ref[k] - 1.0) if (k != "height_rmse" and ref[k]) else (lambda v: v)
        out[k] = {"schedule": e, "reference_seed0": ref[k], "control_mean": mu, "control_std": sd, "control_min": float(c.min()), "control_max": float(c.max()),
                  "z": z, "one_sided": k == "height_rmse", "outlier": (z > 2.0) if k == "height_rmse" else (abs(z) > 2.0),
                  "schedule_rel": rel(e), "control_rel_min": rel(float(c.min())), "control_rel_max": rel(float(c.max()))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("engine", nargs="?", default="relaxed")
    ap.add_argument("cases", nargs="*")
    ap.add_argument("--nctl", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--out", default=None)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 4)
    ap.add_argument("--water-generations", type=int, default=0, help="0 = the engine's default (4)")
    a = ap.parse_args()
    if a.engine in CASES:
        a.cases.insert(0, a.engine); a.engine = "relaxed"
    if a.water_generations:
        from hostsim_lib import lib as _hl
        _hl().hs_set_water_generations(a.water_generations)
    res = {"engine": a.engine, "size": a.size, "nctl": a.nctl, "protocol": __doc__.split("\n\n")[1], "cases": {}}
    for nm in (a.cases or list(CASES)):
        soil, ticks, nw, nd = CASES[nm]
        cfg = load_cfg(soil); n = a.size
        sc = (n / 256.0) ** 2
        nw, nd = int(nw * sc), int(nd * sc)
        o = Oracle(cfg, n, n, seed=0); o.initialize(); s0 = o.snapshot()
        ctl = []
        for k in range(a.nctl):
            c = Oracle(cfg, n, n, seed=0); c.load(s0); c.L.so_srand(c.h, 1000 + 7919 * k); ctl.append(c)
        h = HostSim(cfg, n, n, seed=0); h.load(s0, advance_rand=False)
        t0 = time.time()
        hs_threads = max(2, a.threads // 3)
        def run_oracle(x):
            for _ in range(ticks): x.tick(nw, nd, True, nd > 0)
        def run_engine():
            f = h.relaxed_tick if a.engine == "relaxed" else h.batched_tick
            for _ in range(ticks): f(nw, nd, True, nd > 0, nthreads=hs_threads)
        with ThreadPoolExecutor(max_workers=a.threads) as ex:
            futs = [ex.submit(run_engine)] + [ex.submit(run_oracle, x) for x in [o] + ctl]
            for f in futs: f.result()
        se = o.snapshot(); h0 = se.heights()
        ref = figures(se, o.counters(), h0)
        eng = figures(h.snapshot(), h.counters(), h0)
        cf = [figures(c.snapshot(), c.counters(), h0) for c in ctl]
        z = zscores(eng, ref, cf)
        res["cases"][nm] = {"soil": soil, "ticks": ticks, "nwater": nw, "nwind": nd, "height_std": float(h0.std()), "figures": z, "wall_s": time.time() - t0}
        bad = [k for k in FIGS if z[k]["outlier"]]
        print(f"{nm:12s} {soil[:-5]:26s} " + "  ".join(f"{k}={z[k]['z']:+.2f}" for k in FIGS) + (f"   OUTLIERS: {bad}" if bad else "   all |z|<2") + f"   [{time.time() - t0:.0f}s]", flush=True)
        if a.out:
            json.dump(res, open(a.out, "w"), indent=1)
    return res


if __name__ == "__main__":
    main()
