#!/usr/bin/env python3
"""Is the throughput schedule BIASED against the reference? Ensemble against ensemble, on the CPU.

tools/p2_zscores.py compares ONE engine run (SEED 0's rand() stream) with the reference's rand()-stream distribution; the engine run shares
its spawn positions with the SEED-0 reference run, which may itself be an unusual draw. Here both sides are ensembles over the SAME K
rand() streams on the SAME terrain: the reference-order oracle (pinned to oracle/_ref) and the schedule (device headers on host threads,
tests/hostsim) are each run under srand(s_k), k = 1..K. Per figure: the two ensemble means, Welch's t = (mean_e - mean_r) /
sqrt(var_e/K + var_r/K), and the paired mean of per-stream ratios. |t| < 2: no bias detectable at K streams.

  p2_ensemble.py [relaxed|batched] [--k 8] [--size 256] [--out profiles/r04_p2_ensemble_cpu.json] [case ...]"""
import argparse, json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from common import load_cfg
from hostsim_lib import HostSim
from oracle_lib import Oracle
from p2_zscores import CASES, standing

FIGS = ("steps_water", "nested", "floods", "sections", "standing", "sumh")


def figs(snap, ctr):
    return {"steps_water": ctr["steps_water_all"], "nested": ctr["nested_particles"], "floods": ctr["floods"], "sections": int(snap.nsec),
            "standing": standing(snap), "sumh": float(snap.heights().sum())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("engine", nargs="?", default="relaxed")
    ap.add_argument("cases", nargs="*")
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.engine in CASES:
        a.cases.insert(0, a.engine); a.engine = "relaxed"
    res = {"engine": a.engine, "size": a.size, "k": a.k, "protocol": __doc__.split("\n\n")[1], "cases": {}}
    for nm in (a.cases or list(CASES)):
        soil, ticks, nw, nd = CASES[nm]
        cfg = load_cfg(soil); n = a.size
        sc = (n / 256.0) ** 2; nw, nd = int(nw * sc), int(nd * sc)
        o0 = Oracle(cfg, n, n, seed=0); o0.initialize(); s0 = o0.snapshot()
        seeds = [1000 + 7919 * k for k in range(a.k)]
        refs, engs = [], []
        for sd in seeds:
            c = Oracle(cfg, n, n, seed=0); c.load(s0); c.L.so_srand(c.h, sd); refs.append(c)
            h = HostSim(cfg, n, n, seed=0); h.load(s0, advance_rand=False); h.L.hs_srand(h.h, sd); engs.append(h)
        t0 = time.time()
        def run_o(x):
            for _ in range(ticks): x.tick(nw, nd, True, nd > 0)
        def run_e(h):
            f = h.relaxed_tick if a.engine == "relaxed" else h.batched_tick
            for _ in range(ticks): f(nw, nd, True, nd > 0, nthreads=3)
        with ThreadPoolExecutor(max_workers=max(2, (os.cpu_count() or 4) - 3)) as ex:
            futs = [ex.submit(run_o, x) for x in refs]
            for h in engs: run_e(h)                          # ONE host-sim at a time: its engine switches are process-global (hs_set_relaxed)
            for f in futs: f.result()
        R = [figs(x.snapshot(), x.counters()) for x in refs]
        E = [figs(h.snapshot(), h.counters()) for h in engs]
        rm = []
        for x, h in zip(refs, engs):
            d = h.snapshot().heights() - x.snapshot().heights(); rm.append(float(np.sqrt((d * d).mean())))
        rr = []
        for i in range(a.k):
            for j in range(i + 1, a.k):
                d = refs[i].snapshot().heights() - refs[j].snapshot().heights(); rr.append(float(np.sqrt((d * d).mean())))
        out = {}
        for k in FIGS:
            r = np.array([x[k] for x in R], float); e = np.array([x[k] for x in E], float)
            se = np.sqrt(r.var(ddof=1) / len(r) + e.var(ddof=1) / len(e))
            t = float((e.mean() - r.mean()) / se) if se > 0 else 0.0
            out[k] = {"reference_mean": float(r.mean()), "reference_std": float(r.std(ddof=1)), "engine_mean": float(e.mean()), "engine_std": float(e.std(ddof=1)),
                      "welch_t": t, "mean_ratio": float(e.mean() / r.mean()) if r.mean() else None, "biased": abs(t) > 2.0}
        res["cases"][nm] = {"soil": soil, "ticks": ticks, "nwater": nw, "nwind": nd, "figures": out,
                            "height_rmse_engine_vs_reference_same_stream": {"mean": float(np.mean(rm)), "min": float(np.min(rm)), "max": float(np.max(rm))},
                            "height_rmse_reference_vs_reference_other_stream": {"mean": float(np.mean(rr)), "min": float(np.min(rr)), "max": float(np.max(rr))},
                            "wall_s": time.time() - t0}
        bad = [k for k in FIGS if out[k]["biased"]]
        print(f"{nm:12s} " + "  ".join(f"{k}: t={out[k]['welch_t']:+.2f} ({100 * (out[k]['mean_ratio'] - 1):+.1f}%)" for k in FIGS) +
              f"  rmse same-stream {np.mean(rm):.2e} vs ref-ref {np.mean(rr):.2e}" + (f"   BIASED: {bad}" if bad else "") + f"   [{time.time() - t0:.0f}s]", flush=True)
        if a.out:
            json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
