"""Which phase of the throughput tick biases the water statistics? Hybrid ticks on the host (tests/hostsim): the exact serial phases with ONE
phase replaced by its throughput version, ensembles over the same rand() streams (profiles/r04_p2_analysis.md).
  p2_hybrid.py <case> [mode,mode,...]   modes: exact tp_water tp_water_split<k> tp_grid tp_wind all"""
import os, sys, time
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import numpy as np
from common import load_cfg
from hostsim_lib import HostSim, lib
from oracle_lib import Oracle
from p2_zscores import CASES, standing
case = sys.argv[1]; K = int(os.environ.get('K', 6)); SIZE = int(os.environ.get('SIZE', 256)); TICKS = os.environ.get('TICKS')
soil, ticks, nw, nd = CASES[case]
cfg = load_cfg(soil); n = SIZE; nw = int(nw * (n / 256.0) ** 2); nd = int(nd * (n / 256.0) ** 2); ticks = int(TICKS) if TICKS else ticks
o0 = Oracle(cfg, n, n, seed=0); o0.initialize(); s0 = o0.snapshot()
L = lib()
def run(mode, sd):
    h = HostSim(cfg, n, n, seed=0); h.load(s0, advance_rand=False); L.hs_srand(h.h, sd)
    for t in range(ticks):
        if mode == "exact":
            L.hs_water(h.h, nw); L.hs_grid(h.h)
            if nd: L.hs_wind(h.h, nd)
            L.hs_freq(h.h)
        elif mode == "tp_water":            # relaxed water particles, exact grid + wind
            L.hs_set_relaxed(1); L.hs_batched_water(h.h, nw, 1, 0); L.hs_set_relaxed(0)
            L.hs_grid(h.h)
            if nd: L.hs_wind(h.h, nd)
            L.hs_freq(h.h)
        elif mode == "tpb_water":           # the BATCHED (exclusive) water schedule, exact grid + wind: concurrency without relaxed isolation
            L.hs_batched_water(h.h, nw, 1, 0)
            L.hs_grid(h.h)
            if nd: L.hs_wind(h.h, nd)
            L.hs_freq(h.h)
        elif mode.startswith("tp_water_split"):   # the same in k sequential batches of nw/k particles: less intra-tick concurrency
            k = int(mode[len("tp_water_split"):])
            L.hs_set_relaxed(1)
            for q in range(k): L.hs_batched_water(h.h, nw // k, 1, 0)
            L.hs_set_relaxed(0)
            L.hs_grid(h.h)
            if nd: L.hs_wind(h.h, nd)
            L.hs_freq(h.h)
        elif mode == "tp_grid":             # exact water + wind, throughput grid pass
            L.hs_water(h.h, nw)
            L.hs_set_relaxed(1); L.hs_batched_grid(h.h, 1, 0); L.hs_set_relaxed(0)
            if nd: L.hs_wind(h.h, nd)
            L.hs_freq(h.h)
        elif mode == "tp_wind":
            L.hs_water(h.h, nw); L.hs_grid(h.h)
            if nd: L.hs_set_relaxed(1); L.hs_batched_wind(h.h, nd, 1, 0); L.hs_set_relaxed(0)
            L.hs_freq(h.h)
        elif mode == "all":
            h.relaxed_tick(nw, nd, True, nd > 0, nthreads=1)
    s = h.snapshot(); c = h.counters()
    return {"steps": c["steps_water_all"], "nested": c["nested_particles"], "floods": c["floods"], "standing": standing(s), "nsec": int(s.nsec)}
seeds = [1000 + 7919 * k for k in range(K)]
res = {}
for mode in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("exact", "tp_water", "tp_grid", "tp_wind", "all")):
    t0 = time.time()
    R = [run(mode, sd) for sd in seeds]
    res[mode] = {k: np.mean([r[k] for r in R]) for k in R[0]}
    e = res["exact"]
    print(f"{mode:9s} " + "  ".join(f"{k} {res[mode][k]:.0f} ({100*(res[mode][k]/e[k]-1):+.1f}%)" for k in e) + f"  [{time.time()-t0:.0f}s]", flush=True)
