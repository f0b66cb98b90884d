"""A CPU proxy of the headline-size P2 test (tools/p2_reference.py on the MI355X box): K ensembles at a small map instead of one run at a large one.

The z-scores of tools/p2_reference.py grow with the map because the reference's own rand()-stream spread shrinks like 1/sqrt(area) while the
schedule's relative bias stays: K independent streams at n^2 have the statistical power of ONE run at K*n^2 cells. This tool runs, on the host
(tests/hostsim = the device headers on host threads, exact serial phases from the same headers), hybrid ticks in which ONE phase (or both) is
replaced by its throughput version, K rand() streams per mode in parallel processes, and prints every figure as
  z = (mean over the K streams of the mode - mean over the K streams of the exact tick) / (std of the exact streams / sqrt(K)),
i.e. in units of the control spread of one run at K x the area (16 x 1024^2 = 4096^2). |z| <~ 2.8 is the noise of this proxy (both means carry it).
Round 5 found the grid pass's bias with it (the cap of 24 nested particles per tile and pass: profiles/r05_p2_child_cap.md).

  p2_proxy.py <size> <ticks> <K> mode[,mode]     modes: tp_water tp_grid tp_both    env: TERRAIN_OF=<map size whose terrain regime to use> KW=<water generations> CASE=<tools/p2_zscores.py case>
CHECKER ONLY (tests/hostsim, oracle): nothing here is on the product path."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d_ in ('tests', '', 'tools'): sys.path.insert(0, os.path.join(ROOT, d_))
import numpy as np
from multiprocessing import Pool
case = os.environ.get('CASE', 'rgps')
n, ticks, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
def run(args):
    mode, sd = args
    from common import load_cfg
    from hostsim_lib import HostSim, lib
    from oracle_lib import Oracle
    from p2_zscores import CASES
    soil, _, nw, nd = CASES[case]
    cfg = load_cfg(soil); nw = int(nw * (n / 256.0) ** 2); nd = int(nd * (n / 256.0) ** 2)
    if os.environ.get("TERRAIN_OF"):                         # the terrain REGIME of a larger map: the noise of a TERRAIN_OF^2 map sampled over n^2 cells
        for l in cfg.layers: l.frequency *= n / float(os.environ["TERRAIN_OF"])   # (the layers' noise is a function of i / dim, layermap.h:183-192)
    o0 = Oracle(cfg, n, n, seed=0); o0.initialize(); s0 = o0.snapshot()
    L = lib()
    if os.environ.get("HS_STAGGER"): L.hs_set_water_stagger(*[int(v) for v in os.environ["HS_STAGGER"].split(",")])
    if os.environ.get("HS_RELAX_WATER_STEPS"): L.hs_set_relax_water(int(os.environ["HS_RELAX_WATER_STEPS"]))   # steps per epoch of a relaxed water particle (experiment)
    h = HostSim(cfg, n, n, seed=0); h.load(s0, advance_rand=False); L.hs_srand(h.h, sd)
    kw = int(os.environ.get('KW', '1'))
    for t in range(ticks):
        if mode == "exact":
            L.hs_water(h.h, nw); L.hs_grid(h.h)
        elif mode == "tp_water":
            L.hs_set_relaxed(1)
            for q in range(kw): L.hs_batched_water(h.h, nw // kw, 1, 0)
            L.hs_set_relaxed(0); L.hs_grid(h.h)
        elif mode == "tp_grid":
            L.hs_water(h.h, nw)
            L.hs_set_relaxed(1); L.hs_batched_grid(h.h, 1, 0); L.hs_set_relaxed(0)
        elif mode == "tp_both":
            L.hs_set_relaxed(1)
            for q in range(kw): L.hs_batched_water(h.h, nw // kw, 1, 0)
            L.hs_batched_grid(h.h, 1, 0); L.hs_set_relaxed(0)
        if nd: L.hs_wind(h.h, nd)
        L.hs_freq(h.h)
    s = h.snapshot(); c = h.counters()
    end = np.cumsum(s.count.astype(np.int64)); top = end[s.count > 0] - 1
    air = s.type[top] == 0
    return {"steps": c["steps_water_all"], "nested": c["nested_particles"], "floods": c["floods"], "standing": int(air.sum()), "wvol": float(s.size[top][air].sum()), "nsec": int(s.nsec), "sumh": float(s.heights().sum())}
if __name__ == "__main__":
    seeds = [1000 + 7919 * k for k in range(K)]
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"p2_proxy_exact_{case}_{n}_{ticks}_{K}_{os.environ.get('TERRAIN_OF', n)}.json")
    with Pool(8) as pool:
        if os.path.exists(cache): E = json.load(open(cache))
        else:
            E = pool.map(run, [("exact", sd) for sd in seeds]); json.dump(E, open(cache, "w"))
        em = {k: np.mean([r[k] for r in E]) for k in E[0]}; es = {k: np.std([r[k] for r in E], ddof=1) for k in E[0]}
        print("exact     " + "  ".join(f"{k} {em[k]:.6g}±{100*es[k]/em[k]:.1f}%" for k in em), flush=True)
        for mode in sys.argv[4].split(","):
            t0 = time.time()
            R = pool.map(run, [(mode, sd) for sd in seeds])
            m = {k: np.mean([r[k] for r in R]) for k in R[0]}; ms = {k: np.std([r[k] for r in R], ddof=1) for k in R[0]}
            out = []
            for k in em:
                se = np.sqrt(es[k] ** 2 / len(E) + ms[k] ** 2 / len(R))
                z1 = (m[k] - em[k]) / (es[k] / np.sqrt(K))      # in units of the spread of a K-ensemble mean of the reference (~ one run at K x the area)
                out.append(f"{k} {100*(m[k]/em[k]-1):+.1f}% (z{z1:+.1f})" if k != "sumh" else f"{k} {m[k]-em[k]:+.3f} (z{z1:+.1f})")
            print(f"{mode:9s} " + "  ".join(out) + f"  [{time.time()-t0:.0f}s] env={ {k:v for k,v in os.environ.items() if k.startswith('HS_') or k=='KW'} }", flush=True)
