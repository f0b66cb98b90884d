"""One grid pass, two ways, from the SAME state: at every tick of an exact run on the host (tests/hostsim) the state before the grid pass is
cloned; one clone gets the reference-order pass (water.h:335-343), the other the throughput pass (nine tile colours, nested particles as
generations after the pass). Deterministic and free of the chaos that blurs multi-tick ensembles: what the pass itself does differently.
  p2_onepass.py <size> <ticks> [seed]    env: CASE=<tools/p2_zscores.py case>
CHECKER ONLY."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d_ in ('tests', '', 'tools'): sys.path.insert(0, os.path.join(ROOT, d_))
import numpy as np
from common import load_cfg
from hostsim_lib import HostSim, lib
from oracle_lib import Oracle
from p2_zscores import CASES
case = os.environ.get('CASE', 'rgps')
n, ticks = int(sys.argv[1]), int(sys.argv[2]); sd = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
soil, _, nw, nd = CASES[case]
cfg = load_cfg(soil); nw = int(nw * (n / 256.0) ** 2); nd = int(nd * (n / 256.0) ** 2)
if os.environ.get('TERRAIN_OF'):
    for l in cfg.layers: l.frequency *= n / float(os.environ['TERRAIN_OF'])
o0 = Oracle(cfg, n, n, seed=0); o0.initialize(); s0 = o0.snapshot()
L = lib()
def figs(h):
    s = h.snapshot(); c = h.counters()
    end = np.cumsum(s.count.astype(np.int64)); top = end[s.count > 0] - 1
    air = s.type[top] == 0
    return {"steps": c["steps_water_all"], "nested": c["nested_particles"], "floods": c["floods"], "standing": int(air.sum()), "wvol": float(s.size[top][air].sum())}
main = HostSim(cfg, n, n, seed=0); main.load(s0, advance_rand=False); L.hs_srand(main.h, sd)
tot = {}
for t in range(ticks):
    L.hs_water(main.h, nw)
    S = main.snapshot(); pre = figs(main)
    res = {}
    for nm in ("exact", "tp"):
        h = HostSim(cfg, n, n, seed=0); h.load(S, advance_rand=False); L.hs_srand(h.h, 4242 + t)
        if nm == "exact": L.hs_grid(h.h)
        else:
            L.hs_set_relaxed(1); L.hs_batched_grid(h.h, 1, 0); L.hs_set_relaxed(0)
        res[nm] = figs(h)
    e, p = res["exact"], res["tp"]
    print(f"t{t+1:2d} pre standing {pre['standing']:5d} wvol {pre['wvol']:.4f} | exact: standing {e['standing']:5d} wvol {e['wvol']:.4f} nested {e['nested']:5d} floods {e['floods']:5d} steps {e['steps']:7d} | tp: standing {p['standing']:5d} wvol {p['wvol']:.4f} nested {p['nested']:5d} floods {p['floods']:5d} steps {p['steps']:7d}", flush=True)
    for k in e:
        tot.setdefault(k, [0, 0]); tot[k][0] += e[k]; tot[k][1] += p[k]
    L.hs_grid(main.h)
    if nd: L.hs_wind(main.h, nd)
    L.hs_freq(main.h)
print("SUM  " + "  ".join(f"{k}: exact {v[0]:.6g} tp {v[1]:.6g} ({100*(v[1]/v[0]-1) if v[0] else 0:+.1f}%)" for k, v in tot.items()), " env=", {k:v for k,v in os.environ.items() if k.startswith('HS_')})
