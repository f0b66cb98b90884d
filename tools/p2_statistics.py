#!/usr/bin/env python3
"""Tier-P2 parity of a throughput schedule measured against the reference's OWN variability, on the CPU (256^2 maps):
the device headers run by host threads (tests/hostsim: bit-identical to the device) against the reference-order oracle from the
same terrain and rand() stream, next to the reference-order oracle with two other rand() streams. One markdown row pair per soil.
  p2_statistics.py relaxed|batched [case ...]     (profiles/r02_p2_statistics.md: batched; profiles/r03_p2_statistics_relaxed.md: relaxed)"""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from common import load_cfg
from hostsim_lib import HostSim
from oracle_lib import Oracle
CASES = {"rgps": ("rockgravelpebblessand.soil", 40, 250, 250), "rocksand": ("rocksand.soil", 40, 250, 100), "painted": ("painted.soil", 40, 250, 100),
         "sand": ("sand.soil", 40, 250, 250), "rgpbig": ("rockgravelpebbles_big.soil", 20, 1000, 0), "default_wet": ("default.soil", 150, 250, 0)}
engine = sys.argv[1] if len(sys.argv) > 1 else "relaxed"
names = sys.argv[2:] or list(CASES)
engine_label = None
def standing(s):                                              # water sections on top (type 0)
    end = np.cumsum(s.count.astype(np.int64)); top = end[s.count > 0] - 1
    return int((s.type[top] == 0).sum())
def rel(a, b): return f"{100.0 * (a / b - 1):+.1f} %" if b else ("0" if not a else "inf")
def fac(a, b):
    if b == 0: return "—" if a == 0 else f"{a} vs 0"
    r = a / b
    return f"×{r:.1f}" if r > 2 else f"{100.0 * (r - 1):+.0f} %"
for nm in names:
    soil, ticks, nw, nd = CASES[nm]
    cfg = load_cfg(soil); n = 256
    o = Oracle(cfg, n, n, seed=0); o.initialize(); s0 = o.snapshot()
    ctl = []
    for sd in (12345, 777):
        c = Oracle(cfg, n, n, seed=0); c.load(s0); c.L.so_srand(c.h, sd); ctl.append(c)
    h = HostSim(cfg, n, n, seed=0); h.load(s0, advance_rand=False)
    t0 = time.time()
    for t in range(ticks):
        o.tick(nw, nd, True, nd > 0)
        for c in ctl: c.tick(nw, nd, True, nd > 0)
        (h.relaxed_tick if engine == "relaxed" else h.batched_tick)(nw, nd, True, nd > 0, nthreads=6)
    se, ce = o.snapshot(), o.counters(); he = se.heights()
    def row(s, c):
        d = s.heights() - he
        return (rel(c["steps_water_all"], ce["steps_water_all"]), fac(c["nested_particles"], ce["nested_particles"]), rel(s.nsec, se.nsec),
                fac(standing(s), standing(se)), f"{d.mean():.1e}", f"{np.sqrt((d * d).mean()):.2e}")
    r = row(h.snapshot(), h.counters()); cs = [row(c.snapshot(), c.counters()) for c in ctl]
    print(f"| {soil[:-5]}, {ticks} ({nw} + {nd} / tick) | {engine_label or engine} vs reference | " + " | ".join(r) + f" ({he.std():.3f}) |", flush=True)
    print("| | reference, other streams | " + " | ".join(f"{a} / {b}" for a, b in zip(*cs)) + f" |   <!-- {time.time() - t0:.0f}s -->", flush=True)
