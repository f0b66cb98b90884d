#!/usr/bin/env python3
"""Randomised differential run of the EXACT speculative engine: device sources on host threads (tests/hostsim: stamps, claims,
forwarding, kill closure, roll-back, suspend/resume, wave loop) against the CPU restatement in reference order, over random
map shapes, soils, particle counts, engine modes, thread counts and interleavings. usage: fuzz_exact.py SEED SECONDS"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from common import load_cfg
from hostsim_lib import HostSim
from oracle_lib import Oracle
from soilmachine_amd.snapshot import compare
soils = ["default.soil", "rockgravelpebblessand.soil", "painted.soil", "rocksand.soil", "sand.soil", "bigbutte.soil", "rockgravelpebbles.soil"]
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120.0)
n = 0
while time.time() < t_end:
    soil = rng.choice(soils); cfg = load_cfg(soil)
    dx, dy = rng.choice([17, 33, 48, 64, 80, 97, 130]), rng.choice([19, 32, 50, 64, 96])
    seed, nw, nd = rng.randrange(1000), rng.choice([1, 7, 60, 250]), rng.choice([0, 5, 40, 120])
    ticks, mode, team = rng.choice([1, 2, 4, 8]), rng.choice([0, 1, 2, 3, 7]), rng.choice([1, 5])
    warm = rng.choice([0, 0, 15, 40]) if soil == "default.soil" else 0      # default.soil: start some cases with standing water
    o = Oracle(cfg, dx, dy, seed=seed); o.initialize()
    for _ in range(warm): o.tick(nw, 0, True, False)
    h = HostSim(cfg, dx, dy, seed=seed); h.load(o.snapshot())
    HostSim.set_boundary_threads(team)
    sub, slots = rng.choice([4096, 4096, 97, 33]), rng.choice([1024, 1024, 160, 128])   # round 4: sub-phase size / nested-slot budget (small values force the hand-over)
    h.L.hs_set_spec_sub(sub); h.L.hs_set_spec_maxnest(slots)
    desc = (soil, dx, dy, seed, nw, nd, ticks, mode, team, warm, sub, slots)
    try:
        for t in range(ticks):
            h.spec_tick(nw, nd, True, nd > 0, nthreads=rng.choice([2, 8]), scout=rng.choice([0, 1]) | (mode << 4), shuffle=rng.randrange(50))
            o.tick(nw, nd, True, nd > 0)
            bad = compare(h.snapshot(), o.snapshot())
            if bad:
                print("MISMATCH", desc, "tick", t, bad[:3]); sys.exit(1)
    finally:
        HostSim.set_boundary_threads(1); h.L.hs_set_spec_sub(4096); h.L.hs_set_spec_maxnest(1024)
    n += 1
print("ok", n, "cases")
