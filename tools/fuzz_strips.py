#!/usr/bin/env python3
"""Randomised run of the column-strip protocol on the CPU: G contexts (tests/hostsim as per-rank backend) exchanging halo
columns and migrating particles through the loop-back transport (soilmachine_amd/strips.py, the code that runs over RCCL)
against ONE context running the same (nstrips, a, b) schedule -- random map shapes, 2-4 strips, soils, counts, dry and wet
starts. usage: fuzz_strips.py SEED SECONDS"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import ctypes
from common import load_cfg
from hostsim_lib import HostSim, HostSimStripBackend, lib
from oracle_lib import Oracle
from soilmachine_amd.snapshot import compare
from strips_ref import StripGeometry, StripRank, ThreadComm, merge_owned, run_threads
CK = ("steps_water_all", "steps_water_top", "steps_wind", "floods", "nested_particles", "cascade_calls", "cascade_transfers")
L = lib(); L.hs_set_batch_strips.argtypes = [ctypes.c_int] * 3
soils = ["default.soil", "rockgravelpebblessand.soil", "painted.soil", "rocksand.soil", "sand.soil"]
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120.0)
n = 0
while time.time() < t_end:
    soil = rng.choice(soils); cfg = load_cfg(soil)
    G = rng.choice([2, 2, 3, 4])
    dx = rng.choice([x for x in (224, 256, 272, 336, 352, 400, 448, 480) if x // G >= 2 * 48 + 16]); dy = rng.choice([19, 48, 64, 100])
    seed, nw, nd, ticks = rng.randrange(1000), rng.choice([5, 80, 400]), rng.choice([0, 10, 150]), rng.choice([1, 2])
    warm = rng.choice([0, 25, 60]) if soil == "default.soil" else 0
    o = Oracle(cfg, dx, dy, seed=seed); o.initialize()
    for _ in range(warm): o.tick(250, 0, True, False)
    s0 = o.snapshot()
    desc = (soil, dx, dy, G, seed, nw, nd, ticks, warm)
    L.hs_set_batch_strips(G, 16, 48)
    try:
        h = HostSim(cfg, dx, dy, seed=seed); h.load(s0, advance_rand=True)
        for _ in range(ticks): h.batched_tick(nw, nd, True, nd > 0)
        ref, cref = h.snapshot(), h.counters()
        geom = StripGeometry(dx, G)
        hs = [HostSim(cfg, dx, dy, seed=seed) for _ in range(G)]
        for x in hs: x.load(s0, advance_rand=True)
        comms = ThreadComm.world(G)
        ranks = [StripRank(HostSimStripBackend(hs[r]), comms[r], geom) for r in range(G)]
        run_threads(ranks, lambda rk: [rk.tick(nw, nd, True, nd > 0) for _ in range(ticks)])
    finally:
        L.hs_set_batch_strips(1, 16, 48)
    got = merge_owned(geom, [x.snapshot() for x in hs], dy)
    bad = compare(got, ref)
    cs = [x.counters() for x in hs]
    if bad or {k: sum(c[k] for c in cs) for k in CK} != {k: cref[k] for k in CK} or any(c["rand_calls"] != cref["rand_calls"] for c in cs):
        print("MISMATCH", desc, bad[:3]); sys.exit(1)
    n += 1
print("ok", n, "cases")
