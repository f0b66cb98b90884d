#!/usr/bin/env python3
"""Randomised differential run of the RELAXED schedule: device sources on host threads (tests/hostsim) against the independent
restatement (oracle/soil_oracle.cpp r_water_generation) over random map shapes, soils, counts, tick counts, 1-3 column strips, starting either from the
initial terrain or from a map that already carries lakes. usage: fuzz_relaxed.py SEED SECONDS"""
import sys, time, random
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from common import load_cfg
import ctypes
from hostsim_lib import HostSim, lib
from oracle_lib import Oracle, lib as olib
L = lib(); L.hs_set_batch_strips.argtypes = [ctypes.c_int] * 3
olib().so_batched_set_strips.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3
from soilmachine_amd.snapshot import compare
soils = ["default.soil", "rockgravelpebblessand.soil", "painted.soil", "rocksand.soil", "sand.soil", "bigbutte.soil"]
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120)
KEYS = ["steps_water_top", "steps_water_all", "steps_wind", "nested_particles", "floods", "cascade_calls", "cascade_transfers", "wcascade_calls", "rand_calls"]
n = floods = nstg = 0
while time.time() < t_end:
    soil = rng.choice(soils); cfg = load_cfg(soil)
    dx = rng.choice([5, 17, 33, 48, 64, 80, 97, 130, 224, 260, 352]); dy = rng.choice([4, 19, 32, 50, 64, 96])
    strips = rng.choice([1, 2, 3]) if dx >= 224 else 1
    if strips > 1 and dx // strips < 2 * 48 + 16: strips = 1
    seed = rng.randrange(1000); nw = rng.choice([1, 7, 60, 250, 600]); nd = rng.choice([0, 0, 5, 40])
    ticks = rng.choice([1, 2, 4, 8]); wet = rng.random() < 0.4
    o = Oracle(cfg, dx, dy, seed=seed); o.initialize()
    if wet:                                                   # some ticks of the reference first: standing water, saturated sections
        for _ in range(rng.choice([10, 25, 40])): o.tick(min(nw, 250), 0, True, False)
    if wet:                                                   # both sides start from the SNAPSHOT: the sticky "has held a saturation" bit of a
        s0 = o.snapshot()                                     # column is context state that a snapshot does not carry (it is re-derived on import),
        o = Oracle(cfg, dx, dy, seed=seed); o.load(s0)        # and the fixed active set of the 9-colour grid pass depends on it
    h = HostSim(cfg, dx, dy, seed=seed); h.load(o.snapshot(), advance_rand=True)
    c0 = o.counters()
    wg = rng.choice([0, 0, 1, 2, 4])                          # smx_set_water_generations (0: the engines' own default, eight since round 6, through the tick entry points)
    # staggered generations (smx_set_water_stagger, round 6) through the tick entry points: k batches born `gap` epochs apart inside ONE phase
    stg = (rng.choice([2, 3, 4, 8, 16]), rng.choice([1, 4, 16, 50, 140])) if (wg == 0 and rng.random() < 0.5) else None
    if stg:
        L.hs_set_water_generations(stg[0]); L.hs_set_water_stagger(0, stg[1])
        o.L.so_set_water_generations(o.h, stg[0]); o.L.so_set_water_stagger(o.h, stg[1])
    desc = (soil, dx, dy, seed, nw, nd, ticks, wet, strips, wg, stg)
    L.hs_set_batch_strips(strips, 16, 48); o.L.so_batched_set_strips(o.h, strips, 16, 48)
    for t in range(ticks):
        nt, sh = rng.choice([1, 3, 5]), rng.randrange(5)
        if wg == 0:
            h.relaxed_tick(nw, nd, True, nd > 0, nthreads=nt, shuffle=sh)
            o.relaxed_tick(nw, nd, True, nd > 0)
        else:                                                 # the water phase as wg consecutive generations = wg phase calls
            L.hs_set_relaxed(1); o.L.so_set_relaxed(o.h, 1)
            left = nw
            for q in range(wg):
                nq = left // (wg - q); left -= nq
                if nq: L.hs_batched_water(h.h, nq, nt, sh); o.L.so_batched_water(o.h, nq)
            L.hs_batched_grid(h.h, nt, sh); o.L.so_batched_grid(o.h)
            if nd > 0: L.hs_batched_wind(h.h, nd, nt, sh); o.L.so_batched_wind(o.h, nd)
            L.hs_set_relaxed(0); o.L.so_set_relaxed(o.h, 0)
            L.hs_freq(h.h); o.L.so_map_frequency(o.h); o.L.so_reset_frequency(o.h)
        bad = compare(h.snapshot(), o.snapshot())
        if bad: print("MISMATCH", desc, t, bad[:3]); sys.exit(1)
    L.hs_set_batch_strips(1, 16, 48); L.hs_reset_water_schedule()
    ch, co = h.counters(), o.counters()
    co = {k: co[k] - (c0[k] if k != 'rand_calls' else 0) for k in KEYS}
    if any(ch[k] != co[k] for k in KEYS): print("COUNTERS", desc, {k: (ch[k], co[k]) for k in KEYS if ch[k] != co[k]}); sys.exit(1)
    if o.batched_stats()["guard_violations"]: print("GUARD", desc, o.batched_stats()); sys.exit(1)
    n += 1; floods += co["floods"]; nstg += 1 if stg else 0
print("ok", n, "cases,", nstg, "of them staggered,", floods, "floods")
