#!/usr/bin/env python3
"""Randomised differential run of the batched engine: device sources on host threads (tests/hostsim) against the CPU restatement
(oracle/soil_oracle.cpp) over random map shapes, soils, counts, margins, granularities, strips. usage: fuzz_batched.py SEED SECONDS"""
import sys, time, random
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from common import load_cfg
from hostsim_lib import HostSim, lib
from oracle_lib import Oracle
from soilmachine_amd.snapshot import compare
soils = ["default.soil","rockgravelpebblessand.soil","painted.soil","rocksand.soil","sand.soil","bigbutte.soil"]
rng = random.Random(int(sys.argv[1]) if len(sys.argv)>1 else 1)
L = lib()
import ctypes
L.hs_set_batch_dilate.argtypes=[ctypes.c_int]; L.hs_set_batch_cell_above.argtypes=[ctypes.c_uint32]; L.hs_set_batch_strips.argtypes=[ctypes.c_int]*3
from oracle_lib import lib as olib
OL=olib(); OL.so_batched_set_cell_above.argtypes=[ctypes.c_void_p, ctypes.c_uint32]; OL.so_batched_set_strips.argtypes=[ctypes.c_void_p]+[ctypes.c_int]*3
t_end = time.time() + float(sys.argv[2]) if len(sys.argv)>2 else time.time()+120
n=0
while time.time() < t_end:
    soil = rng.choice(soils); cfg = load_cfg(soil)
    dx = rng.choice([17, 33, 48, 64, 80, 97, 130, 224, 260, 352]); dy = rng.choice([19, 32, 50, 64, 96])
    seed = rng.randrange(1000); nw = rng.choice([1, 7, 60, 250]); nd = rng.choice([0, 5, 40, 120])
    ticks = rng.choice([1, 2, 4]); dil = rng.choice([0, 0, 1, 2]); cell_above = rng.choice([0xFFFFFFFF, 0xFFFFFFFF, 30])
    strips = rng.choice([1, 2, 3]) if dx >= 224 else 1
    if strips > 1 and dx // strips < 2*48+16: strips = 1
    if strips > 1 and dil > 1: dil = 1                          # (wider margins do not fit the seam zones: rejected at the C-ABI)
    o = Oracle(cfg, dx, dy, seed=seed); o.initialize()
    h = HostSim(cfg, dx, dy, seed=seed); h.load(o.snapshot())
    L.hs_set_batch_dilate(dil); o.batched_set_dilate(dil)
    L.hs_set_batch_cell_above(cell_above); o.L.so_batched_set_cell_above(o.h, cell_above)
    L.hs_set_batch_strips(strips, 16, 48); o.L.so_batched_set_strips(o.h, strips, 16, 48)
    desc = (soil, dx, dy, seed, nw, nd, ticks, dil, cell_above, strips)
    try:
        for t in range(ticks):
            h.batched_tick(nw, nd, True, nd > 0, nthreads=rng.choice([1,3]), shuffle=rng.randrange(5))
            o.batched_tick(nw, nd, True, nd > 0)
            bad = compare(h.snapshot(), o.snapshot())
            if bad: print("MISMATCH", desc, t, bad[:3]); sys.exit(1)
        if o.batched_stats()["guard_violations"]: print("GUARD", desc, o.batched_stats()); sys.exit(1)
    finally:
        L.hs_set_batch_dilate(0); L.hs_set_batch_cell_above(0xFFFFFFFF); L.hs_set_batch_strips(1, 16, 48)
    n+=1
print("ok", n, "cases")
