"""GPU check of the batched engine: device == host-sim (same headers on the CPU), bit for bit; then bench lines."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import load_cfg, golden_snapshot
from hostsim_lib import HostSim
from soilmachine_amd import capi
from soilmachine_amd.machine import Layermap, SoilMachine
from soilmachine_amd.snapshot import compare

def check(cfgname, case, tick0, nw, nd, wind, ticks, dilate=0):
    cfg = load_cfg(cfgname); g = golden_snapshot(case, tick0)
    h = HostSim(cfg, g.dimx, g.dimy, seed=0); h.load(g, advance_rand=True)
    import ctypes
    h.L.hs_set_batch_dilate.argtypes = [ctypes.c_int]; h.L.hs_set_batch_dilate(dilate)
    m = Layermap(cfg, g.dimx, g.dimy, seed=0, initialize=False, engine=capi.ENGINE_BATCHED); m.load(g, rand_seed=0)
    m.set_batch_dilate(dilate)
    sm = SoilMachine.__new__(SoilMachine); sm.cfg, sm.map, sm.nwater, sm.nwind, sm.dowater, sm.dowind = cfg, m, nw, nd, True, wind
    for t in range(ticks):
        h.batched_tick(nw, nd, True, wind); sm.tick(1, sync=True)
        bad = compare(m.snapshot(), h.snapshot())
        if bad: print("MISMATCH", case, "tick", t, bad[:3]); return False
    ch, cd = h.counters(), m.counters()
    keys = ["steps_water_top","steps_water_all","steps_wind","nested_particles","floods","cascade_calls","cascade_transfers","wcascade_calls","grid_active_cells","rand_calls","pool_free"]
    diff = {k: (ch[k], cd[k]) for k in keys if ch[k] != cd[k]}
    print("OK" if not diff else "COUNTERS DIFFER", case, tick0, "dilate", dilate, {k: cd[k] for k in keys[:5]}, diff, m.batch_stats(), flush=True)
    return not diff

ok = True
ok &= check("rockgravelpebblessand.soil", "rgps64", 0, 100, 50, True, 3)
ok &= check("rockgravelpebblessand.soil", "rgps64", 3, 100, 50, True, 3, dilate=1)
ok &= check("default.soil", "default64", 20, 250, 0, False, 6)        # a state with lakes: floods, nested particles, wet grid pass
ok &= check("default.soil", "default64s7", 40, 150, 0, False, 6)
ok &= check("painted.soil", "painted64", 5, 100, 50, True, 3)
ok &= check("rocksand.soil", "rocksand48x80", 5, 60, 40, True, 3)
print("ALL_OK" if ok else "FAILED", flush=True)
