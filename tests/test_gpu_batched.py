"""The BATCHED (throughput) engine on the MI355X, through the C-ABI: bit-exact against its CPU restatement
(oracle/soil_oracle.cpp so_batched_*), deterministic, structurally sound at full size. Parity against the REFERENCE is
tier P2 (statistical) by design -- see tests/test_batched.py and DESIGN.md; the one exception is tested here too: with a
margin as large as the map the engine runs the particles in the reference's order and reproduces the reference."""
import numpy as np
import pytest

from common import golden_snapshot, load_cfg
from oracle_lib import Oracle
from soilmachine_amd import capi
from soilmachine_amd.machine import Layermap, SoilMachine
from soilmachine_amd.snapshot import compare
from test_gpu_fullsize import column_invariants

pytestmark = pytest.mark.gpu
KEYS = ["steps_water_top", "steps_water_all", "steps_wind", "nested_particles", "floods", "cascade_calls",
        "cascade_transfers", "wcascade_calls", "rand_calls", "pool_free"]


def machine_from(cfg, g, nw, nd, wind, dilate=0):
    m = Layermap(cfg, g.dimx, g.dimy, seed=0, initialize=False, engine=capi.ENGINE_BATCHED)
    m.load(g, rand_seed=0)
    m.set_batch_dilate(dilate)
    sm = SoilMachine.__new__(SoilMachine)
    sm.cfg, sm.map, sm.nwater, sm.nwind, sm.dowater, sm.dowind = cfg, m, nw, nd, True, wind
    return sm


@pytest.mark.parametrize("soil,case,t0,nw,nd,wind,ticks,dilate", [
    ("rockgravelpebblessand.soil", "rgps64", 0, 100, 50, True, 3, 0),
    ("rockgravelpebblessand.soil", "rgps64", 3, 100, 50, True, 2, 1),
    ("default.soil", "default64", 20, 250, 0, False, 8, 0),          # standing water: floods, nested particles, wet grid pass
    ("default.soil", "default64s7", 40, 150, 0, False, 5, 0),
    ("painted.soil", "painted64", 5, 100, 50, True, 3, 0),
    ("rocksand.soil", "rocksand48x80", 5, 60, 40, True, 3, 0),
])
def test_device_equals_restated_schedule(soil, case, t0, nw, nd, wind, ticks, dilate):
    cfg = load_cfg(soil)
    g = golden_snapshot(case, t0)
    sm = machine_from(cfg, g, nw, nd, wind, dilate)
    o = Oracle(cfg, g.dimx, g.dimy, seed=0)
    o.load(g)
    o.batched_set_dilate(dilate)
    for t in range(ticks):
        sm.tick(1, sync=True)
        o.batched_tick(nw, nd, True, wind)
        bad = compare(sm.map.snapshot(), o.snapshot())
        assert not bad, f"{case} tick {t}: {bad}"
    cd, co = sm.map.counters(), o.counters()
    assert {k: cd[k] for k in KEYS} == {k: co[k] for k in KEYS}
    st, bs = o.batched_stats(), sm.map.batch_stats()
    assert st["guard_violations"] == 0
    assert (bs["epochs"], bs["generations"], bs["children_lost"]) == (st["epochs"], st["generations"], 0)
    if case == "default64":
        assert co["floods"] > 500 and co["nested_particles"] > 1000


def test_512_multilayer_two_ticks_equal_restated_schedule():
    """a map big enough for thousands of concurrent particles per epoch and for the wind schedule to change granularity"""
    cfg = load_cfg("rockgravelpebblessand.soil")
    n, nw, nd = 512, 3000, 4500
    sm = SoilMachine(cfg, n, seed=0, nwater=nw, nwind=nd, engine=capi.ENGINE_BATCHED)
    o = Oracle(cfg, n, n, seed=0)
    o.initialize()
    for _ in range(2):
        sm.tick(1, sync=True)
        o.batched_tick(nw, nd)
    assert not compare(sm.map.snapshot(), o.snapshot())
    cd, co = sm.map.counters(), o.counters()
    assert {k: cd[k] for k in KEYS} == {k: co[k] for k in KEYS}
    assert o.batched_stats()["guard_violations"] == 0


@pytest.mark.parametrize("soil,case,seed,nw,nd,t_end", [("rockgravelpebblessand.soil", "rgps64", 0, 100, 50, 3),
                                                        ("rocksand.soil", "rocksand48x80", 3, 60, 40, 5)])
def test_fully_serialised_schedule_reproduces_the_reference(soil, case, seed, nw, nd, t_end):
    cfg = load_cfg(soil)
    g0 = golden_snapshot(case, 0)
    m = Layermap(cfg, g0.dimx, g0.dimy, seed=seed, initialize=False, engine=capi.ENGINE_BATCHED)
    m.load(g0, rand_seed=seed)
    m.set_batch_dilate(64)
    sm = SoilMachine.__new__(SoilMachine)
    sm.cfg, sm.map, sm.nwater, sm.nwind, sm.dowater, sm.dowind = cfg, m, nw, nd, True, True
    sm.tick(t_end, sync=True)
    assert m.counters()["floods"] == 0
    bad = compare(m.snapshot(), golden_snapshot(case, t_end))
    assert not bad, bad


@pytest.mark.slow
def test_4096_batched_determinism_invariants_and_statistics():
    """BASELINE configs[3] with area-scaled particle counts: two contexts land on the same bits; the state keeps every
    Layermap invariant; mean height and section count stay close to the exact engine's (same seed, same counts)."""
    cfg = load_cfg("rockgravelpebblessand.soil")
    n, nw, nd, ticks = 4096, 16000, 4000, 2
    a = SoilMachine(cfg, n, seed=0, nwater=nw, nwind=nd, pool=3 * n * n, engine=capi.ENGINE_BATCHED)
    b = SoilMachine(cfg, n, seed=0, nwater=nw, nwind=nd, pool=3 * n * n, engine=capi.ENGINE_BATCHED)
    a.tick(ticks, sync=True); b.tick(ticks, sync=True)
    da, db = a.map.digest(), b.map.digest()
    assert da == db
    assert a.map.counters() == b.map.counters()
    sa = a.map.snapshot()
    del b
    column_invariants(sa)
    ca = a.map.counters()
    assert ca["pool_overflow"] == 0 and a.map.batch_stats()["children_lost"] == 0
    ex = SoilMachine(cfg, n, seed=0, nwater=nw, nwind=nd, pool=3 * n * n, engine=capi.ENGINE_SPECULATIVE)
    ex.tick(ticks, sync=True)
    ce, de = ex.map.counters(), ex.map.digest()
    assert abs(ca["steps_water_all"] - ce["steps_water_all"]) < 0.03 * ce["steps_water_all"]
    assert abs(ca["steps_wind"] - ce["steps_wind"]) < 0.05 * ce["steps_wind"]
    assert abs(da["sumh"] - de["sumh"]) < 1e-5 * de["sumh"]
    assert abs(da["nsec"] - de["nsec"]) < 0.01 * de["nsec"]


@pytest.mark.slow
def test_256_default_1000_ticks_in_the_wet_regime_equal_restated_schedule():
    """Depth: 1000 ticks of BASELINE configs[0]'s map (256^2 default.soil, 250 particles per tick, hydraulic only) -- lakes from
    tick ~60 on, > 10^6 nested particles, > 10^6 epochs -- device == restated schedule on the full state, every 250 ticks."""
    cfg = load_cfg("default.soil")
    n, nw = 256, 250
    sm = SoilMachine(cfg, n, seed=0, nwater=nw, nwind=0, dowind=False, engine=capi.ENGINE_BATCHED)
    o = Oracle(cfg, n, n, seed=0)
    o.initialize()
    for block in range(4):
        sm.tick(250, sync=True)
        for _ in range(250):
            o.batched_tick(nw, 0, True, False)
        bad = compare(sm.map.snapshot(), o.snapshot())
        assert not bad, f"after {250 * (block + 1)} ticks: {bad}"
    cd, co = sm.map.counters(), o.counters()
    assert {k: cd[k] for k in KEYS} == {k: co[k] for k in KEYS}
    st, bs = o.batched_stats(), sm.map.batch_stats()
    assert st["guard_violations"] == 0 and (bs["epochs"], bs["generations"], bs["children_lost"]) == (st["epochs"], st["generations"], 0)
    assert co["nested_particles"] > 10**6 and cd["pool_overflow"] == 0

