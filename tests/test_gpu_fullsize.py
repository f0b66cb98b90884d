"""Full-size checks (BASELINE.json sizes): bit-exact against the oracle where the oracle finishes in seconds,
and size-independent structural properties of the Layermap elsewhere."""
import numpy as np
import pytest

from common import load_cfg
from oracle_lib import Oracle
from soilmachine_amd.machine import Layermap, SoilMachine
from soilmachine_amd.snapshot import compare

pytestmark = pytest.mark.gpu


def column_invariants(s):
    """Invariants every reference state satisfies (layermap.h:230-339): sizes > 0, floors are the running sum
    of the sections below (bitwise), water ("Air", type 0) only ever on top of a column."""
    end = np.cumsum(s.count.astype(np.int64))
    start = end - s.count
    assert (s.size > 0).all()
    first = np.zeros(s.nsec, bool)
    first[start[s.count > 0]] = True
    assert (s.floor[first] == 0).all()
    nxt = ~first
    idx = np.nonzero(nxt)[0]
    assert np.array_equal(s.floor[idx], s.floor[idx - 1] + s.size[idx - 1])
    is_top = np.zeros(s.nsec, bool)
    is_top[end[s.count > 0] - 1] = True
    assert not ((s.type == 0) & ~is_top).any()
    adj_same = (s.type[idx] == s.type[idx - 1])
    assert not adj_same.any()                          # equal neighbours always merge (layermap.h:249-253)


@pytest.mark.slow
def test_1024_default_5_ticks_reference_digest():
    from common import digests
    d = digests()["default1024_t5"]
    cfg = load_cfg("default.soil")
    sm = SoilMachine(cfg, 1024, seed=0, nwater=d["nwater"], nwind=0, dowind=False)
    sm.tick(5, sync=True)
    s = sm.map.snapshot()
    g = s.digest()
    assert (g["nsec"], g["typehash"], g["sumh"], s.rand_calls) == (d["nsec"], d["typehash"], d["sumh"], d["rand_calls"])


@pytest.mark.slow
def test_4096_multilayer_one_tick_bit_exact_and_invariants():
    cfg = load_cfg("rockgravelpebblessand.soil")
    n = 4096
    from soilmachine_amd import capi
    sm = SoilMachine(cfg, n, seed=0, pool=3 * n * n, engine=capi.ENGINE_SPECULATIVE)
    o = Oracle(cfg, n, n, seed=0, pool=3 * n * n)
    o.initialize()
    s0 = sm.map.snapshot()
    assert not compare(s0, o.snapshot())               # 16.8 M cells x 2 noise layers, bit-exact
    column_invariants(s0)
    sm.tick(1, sync=True)
    o.tick(cfg.NWATER, cfg.NWIND, True, True)
    s1 = sm.map.snapshot()
    assert not compare(s1, o.snapshot())
    column_invariants(s1)
    # the grid pass on a dry map must be an exact no-op (classification finds no active cell)
    before = sm.map.counters()["grid_active_cells"]
    sm.grid_pass(); sm.map.sync()
    assert sm.map.counters()["grid_active_cells"] == before
    assert not compare(sm.map.snapshot(), s1)


def test_determinism_two_contexts():
    cfg = load_cfg("rockgravelpebblessand.soil")
    a = SoilMachine(cfg, 512, seed=9)
    b = SoilMachine(cfg, 512, seed=9)
    a.tick(2, sync=True); b.tick(2, sync=True)
    sa = a.map.snapshot()
    assert not compare(sa, b.map.snapshot())
    column_invariants(sa)
