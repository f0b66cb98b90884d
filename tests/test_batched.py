"""The BATCHED (throughput) engine on the CPU: soil_batch.h compiled for the host (tests/hostsim, the very headers the
HIP kernels are built from) against its independent restatement in the oracle (oracle/soil_oracle.cpp so_batched_*).

What is and is not claimed (DESIGN.md "Batched engine"):
  * deterministic: the result does not depend on thread count or on the order in which a kernel's particles run;
  * bit-exact against the restated schedule, including lakes, floods and nested particles;
  * every step is isolated: no cell is touched outside the tiles the acting particle holds (the oracle checks every
    access of its step functions against the reservation);
  * with a margin as large as the map the schedule degenerates to the reference's particle-after-particle order and
    reproduces the REFERENCE bit for bit (phases without nested particles);
  * against the reference in general it is parity tier P2 only (statistical) -- see test_batched_statistics.
"""
import ctypes

import numpy as np
import pytest

from common import golden_snapshot, load_cfg
from hostsim_lib import HostSim, lib
from oracle_lib import Oracle
from soilmachine_amd.snapshot import compare

KEYS = ["steps_water_top", "steps_water_all", "steps_wind", "nested_particles", "floods", "cascade_calls",
        "cascade_transfers", "wcascade_calls", "rand_calls", "pool_free"]


def set_dilate(d):
    L = lib()
    L.hs_set_batch_dilate.argtypes = [ctypes.c_int]
    L.hs_set_batch_dilate(d)


def set_fine_above(n, oracle=None):
    """schedule knob: water reserves single cells while more than n particles run (default: never)"""
    L = lib()
    L.hs_set_batch_cell_above.argtypes = [ctypes.c_uint32]
    L.hs_set_batch_cell_above(n)
    if oracle is not None:
        oracle.L.so_batched_set_cell_above.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        oracle.L.so_batched_set_cell_above(oracle.h, n)


def pair(soil, case, tick0, dilate=0):
    cfg = load_cfg(soil)
    g = golden_snapshot(case, tick0)
    set_dilate(dilate)
    h = HostSim(cfg, g.dimx, g.dimy, seed=0)
    h.load(g, advance_rand=True)
    h.batch_stats()                                           # (the host-sim's counters are process-global: read = reset)
    o = Oracle(cfg, g.dimx, g.dimy, seed=0)
    o.load(g)
    o.batched_set_dilate(dilate)
    return h, o


CASES = [
    # soil, golden case, start tick, nwater, nwind, wind?, ticks, dilate, host threads, shuffle seed
    ("rockgravelpebblessand.soil", "rgps64", 0, 100, 50, True, 3, 0, 1, 0),
    ("rockgravelpebblessand.soil", "rgps64", 3, 100, 50, True, 2, 1, 4, 7),
    ("default.soil", "default64", 20, 250, 0, False, 8, 0, 1, 0),          # standing water: floods, nested particles, wet grid pass
    ("default.soil", "default64", 20, 250, 0, False, 4, 0, 4, 3),
    ("default.soil", "default64s7", 40, 150, 0, False, 5, 0, 1, 5),
    ("painted.soil", "painted64", 5, 100, 50, True, 3, 0, 3, 0),
    ("rocksand.soil", "rocksand48x80", 5, 60, 40, True, 3, 0, 3, 11),      # non-square map, tiles cut by the border
]


@pytest.mark.parametrize("soil,case,t0,nw,nd,wind,ticks,dilate,threads,shuffle", CASES)
def test_device_headers_equal_restated_schedule(soil, case, t0, nw, nd, wind, ticks, dilate, threads, shuffle):
    h, o = pair(soil, case, t0, dilate)
    try:
        for t in range(ticks):
            h.batched_tick(nw, nd, True, wind, nthreads=threads, shuffle=shuffle)
            o.batched_tick(nw, nd, True, wind)
            bad = compare(h.snapshot(), o.snapshot())
            assert not bad, f"{case} tick {t}: {bad}"
        ch, co = h.counters(), o.counters()
        assert {k: ch[k] for k in KEYS} == {k: co[k] for k in KEYS}
        st = o.batched_stats()
        assert st["guard_violations"] == 0                  # isolation: nothing touched outside the held tiles
        hs = h.batch_stats()
        assert hs["epochs"] == st["epochs"] and hs["generations"] == st["generations"] and hs["children_lost"] == 0
        if case == "default64":
            assert co["floods"] > 500 and co["nested_particles"] > 1000   # the wet regime really was exercised
    finally:
        set_dilate(0)
        h.batch_stats()


@pytest.mark.parametrize("soil,case,t0,nw,nd,wind,fine", [
    ("default.soil", "default64", 20, 250, 0, False, 100),                 # per-cell reservations, then the switch to tiles
    ("rockgravelpebblessand.soil", "rgps64", 0, 300, 4300, True, 120),     # > 4096 wind particles: the wind schedule regrains too
])
def test_schedule_changes_granularity_mid_generation(soil, case, t0, nw, nd, wind, fine):
    """While many particles run the schedule may reserve single cells / smaller tiles; when their number drops it
    switches granularity: planes are cleared and every running particle re-issues its reservation."""
    h, o = pair(soil, case, t0)
    try:
        set_fine_above(fine, o)
        for t in range(2):
            h.batched_tick(nw, nd, True, wind, nthreads=2, shuffle=4)
            o.batched_tick(nw, nd, True, wind)
            bad = compare(h.snapshot(), o.snapshot())
            assert not bad, f"{case} tick {t}: {bad}"
        assert o.batched_stats()["guard_violations"] == 0
        ch, co = h.counters(), o.counters()
        assert {k: ch[k] for k in KEYS} == {k: co[k] for k in KEYS}
    finally:
        set_fine_above(0xFFFFFFFF)
        h.batch_stats()


def test_result_is_independent_of_threads_and_order():
    cfg = load_cfg("default.soil")
    g = golden_snapshot("default64", 20)
    snaps = []
    for threads, shuffle in [(1, 0), (1, 9), (8, 0), (5, 123)]:
        h = HostSim(cfg, 64, 64, seed=0)
        h.load(g, advance_rand=True)
        for _ in range(3):
            h.batched_tick(250, 0, True, False, nthreads=threads, shuffle=shuffle)
        snaps.append((h.snapshot(), h.counters()))
    for s, c in snaps[1:]:
        assert not compare(snaps[0][0], s)
        assert c == snaps[0][1]


@pytest.mark.parametrize("soil,case,nw,nd", [("rockgravelpebblessand.soil", "rgps64", 100, 50), ("rocksand.soil", "rocksand48x80", 60, 40)])
def test_fully_serialised_schedule_reproduces_the_reference(soil, case, nw, nd):
    """Margin >= map: the lowest slot always wins everything, i.e. particles run one after the other in spawn order =
    the reference's order. Golden snapshots of the REFERENCE must come out bit for bit (these phases have no floods)."""
    cfg = load_cfg(soil)
    g0 = golden_snapshot(case, 0)
    try:
        set_dilate(64)
        h = HostSim(cfg, g0.dimx, g0.dimy, seed=0 if case == "rgps64" else 3)
        h.load(g0, advance_rand=True)
        t_end = 3 if case == "rgps64" else 5
        for _ in range(t_end):
            h.batched_tick(nw, nd, True, True)
        assert h.counters()["floods"] == 0
        bad = compare(h.snapshot(), golden_snapshot(case, t_end))
        assert not bad, bad
    finally:
        set_dilate(0)
        h.batch_stats()


def test_batched_statistics_against_the_exact_engine():
    """Parity tier P2: the batched schedule is NOT the reference's order. What it must keep: every
    structural invariant of a Layermap, the amount of work (steps within a few percent), and heights that stay as
    close to the exact run as the exact run stays to itself under a 1-ulp perturbation (the reference's own noise
    floor: RMSE 1.0e-3 after 20 ticks at 256^2)."""
    from test_gpu_fullsize import column_invariants
    cfg = load_cfg("rockgravelpebblessand.soil")
    n = 128
    o = Oracle(cfg, n, n, seed=0)
    o.initialize()
    s0 = o.snapshot()
    ex = Oracle(cfg, n, n, seed=0); ex.load(s0)
    ba = Oracle(cfg, n, n, seed=0); ba.load(s0)
    pe = Oracle(cfg, n, n, seed=0)                           # the exact engine on a map perturbed by 1e-12 (chaos floor)
    sp = o.snapshot(); sp.size = sp.size * (1.0 + 1e-12); pe.load(sp)
    ticks, nw, nd = 8, 200, 60
    for _ in range(ticks):
        ex.tick(nw, nd); ba.batched_tick(nw, nd); pe.tick(nw, nd)
    se, sb, sq = ex.snapshot(), ba.snapshot(), pe.snapshot()
    column_invariants(sb)
    he, hb, hq = se.heights(), sb.heights(), sq.heights()
    rmse_b = float(np.sqrt(np.mean((he - hb) ** 2)))
    rmse_floor = float(np.sqrt(np.mean((he - hq) ** 2)))
    ce, cb = ex.counters(), ba.counters()
    assert abs(cb["steps_water_all"] - ce["steps_water_all"]) < 0.05 * ce["steps_water_all"]
    assert abs(cb["steps_wind"] - ce["steps_wind"]) < 0.25 * ce["steps_wind"] + 500
    assert abs(hb.mean() - he.mean()) < 2e-4 * he.mean()
    assert abs(sb.nsec - se.nsec) < 0.05 * se.nsec
    assert rmse_b < max(5.0 * rmse_floor, 5e-3), (rmse_b, rmse_floor)
    assert ba.batched_stats()["guard_violations"] == 0


def test_200_ticks_into_the_wet_regime_equal_restated_schedule():
    """Depth on the CPU: 128^2 default.soil, 250 particles per tick for 200 ticks -- lakes, > 2.5 * 10^5 nested particles,
    ~2.5 * 10^5 epochs; the device sources on host threads == the restated schedule on the full state every 100 ticks."""
    cfg = load_cfg("default.soil")
    n = 128
    o = Oracle(cfg, n, n, seed=0)
    o.initialize()
    h = HostSim(cfg, n, n, seed=0)
    h.load(o.snapshot())
    for block in range(2):
        for _ in range(100):
            h.batched_tick(250, 0, True, False, nthreads=4, shuffle=block + 1)
            o.batched_tick(250, 0, True, False)
        bad = compare(h.snapshot(), o.snapshot())
        assert not bad, f"after {100 * (block + 1)} ticks: {bad}"
    ch, co = h.counters(), o.counters()
    assert {k: ch[k] for k in KEYS} == {k: co[k] for k in KEYS}
    assert co["nested_particles"] > 2.5 * 10**5 and o.batched_stats()["guard_violations"] == 0


def test_statistics_in_the_wet_regime_track_the_reference():
    """Tier P2 where it is hardest: 256^2 default.soil, 150 ticks (lakes, ~2 * 10^5 nested particles). The batched schedule
    must reproduce the reference's AMOUNTS -- floods, nested particles, standing water, total height -- not just look
    plausible. (Round 2 tried running floods as a sub-phase after every 32-epoch chunk: 18 % faster on the device, but a
    flood is a feedback -- the puddle it leaves ends the particles that arrive next -- and delaying it cost 17 % of the
    floods and 25 % of the nested particles, +21 % standing water; this test is what such a change has to pass.)
    The amounts are held to the ratio bounds of rounds 2-4 AND to the reference's own spread (six control streams: the same terrain, another
    rand() stream): see the loop below."""
    from concurrent.futures import ThreadPoolExecutor
    cfg = load_cfg("default.soil")
    n = 256
    a = Oracle(cfg, n, n, seed=0); a.initialize()
    s0 = a.snapshot()
    b = Oracle(cfg, n, n, seed=0); b.initialize()
    ctl = []
    for k in range(6):
        c = Oracle(cfg, n, n, seed=0); c.load(s0); c.L.so_srand(c.h, 1000 + 7919 * k); ctl.append(c)
    def run(x, batched=False):
        for _ in range(150):
            (x.batched_tick if batched else x.tick)(250, 0, True, False)
    with ThreadPoolExecutor(max_workers=8) as ex:          # (ctypes releases the GIL)
        futs = [ex.submit(run, a), ex.submit(run, b, True)] + [ex.submit(run, c) for c in ctl]
        for f in futs: f.result()
    def figs(x):
        c, s = x.counters(), x.snapshot()
        return {"floods": c["floods"], "nested": c["nested_particles"], "water": float(s.size[s.type == 0].sum()), "sections": int(s.count.sum()),
                "sumh": float(np.asarray(x.heights()).sum())}
    fa, fb, fc = figs(a), figs(b), [figs(c) for c in ctl]
    assert fa["floods"] > 50000 and fa["nested"] > 250000
    # Round 6: the RATIO bounds of rounds 2-4 are back -- floods +-3 %, nested particles +-5 %, standing water +-12 %,
    # sections +-1 % against the mean of the reference's seven streams (SEED 0 + six controls) -- each widened only where the reference's OWN
    # stream-to-stream spread is wider than the bound (1.5 standard deviations of those streams: nobody can be asked to sit closer to the
    # reference than the reference sits to itself; its floods spread by 7 %), plus |z| < 2.5 on every figure. Round 5's |z| < 3.5 admitted a
    # 12-16 % deficit of nested particles; with eight water generations (the default since round 6) the exclusive schedule is at +4 %.
    BOUND = {"floods": 0.03, "nested": 0.05, "water": 0.12, "sections": 0.01, "sumh": 1e-4}
    for k in fa:
        v = np.array([f[k] for f in fc] + [fa[k]], float)
        mu, sd = v.mean(), v.std(ddof=1)
        rel, z = fb[k] / mu - 1.0, (fb[k] - mu) / sd
        assert abs(rel) <= max(BOUND[k], 1.5 * sd / mu), (k, rel, BOUND[k], 1.5 * sd / mu, fb[k], fa[k], v.min(), v.max())
        assert abs(z) < 2.5, (k, z, fb[k], fa[k], v.min(), v.max())
    ha, hb = np.asarray(a.heights()), np.asarray(b.heights())
    rm = max(float(np.sqrt(((np.asarray(c.heights()) - ha) ** 2).mean())) for c in ctl)
    assert float(np.sqrt(((ha - hb) ** 2).mean())) < 1.25 * rm          # no further from SEED 0 than another stream of the reference is


def test_random_configurations_host_sim_equals_restatement():
    """A seeded sweep over what the fixed cases do not vary together: map shapes that are not multiples of any tile (17x19 ...
    130x96), six soil files, particle counts from 1 up, margins, single-cell reservations, 1-3 column strips, thread counts and
    execution orders. (The same loop ran ~10^3 cases while the engine was written; here 30 of them.)"""
    import random
    from oracle_lib import lib as olib
    L = lib()
    L.hs_set_batch_strips.argtypes = [ctypes.c_int] * 3
    OL = olib()
    OL.so_batched_set_strips.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3
    soils = ["default.soil", "rockgravelpebblessand.soil", "painted.soil", "rocksand.soil", "sand.soil", "bigbutte.soil"]
    rng = random.Random(20260926)
    try:
        for case in range(30):
            cfg = load_cfg(rng.choice(soils))
            dx, dy = rng.choice([17, 33, 48, 64, 80, 97, 130, 224]), rng.choice([19, 32, 50, 64, 96])
            seed, nw, nd = rng.randrange(1000), rng.choice([1, 7, 60, 250]), rng.choice([0, 5, 40, 120])
            ticks, dil, fine = rng.choice([1, 2, 3]), rng.choice([0, 0, 1, 2]), rng.choice([0xFFFFFFFF, 0xFFFFFFFF, 30])
            strips = rng.choice([1, 2, 3]) if dx >= 130 else 1
            if strips > 1 and dx // strips < 2 * 48 + 16:
                strips = 1
            if strips > 1 and dil > 1:                          # (a margin of 2 tiles does not fit the seam zones: the C-ABI rejects it)
                dil = 1
            o = Oracle(cfg, dx, dy, seed=seed)
            o.initialize()
            h = HostSim(cfg, dx, dy, seed=seed)
            h.load(o.snapshot())
            set_dilate(dil); o.batched_set_dilate(dil)
            set_fine_above(fine, o)
            L.hs_set_batch_strips(strips, 16, 48); OL.so_batched_set_strips(o.h, strips, 16, 48)
            for t in range(ticks):
                h.batched_tick(nw, nd, True, nd > 0, nthreads=rng.choice([1, 3]), shuffle=rng.randrange(5))
                o.batched_tick(nw, nd, True, nd > 0)
                bad = compare(h.snapshot(), o.snapshot())
                assert not bad, f"case {case} ({dx}x{dy}, seed {seed}, {nw}+{nd}, margin {dil}, strips {strips}) tick {t}: {bad}"
            assert o.batched_stats()["guard_violations"] == 0
    finally:
        set_dilate(0); set_fine_above(0xFFFFFFFF); L.hs_set_batch_strips(1, 16, 48)
