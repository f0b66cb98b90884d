"""Parity tests proper: the hand-written HIP path (through the C-ABI of include/soilmx.h) against the CPU
oracle and the committed golden fixtures, on a real MI355X. Everything is BIT-EXACT: section counts, soil
type order, size / floor / saturation doubles, frequency planes and the number of rand() draws."""
import numpy as np
import pytest

from common import SNAP_CASES, digests, golden_snapshot, load_cfg, case_dims
from oracle_lib import Oracle
from soilmachine_amd import capi
from soilmachine_amd.machine import Layermap, SoilMachine
from soilmachine_amd.snapshot import compare

pytestmark = pytest.mark.gpu
DIG = digests()
ENGINES = [capi.ENGINE_SERIAL, capi.ENGINE_SPECULATIVE]


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(f"u{a.dtype.itemsize}")


# ---------------------------------------------------------------- terrain + read-side primitives
@pytest.mark.parametrize("soil,dimx,dimy,seed", [
    ("default.soil", 256, 256, 0), ("rockgravelpebblessand.soil", 128, 128, 0), ("painted.soil", 96, 64, 3),
    ("bigbutte2.soil", 64, 80, 12345), ("rocksand.soil", 33, 47, 7)])
def test_terrain_init_bit_exact(soil, dimx, dimy, seed):
    cfg = load_cfg(soil)
    m = Layermap(cfg, dimx, dimy, seed=seed)
    o = Oracle(cfg, dimx, dimy, seed=seed)
    o.initialize()
    bad = compare(m.snapshot(), o.snapshot())
    assert not bad, bad


def test_heights_surface_normals_bilinear_bit_exact():
    cfg = load_cfg("rockgravelpebblessand.soil")
    n = 96
    m = Layermap(cfg, n, n, seed=2)
    o = Oracle(cfg, n, n, seed=2)
    o.initialize()
    assert np.array_equal(bits(m.heights()), bits(o.heights()))
    surf = np.array([o.L.so_surface(o.h, x, y) for x in range(n) for y in range(n)], np.uint32)
    assert np.array_equal(m.surface(), surf)
    # normals for every cell, including the k = 1, 2 border cases (layermap.h:347-375)
    ref = np.zeros((n * n, 3), np.float32)
    tmp = np.zeros(3, np.float32)
    for x in range(n):
        for y in range(n):
            o.L.so_normal(o.h, x, y, tmp.ctypes.data)
            ref[x * n + y] = tmp
    assert np.array_equal(bits(m.normals()), bits(ref))
    rng = np.random.default_rng(0)
    pos = (rng.random((4000, 2)) * (n - 1.001)).astype(np.float32)
    pos[:50] = np.floor(pos[:50])                                  # exact integer positions too
    want = np.array([o.L.so_height_bilinear(o.h, float(p[0]), float(p[1])) for p in pos])
    assert np.array_equal(bits(m.heights_bilinear(pos)), bits(want))


def test_vertex_stream_matches_reference_update_rule():
    """smx_fill_vertices = Layermap::update(Vertexpool&) (layermap.h:475-555, no SLICE cut) for every column: the 44-byte
    Vertex records rebuilt here from the ORACLE's heights / surface types / normals, bit for bit; non-square map with
    carved-out empty columns so that the (x, 0, y), (0,1,0), soils[0] branch of layermap.h:545 is taken too."""
    cfg = load_cfg("rockgravelpebblessand.soil")
    dimx, dimy = 70, 96                                            # not a multiple of the 256-cell block
    m = Layermap(cfg, dimx, dimy, seed=4)
    o = Oracle(cfg, dimx, dimy, seed=4)
    o.initialize()
    for (x, y) in [(0, 0), (5, 7), (69, 95), (33, 1)]:             # dig a few columns down to nothing
        for _ in range(8):                                         # one section per call (layermap.h:310-339)
            m.remove(x, y, 10.0)
            o.L.so_remove(o.h, x, y, 10.0)
    colors = np.linspace(0.05, 0.95, 4 * len(cfg.soils), dtype=np.float32).reshape(-1, 4)
    got = m.vertices(colors)
    h = o.heights()
    want = np.zeros((dimx * dimy, 11), np.float32)
    tmp = np.zeros(3, np.float32)
    n_empty = 0
    for x in range(dimx):
        for y in range(dimy):
            c = x * dimy + y
            ty = int(o.L.so_surface(o.h, x, y))
            if h[c] == 0.0 and ty == 0:
                want[c] = [x, 0, y, 0, 1, 0, *colors[0], 0]
                n_empty += 1
            else:
                o.L.so_normal(o.h, x, y, tmp.ctypes.data)
                want[c, 0:3] = [x, np.float32(cfg.SCALE * h[c]), y]
                want[c, 3:6] = tmp
                want[c, 6:10] = colors[ty]
                want[c, 10] = ty
    assert n_empty >= 4
    assert np.array_equal(bits(got), bits(want))


# ---------------------------------------------------------------- column operations (layermap.h:230-339)
def test_add_remove_sequences_bit_exact():
    cfg = load_cfg("rockgravelpebblessand.soil")
    n = 8
    m = Layermap(cfg, n, n, seed=0, initialize=False)
    o = Oracle(cfg, n, n, seed=0)
    rng = np.random.default_rng(5)
    nso = len(cfg.soils)
    for i in range(400):
        x, y = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        if rng.random() < 0.55:
            ty = int(rng.integers(0, nso))                         # includes Air == water: swap path
            sz = float(rng.choice([0.0, -0.1, 0.25, 1e-9, 0.5 * rng.random()]))
            m.add(x, y, sz, ty)
            o.L.so_add(o.h, x, y, sz, ty)
        else:
            hh = float(rng.choice([0.0, -1.0, 0.1, 0.25, 2.0 * rng.random()]))
            assert m.remove(x, y, hh) == o.L.so_remove(o.h, x, y, hh)
    bad = compare(m.snapshot(), o.snapshot())
    assert not bad, bad


def _patch(m, o, heights, types):
    for (x, y), cols in heights.items():
        for sz, ty in cols:
            m.add(x, y, sz, ty)
            o.L.so_add(o.h, x, y, sz, ty)


def test_cascades_and_seep_on_hand_built_patches():
    cfg = load_cfg("rockgravelpebblessand.soil")      # 0 Air, 1 Rock, 2 Gravel, 3 Pebbles, 4 Red Sand
    n = 6
    m = Layermap(cfg, n, n, seed=0, initialize=False)
    o = Oracle(cfg, n, n, seed=0)
    rng = np.random.default_rng(1)
    cols = {}
    for x in range(n):
        for y in range(n):
            base = [(0.3 + 0.2 * rng.random(), 1), (0.05 * rng.random(), int(rng.integers(2, 5)))]
            if (x + y) % 3 == 0:
                base.append((0.02 + 0.05 * rng.random(), 0))     # water on top
            cols[(x, y)] = base
    cols[(2, 2)] = [(1.5, 1), (0.2, 4), (0.3, 0)]                 # a peak with water: spills
    cols[(3, 3)] = [(0.4, 1), (0.4, 1)]                           # tie heights with (3,4)
    cols[(3, 4)] = [(0.8, 1)]
    _patch(m, o, cols, None)
    assert not compare(m.snapshot(), o.snapshot())
    for (px, py, tl) in [(2.0, 2.0, 0), (2.4, 2.6, 1), (0.0, 0.0, 1), (5.0, 5.0, 0), (3.0, 3.0, 1)]:
        m._chk(m.L.smx_particle_cascade(m.h, px, py, tl))
        o.L.so_particle_cascade(o.h, px, py, tl)
        bad = compare(m.snapshot(), o.snapshot())
        assert not bad, (px, py, tl, bad)
    for (x, y, sp) in [(2, 2, 3), (0, 0, 3), (3, 0, 0), (5, 5, 2), (2, 3, 1)]:
        m._chk(m.L.smx_water_cascade(m.h, x, y, sp))
        o.L.so_water_cascade(o.h, x, y, sp)
        m._chk(m.L.smx_seep(m.h, x, y))
        o.L.so_seep(o.h, x, y)
        bad = compare(m.snapshot(), o.snapshot())
        assert not bad, (x, y, sp, bad)


# ---------------------------------------------------------------- whole ticks
@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("case", sorted(SNAP_CASES))
def test_ticks_reproduce_reference_snapshots(case, engine):
    soil, seed, dowind, ticks = SNAP_CASES[case]
    cfg = load_cfg(soil)
    d = DIG[case]
    dimx, dimy = case_dims(d, cfg)
    sm = SoilMachine(cfg, dimx=dimx, dimy=dimy, seed=seed, nwater=d["nwater"], nwind=d["nwind"], dowind=dowind,
                     engine=engine)
    t = 0
    for tk in ticks:
        sm.tick(tk - t, sync=True)
        t = tk
        bad = compare(sm.map.snapshot(), golden_snapshot(case, tk))
        assert not bad, f"{case} tick {tk}: {bad}"
    c = sm.map.counters()
    assert c["steps_water_top"] == d["steps_water_top"] and c["steps_wind"] == d["steps_wind"]
    assert c["pool_overflow"] == 0


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("case", ["default256_t20_wind", "default256_t20", "rgps256_t10", "rgpbig256_t10",
                                  "sand256_t10", "bigbutte2_128_t10"])
def test_ticks_reproduce_reference_digests(case, engine):
    d = DIG[case]
    cfg = load_cfg(d["soil"])
    dimx, dimy = case_dims(d, cfg)
    kw = d["kw"]
    sm = SoilMachine(cfg, dimx=dimx, dimy=dimy, seed=kw.get("seed", 0), nwater=d["nwater"], nwind=d["nwind"],
                     dowind=kw.get("wind", True), engine=engine)
    sm.tick(d["ticks"], sync=True)
    s = sm.map.snapshot()
    g = s.digest()
    assert (g["nsec"], g["typehash"], g["sumh"], s.rand_calls) == (d["nsec"], d["typehash"], d["sumh"], d["rand_calls"])


@pytest.mark.slow
@pytest.mark.parametrize("engine", ENGINES)
def test_lakes_and_nested_particles_100_ticks(engine):
    d = DIG["default256_t100"]
    cfg = load_cfg(d["soil"])
    sm = SoilMachine(cfg, 256, seed=0, nwater=d["nwater"], nwind=0, dowind=False, engine=engine)
    sm.tick(100, sync=True)
    s = sm.map.snapshot()
    g = s.digest()
    assert (g["nsec"], g["typehash"], g["sumh"], s.rand_calls) == (d["nsec"], d["typehash"], d["sumh"], d["rand_calls"])
    c = sm.map.counters()
    assert c["nested_particles"] > 100000 and c["grid_active_cells"] > 0


@pytest.mark.parametrize("engine", ENGINES)
def test_phase_by_phase_equals_tick_and_oracle(engine):
    cfg = load_cfg("rocksand.soil")
    n = 80
    a = SoilMachine(cfg, n, seed=4, nwater=120, nwind=80, engine=engine)
    b = SoilMachine(cfg, n, seed=4, nwater=120, nwind=80, engine=engine)
    o = Oracle(cfg, n, n, seed=4)
    o.initialize()
    for _ in range(4):
        a.tick(1)
        b.water(); b.grid_pass(); b.wind(); b.map_frequency(); b.reset_frequency()
        o.tick(120, 80, True, True)
    sa, sb, so = a.map.snapshot(), b.map.snapshot(), o.snapshot()
    assert not compare(sa, sb)
    assert not compare(sa, so)
    oc, gc = o.counters(), a.map.counters()
    for k in ("steps_water_top", "steps_water_all", "steps_wind", "nested_particles", "floods", "cascade_calls",
              "cascade_transfers", "wcascade_calls", "rand_calls"):
        assert oc[k] == gc[k], k


def test_import_resume_midrun():
    """Import a mid-run reference state, resume the rand stream at its offset, continue -> reference state."""
    cfg = load_cfg("default.soil")
    d = DIG["default64"]
    s5 = golden_snapshot("default64", 5)
    m = Layermap(cfg, 64, 64, seed=0, initialize=False)
    m.load(s5, rand_seed=0)
    assert not compare(m.snapshot(), s5)              # import/export round trip is loss-free
    sm = SoilMachine.__new__(SoilMachine)
    sm.cfg, sm.map, sm.nwater, sm.nwind, sm.dowater, sm.dowind = cfg, m, d["nwater"], 0, True, False
    sm.tick(15, sync=True)
    bad = compare(m.snapshot(), golden_snapshot("default64", 20))
    assert not bad, bad


def test_pool_exhaustion_is_reported_not_silent():
    cfg = load_cfg("default.soil")
    m = Layermap(cfg, 16, 16, seed=0, pool=300)       # 256 cells need 256 sections; 44 spare
    o = Oracle(cfg, 16, 16, seed=0, pool=300)
    o.initialize()
    assert not compare(m.snapshot(), o.snapshot())
    for i in range(60):                               # push alternating types until the pool is empty
        for mm in (m,):
            mm.add(i % 16, 3, 0.01, 0 if i % 2 else 1)
        o.L.so_add(o.h, i % 16, 3, 0.01, 0 if i % 2 else 1)
    assert not compare(m.snapshot(), o.snapshot())
    assert m.counters()["pool_overflow"] == o.counters()["pool_overflow"]
    assert m.counters()["pool_free"] == o.counters()["pool_free"]


def test_speculative_engine_statistics_and_exactness_1024():
    """The speculative engine really runs speculatively (rounds > 0, most particles commit without a serial
    fallback) and lands on the same bits as the oracle on a map big enough to have parallelism."""
    cfg = load_cfg("rockgravelpebblessand.soil")
    n = 1024
    sm = SoilMachine(cfg, n, seed=0, engine=capi.ENGINE_SPECULATIVE)
    o = Oracle(cfg, n, n, seed=0)
    o.initialize()
    sm.tick(2, sync=True)
    for _ in range(2):
        o.tick(cfg.NWATER, cfg.NWIND, True, True)
    assert not compare(sm.map.snapshot(), o.snapshot())
    c = sm.map.counters()
    assert c["spec_rounds"] > 0
    oc = o.counters()
    for k in ("steps_water_top", "steps_water_all", "steps_wind", "nested_particles", "floods", "cascade_calls",
              "cascade_transfers", "rand_calls"):
        assert oc[k] == c[k], k


def test_speculative_sub_phases_replace_the_serial_fallback():
    """Round 4: when the nested particles of a phase outnumber the slots one speculative (sub-)phase owns, the
    sub-phase commits, ends and is re-armed from the continued rand() stream -- the rest of the phase no longer goes to the one-lane
    serial walker. 256^2 with lakes, 4000 + 300 particles per tick (thousands of nested particles per phase): full state, counters
    and rand() draws equal the reference-order oracle's; no serial fallback; sub-phases were in fact cut."""
    cfg = load_cfg("rockgravelpebblessand.soil")
    n = 256
    o = Oracle(cfg, n, n, seed=0)
    o.initialize()
    for _ in range(6):
        o.tick(cfg.NWATER, cfg.NWIND, True, True)
    sm = SoilMachine(cfg, n, seed=0, nwater=4000, nwind=300, engine=capi.ENGINE_SPECULATIVE)
    sm.map._chk(sm.map.L.smx_set_spec_limits(sm.map.h, 0, 1024))       # round 3's slot budget: forces the hand-over (default: 8192)
    sm.map.load(o.snapshot(), rand_seed=0)
    base = o.counters()
    for t in range(2):
        sm.tick(1, sync=True)
        o.tick(4000, 300, True, True)
        assert not compare(sm.map.snapshot(), o.snapshot()), f"tick {t}"
    oc, c = o.counters(), sm.map.counters()
    for k in ("steps_water_all", "steps_wind", "nested_particles", "floods"):
        assert oc[k] - base[k] == c[k], k
    assert oc["nested_particles"] - base["nested_particles"] > 2048
    assert c["spec_fallbacks"] == 0 and c["spec_serial_particles"] == 0 and c["spec_subphases_cut"] >= 2
