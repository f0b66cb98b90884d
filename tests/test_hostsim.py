"""The DEVICE sources (soil_core.h + policies), compiled for the host, against the golden fixtures and the
oracle: unit-tests the kernel logic (recursion-free cascade state machine, sorting network, inline-top column
primitives) without a GPU. The product never runs this build."""
import numpy as np
import pytest

from common import SNAP_CASES, digests, golden_snapshot, load_cfg, case_dims
from hostsim_lib import HostSim
from oracle_lib import Oracle
from soilmachine_amd.snapshot import compare

DIG = digests()


@pytest.mark.parametrize("case", sorted(SNAP_CASES))
def test_device_sources_reproduce_reference_snapshots(case):
    soil, seed, dowind, ticks = SNAP_CASES[case]
    cfg = load_cfg(soil)
    d = DIG[case]
    dimx, dimy = case_dims(d, cfg)
    o = Oracle(cfg, dimx, dimy, seed=seed)
    o.initialize()
    hs = HostSim(cfg, dimx, dimy, seed=seed)
    hs.load(o.snapshot())
    t = 0
    for tk in ticks:
        while t < tk:
            hs.tick(d["nwater"], d["nwind"], True, dowind)
            t += 1
        bad = compare(hs.snapshot(), golden_snapshot(case, tk))
        assert not bad, f"{case} tick {tk}: {bad}"
    c = hs.counters()
    assert c["steps_water_top"] == d["steps_water_top"] and c["steps_wind"] == d["steps_wind"]


@pytest.mark.parametrize("case", ["default256_t20", "rgps256_t10", "sand256_t10"])
def test_device_sources_reproduce_reference_digests(case):
    d = DIG[case]
    cfg = load_cfg(d["soil"])
    dimx, dimy = case_dims(d, cfg)
    kw = d["kw"]
    o = Oracle(cfg, dimx, dimy, seed=kw.get("seed", 0))
    o.initialize()
    hs = HostSim(cfg, dimx, dimy, seed=kw.get("seed", 0))
    hs.load(o.snapshot())
    for _ in range(d["ticks"]):
        hs.tick(d["nwater"], d["nwind"], True, kw.get("wind", True))
        o.tick(d["nwater"], d["nwind"], True, kw.get("wind", True))
    s = hs.snapshot()
    g = s.digest()
    assert (g["nsec"], g["typehash"], g["sumh"], s.rand_calls) == (d["nsec"], d["typehash"], d["sumh"], d["rand_calls"])
    oc, hc = o.counters(), hs.counters()
    for k in ("steps_water_all", "nested_particles", "floods", "cascade_calls", "cascade_transfers", "wcascade_calls"):
        assert oc[k] == hc[k], k


# ---------------------------------------------------------------- speculative engine (threads emulate the waves)
@pytest.fixture(params=[1, 5], ids=["boundary-scalar", "boundary-team5"])
def boundary_team(request):
    """The round boundary as one thread, and as a team of 5 threads with a barrier (the device runs 1024)."""
    HostSim.set_boundary_threads(request.param)
    yield request.param
    HostSim.set_boundary_threads(1)


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 7])            # bit 2: the wave-loop structure of soil_coop.h
@pytest.mark.parametrize("case", ["default64", "rgps64", "rocksand48x80"])
def test_speculative_protocol_is_exact(case, mode, boundary_team):
    """stamps / claims / forwarding / kill closure / rollback / suspend-resume: whatever interleaving the 8 threads
    produce and whichever engine mode is on, the state after every tick equals the reference's."""
    soil, seed, dowind, ticks = SNAP_CASES[case]
    cfg = load_cfg(soil)
    d = DIG[case]
    dimx, dimy = case_dims(d, cfg)
    o = Oracle(cfg, dimx, dimy, seed=seed)
    o.initialize()
    hs = HostSim(cfg, dimx, dimy, seed=seed)
    hs.load(o.snapshot())
    t = 0
    for tk in ticks:
        while t < tk:
            hs.spec_tick(d["nwater"], d["nwind"], True, dowind, nthreads=8, scout=1 | (mode << 4), shuffle=11 + t)
            t += 1
        bad = compare(hs.snapshot(), golden_snapshot(case, tk))
        assert not bad, f"{case} mode {mode} tick {tk}: {bad}"
    c = hs.counters()
    assert c["steps_water_top"] == d["steps_water_top"] and c["steps_wind"] == d["steps_wind"]


def test_speculative_protocol_larger_map_has_parallelism():
    cfg = load_cfg("rockgravelpebblessand.soil")
    n = 512
    o = Oracle(cfg, n, n, seed=0)
    o.initialize()
    hs = HostSim(cfg, n, n, seed=0)
    hs.load(o.snapshot())
    st = hs.spec_tick(cfg.NWATER, cfg.NWIND, nthreads=8)
    o.tick(cfg.NWATER, cfg.NWIND, True, True)
    assert not compare(hs.snapshot(), o.snapshot())
    rounds_w, executed_w, fell_back = int(st[0]), int(st[4]), int(st[3])
    assert (fell_back & 0xFFFF) == 0 and rounds_w < cfg.NWATER // 4          # nothing went to the serial walker; far fewer rounds than particles
    oc, hc = o.counters(), hs.counters()
    for k in ("steps_water_all", "steps_wind", "nested_particles", "cascade_transfers", "rand_calls"):
        assert oc[k] == hc[k], k


def test_nested_particles_void_only_their_slots_and_lakes_do_not_fall_back(boundary_team):
    """rand() slots + solo rounds + take-over-chain clearing (soil_spec.h): 1000 water particles on a 256^2 map is far
    denser than any BASELINE config -- lakes form after three ticks, every tick spawns hundreds of nested particles and
    lake cells are handed on through dozens of tentative holders. The engine must stay exact, must not give the phase
    to the serial walker, and a nested particle must no longer cost a re-run of every later particle."""
    cfg = load_cfg("rockgravelpebblessand.soil")
    n = 256
    o = Oracle(cfg, n, n, seed=0)
    o.initialize()
    hs = HostSim(cfg, n, n, seed=0)
    hs.load(o.snapshot())
    nested_before = 0
    for t in range(7):
        st = hs.spec_tick(cfg.NWATER, cfg.NWIND, True, True, nthreads=8, shuffle=5 + t)
        o.tick(cfg.NWATER, cfg.NWIND, True, True)
        assert not compare(hs.snapshot(), o.snapshot()), f"tick {t}"
        oc, hc = o.counters(), hs.counters()
        for k in ("steps_water_all", "steps_wind", "nested_particles", "rand_calls", "pool_free"):
            assert oc[k] == hc[k], (t, k)
        assert (int(st[3]) & 0xFFFF) == 0 and (int(st[9]) & 0xFFFF) == 0, f"tick {t}: fell back to the serial walker"
        executed_w = int(st[4])
        assert executed_w < 6 * cfg.NWATER, f"tick {t}: {executed_w} executions for {cfg.NWATER} particles"
        nested_before = hc["nested_particles"]
    assert nested_before > 1000                                # the regime this test is about was reached


def test_more_particles_than_the_boundary_window():
    """A phase with more slots than SPEC_WINDOW (4096): the boundary's window slides with the committed prefix and
    the slot chain continues across its end -- same state as the reference, no serial fallback."""
    cfg = load_cfg("rockgravelpebblessand.soil")
    n, nwater = 768, 4600
    o = Oracle(cfg, n, n, seed=0)
    o.initialize()
    hs = HostSim(cfg, n, n, seed=0)
    hs.load(o.snapshot())
    st = hs.spec_tick(nwater, 0, True, False, nthreads=8, shuffle=3)
    o.tick(nwater, 0, True, False)
    assert not compare(hs.snapshot(), o.snapshot())
    assert (int(st[3]) & 0xFFFF) == 0                          # did not fall back
    oc, hc = o.counters(), hs.counters()
    for k in ("steps_water_all", "nested_particles", "rand_calls", "pool_free"):
        assert oc[k] == hc[k], k


@pytest.mark.parametrize("sub", [4096, 700])
def test_sub_phases_replace_the_serial_fallback_when_nested_slots_run_out(sub):
    """Round 4: a phase whose nested particles outnumber the slots one speculative (sub-)phase owns (SPEC_MAXNEST =
    1024) no longer hands its rest to the one-lane serial walker: the sub-phase commits what it has, ends, and the next one is armed
    from the continued rand() stream. 4000 particles per tick on a 256^2 map with lakes: thousands of nested particles per phase.
    State, counters and rand() draws equal the reference-order oracle's; nothing runs serially. `sub` also cuts phases by size."""
    cfg = load_cfg("rockgravelpebblessand.soil")
    n = 256
    o = Oracle(cfg, n, n, seed=0)
    o.initialize()
    for _ in range(6):
        o.tick(cfg.NWATER, cfg.NWIND, True, True)              # lakes (the oracle alone: fast)
    hs = HostSim(cfg, n, n, seed=0)
    hs.load(o.snapshot())
    hs.L.hs_set_spec_sub(sub)
    try:
        cuts = 0
        base = o.counters()                                      # (the host-sim's counters start at the snapshot)
        for t in range(2):
            n0 = o.counters()["nested_particles"]
            st = hs.spec_tick(4000, 300, True, True, nthreads=8, shuffle=9 + t)
            o.tick(4000, 300, True, True)
            assert not compare(hs.snapshot(), o.snapshot()), f"tick {t}"
            oc, hc = o.counters(), hs.counters()
            for k in ("steps_water_all", "steps_wind", "nested_particles", "floods"):
                assert oc[k] - base[k] == hc[k], (t, k)
            assert oc["rand_calls"] == hc["rand_calls"] and oc["pool_free"] == hc["pool_free"]
            assert (int(st[3]) & 0xFFFF) == 0 and (int(st[9]) & 0xFFFF) == 0, f"tick {t}: something ran on the serial walker"
            cuts += int(st[3]) >> 16
            assert oc["nested_particles"] - n0 > 1024            # more nested particles than one sub-phase has slots
        assert cuts >= 2 or sub < 4096                          # (short sub-phases end by size before their nested slots run out)
    finally:
        hs.L.hs_set_spec_sub(4096)
