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
