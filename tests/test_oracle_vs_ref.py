"""Live differential test oracle restatement <-> the compiled reference (oracle/_ref/soil_ref).
Skipped where the reference binary is absent (it is built from /root/reference by oracle/Makefile)."""
import os
import tempfile

import pytest

from common import load_cfg
from oracle_lib import Oracle, have_ref, run_ref
from soilmachine_amd.snapshot import compare, read_snapshot

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref/soil_ref not built (needs /root/reference)")
REFSOIL = "/root/reference/soil"


@pytest.mark.parametrize("soil,size,seed,ticks,nwater,nwind", [
    ("default.soil", 96, 11, 12, 200, 0),
    ("rocksand.soil", 80, 2, 6, 120, 80),
    ("rockgravelpebbles.soil", 72, 5, 8, 150, 0),
    ("sand.soil", 64, 9, 6, 50, 120),
    ("bigbutte.soil", 64, 4, 6, 100, 0),
])
def test_restatement_equals_reference(soil, size, seed, ticks, nwater, nwind):
    if not os.path.exists(os.path.join(REFSOIL, soil)):
        pytest.skip("reference soil files absent")
    cfg = load_cfg(soil)
    with tempfile.TemporaryDirectory() as td:
        run_ref(os.path.join(REFSOIL, soil), seed=seed, size=size, ticks=ticks, nwater=nwater, nwind=nwind,
                wind=nwind > 0, dump_prefix=os.path.join(td, "r"), dump_at=[0, ticks])
        r0 = read_snapshot(os.path.join(td, "r.t0.snap"))
        r1 = read_snapshot(os.path.join(td, f"r.t{ticks}.snap"))
    o = Oracle(cfg, size, size, seed=seed)
    o.initialize()
    assert not compare(o.snapshot(), r0)
    for _ in range(ticks):
        o.tick(nwater, nwind, True, nwind > 0)
    assert not compare(o.snapshot(), r1)


def test_reference_driver_resumes_from_a_snapshot_bit_exactly():
    """oracle/ref_driver.cpp --load (what bench.py's cpu_baseline uses to time the CPU path on the GPU line's own state): k ticks, snapshot,
    load, n - k ticks == n ticks uninterrupted -- full state, rand() draws, and the JSON figures the P2 envelope reads."""
    soil = os.path.join(REFSOIL, "rockgravelpebblessand.soil")
    if not os.path.exists(soil):
        pytest.skip("reference soil files absent")
    kw = dict(seed=3, size=96, nwater=300, nwind=120, lean=True)
    with tempfile.TemporaryDirectory() as td:
        whole = run_ref(soil, ticks=9, dump_prefix=os.path.join(td, "w"), dump_at=[4, 9], **kw)
        part = run_ref(soil, ticks=5, load=os.path.join(td, "w.t4.snap"), dump_prefix=os.path.join(td, "p"), dump_at=[5], **kw)
        assert not compare(read_snapshot(os.path.join(td, "p.t5.snap")), read_snapshot(os.path.join(td, "w.t9.snap")))
        for k in ("sumh", "nsec", "typehash", "rand_calls", "next_rand", "standing", "water_volume"):
            assert part[k] == whole[k], k
        # another rand() stream on the same terrain (the P2 control) leaves the terrain alone and changes the run
        base = run_ref(soil, ticks=0, dump_prefix=os.path.join(td, "b"), dump_at=[0], **kw)
        other = run_ref(soil, ticks=9, rand_seed=77, dump_prefix=os.path.join(td, "o"), dump_at=[0], **kw)
        t0, o0 = read_snapshot(os.path.join(td, "b.t0.snap")), read_snapshot(os.path.join(td, "o.t0.snap"))
        o0.rand_calls = t0.rand_calls
        assert not compare(o0, t0) and base["nsec"] == t0.nsec
        assert other["typehash"] != whole["typehash"] or other["sumh"] != whole["sumh"]
