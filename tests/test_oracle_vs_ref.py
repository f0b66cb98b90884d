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
