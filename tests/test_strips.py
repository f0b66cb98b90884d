"""Column strips on the CPU: the host-sim (device headers compiled for the host) as the per-rank backend.

  * the strip schedule itself (INTERIOR / SEAM chunks) against the oracle's restatement: tests/test_batched.py-style;
  * G contexts exchanging halo columns + migrating particles through the loop-back transport (threads) land on the same
    bits as ONE context running the same (nstrips, a, b) schedule;
  * the same over torch.distributed (gloo, world size 2, two processes).
"""
import ctypes
import os
import sys

import numpy as np
import pytest

from common import load_cfg
from hostsim_lib import HostSim, HostSimStripBackend, lib
from oracle_lib import Oracle
from soilmachine_amd.snapshot import Snapshot, compare
from strips_ref import StripGeometry, StripRank, ThreadComm, TorchComm, merge_owned, run_threads

CK = ("steps_water_all", "steps_water_top", "steps_wind", "floods", "nested_particles", "cascade_calls", "cascade_transfers")


def dry_state(n=256, soil="rockgravelpebblessand.soil"):
    cfg = load_cfg(soil)
    o = Oracle(cfg, n, n, seed=0)
    o.initialize()
    return cfg, o.snapshot()


def wet_state(n=256, ticks=70):
    """default.soil after `ticks` exact ticks: lakes, so floods + nested particles + a wet grid pass"""
    cfg = load_cfg("default.soil")
    o = Oracle(cfg, n, n, seed=0)
    o.initialize()
    for _ in range(ticks):
        o.tick(250, 0, True, False)
    return cfg, o.snapshot()


def single_context(cfg, s0, n, G, nw, nd, ticks):
    L = lib()
    L.hs_set_batch_strips(G, 16, 48)
    try:
        h = HostSim(cfg, n, n, seed=0)
        h.load(s0, advance_rand=True)
        for _ in range(ticks):
            h.batched_tick(nw, nd, True, nd > 0)
        return h.snapshot(), h.counters()
    finally:
        L.hs_set_batch_strips(1, 16, 48)


@pytest.mark.parametrize("G,n", [(2, 256), (3, 384)])
def test_strip_schedule_device_headers_equal_oracle(G, n):
    cfg, s0 = dry_state(n)
    ref, cref = single_context(cfg, s0, n, G, 500, 250, 2)
    o = Oracle(cfg, n, n, seed=0)
    o.load(s0)
    o.L.so_batched_set_strips.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3
    o.L.so_batched_set_strips(o.h, G, 16, 48)
    for _ in range(2):
        o.batched_tick(500, 250, True, True)
    assert not compare(ref, o.snapshot())
    co = o.counters()
    assert {k: cref[k] for k in CK} == {k: co[k] for k in CK}
    assert o.batched_stats()["guard_violations"] == 0


@pytest.mark.parametrize("kind,G,n,nw,nd,ticks", [("dry", 2, 256, 500, 250, 2), ("wet", 2, 256, 250, 0, 3), ("dry", 3, 384, 700, 350, 2)])
def test_loopback_strips_equal_single_context(kind, G, n, nw, nd, ticks):
    cfg, s0 = wet_state(n) if kind == "wet" else dry_state(n)
    ref, cref = single_context(cfg, s0, n, G, nw, nd, ticks)
    geom = StripGeometry(n, G)
    hs = [HostSim(cfg, n, n, seed=0) for _ in range(G)]
    for x in hs:
        x.load(s0, advance_rand=True)
    comms = ThreadComm.world(G)
    ranks = [StripRank(HostSimStripBackend(hs[r]), comms[r], geom) for r in range(G)]
    try:
        run_threads(ranks, lambda rk: [rk.tick(nw, nd, True, nd > 0) for _ in range(ticks)])
    finally:
        lib().hs_set_batch_strips(1, 16, 48)
    got = merge_owned(geom, [x.snapshot() for x in hs], n)
    assert not compare(got, ref)
    cs = [x.counters() for x in hs]
    assert {k: sum(c[k] for c in cs) for k in CK} == {k: cref[k] for k in CK}
    assert all(c["rand_calls"] == cref["rand_calls"] for c in cs)          # every rank drew the same stream
    assert ranks[0].stats["seam_chunks"] > 0 and sum(c.bytes_sent for c in comms) > 0
    if kind == "wet":
        assert cref["floods"] > 200 and cref["nested_particles"] > 500


def _gloo_worker(rank, world, port, path, nw, nd, ticks, n, soil):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(path)
    cfg = load_cfg(soil)
    s0 = Snapshot(n, n, cfg.SCALE, len(cfg.soils), int(z["rand_calls"]), 0, z["count"], z["type"], z["size"], z["floor"], z["sat"],
                  z["wfreq"], z["wtrack"], z["windfreq"])
    h = HostSim(cfg, n, n, seed=0)
    h.load(s0, advance_rand=True)
    rk = StripRank(HostSimStripBackend(h), TorchComm(dist), StripGeometry(n, world))
    for _ in range(ticks):
        rk.tick(nw, nd, True, nd > 0)
    s = h.snapshot()
    np.savez(path + f".rank{rank}.npz", count=s.count, type=s.type, size=s.size, floor=s.floor, sat=s.sat, wfreq=s.wfreq,
             wtrack=s.wtrack, windfreq=s.windfreq, rand_calls=s.rand_calls, steps=h.counters()["steps_water_all"] + h.counters()["steps_wind"])
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world_size_2_equals_single_context(tmp_path):
    import torch.multiprocessing as mp
    n, nw, nd, ticks, soil = 256, 400, 200, 2, "rockgravelpebblessand.soil"
    cfg, s0 = dry_state(n, soil)
    path = str(tmp_path / "s0.npz")
    np.savez(path, count=s0.count, type=s0.type, size=s0.size, floor=s0.floor, sat=s0.sat, wfreq=s0.wfreq, wtrack=s0.wtrack,
             windfreq=s0.windfreq, rand_calls=s0.rand_calls)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_gloo_worker, args=(2, port, path, nw, nd, ticks, n, soil), nprocs=2, join=True)
    ref, cref = single_context(cfg, s0, n, 2, nw, nd, ticks)
    snaps = []
    steps = 0
    for r in range(2):
        z = np.load(path + f".rank{r}.npz")
        snaps.append(Snapshot(n, n, cfg.SCALE, len(cfg.soils), int(z["rand_calls"]), 0, z["count"], z["type"], z["size"], z["floor"],
                              z["sat"], z["wfreq"], z["wtrack"], z["windfreq"]))
        steps += int(z["steps"])
    got = merge_owned(StripGeometry(n, 2), snaps, n)
    assert not compare(got, ref)
    assert steps == cref["steps_water_all"] + cref["steps_wind"]
