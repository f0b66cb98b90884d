"""Column strips on the MI355X. gpurun gives ONE device, so the strips run as several contexts on it:
through the loop-back transport (threads) and through torch.distributed (gloo, two processes sharing cuda:0 -- RCCL
refuses two ranks on one device). Either way the result must equal ONE context running the same (nstrips, inset, seam)
schedule, and that schedule must equal its CPU restatement. Two drivers: the library's own (csrc/soil_strips_host.h,
smx_strips_*: C++ protocol, device-resident messages; here over the host-callback transport -- the RCCL transport is the same
protocol with ncclSend/ncclRecv in place of the callbacks and is exercised with the one rank a single device allows) and round 2's
Python driver (tests/strips_ref.py StripRank, the protocol's executable description, also run by the host-sim)."""
import ctypes
import os
import sys

import numpy as np
import pytest

from common import load_cfg
from oracle_lib import Oracle
from soilmachine_amd import capi
from soilmachine_amd.machine import Layermap, SoilMachine
from soilmachine_amd.snapshot import Snapshot, compare
from strips_ref import DeviceStripBackend, LibStripRank, StripGeometry, StripRank, ThreadComm, TorchComm, merge_owned, run_threads

pytestmark = pytest.mark.gpu
CK = ("steps_water_all", "steps_water_top", "steps_wind", "floods", "nested_particles", "cascade_calls", "cascade_transfers")


def start_state(kind, n):
    if kind == "wet":
        cfg = load_cfg("default.soil")
        sm = SoilMachine(cfg, n, seed=0, nwater=250, nwind=0, dowind=False, engine=capi.ENGINE_SPECULATIVE)
        sm.tick(60, sync=True)                                   # exact engine: lakes
        return cfg, sm.map.snapshot()
    cfg = load_cfg("rockgravelpebblessand.soil")
    return cfg, Layermap(cfg, n, n, seed=0).snapshot()


def device_ctx(cfg, s0, n, engine=capi.ENGINE_BATCHED, x_range=None, device=0):
    """x_range = (lo, hi): a STRIP context that holds only these columns of the map (smx_create_strip): memory follows the strip"""
    m = Layermap(cfg, n, n, seed=0, initialize=False, engine=engine, x_range=x_range, device=device)
    m.load(s0, rand_seed=0)
    return m


def single_context(cfg, s0, n, G, nw, nd, ticks, engine=capi.ENGINE_BATCHED):
    m = device_ctx(cfg, s0, n, engine)
    m.set_batch_strips(G)
    sm = SoilMachine.__new__(SoilMachine)
    sm.cfg, sm.map, sm.nwater, sm.nwind, sm.dowater, sm.dowind = cfg, m, nw, nd, True, nd > 0
    sm.tick(ticks, sync=True)
    return m.snapshot(), m.counters()


@pytest.mark.parametrize("kind,G,n,nw,nd,ticks", [("dry", 2, 256, 500, 250, 2), ("wet", 2, 256, 250, 0, 3), ("dry", 3, 384, 700, 350, 2)])
def test_strip_schedule_on_device_equals_oracle_and_loopback(kind, G, n, nw, nd, ticks):
    cfg, s0 = start_state(kind, n)
    ref, cref = single_context(cfg, s0, n, G, nw, nd, ticks)
    # (1) the schedule itself against its CPU restatement
    o = Oracle(cfg, n, n, seed=0)
    o.load(s0)
    o.L.so_batched_set_strips.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3
    o.L.so_batched_set_strips(o.h, G, 16, 48)
    for _ in range(ticks):
        o.batched_tick(nw, nd, True, nd > 0)
    assert not compare(ref, o.snapshot())
    co = o.counters()
    assert {k: cref[k] for k in CK} == {k: co[k] for k in CK}
    # (2) G contexts exchanging halo columns and migrating particles (loop-back transport)
    geom = StripGeometry(n, G)
    ms = [device_ctx(cfg, s0, n) for _ in range(G)]
    comms = ThreadComm.world(G)
    ranks = [StripRank(DeviceStripBackend(ms[r]), comms[r], geom) for r in range(G)]
    run_threads(ranks, lambda rk: [rk.tick(nw, nd, True, nd > 0) for _ in range(ticks)])
    got = merge_owned(geom, [m.snapshot() for m in ms], n)
    assert not compare(got, ref)
    cs = [m.counters() for m in ms]
    assert {k: sum(c[k] for c in cs) for k in CK} == {k: cref[k] for k in CK}
    assert all(c["rand_calls"] == cref["rand_calls"] for c in cs)
    assert ranks[0].stats["seam_chunks"] > 0 and sum(c.bytes_sent for c in comms) > 0
    if kind == "wet":
        assert cref["floods"] > 100 and cref["nested_particles"] > 300


@pytest.mark.parametrize("kind,G,n,nw,nd,ticks", [("dry", 2, 256, 500, 250, 2), ("wet", 2, 256, 250, 0, 3), ("dry", 3, 384, 700, 350, 2)])
def test_library_strip_driver_over_loopback_equals_single_context(kind, G, n, nw, nd, ticks):
    """smx_strips_attach + smx_strips_tick: the C++ driver, G contexts on the one device, messages handed over by thread queues"""
    cfg, s0 = start_state(kind, n)
    ref, cref = single_context(cfg, s0, n, G, nw, nd, ticks)
    geom = StripGeometry(n, G)
    ms = [device_ctx(cfg, s0, n, x_range=geom.held(r, n)) for r in range(G)]      # every context holds its strip + halo only
    assert all(m.x_range is not None for m in ms)
    comms = ThreadComm.world(G)
    ranks = [None] * G

    def attach_and_run(r):
        ranks[r] = LibStripRank(ms[r], comms[r], geom)
        for _ in range(ticks):
            ranks[r].tick(nw, nd, True, nd > 0)

    class R:                                                      # (run_threads wants objects with .comm)
        def __init__(self, r): self.r, self.comm = r, comms[r]
    run_threads([R(r) for r in range(G)], lambda o: attach_and_run(o.r))
    got = merge_owned(geom, [m.snapshot() for m in ms], n)
    assert not compare(got, ref)
    cs = [m.counters() for m in ms]
    assert {k: sum(c[k] for c in cs) for k in CK} == {k: cref[k] for k in CK}
    assert all(c["rand_calls"] == cref["rand_calls"] for c in cs)
    st = ranks[0].stats
    assert st["seam_chunks"] > 0 and st["bytes_sent"] > 0 and st["messages"] > 0


@pytest.mark.parametrize("kind,G,n,nw,nd,ticks", [("dry", 2, 256, 500, 250, 2), ("wet", 2, 256, 250, 0, 3)])
def test_relaxed_schedule_in_strips_library_driver_equals_single_context(kind, G, n, nw, nd, ticks):
    """SMX_ENGINE_RELAXED under the strip schedule: a particle steps iff its step lies inside what the running kind of chunk may touch
    (csrc/soil_relax.h); G contexts exchanging halos == ONE context running the same schedule, bit for bit."""
    cfg, s0 = start_state(kind, n)
    ref, cref = single_context(cfg, s0, n, G, nw, nd, ticks, capi.ENGINE_RELAXED)
    geom = StripGeometry(n, G)
    ms = [device_ctx(cfg, s0, n, capi.ENGINE_RELAXED, x_range=geom.held(r, n)) for r in range(G)]
    comms = ThreadComm.world(G)

    class R:
        def __init__(self, r): self.r, self.comm = r, comms[r]

    def go(o):
        rk = LibStripRank(ms[o.r], comms[o.r], geom)
        for _ in range(ticks):
            rk.tick(nw, nd, True, nd > 0)
    run_threads([R(r) for r in range(G)], go)
    got = merge_owned(geom, [m.snapshot() for m in ms], n)
    assert not compare(got, ref)
    cs = [m.counters() for m in ms]
    assert {k: sum(c[k] for c in cs) for k in CK} == {k: cref[k] for k in CK}
    if kind == "wet":
        assert cref["floods"] > 100 and cref["nested_particles"] > 200


def test_rccl_transport_with_the_one_rank_a_single_device_allows():
    """librccl is loaded, the communicator is created, the strip driver ticks through it: with one rank there is no neighbour, so the
    result must be the plain batched engine's. (Two ranks need two devices: RCCL refuses duplicates.)"""
    cfg, s0 = start_state("dry", 256)
    ref, cref = single_context(cfg, s0, 256, 1, 500, 250, 2)

    class One:
        rank, size = 0, 1
        def allgather_bytes(self, b): return [b]
    m = device_ctx(cfg, s0, 256)
    rk = LibStripRank(m, One(), StripGeometry(256, 1), transport="rccl")
    for _ in range(2):
        rk.tick(500, 250, True, True)
    assert not compare(m.snapshot(), ref)
    assert {k: m.counters()[k] for k in CK} == {k: cref[k] for k in CK}
    # ... and the exchange itself, which one rank's ticks never reach: synthetic seam messages from the rank to itself through RcclTransport::exchange -- the
    # grouped header send/receive, the size synchronisation, the grouped payloads -- byte for byte, from empty to tens of megabytes, repeatedly (the header ring)
    g0 = rk.stats["transport_groups"]
    for nrec, nbytes, npart in ((0, 0, 0), (1, 32, 0), (0, 0, 1), (1000, 96_000, 17), (300_000, 40_000_000, 4_000), (5, 160, 3)) + ((7, 224, 2),) * 70:
        rk.selfcheck(nrec, nbytes, npart)
    assert rk.stats["transport_groups"] - g0 == 2 * 76 - 1           # (the empty message has no payload group)
    rk.tick(500, 250, True, True)                                    # the communicator is still usable
    rk.detach()


def _nccl_world_1(_i, port, path, out):
    """bench.py --gpus N's control plane with the one rank a single device allows: torch.distributed backend "nccl" (= RCCL), TorchComm on DEVICE tensors
    (unique-id all-gather, all-reduces), the library's own RCCL communicator beside torch's, the strip driver ticking through it."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    comm = TorchComm(dist, torch.device("cuda", 0))
    assert comm.allreduce_sum(41) == 41 and comm.allgather_bytes(b"seam") == [b"seam"] and comm.allgather_bytes(b"") == [b""]
    assert list(comm.allreduce_min_u32(np.array([7, 3, 9], np.uint32))) == [7, 3, 9]
    z = np.load(path)
    cfg = load_cfg("rockgravelpebblessand.soil")
    n = 256
    s0 = Snapshot(n, n, cfg.SCALE, len(cfg.soils), int(z["rand_calls"]), 0, z["count"], z["type"], z["size"], z["floor"], z["sat"],
                  z["wfreq"], z["wtrack"], z["windfreq"])
    m = device_ctx(cfg, s0, n, capi.ENGINE_RELAXED)
    rk = LibStripRank(m, comm, StripGeometry(n, 1), transport="rccl")
    for _ in range(2):
        rk.tick(500, 250, True, True)
    rk.selfcheck(1000, 96_000, 17)
    s = m.snapshot(); c = m.counters()
    np.savez(out, count=s.count, type=s.type, size=s.size, floor=s.floor, sat=s.sat, wfreq=s.wfreq, wtrack=s.wtrack, windfreq=s.windfreq,
             rand_calls=s.rand_calls, steps=c["steps_water_all"] + c["steps_wind"])
    dist.barrier()
    dist.destroy_process_group()


def test_torch_distributed_nccl_control_plane_with_one_rank(tmp_path):
    """What `bench.py --gpus N` under torchrun does before and around smx_strips_tick, on the one device the suite has: process group "nccl", TorchComm on device
    tensors, the library's RCCL communicator next to torch's -- the result must be the single context's."""
    import torch.multiprocessing as mp
    n = 256
    cfg, s0 = start_state("dry", n)
    path, out = str(tmp_path / "s0.npz"), str(tmp_path / "out.npz")
    np.savez(path, count=s0.count, type=s0.type, size=s0.size, floor=s0.floor, sat=s0.sat, wfreq=s0.wfreq, wtrack=s0.wtrack,
             windfreq=s0.windfreq, rand_calls=s0.rand_calls)
    mp.spawn(_nccl_world_1, args=(29500 + (os.getpid() % 2000), path, out), nprocs=1, join=True)
    ref, cref = single_context(cfg, s0, n, 1, 500, 250, 2, capi.ENGINE_RELAXED)
    z = np.load(out)
    got = Snapshot(n, n, cfg.SCALE, len(cfg.soils), int(z["rand_calls"]), 0, z["count"], z["type"], z["size"], z["floor"], z["sat"], z["wfreq"], z["wtrack"], z["windfreq"])
    assert not compare(got, ref)
    assert int(z["steps"]) == cref["steps_water_all"] + cref["steps_wind"]


def _worker(rank, world, port, path, nw, nd, ticks, n, soil, lib_driver=False, rccl=False, engine=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)       # (gloo hands the RCCL unique id around; the halos go over RCCL itself)
    z = np.load(path)
    cfg = load_cfg(soil)
    s0 = Snapshot(n, n, cfg.SCALE, len(cfg.soils), int(z["rand_calls"]), 0, z["count"], z["type"], z["size"], z["floor"], z["sat"],
                  z["wfreq"], z["wtrack"], z["windfreq"])
    if rccl:                                                           # one device per rank, the rank's context holds its strip only
        geom = StripGeometry(n, world)
        m = device_ctx(cfg, s0, n, engine if engine is not None else capi.ENGINE_BATCHED, x_range=geom.held(rank, n), device=rank)
        rk = LibStripRank(m, TorchComm(dist), geom, transport="rccl")
    else:
        m = device_ctx(cfg, s0, n)
        rk = LibStripRank(m, TorchComm(dist), StripGeometry(n, world)) if lib_driver else StripRank(DeviceStripBackend(m), TorchComm(dist), StripGeometry(n, world))
    for _ in range(ticks):
        rk.tick(nw, nd, True, nd > 0)
    s = m.snapshot()
    c = m.counters()
    np.savez(path + f".rank{rank}.npz", count=s.count, type=s.type, size=s.size, floor=s.floor, sat=s.sat, wfreq=s.wfreq,
             wtrack=s.wtrack, windfreq=s.windfreq, rand_calls=s.rand_calls, steps=c["steps_water_all"] + c["steps_wind"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lib_driver", [False, True])
def test_two_processes_torch_distributed_equal_single_context(tmp_path, lib_driver):
    import torch.multiprocessing as mp
    n, nw, nd, ticks, soil = 256, 400, 200, 2, "rockgravelpebblessand.soil"
    cfg, s0 = start_state("dry", n)
    path = str(tmp_path / "s0.npz")
    np.savez(path, count=s0.count, type=s0.type, size=s0.size, floor=s0.floor, sat=s0.sat, wfreq=s0.wfreq, wtrack=s0.wtrack,
             windfreq=s0.windfreq, rand_calls=s0.rand_calls)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, path, nw, nd, ticks, n, soil, lib_driver), nprocs=2, join=True)
    ref, cref = single_context(cfg, s0, n, 2, nw, nd, ticks)
    snaps, steps = [], 0
    for r in range(2):
        z = np.load(path + f".rank{r}.npz")
        snaps.append(Snapshot(n, n, cfg.SCALE, len(cfg.soils), int(z["rand_calls"]), 0, z["count"], z["type"], z["size"], z["floor"],
                              z["sat"], z["wfreq"], z["wtrack"], z["windfreq"]))
        steps += int(z["steps"])
    assert not compare(merge_owned(StripGeometry(n, 2), snaps, n), ref)
    assert steps == cref["steps_water_all"] + cref["steps_wind"]


@pytest.mark.parametrize("engine", ["batched", "relaxed"])
def test_rccl_two_ranks_equal_single_context(tmp_path, engine):
    """The RCCL transport with MORE THAN ONE rank: two processes, one device each, strip contexts
    (smx_create_strip), smx_strips_attach_rccl -- halo columns and migrating particles by ncclSend / ncclRecv between the devices' own
    buffers -- against ONE context running the same two-strip schedule, bit for bit. Needs two GPUs: skipped on the one-GPU boxes this
    suite usually sees, enabled by itself on the multi-GPU node before bench.py --gpus N relies on the transport."""
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices (RCCL refuses two ranks on one)")
    eng = capi.ENGINE_RELAXED if engine == "relaxed" else capi.ENGINE_BATCHED
    n, nw, nd, ticks, soil = 256, 400, 200, 3, "rockgravelpebblessand.soil"
    cfg, s0 = start_state("dry", n)
    path = str(tmp_path / "s0.npz")
    np.savez(path, count=s0.count, type=s0.type, size=s0.size, floor=s0.floor, sat=s0.sat, wfreq=s0.wfreq, wtrack=s0.wtrack,
             windfreq=s0.windfreq, rand_calls=s0.rand_calls)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, path, nw, nd, ticks, n, soil, True, True, eng), nprocs=2, join=True)
    ref, cref = single_context(cfg, s0, n, 2, nw, nd, ticks, eng)
    snaps, steps = [], 0
    for r in range(2):
        z = np.load(path + f".rank{r}.npz")
        snaps.append(Snapshot(n, n, cfg.SCALE, len(cfg.soils), int(z["rand_calls"]), 0, z["count"], z["type"], z["size"], z["floor"],
                              z["sat"], z["wfreq"], z["wtrack"], z["windfreq"]))
        steps += int(z["steps"])
    assert not compare(merge_owned(StripGeometry(n, 2), snaps, n), ref)
    assert steps == cref["steps_water_all"] + cref["steps_wind"]
