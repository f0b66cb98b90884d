#!/usr/bin/env python3
"""Score a protocol change of the speculative engine WITHOUT a GPU: run ticks of the host simulation (the device
headers compiled for the host, tests/hostsim) next to the oracle, check exactness, and print per phase
[rounds, aborts, kills, fell_back, executions, critical-path steps] -- the last one (sum over rounds of the longest
advance of any particle) is the proxy for the phase's GPU time (x the per-step latency, + rounds x the per-round overhead).

  tools/hostsim_proxy.py --size 4096 --ticks 3                       # the bench workload
  tools/hostsim_proxy.py --size 256 --ticks 8 --boundary-threads 7   # dense lake regime, boundary as a team
  tools/hostsim_proxy.py --size 1024 --nwater 6000 --nwind 0         # more particles than the boundary window
  tools/hostsim_proxy.py --deps --size 4096 --nwater 4000            # oracle dependency analysis (DAG depth, critical path)
Engine modes: bit0 blocked-particle filter, bit1 suspend/resume, bit2 wave-loop structure of soil_coop.h.
"""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import load_cfg                                    # noqa: E402
from hostsim_lib import HostSim                                # noqa: E402
import oracle_lib                                              # noqa: E402
from oracle_lib import Oracle                                  # noqa: E402
from soilmachine_amd.snapshot import compare                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--soil", default="rockgravelpebblessand.soil")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--ticks", type=int, default=3)
    ap.add_argument("--nwater", type=int, default=None)
    ap.add_argument("--nwind", type=int, default=None)
    ap.add_argument("--mode-water", type=int, default=3)
    ap.add_argument("--mode-wind", type=int, default=2)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--boundary-threads", type=int, default=1)
    ap.add_argument("--deps", action="store_true", help="oracle dependency analysis of the water phase instead")
    a = ap.parse_args()
    cfg = load_cfg(a.soil)
    nw = cfg.NWATER if a.nwater is None else a.nwater
    nd = cfg.NWIND if a.nwind is None else a.nwind
    n = a.size
    o = Oracle(cfg, n, n, seed=0)
    o.initialize()
    if a.deps:
        L = oracle_lib.lib()
        L.so_dep_enable.argtypes = [C.c_void_p, C.c_int]; L.so_dep_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.so_water_particles.argtypes = [C.c_void_p, C.c_int]
        o.tick(nw, 0, True, False)
        L.so_dep_enable(o.h, 1)
        L.so_water_particles(o.h, nw)
        out = (C.c_uint64 * 7)()
        L.so_dep_stats(o.h, out)
        print(nw, dict(zip(["tasks", "dag_depth_inplace", "dag_depth_multiversion", "conflict_free", "critical_path_steps",
                            "steps", "sum_of_levels"], list(out))))
        return
    hs = HostSim(cfg, n, n, seed=0)
    hs.load(o.snapshot())
    HostSim.set_boundary_threads(a.boundary_threads)
    for t in range(a.ticks):
        t0 = time.time()
        st = hs.spec_tick(nw, nd, True, nd > 0, nthreads=a.threads, scout=1 | (a.mode_water << 4) | (a.mode_wind << 8), shuffle=5 + t)
        t1 = time.time()
        o.tick(nw, nd, True, nd > 0)
        bad = compare(hs.snapshot(), o.snapshot())
        hc = hs.counters()
        print(f"tick {t}: water {st[:6].tolist()}  wind {st[6:].tolist()}  nested(total) {hc['nested_particles']}  "
              f"{'EXACT' if not bad else bad}  {t1 - t0:.1f}s", flush=True)
        assert not bad


if __name__ == "__main__":
    main()
