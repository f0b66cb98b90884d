#!/usr/bin/env python3
"""bench.py -- particle-steps/s and ms/tick of the SoilMachine particle-transport hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
  * a "step" is one tick of SoilMachine.cpp:283-329 (water particles -> grid pass -> wind particles ->
    frequency update) over the synthetic 4096^2 layermap of BASELINE.json configs[3]
    (soil/rockgravelpebblessand.soil: 4 soils, 2 noise layers, NWATER 1000, NWIND 250, SCALE 120, SEED 0);
    terrain comes from Layermap::initialize on the device, state is resident in HBM before the timed region.
  * value = particle-steps/s, whole job: a particle-step is one WaterParticle/WindParticle move()==true followed
    by its interact() (top-level AND nested particles), summed over all ranks / max-over-ranks wall time.
  * N > 1: the exact engine executes the reference's sequential semantics, which do not shard (DESIGN.md
    "Multi-GPU"): --gpus N runs N independent replicas (rank r uses SEED r), no data-path collective, weak scaling.
  * "roofline": algorithmic bytes of the dominant kernel / its HIP-event duration vs the 8 TB/s HBM peak.
  * "cpu_baseline": the reference's own CPU loop (oracle/_ref/soil_ref_lean = the unmodified reference headers,
    render-only Layermap::update elided) timed on this box's host, 1 core, on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SOIL = "rockgravelpebblessand.soil"
B_WATER_STEP = 192      # algorithmic bytes per water step   
B_WIND_STEP = 256       # algorithmic bytes per wind step
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec peak


def cpu_baseline(size: int, nwater: int, nwind: int, seed: int, ticks: int, pool: int) -> dict | None:
    """Time the reference's own CPU path (bounded sample) on this host. Test/bench infrastructure only."""
    exe = os.path.join(ROOT, "oracle", "_ref", "soil_ref_lean")
    kind = "reference"
    if not os.path.exists(exe):
        return None
    cmd = [exe, "--soil", os.path.join(ROOT, "soilmachine_amd", "soils", SOIL), "--seed", str(seed), "--size", str(size),
           "--ticks", str(ticks), "--nwater", str(nwater), "--nwind", str(nwind), "--pool", str(pool)]
    if hasattr(os, "sched_setaffinity"):
        pass
    t0 = time.time()
    out = subprocess.run(["taskset", "-c", "0"] + cmd, capture_output=True, text=True)
    if out.returncode != 0:
        out = subprocess.run(cmd, capture_output=True, text=True)
    wall = time.time() - t0
    j = None
    for line in out.stdout.splitlines():
        if line.startswith("JSON "):
            j = json.loads(line[5:])
    if j is None:
        return None
    t_tick = j["t_water"] + j["t_grid"] + j["t_wind"] + j["t_freq"]
    steps_top = j["steps_water_top"] + j["steps_wind"]
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": steps_top / t_tick if t_tick > 0 else None, "unit": "particle-steps/s", "cores": 1, "kind": kind,
        "sample": f"{ticks} tick(s) of the same workload ({size}x{size}, {SOIL}, nwater={nwater}, nwind={nwind}, SEED {seed}) "
                  f"with oracle/_ref/soil_ref_lean, 1 thread; steps = top-level move()==true (nested ones are not countable "
                  f"in the unmodified reference)",
        "ms_per_tick": 1e3 * t_tick / ticks, "ms_water": 1e3 * j["t_water"] / ticks, "ms_grid": 1e3 * j["t_grid"] / ticks,
        "ms_wind": 1e3 * j["t_wind"] / ticks, "ms_freq": 1e3 * j["t_freq"] / ticks,
        "particle_phase_steps_per_s": steps_top / max(j["t_water"] + j["t_wind"], 1e-9),
        "host_cpu": model, "host_cores_total": os.cpu_count(), "wall_s": wall, "init_s": j["t_init"],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--nwater", type=int, default=None)
    ap.add_argument("--nwind", type=int, default=None)
    ap.add_argument("--engine", type=str, default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-ticks", type=int, default=1)
    args = ap.parse_args()

    import torch
    from soilmachine_amd import distrib
    rank, local_rank, world = distrib.env_rank()
    dist = distrib.init(world, local_rank, use_cuda=torch.cuda.is_available())

    from soilmachine_amd import capi
    from soilmachine_amd.machine import SoilMachine
    from soilmachine_amd.soilfile import loadsoil

    cfg = loadsoil(os.path.join(ROOT, "soilmachine_amd", "soils", SOIL))
    nwater = cfg.NWATER if args.nwater is None else args.nwater
    nwind = cfg.NWIND if args.nwind is None else args.nwind
    n = args.size
    engine = {"auto": capi.ENGINE_SPECULATIVE, "serial": capi.ENGINE_SERIAL, "spec": capi.ENGINE_SPECULATIVE,
              "batched": capi.ENGINE_BATCHED}[args.engine]
    pool = 3 * n * n
    seed = distrib.replica_seed(rank)   # replicas: independent worlds, SEED = rank (rank 0 == BASELINE's SEED 0)
    sm = SoilMachine(cfg, n, seed=seed, nwater=nwater, nwind=nwind, pool=pool, device=local_rank, engine=engine)
    m = sm.map

    def barrier():
        m.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sm.tick(args.warmup, sync=True)
    c0 = m.counters()
    m.timing_reset()
    barrier()
    t0 = time.perf_counter()
    sm.tick(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    c1 = m.counters()
    tm = m.timing()

    steps_w = c1["steps_water_all"] - c0["steps_water_all"]
    steps_d = c1["steps_wind"] - c0["steps_wind"]
    steps = steps_w + steps_d
    dev = torch.device("cuda", local_rank) if torch.cuda.is_available() else torch.device("cpu")
    tmax, total_steps = distrib.aggregate(dist, dt, steps, dev)

    if rank == 0:
        # dominant kernel = the particle kernel with the largest summed HIP-event time (every launch is bracketed
        # by its own event pair on the context's stream inside libsoilmx)
        spec = engine == capi.ENGINE_SPECULATIVE
        kspec = "k_spec_exec" if os.environ.get("SMX_COOP", "1") == "0" else "k_spec_exec_coop"   # soilmx.hip spec_phase
        kernels = {"water": (tm["ms_kernel_water"], tm["launches_kernel_water"], steps_w * B_WATER_STEP,
                             f"{kspec}<water>" if spec else "k_water_serial"),
                   "wind": (tm["ms_kernel_wind"], tm["launches_kernel_wind"], steps_d * B_WIND_STEP,
                            f"{kspec}<wind>" if spec else "k_wind_serial")}
        dom = max(kernels, key=lambda k: kernels[k][0])
        ms, launches, nbytes, kname = kernels[dom]
        ach = (nbytes / max(launches, 1)) / (ms / max(launches, 1) * 1e-3) / 1e9 if ms > 0 else 0.0
        traffic = None                                        # HBM bytes per launch from the PMC passes (profiles/r01_pmc.json)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))
            if pmc.get("workload_size") == n and kname in pmc.get("kernels", {}):
                traffic = pmc["kernels"][kname]["hbm_bytes_per_launch"]
        except (OSError, ValueError):
            pass
        roofline = {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                    "avg_launch_ms": ms / max(launches, 1), "launches": launches,
                    "algorithmic_bytes_per_launch": nbytes / max(launches, 1),
                    "note": "exact engine: the particle kernels are instruction-issue / dependent-latency bound, "
                            "not bandwidth bound (DESIGN.md); the O(cells) kernels are the HBM-streaming ones"}
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(n, nwater, nwind, 0, args.cpu_ticks, pool)
        out = {
            "metric": "particle-steps/s", "value": total_steps / tmax, "unit": "particle-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * tmax / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{n}x{n} layermap, soil/{SOIL}, SEED 0, nwater={nwater}, nwind={nwind} per tick, "
                                   f"exact (reference-order) engine, tick = water + grid pass + wind + frequency",
                       "engine": args.engine, "parallelism": "replicas" if world > 1 else "single"},
            "ms_per_tick": 1e3 * tmax / args.steps,
            "phase_ms_per_tick": {"water": tm["ms_water"] / args.steps, "grid": tm["ms_grid"] / args.steps,
                                  "wind": tm["ms_wind"] / args.steps, "freq": tm["ms_freq"] / args.steps},
            "steps_per_tick": {"water": steps_w / args.steps, "wind": steps_d / args.steps,
                               "nested_particles": (c1["nested_particles"] - c0["nested_particles"]) / args.steps},
            "engine_stats": dict({k: c1[k] - c0[k] for k in ("spec_rounds", "spec_aborts", "spec_fallbacks", "spec_executed")}, **m.batch_stats()),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
