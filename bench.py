#!/usr/bin/env python3
"""bench.py -- particle-steps/s and ms/tick of the SoilMachine particle-transport hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0 -- the LAST line of stdout, compact
(~4 KB: compact_line below); the detailed record (per-tick entries, extra lines, parity blocks, LBM) goes to a side file
(gpurun_out/bench_detail_n<N>.json, named on the line).
  * a "step" of the bench = one tick of SoilMachine.cpp:283-329 (water particles -> grid pass -> wind particles ->
    frequency update) over the synthetic 4096^2 layermap of BASELINE.json configs[3]
    (soil/rockgravelpebblessand.soil: 4 soils, 2 noise layers, SCALE 120, SEED 0); terrain comes from
    Layermap::initialize on the device, state is resident in HBM before the timed region.
  * value = particle-steps/s, whole job: a particle-step is one WaterParticle/WindParticle move()==true followed by its
    interact(), top-level AND nested particles. The unmodified reference can only count top-level steps,
    so the CPU baseline is top-level steps/s and the GPU line carries `value_top_level` next to `value` for that ratio.
  * headline line: the RELAXED schedule of the batched engine (SMX_ENGINE_RELAXED, csrc/soil_relax.h) in its DEFAULT configuration (eight
    water generations since round 6, nested particles inside their parents, 8x8 grid tiles: the one measured against the reference) with
    AREA-SCALED particle counts (NWATER = 250*(N/256)^2 = 64 000, NWIND = 16 000 at 4096^2). Deterministic, but NOT the
    reference's particle order: parity tier P2 (statistical) -- `parity_vs_reference` measures it in the run against the reference itself
    (1024^2, 16 control streams) and `parity_gate` applies the tests' rule to it and to the committed record at the headline's size
    (profiles/r06_p2_reference_4096.json, 33 reference streams); `parity_vs_exact` against the exact engine on the device. The EXACT engine (bit-identical to the reference, the parity anchor) is reported under "lines" at the soil
    file's own counts and under config.bit_exact_on_this_workload for the headline's counts, next to the batched schedule of round 2 and
    the 1024^2 default.soil lines of BASELINE configs[1] and [2].
  * every line carries the state digest after its timed region (section count, type hash, sum of heights, rand() draws).
  * "roofline": algorithmic bytes of the dominant kernel / its HIP-event duration (events on the library's own stream,
    bracketing nothing but launches of that kernel) vs the 8 TB/s HBM peak; "phases" gives the same for the O(cells)
    streaming kernels. `traffic` comes from a separate rocprofv3 --pmc pass of the same command (the newest profiles/r*_pmc.json whose workload matches), labelled as such.
    achieved = bytes of ALL launches / ALL launches / the average duration of the bracketed launches (kernel_roofline).
  * "cpu_baseline": the reference's own CPU loop (oracle/_ref/soil_ref_lean = the unmodified reference headers,
    render-only Layermap::update elided) timed on this box's host on THE SAME tick the timed window starts with: the state after the
    warm-up is checkpointed (smx_save, outside the timed region) and the reference continues it (ref_driver --load) at the same counts,
    1 core pinned; an all-cores line (independent processes from that state -- the reference has no threads) and the "as-is" line.
  * "parity_vs_reference": z-scores of the timed engine against the reference itself (tools/p2_reference.py: 1024^2, 16 control streams).
  * N > 1: one process per GPU. The throughput engines run ONE map in N column strips, driven inside the library (smx_strips_*: halo
    columns + migrating particles point to point with ncclSend/ncclRecv on device buffers, live count by ncclAllReduce). Default "scaling": "weak" -- the map is
    (N*size) x size, every GPU owns a size x size strip and its share of N x the particles, so per-GPU work is the N=1 line's;
    --strong splits the N=1 map instead ("strong"). The exact engines do not shard (sequential semantics): with --engine
    spec/serial the ranks run independent replicas (SEED = rank, "weak").
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
B_FLOOD = 800           # algorithmic bytes per flood (water.h:123-145 + the water-table cascade it starts): the 5x5 cells of 32-byte records its reach covers
B_ACTIVE_CELL = 32      # grid pass: one top record per active cell
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec peak


def host_cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def run_ref_lean(soil: str, size: int, nwater: int, nwind: int, seed: int, ticks: int, pool: int, core: int | None, exe_name: str = "soil_ref_lean",
                 load: str | None = None, rand_seed: int | None = None):
    exe = os.path.join(ROOT, "oracle", "_ref", exe_name)
    cmd = [exe, "--soil", os.path.join(ROOT, "soilmachine_amd", "soils", soil), "--seed", str(seed), "--size", str(size),
           "--ticks", str(ticks), "--nwater", str(nwater), "--nwind", str(nwind), "--pool", str(pool)]
    if load:
        cmd += ["--load", load]                                  # continue from a state written by smx_save (oracle/ref_driver.cpp --load)
    if rand_seed is not None:
        cmd += ["--rand-seed", str(rand_seed)]
    if core is not None:
        cmd = ["taskset", "-c", str(core)] + cmd
    return subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)


def parse_ref(out: str):
    for line in out.splitlines():
        if line.startswith("JSON "):
            return json.loads(line[5:])
    return None


def cpu_baseline(soil: str, size: int, nwater: int, nwind: int, ticks: int, pool: int, all_cores: bool, load: str | None = None, tick0: int = 0,
                 gpu_same_tick: dict | None = None, core: int = 0) -> dict | None:
    """Time the reference's own CPU path on this host. Test/bench infrastructure only.
    load = a checkpoint of the GPU line's state after its warm-up (smx_save): the CPU then runs the SAME tick(s) the GPU line times first,
    at the SAME particle counts. Without it: a bounded sample from the initial terrain.
    The unmodified reference can only count TOP-LEVEL steps (nested particles run inside WaterParticle::cascade):
    cpu value = top-level steps/s, to be compared with the GPU line's `value_top_level` (same definition)."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "soil_ref_lean")):
        return None
    t0 = time.time()
    p = run_ref_lean(soil, size, nwater, nwind, 0, ticks, pool, core, load=load)   # (pinned: two samples taken side by side sit on different cores)
    j = parse_ref(p.communicate()[0])
    if j is None:
        p = run_ref_lean(soil, size, nwater, nwind, 0, ticks, pool, None, load=load)
        j = parse_ref(p.communicate()[0])
    if j is None:
        return None
    wall = time.time() - t0
    t_tick = j["t_water"] + j["t_grid"] + j["t_wind"] + j["t_freq"]
    steps = j["steps_water_top"] + j["steps_wind"]
    window = (f"tick(s) {tick0 + 1}..{tick0 + ticks} continued from the state the GPU line's {tick0} warm-up ticks produced (smx_save -> ref_driver --load): the same state, "
              f"the same counts and the first tick(s) of the GPU line's timed window") if load else f"{ticks} tick(s) from the initial terrain"
    out = {
        "value": steps / t_tick if t_tick > 0 else None, "unit": "particle-steps/s", "cores": 1, "kind": "reference",
        "sample": f"{window}; {size}x{size}, {soil}, nwater={nwater}, nwind={nwind} with oracle/_ref/soil_ref_lean "
                  f"(the reference's own headers, render-only Layermap::update elided), 1 thread pinned; steps = TOP-LEVEL "
                  f"move()==true (nested particles are not countable in the unmodified reference): compare with the GPU "
                  f"line's value_top_level; whole tick in the denominator",
        "ms_per_tick": 1e3 * t_tick / ticks, "ms_water": 1e3 * j["t_water"] / ticks, "ms_grid": 1e3 * j["t_grid"] / ticks,
        "ms_wind": 1e3 * j["t_wind"] / ticks, "ms_freq": 1e3 * j["t_freq"] / ticks,
        "particle_phase_steps_per_s": steps / max(j["t_water"] + j["t_wind"], 1e-9),
        "steps_per_tick": steps / ticks, "nested_per_tick": j.get("nested", 0) / ticks, "host_cpu": host_cpu_model(), "host_cores_total": os.cpu_count(), "wall_s": wall,
        "init_s": j["t_init"],
    }
    if gpu_same_tick:
        out["gpu_same_tick"] = gpu_same_tick                      # the GPU line's own record of that tick (ms, steps)
    if not load:
        out["regime_note"] = ("NOT the GPU line's tick window: a bounded sample from the initial terrain (dry map: the O(cells) grid pass dominates ms_per_tick); "
                              "particle_phase_steps_per_s is the like-for-like figure")
    procs = []
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "soil_ref")):   # SURVEY 8d "as-is": Layermap::update's render work kept (layermap.h:475-549)
        procs.append(("as_is", run_ref_lean(soil, size, nwater, nwind, 0, ticks, pool, 0, "soil_ref", load=load)))
    nproc = 0
    t1 = time.time()
    if all_cores:
        # the reference is single-threaded: "all cores" = independent processes, as many as fit the box's memory (each holds
        # its own layermap + POOLSIZE): other SEEDs from the initial terrain, or -- from a checkpoint -- the same state under other rand() streams
        ncpu = os.cpu_count() or 1
        try:
            mem_gb = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2**30
        except (ValueError, OSError):
            mem_gb = 64.0
        per_proc_gb = (pool * 56 + size * size * 60) / 2**30 + 0.5
        nproc = int(max(1, min(ncpu - 1, 32, (0.5 * mem_gb) // per_proc_gb)))
        for q in range(nproc):
            procs.append(("all", run_ref_lean(soil, size, nwater, nwind, 0 if load else q, ticks, pool, (q + 1) % ncpu, load=load, rand_seed=(5000 + q) if (load and q) else None)))
    js = [(k, parse_ref(q.communicate()[0])) for k, q in procs]
    for k, ja in js:
        if k == "as_is" and ja is not None:
            ta = ja["t_water"] + ja["t_grid"] + ja["t_wind"] + ja["t_freq"]
            out["as_is"] = {"value": (ja["steps_water_top"] + ja["steps_wind"]) / ta if ta > 0 else None, "unit": "particle-steps/s", "cores": 1, "kind": "reference",
                            "sample": "the same sample with oracle/_ref/soil_ref: the reference's headers UNCHANGED, Layermap::update(ivec2, Vertexpool&) doing its "
                                      "render-side work after every edit as in the real application" + (" (timed while the all-cores processes run)" if nproc else ""),
                            "ms_per_tick": 1e3 * ta / ticks, "ms_water": 1e3 * ja["t_water"] / ticks, "ms_grid": 1e3 * ja["t_grid"] / ticks, "ms_wind": 1e3 * ja["t_wind"] / ticks}
    ja = [x for k, x in js if k == "all" and x is not None]
    if ja:
        agg = sum((x["steps_water_top"] + x["steps_wind"]) / max(x["t_water"] + x["t_grid"] + x["t_wind"] + x["t_freq"], 1e-9) for x in ja)
        out["all_cores"] = {"value": agg, "unit": "particle-steps/s (sum over independent processes)", "cores": len(ja),
                            "note": f"{len(ja)} independent soil_ref_lean processes, one pinned core each, the same sample"
                                    f"{' under other rand() streams' if load else ' (SEED 0..' + str(len(ja) - 1) + ')'}; the reference has no threads", "wall_s": time.time() - t1}
    return out


def measure(cfg, soil, n, seed, nwater, nwind, dowind, engine, pool, device, steps, warmup, barrier=None, rand_seed=None, keep_heights=False, save_after_warmup=None,
            water_generations=None, save_after_window=None, water_stagger=None):
    """warmup ticks, then `steps` timed ticks on a fresh context. Returns a dict (rank-local numbers).
    rand_seed: re-seed the libc stream after the terrain is built (same map, another rand() stream: the parity control)."""
    from soilmachine_amd.machine import SoilMachine
    sm = SoilMachine(cfg, n, seed=seed, nwater=nwater, nwind=nwind, dowind=dowind, pool=pool, device=device, engine=engine)
    m = sm.map
    if rand_seed is not None:
        m._chk(m.L.smx_srand(m.h, int(rand_seed)))
    if water_generations is not None:
        m.set_water_generations(water_generations)
    if water_stagger is not None:
        m.set_water_stagger(water_stagger)
    k_gen = m.water_generations()                               # the value in force (smx_get_water_generations), reported as measured
    k_stagger = m.water_stagger()
    sm.tick(warmup, sync=True)
    if save_after_warmup:                                       # the state the timed window starts from: what cpu_baseline continues (outside the timed region)
        try:
            m.save(save_after_warmup)
        except Exception:
            save_after_warmup = None
    c0 = m.counters()
    b0 = m.batch_stats()
    m.timing_reset()
    if barrier:
        barrier(m)
    t0 = time.perf_counter()
    per_tick = []                                               # one entry per timed tick: what the driver's window (ticks W+1 .. W+K) is made of
    tl, cl, tml = t0, c0, {k: 0.0 for k in ("ms_water", "ms_grid", "ms_wind", "ms_freq")}
    for k in range(steps):
        sm.tick(1)
        m.sync()                                                # (a tick ends with host synchronisations of its own: this adds none that matter)
        tn, cn, tmn = time.perf_counter(), m.counters(), m.timing()
        per_tick.append({"tick": warmup + k + 1, "ms": 1e3 * (tn - tl),
                         "ms_water": tmn["ms_water"] - tml["ms_water"], "ms_grid": tmn["ms_grid"] - tml["ms_grid"], "ms_wind": tmn["ms_wind"] - tml["ms_wind"],
                         "steps_water": cn["steps_water_all"] - cl["steps_water_all"], "steps_wind": cn["steps_wind"] - cl["steps_wind"],
                         "floods": cn["floods"] - cl["floods"], "nested": cn["nested_particles"] - cl["nested_particles"]})
        tl, cl, tml = tn, cn, tmn
    if barrier:
        barrier(m)
    else:
        m.sync()
    dt = time.perf_counter() - t0
    c1, tm, b1 = m.counters(), m.timing(), m.batch_stats()
    steps_w = c1["steps_water_all"] - c0["steps_water_all"]
    steps_d = c1["steps_wind"] - c0["steps_wind"]
    r = {
        "dt": dt, "steps_w": steps_w, "steps_d": steps_d, "tm": tm, "digest": m.digest(),
        "steps_w_top": c1["steps_water_top"] - c0["steps_water_top"],
        "nested": c1["nested_particles"] - c0["nested_particles"], "floods": c1["floods"] - c0["floods"],
        "pool_overflow": c1["pool_overflow"], "per_tick": per_tick, "water_generations": k_gen, "water_stagger": k_stagger,
        "flood_nested_steps": c1.get("flood_nested_steps", 0) - c0.get("flood_nested_steps", 0),
        "grid_nested_steps": c1.get("grid_nested_steps", 0) - c0.get("grid_nested_steps", 0),
        "grid_active_cells": c1["grid_active_cells"] - c0["grid_active_cells"],
        "engine_stats": dict({k: c1[k] - c0[k] for k in ("spec_rounds", "spec_aborts", "spec_fallbacks", "spec_executed")},
                             **{k: b1[k] - b0[k] for k in b1}),
    }
    if keep_heights:
        r["heights"] = m.heights()
    if save_after_window:
        # OUTSIDE the timed region: the state the window ends with is checkpointed and ONE more tick runs on the device -- the tick the CPU
        # baseline's second sample continues from that checkpoint (the tick grows inside the window: lakes accumulate)
        try:
            m.save(save_after_window)
            ca = m.counters(); ta = time.perf_counter()
            sm.tick(1); m.sync()
            cb = m.counters()
            r["post_tick"] = {"tick": warmup + steps + 1, "ms": 1e3 * (time.perf_counter() - ta), "steps_water": cb["steps_water_all"] - ca["steps_water_all"],
                              "steps_water_top": cb["steps_water_top"] - ca["steps_water_top"], "steps_wind": cb["steps_wind"] - ca["steps_wind"]}
        except Exception as e:
            r["post_tick"] = {"error": str(e)}
    sm.map.close()
    return r


def parity_block(cfg, soil, n, nwater, nwind, pool, device, ticks, nstreams, engines):
    """Device-side parity of the throughput schedules against the EXACT engine: every run starts from the same
    terrain (SEED 0) and ticks `ticks` times; `control` = the exact engine itself with other rand() streams (what the reference's
    own chaos does to the same figures). Heights in map units; *_rel = value / exact - 1."""
    import numpy as np
    from soilmachine_amd import capi
    def run(engine, rand_seed=None):
        r = measure(cfg, soil, n, 0, nwater, nwind, nwind > 0, engine, pool, device, ticks, 0, rand_seed=rand_seed, keep_heights=True)
        return r
    t0 = time.time()
    ex = run(capi.ENGINE_SPECULATIVE)
    def versus(r):
        d = r["heights"] - ex["heights"]
        return {"height_rmse": float(np.sqrt(np.mean(d * d))), "height_max_abs": float(np.max(np.abs(d))),
                "nsec_rel": r["digest"]["nsec"] / ex["digest"]["nsec"] - 1.0, "sumh_rel": r["digest"]["sumh"] / ex["digest"]["sumh"] - 1.0,
                "steps_water_rel": r["steps_w"] / max(ex["steps_w"], 1) - 1.0, "steps_wind_rel": (r["steps_d"] / ex["steps_d"] - 1.0) if ex["steps_d"] else None,
                "floods": r["floods"], "nested": r["nested"], "ms_per_tick": 1e3 * r["dt"] / ticks}
    out = {"workload": f"{n}x{n}, soil/{soil}, SEED 0, nwater={nwater}, nwind={nwind}, ticks 1-{ticks} from the initial terrain",
           "exact": {"nsec": ex["digest"]["nsec"], "sumh": ex["digest"]["sumh"], "steps_water": ex["steps_w"], "steps_wind": ex["steps_d"], "floods": ex["floods"],
                     "nested": ex["nested"], "ms_per_tick": 1e3 * ex["dt"] / ticks},
           "control_exact_other_rand_stream": [versus(run(capi.ENGINE_SPECULATIVE, rand_seed=1000 + k)) for k in range(nstreams)]}
    for name, eng in engines:
        out[name] = versus(run(eng))
    ctl = [c["height_rmse"] for c in out["control_exact_other_rand_stream"]]
    out["control_height_rmse_max"] = max(ctl) if ctl else None
    out["wall_s"] = time.time() - t0
    return out


def measure_strips(cfg, dimx, n, nwater, nwind, pool, device, steps, warmup, dist, barrier, torch, engine):
    """ONE map sharded into world-size column strips, driven INSIDE the library (csrc/soil_strips_host.h, smx_strips_*): halo columns
    and migrating particles go from the pack kernel to the neighbour's unpack kernel over RCCL (ncclSend / ncclRecv on device buffers,
    backend nccl) -- or through host callbacks on torch.distributed gloo when there is no RCCL (CPU-side tests)."""
    from soilmachine_amd.machine import Layermap
    from soilmachine_amd.strips import LibStripRank, StripGeometry, TorchComm
    world = dist.get_world_size()
    geom = StripGeometry(dimx, world)
    held = geom.held(dist.get_rank(), n)                            # this rank's context holds its strip + halo only (smx_create_strip)
    if held is not None:
        pool = max(4_000_000, 3 * (held[1] - held[0]) * n)
    m = Layermap(cfg, dimx, n, seed=0, pool=pool, device=device, engine=engine, x_range=held)
    nccl = dist.get_backend() == "nccl"
    dev = torch.device("cuda", device) if (torch.cuda.is_available() and nccl) else None
    rk = LibStripRank(m, TorchComm(dist, dev), geom, transport="rccl" if nccl else "host")
    for _ in range(warmup):
        rk.tick(nwater, nwind, True, nwind > 0)
    m.sync()
    c0, b0 = m.counters(), m.batch_stats()
    m.timing_reset()
    barrier(m)
    t0 = time.perf_counter()
    for _ in range(steps):
        rk.tick(nwater, nwind, True, nwind > 0)
    barrier(m)
    dt = time.perf_counter() - t0
    c1, tm, b1 = m.counters(), m.timing(), m.batch_stats()
    r = {"dt": dt, "steps_w": c1["steps_water_all"] - c0["steps_water_all"], "steps_d": c1["steps_wind"] - c0["steps_wind"], "tm": tm,
         "digest": {"note": "strips: each rank holds its own strip; no global digest in the timed run"},
         "steps_w_top": c1["steps_water_top"] - c0["steps_water_top"], "water_generations": m.water_generations(), "water_stagger": m.water_stagger(),
         "flood_nested_steps": c1.get("flood_nested_steps", 0) - c0.get("flood_nested_steps", 0),
         "grid_nested_steps": c1.get("grid_nested_steps", 0) - c0.get("grid_nested_steps", 0),
         "grid_active_cells": c1["grid_active_cells"] - c0["grid_active_cells"],
         "nested": c1["nested_particles"] - c0["nested_particles"], "floods": c1["floods"] - c0["floods"], "pool_overflow": c1["pool_overflow"],
         "engine_stats": dict({k: b1[k] - b0[k] for k in b1}, strip_stats=rk.stats, held_columns=list(held) if held else [0, dimx], pool_sections=pool, transport="rccl (ncclSend/ncclRecv, device buffers)" if nccl else "host callbacks (gloo)")}
    rk.detach()
    m.close()
    return r


def measure_lbm(device, dims=(512, 320, 512), steps=40):
    """Row f4 (LBM wind, csrc/soil_lbm.h): cell updates/s of the fused collide+stream kernel on a lattice that does not fit the
    caches, HIP events around the step kernels; 176 algorithmic bytes per cell and step (19 + 19 distributions, mask, rho, v)."""
    import numpy as np
    from soilmachine_amd.lbm import LbmWind
    nx, ny, nz = dims
    g = LbmWind(nx, ny, nz, device=device)
    b = np.zeros((nx, ny, nz), np.float32)
    b[:, :ny // 8, :] = 1.0
    g.set_boundary(b); g.initialize(); g.step(5); g.timing(reset=True)
    g.step(steps)
    tm = g.timing()
    g.close()
    us = 1e3 * tm["ms_steps"] / max(tm["steps"], 1)
    gbs = nx * ny * nz * 176 / (us * 1e-6) / 1e9
    return {"kernel": "k_lbm_step (D3Q19 TRT collide+stream fused; reference: lbmwind.h:174-188, shader/LBM/*.cs)", "lattice": f"{nx}x{ny}x{nz}",
            "steps": int(tm["steps"]), "us_per_step": us, "cell_updates_per_s": nx * ny * nz / (us * 1e-6),
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_cell_step": 176}}


def headline_parity_record(k_gen, strips=1, stagger=0):
    """The committed record of tools/p2_reference.py at the headline's own size and counts (4096^2, 64 000 + 16 000, 32 control streams) for the
    configuration this run times: read from the newest profiles/r*_p2_reference_4096.json that holds it -- never a literal."""
    pdir = os.path.join(ROOT, "profiles")
    name = f"k{k_gen}" + (f"e{stagger}" if stagger else "") + (f"s{strips}" if strips > 1 else "")
    for pf in sorted((f for f in os.listdir(pdir) if f.endswith("_p2_reference_4096.json")), reverse=True):
        try:
            rec = json.load(open(os.path.join(pdir, pf)))
            b = (rec.get("variants") or {}).get(name)
            if b is None and not rec.get("variants") and rec.get("water_generations") == k_gen and strips == 1 and not stagger:
                b = rec
            if b is None:
                continue
            return {"file": f"profiles/{pf}", "variant": name, "workload": rec.get("workload"), "controls": rec.get("controls"), "outliers": b.get("outliers"),
                    "gate": b.get("gate"), "z": {k: round(v["z"], 2) for k, v in b["figures"].items()},
                    "reference_streams_passing_the_gate": (rec.get("calibration") or {}).get("_streams_passing_the_gate")}
        except (OSError, ValueError, KeyError):
            continue
    return None


def kernel_roofline(nbytes, ms_timed, launches_timed, launches_all):
    """One particle kernel against the HBM roofline: ALGORITHMIC bytes per launch (the bytes of every launch / the number of ALL launches)
    over the AVERAGE launch duration (HIP events around the bracketed launches only: the relaxed water step kernel is bracketed every 8th
    launch, csrc/soilmx.hip run_chunk). Round 4 divided the bytes of all launches by the time of the bracketed ones."""
    if not launches_timed or ms_timed <= 0:
        return None
    n_all = launches_all or launches_timed
    avg_s = ms_timed * 1e-3 / launches_timed
    per_launch = nbytes / n_all
    g = per_launch / avg_s / 1e9
    return {"achieved": g, "unit": "GB/s", "frac": g / HBM_PEAK_GBS, "launches": int(n_all), "launches_timed": int(launches_timed),
            "avg_launch_us": 1e6 * avg_s, "algorithmic_bytes_per_launch": per_launch}


def line_of(r, name, workload, engine_name, steps):
    dt, tm = r["dt"], r["tm"]
    tot = r["steps_w"] + r["steps_d"]
    return {
        "name": name, "workload": workload, "engine": engine_name, "value": tot / dt, "unit": "particle-steps/s",
        "value_top_level": (r["steps_w_top"] + r["steps_d"]) / dt,
        "ms_per_tick": 1e3 * dt / steps,
        "phase_ms_per_tick": {"water": tm["ms_water"] / steps, "grid": tm["ms_grid"] / steps, "wind": tm["ms_wind"] / steps,
                              "freq": tm["ms_freq"] / steps},
        "steps_per_tick": {"water": r["steps_w"] / steps, "wind": r["steps_d"] / steps, "nested_particles": r["nested"] / steps,
                           "floods": r["floods"] / steps},
        "engine_stats": r["engine_stats"], "digest": r["digest"], "pool_overflow": r["pool_overflow"],
        # particle kernels of this line against the HBM roofline: algorithmic bytes (192 B per water step, 256 B per wind step,
        # SURVEY 8d) over the HIP-event time of nothing but those kernels
        "roofline_particles": {k: kernel_roofline(nb, ms, ln, lt) for k, nb, ms, ln, lt in (
            ("water", r["steps_w"] * B_WATER_STEP, tm["ms_kernel_water"], tm["launches_kernel_water"], tm.get("launches_step_water", 0)),
            ("wind", r["steps_d"] * B_WIND_STEP, tm["ms_kernel_wind"], tm["launches_kernel_wind"], tm.get("launches_step_wind", 0)))},
        "parity": "bit-exact vs the reference (tests/golden)" if engine_name not in ("batched", "relaxed") else
                  ("P2: deterministic, bit-exact vs its CPU restatement, statistical vs the reference" if engine_name == "batched" else
                   "P2: deterministic (device == host threads, any interleaving), statistical vs the reference"),
    }


def wet_regime_line(dcfg, device, engine, eng_name, warm=100, ticks=20, cpu_ticks=3):
    """The established wet regime at the reference's own map size: 256^2 default.soil, NWATER = 250, lakes from tick ~60 -- `ticks`
    timed ticks after `warm`, and the reference's CPU loop (1 core) on the first `cpu_ticks` of them from the very same state (checkpoint)."""
    n, nw = 256, 250
    ckpt = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"soilmx_bench_wet_{os.getpid()}.smx")
    r = measure(dcfg, "default.soil", n, 0, nw, 0, False, engine, 10_000_000, device, ticks, warm, save_after_warmup=ckpt)
    ln = line_of(r, "wet_regime_256_default", f"256x256, soil/default.soil, NWATER=250, NWIND=0, ticks {warm + 1}-{warm + ticks} (the established wet regime: lakes from "
                 f"tick ~60; the map of BASELINE configs[0])", eng_name, ticks)
    ln["steps"], ln["warmup"] = ticks, warm
    cpu = None
    if os.path.exists(ckpt):
        try:
            cpu = cpu_baseline("default.soil", n, nw, 0, cpu_ticks, 10_000_000, all_cores=False, load=ckpt, tick0=warm)
        finally:
            try:
                os.remove(ckpt)
            except OSError:
                pass
    ln["cpu_baseline"] = ({"value": cpu["value"], "unit": cpu["unit"], "cores": 1, "kind": "reference", "ms_per_tick": cpu["ms_per_tick"], "sample": cpu["sample"][:200]} if cpu else None)
    return ln


def _r(x, nd=4):
    """round floats for the compact line (NaN / inf -> None: strict JSON)"""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{nd + 2}g}") if abs(x) < 1 else round(x, nd)
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def compact_line(d: dict, detail_path) -> dict:
    """The ONE line the driver parses: the contract's keys + roofline + cpu_baseline, nothing that was not measured in this run except
    `roofline.traffic` (a separate rocprofv3 --pmc job, labelled). Everything else is in the side file."""
    rf, cpu, pref = d["roofline"], d.get("cpu_baseline"), d.get("parity_vs_reference")
    out = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {k: d["config"].get(k) for k in ("workload", "engine", "water_generations", "water_stagger", "parallelism", "parity")}
    out["value_top_level"] = d.get("value_top_level")
    out["phase_ms_per_tick"] = d["phase_ms_per_tick"]
    out["steps_per_tick"] = d["steps_per_tick"]
    out["digest"] = d["digest"] if "nsec" in d["digest"] else None
    ph = {}
    for k, v in (rf.get("phases") or {}).items():
        if v:
            ph[k] = {"kernel": v["kernel"].split(" ")[0], "achieved": v["achieved"], "frac": v["frac"], "avg_launch_us": v.get("avg_launch_us"), "launches": v.get("launches")}
            if v.get("ms_per_tick") is not None:
                ph[k]["ms_per_tick"] = v["ms_per_tick"]
    out["roofline"] = {"bound": rf["bound"], "kernel": rf["kernel"].split(" ")[0], "achieved": rf["achieved"], "peak": rf["peak"], "unit": rf["unit"], "frac": rf["frac"],
                       "traffic": rf["traffic"], "traffic_source": (rf["traffic_source"] or "")[:120], "avg_launch_ms": rf["avg_launch_ms"], "launches": rf["launches"],
                       "launches_timed": rf.get("launches_timed"), "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"],
                       "whole_tick": {k: rf["whole_tick"][k] for k in ("achieved", "frac", "algorithmic_bytes_per_tick")}, "phases": ph,
                       "time_share_of_tick": rf.get("time_share_of_tick")}
    pt = d.get("per_tick") or []
    if pt:
        out["ms_first_tick"], out["ms_last_tick"] = pt[0]["ms"], pt[-1]["ms"]      # the trend inside the timed window (lakes keep growing: VERDICT r5 weak 5)
    if cpu:
        out["cpu_baseline"] = {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"], "ms_per_tick": cpu["ms_per_tick"],
                               "host_cpu": cpu.get("host_cpu"), "host_cores_total": cpu.get("host_cores_total"),
                               "sample": cpu["sample"][:300], "steps_counted": "top-level (compare with value_top_level)",
                               "all_cores": ({"value": cpu["all_cores"]["value"], "cores": cpu["all_cores"]["cores"]} if cpu.get("all_cores") else None),
                               "after_window": ({"value": cpu["after_window"]["value"], "ms_per_tick": cpu["after_window"]["ms_per_tick"],
                                                 "gpu_ms_same_tick": (cpu["after_window"].get("gpu_same_tick") or {}).get("ms")} if cpu.get("after_window") else None)}
    else:
        out["cpu_baseline"] = None
    if pref and "error" not in pref:
        zs = {k: v.get("z") for k, v in (pref.get("figures") or {}).items()}
        out["parity_vs_reference"] = {"workload": pref.get("workload", "")[:160], "controls": pref.get("controls"), "outliers": pref.get("outliers"),
                                      "z": zs, "gate": pref.get("gate"), "skipped": pref.get("skipped"),
                                      "headline_size_record": pref.get("headline_size_record")}
    elif pref:
        out["parity_vs_reference"] = {"error": str(pref.get("error"))[:200]}
    out["parity_gate"] = d.get("parity_gate")
    out["work_dropped"] = d["work_dropped"]
    if d.get("lines"):
        out["lines"] = [dict({"name": l_["name"], "value": l_.get("value"), "ms_per_tick": l_.get("ms_per_tick")},
                             **({"cpu_1core_ms_per_tick": l_["cpu_baseline"]["ms_per_tick"]} if l_.get("cpu_baseline") else {}),
                             **({"parity_gate": ("pass" if l_["parity_gate_headline_size_record"]["gate"]["pass"] else "fail")} if l_.get("parity_gate_headline_size_record") else {})) for l_ in d["lines"] if l_]
    out["detail_file"] = detail_path
    return _r(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--soil", type=str, default=SOIL)
    ap.add_argument("--nwater", type=int, default=None)
    ap.add_argument("--nwind", type=int, default=None)
    ap.add_argument("--engine", type=str, default="relaxed", choices=["auto", "serial", "spec", "batched", "relaxed"])
    ap.add_argument("--only", type=str, default=None, help="measure just this extra line (by name) and print it")
    ap.add_argument("--full", action="store_true", help="all extra lines (16384^2, both throughput schedules on every map)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lines", action="store_true", help="only the headline measurement")
    ap.add_argument("--cpu-ticks", type=int, default=1)
    ap.add_argument("--strong", action="store_true", help="N > 1: keep the map at --size^2 and split it (default: the map grows with N)")
    ap.add_argument("--water-generations", type=int, default=None, help="throughput engines: the tick's water particles as k consecutive generations (smx_set_water_generations; "
                                                                        "default: the engine's own (8 since round 6: inside the reference's spread on all eight figures at the headline workload, "
                                                                        "profiles/r06_p2_reference_4096.json); 4 = round 5's default, 1 = round 4's schedule: faster, outside it)")
    ap.add_argument("--water-stagger", type=int, default=None, help="relaxed engine: the water generations born this many epochs apart inside ONE phase (smx_set_water_stagger; 0 = one after the other; default: the library's)")
    args = ap.parse_args()

    import torch
    from soilmachine_amd import distrib
    rank, local_rank, world = distrib.env_rank()
    dist = distrib.init(world, local_rank, use_cuda=torch.cuda.is_available())
    if torch.cuda.is_available():
        local_rank %= max(torch.cuda.device_count(), 1)          # (several ranks may share a device when SMX_DIST_BACKEND=gloo)

    from soilmachine_amd import capi
    from soilmachine_amd.soilfile import loadsoil

    soil = args.soil
    cfg = loadsoil(os.path.join(ROOT, "soilmachine_amd", "soils", soil))
    n = args.size
    eng_name = "spec" if args.engine == "auto" else args.engine
    ENG = {"serial": capi.ENGINE_SERIAL, "spec": capi.ENGINE_SPECULATIVE, "batched": capi.ENGINE_BATCHED, "relaxed": capi.ENGINE_RELAXED}
    engine = ENG[eng_name]
    throughput = eng_name in ("batched", "relaxed")
    area = (n / 256.0) ** 2
    if throughput:                  # throughput configuration: area-scaled counts, same ratio as the .soil file
        nwater = int(250 * area) if args.nwater is None else args.nwater
        nwind = int(250 * area * cfg.NWIND / max(cfg.NWATER, 1)) if args.nwind is None else args.nwind
    else:
        nwater = cfg.NWATER if args.nwater is None else args.nwater
        nwind = cfg.NWIND if args.nwind is None else args.nwind
    pool = max(10_000_000, 3 * n * n)
    seed = distrib.replica_seed(rank)   # replicas: independent worlds, SEED = rank (rank 0 == BASELINE's SEED 0)

    def barrier(m):
        m.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    strips = (world > 1 or dist is not None) and throughput     # (dist with one rank: SMX_BENCH_FORCE_DIST=1, the multi-rank path on a single device)
    dimx = n
    ckpt = None
    if strips:
        # ONE map in `world` column strips. Default (weak scaling): the map grows with N -- every GPU owns a size x size strip of
        # a (N*size) x size map and its share of N x the particles, i.e. per-GPU work is what the N=1 line measures. --strong
        # keeps the N=1 map and splits it: the phases are latency chains, which a smaller strip does not shorten.
        if not args.strong:
            dimx, nwater, nwind = n * world, nwater * world, nwind * world
            pool = max(10_000_000, 3 * dimx * n)
        r = measure_strips(cfg, dimx, n, nwater, nwind, pool, local_rank, args.steps, args.warmup, dist, barrier, torch, engine)
    else:
        want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "soil_ref_lean"))
        ckpt = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"soilmx_bench_{os.getpid()}.smx") if want_cpu else None
        ckpt2 = (ckpt + ".end") if (ckpt and args.steps > 1) else None
        r = measure(cfg, soil, n, seed, nwater, nwind, nwind > 0, engine, pool, local_rank, args.steps, args.warmup, barrier, save_after_warmup=ckpt,
                    water_generations=args.water_generations if throughput else None, save_after_window=ckpt2, water_stagger=args.water_stagger if throughput else None)
        if ckpt and not os.path.exists(ckpt):
            ckpt = None
        if ckpt2 and not os.path.exists(ckpt2):
            ckpt2 = None
    steps = r["steps_w"] + r["steps_d"]
    dev = torch.device("cuda", local_rank) if (torch.cuda.is_available() and (dist is None or dist.get_backend() == "nccl")) else torch.device("cpu")
    tmax, total_steps = distrib.aggregate(dist, r["dt"], steps, dev)
    strips4 = None
    if strips and not args.no_lines and not args.strong and args.size == 4096 and args.only is None:
        # BASELINE configs[4] as the config text states it: ONE 8192^2 rockgravelpebbles_big map split into N column strips (a STRONG split of a
        # fixed map, next to the weak headline), area-scaled NWATER, 2 timed ticks after 1. Every rank takes part; rank 0 reports it in `lines`.
        bcfg4 = loadsoil(os.path.join(ROOT, "soilmachine_amd", "soils", "rockgravelpebbles_big.soil"))
        n4, nw4 = 8192, int(250 * (8192 / 256.0) ** 2)
        r4 = measure_strips(bcfg4, n4, n4, nw4, 0, max(10_000_000, 3 * n4 * n4), local_rank, 2, 1, dist, barrier, torch, engine)
        t4, s4 = distrib.aggregate(dist, r4["dt"], r4["steps_w"] + r4["steps_d"], dev)
        strips4 = {"name": f"configs4_8192_rockgravelpebbles_big_{world}_strips", "engine": eng_name, "scaling": "strong",
                   "workload": f"8192x8192, soil/rockgravelpebbles_big.soil, area-scaled NWATER={nw4} / NWIND=0, ONE map in {world} column strips of {n4 // world}x{n4} (BASELINE configs[4])",
                   "value": s4 / t4, "unit": "particle-steps/s", "ms_per_tick": 1e3 * t4 / 2, "steps": 2, "warmup": 1, "n_gpus": world,
                   "engine_stats_rank0": r4["engine_stats"], "single_gpu_reference": "lines[configs4_8192_rockgravelpebbles_big_relaxed] of the N=1 run's detail file"}

    if rank == 0:
        tm = r["tm"]
        kname = {"serial": ("k_water_serial", "k_wind_serial"), "spec": ("k_spec_exec_coop<water>", "k_spec_exec_coop<wind>"),
                 "batched": ("k_batch_epoch<water>", "k_batch_epoch<wind>"),
                 "relaxed": ("k_relax_step<water>", "k_relax_step<wind> (+ k_batch_epoch<wind> for the exclusive tail)" if os.environ.get("SMX_RELAX_WIND_MIN")
                             else "k_batch_epoch<wind> (the relaxed engine runs wind on the exclusive schedule by default: profiles/r03_relaxed.md)")}[eng_name]
        kr = {"water": kernel_roofline(r["steps_w"] * B_WATER_STEP, tm["ms_kernel_water"], tm["launches_kernel_water"], tm.get("launches_step_water", 0)),
              "wind": kernel_roofline(r["steps_d"] * B_WIND_STEP, tm["ms_kernel_wind"], tm["launches_kernel_wind"], tm.get("launches_step_wind", 0))}
        knames = {"water": kname[0], "wind": kname[1]}
        if eng_name == "relaxed":
            # the launches that carry the relaxed water phase and the grid pass besides the step kernel (round 6: each priced with ITS algorithmic
            # bytes -- floods: the steps of the nested particles that ran inside them x 192 B + 800 B per flood; grid tiles: 32 B per active cell + their
            # nested steps x 192 B -- over ITS launches; k_relax_floods is bracketed every 7th launch, the others at every launch)
            kr["floods"] = kernel_roofline(r.get("flood_nested_steps", 0) * B_WATER_STEP + r["floods"] * B_FLOOD, tm.get("ms_kernel_floods", 0.0),
                                           tm.get("launches_kernel_floods", 0), tm.get("launches_floods_all", 0))
            kr["grid_tiles"] = kernel_roofline(r.get("grid_active_cells", 0) * B_ACTIVE_CELL + r.get("grid_nested_steps", 0) * B_WATER_STEP, tm["ms_kernel_gridtiles"],
                                               tm["launches_kernel_gridtiles"], tm["launches_kernel_gridtiles"])
            kr["tail"] = kernel_roofline(0.0, tm.get("ms_kernel_tail", 0.0), tm.get("launches_kernel_tail", 0), tm.get("launches_kernel_tail", 0))   # (time only: its steps are not counted apart)
            kr["epochs"] = kernel_roofline(0.0, tm.get("ms_kernel_epochs", 0.0), tm.get("launches_kernel_epochs", 0), tm.get("launches_kernel_epochs", 0))
            knames.update(floods="k_relax_floods", grid_tiles="k_grid_tiles_flow", tail="k_relax_tail", epochs="k_relax_epochs")
        # the dominant kernel = the one with the most device time: average duration x ALL its launches
        ktime = {k: (v["avg_launch_us"] * v["launches"] if v else 0.0) for k, v in kr.items()}
        dom = max(ktime, key=lambda k: ktime[k])
        kn = knames[dom]
        kd = kr[dom] or {"achieved": 0.0, "frac": 0.0, "launches": 0, "launches_timed": 0, "avg_launch_us": 0.0, "algorithmic_bytes_per_launch": 0.0}
        ach = kd["achieved"]
        traffic, traffic_src = None, "not collected in this process (rocprofv3 --pmc runs as its own job, tools/gpu_run.sh pmc); no committed pass matches this workload"
        for pf in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc.json")), reverse=True):   # the newest round's pass first
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", pf)))
                k = pmc.get("kernels", {}).get(kn.split(" ")[0])
                if k and pmc.get("workload") == f"{n}/{nwater}/{nwind}/{eng_name}":
                    traffic = k["hbm_bytes_per_launch"]
                    traffic_src = f"profiles/{pf} (separate rocprofv3 --pmc passes of this command, not measured in this run)"
                    break
            except (OSError, ValueError, KeyError):
                pass
        ncells = n * n
        phases = {}
        def stream(ms_k, launches_k, bytes_per_launch, what):
            if launches_k and ms_k > 0:
                g = bytes_per_launch * launches_k / (ms_k * 1e-3) / 1e9
                return {"kernel": what, "achieved": g, "unit": "GB/s", "frac": g / HBM_PEAK_GBS, "avg_launch_us": 1e3 * ms_k / launches_k,
                        "algorithmic_bytes_per_launch": bytes_per_launch}
            return None
        phases["grid_classify"] = stream(tm["ms_kernel_classify"], tm["launches_kernel_classify"], ncells * 1.125, "k_grid_classify8 (1 flag byte/cell in, 1 bit/cell out)")
        phases["map_frequency"] = stream(tm["ms_kernel_mapfreq"], tm["launches_kernel_mapfreq"], ncells * 12.0,
                                         "k_map_frequency (freq r/w + track r, 12 B/cell; %d MB working set%s)" % (ncells * 12 // 2**20, ", below the 256 MiB Infinity Cache: see lines[*8192*] for one past it" if ncells * 12 < 256 * 2**20 else ""))
        for key, kk, kname_ in (("particles_water", "water", kname[0]), ("particles_wind", "wind", kname[1])):
            phases[key] = dict(kr[kk], kernel=kname_, algorithmic_bytes_per_step=B_WATER_STEP if kk == "water" else B_WIND_STEP) if kr[kk] else None
        for kk in ("floods", "grid_tiles", "tail", "epochs"):
            if kr.get(kk):
                phases[kk] = dict(kr[kk], kernel=knames[kk], ms_per_tick=ktime[kk] * 1e-3 / args.steps)
        tick_us = 1e6 * tmax / args.steps
        time_share = {knames[k].split(" ")[0]: ktime[k] / args.steps / tick_us for k in ktime if ktime[k] > 0}
        # whole-tick figure of SURVEY 8d: B_tick = cells * (32 + 24 * sections/cell) + 192 * water steps + 256 * wind steps, over the tick
        spc = r["digest"]["nsec"] / ncells if isinstance(r["digest"].get("nsec"), int) else 1.5
        b_tick = ncells * (32 + 24 * spc) + (B_WATER_STEP * r["steps_w"] + B_WIND_STEP * r["steps_d"]) / args.steps
        tick_gbs = b_tick / (tmax / args.steps) / 1e9
        roofline = {"bound": "hbm", "kernel": kn, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": kd["avg_launch_us"] * 1e-3, "launches": kd["launches"],
                    "launches_timed": kd["launches_timed"], "algorithmic_bytes_per_launch": kd["algorithmic_bytes_per_launch"], "phases": phases,
                    "time_share_of_tick": time_share,
                    "whole_tick": {"algorithmic_bytes_per_tick": b_tick, "achieved": tick_gbs, "unit": "GB/s", "frac": tick_gbs / HBM_PEAK_GBS},
                    "note": "kernel = the launch family with the most device time in the timed window (time_share_of_tick). achieved = algorithmic bytes per launch (SURVEY 8d: 192 B per "
                            "water step, 256 B per wind step; a flood launch: its nested particles' steps x 192 B + 800 B per flood; the grid tiles: 32 B per active cell + nested steps x 192 B; "
                            "x the units of ALL launches / ALL launches) / the average launch duration (HIP events on the library's stream around the bracketed launches: every launch of the "
                            "wind, tile and tail kernels, every 8th of k_relax_step<water>, every 7th of k_relax_floods). A particle step is a ~1000-instruction f64 chain around scattered 32-byte records: latency- and issue-bound, "
                            "not a stream (DESIGN.md 3); the O(cells) kernels are the HBM streams (phases)."}

        workload = (f"{dimx}x{n} layermap{f' (one map, {world} column strips of {dimx // world}x{n})' if strips else ''}, soil/{soil}, SEED 0, nwater={nwater}, nwind={nwind} per tick"
                    f"{' (area-scaled, SURVEY 8d)' if throughput and args.nwater is None else ''}, "
                    f"tick = water + grid pass + wind + frequency")
        head = line_of(r, "headline", workload, eng_name, args.steps)
        lines = [strips4] if strips4 else []
        parity = pref = None
        if not args.no_lines and world == 1:
            k3, w3 = max(2, min(args.steps, 3)), max(1, min(args.warmup, 2))
            dcfg = loadsoil(os.path.join(ROOT, "soilmachine_amd", "soils", "default.soil"))
            bcfg = loadsoil(os.path.join(ROOT, "soilmachine_amd", "soils", "rockgravelpebbles_big.soil"))
            fc = f"{n}x{n}, soil/{soil}, the file's own NWATER={cfg.NWATER} / NWIND={cfg.NWIND}"
            extra = []   # name, cfg, soil, size, nwater, nwind, engine, pool, steps, warmup, workload, in the default run?
            if throughput:
                # the bit-exact engine on the HEADLINE's workload (same map, same counts): what parity-green throughput is today
                # (37 s per tick at 4096^2 / 64 000 + 16 000: measured once, profiles/r04_exact_area_scaled_4096.json; --full or --only re-measures it)
                extra.append(("exact_area_scaled", cfg, soil, n, nwater, nwind, "spec", pool, 1, 1, workload, False))
                other = "batched" if eng_name == "relaxed" else "relaxed"
                extra.append((f"{other}_area_scaled", cfg, soil, n, nwater, nwind, other, pool, k3, w3, workload, False))   # (--full: the exclusive schedule is an internal of the relaxed one -- wind, floods --, not a shipped configuration: DESIGN.md 4.2)
                extra.append(("exact_file_counts", cfg, soil, n, cfg.NWATER, cfg.NWIND, "spec", pool, k3, w3, fc + " (round-1 headline workload)", True))
                extra.append((f"{eng_name}_file_counts", cfg, soil, n, cfg.NWATER, cfg.NWIND, eng_name, pool, k3, w3, fc, True))
            for nm, wind in (("configs1_1024_default_hydraulic", 0), ("configs2_1024_default_coupled", 250)):
                for e in ("spec", eng_name if throughput else "relaxed"):
                    extra.append((f"{nm}_{'exact' if e == 'spec' else e}", dcfg, "default.soil", 1024, 250, wind, e, 10_000_000, k3, w3,
                                  f"1024x1024, soil/default.soil, NWATER=250, NWIND={wind} (BASELINE {nm.split('_')[0]})", True))
            if throughput:
                # the same particle density on larger maps; configs[4]'s map size with its own soil file and with the headline's
                for nm, c_, s_, n_, dflt in ((f"configs4_8192_rockgravelpebbles_big_{eng_name}", bcfg, "rockgravelpebbles_big.soil", 8192, True),
                                             (f"area_8192_{eng_name}", cfg, soil, 8192, False), (f"area_16384_{eng_name}", cfg, soil, 16384, False)):
                    a_ = (n_ / 256.0) ** 2
                    nw_, nd_ = int(250 * a_), int(250 * a_ * c_.NWIND / max(c_.NWATER, 1))
                    extra.append((nm, c_, s_, n_, nw_, nd_, eng_name, max(10_000_000, 3 * n_ * n_), 2, 1,
                                  f"{n_}x{n_}, soil/{s_}, area-scaled NWATER={nw_} / NWIND={nd_}", dflt))
            for (nm, c_, s_, n_, nw_, nd_, e_, pool_, k_, w_, wl_, dflt) in extra:
                if args.only is not None and nm != args.only:
                    continue
                if args.only is None and not (dflt or args.full):
                    continue
                rr = measure(c_, s_, n_, 0, nw_, nd_, nd_ > 0, ENG[e_], pool_, local_rank, k_, w_)
                ln = line_of(rr, nm, wl_, "exact (speculative)" if e_ == "spec" else e_, k_)
                ln["steps"], ln["warmup"] = k_, w_
                ln["map_frequency_GBs"] = (n_ * n_ * 12.0 * rr["tm"]["launches_kernel_mapfreq"] / (rr["tm"]["ms_kernel_mapfreq"] * 1e-3) / 1e9) if rr["tm"]["ms_kernel_mapfreq"] > 0 else None
                if nm.startswith("configs") and e_ == "spec" and n_ <= 1024 and not args.no_cpu_baseline:
                    # BASELINE configs[1] / [2] as the reference runs them: its own CPU loop on one core, the same SEED-0 start, the same ticks (the exact engine's
                    # state IS the reference's, bit for bit: no checkpoint needed) -- VERDICT r5 weak 7: these lines carried no CPU figure beside them
                    try:
                        cb = cpu_baseline(s_, n_, nw_, nd_, w_ + k_, pool_, all_cores=False)
                        if cb:
                            ln["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": 1, "kind": "reference", "ms_per_tick": cb["ms_per_tick"],
                                                  "sample": f"ticks 1..{w_ + k_} from the initial terrain (the GPU line times ticks {w_ + 1}..{w_ + k_} of the same run)"}
                    except Exception:
                        pass
                lines.append(ln)
            if eng_name == "relaxed" and args.only is None and n == 4096 and args.water_generations is None and args.water_stagger is None:
                # the staggered variant of the schedule (smx_set_water_stagger, round 6): sixteen generations born 140 epochs apart inside ONE phase -- the same
                # window, with ITS parity verdict read from its record (it passes the gate for SEED 0 and for the two-strip schedule; over four device streams its
                # mean is a little further from the reference's than the default's: DESIGN.md 5)
                try:
                    rs = measure(cfg, soil, n, 0, nwater, nwind, nwind > 0, engine, pool, local_rank, args.steps, args.warmup, water_generations=16, water_stagger=140)
                    ls = line_of(rs, "headline_staggered_k16_e140", workload, eng_name, args.steps)
                    ls["steps"], ls["warmup"] = args.steps, args.warmup
                    ls["water_generations"], ls["water_stagger"] = rs["water_generations"], rs["water_stagger"]
                    hs_ = headline_parity_record(16, 1, 140)
                    ls["parity_gate_headline_size_record"] = ({"file": hs_["file"], "variant": hs_["variant"], "gate": hs_["gate"], "outliers": hs_["outliers"]} if hs_ else None)
                    lines.append(ls)
                except Exception as e:
                    lines.append({"name": "headline_staggered_k16_e140", "error": str(e)})
            if throughput and args.only is None:
                try:
                    lines.append(wet_regime_line(dcfg, local_rank, ENG[eng_name], eng_name))
                except Exception as e:                              # (an extra line must not take the headline down with it)
                    lines.append({"name": "wet_regime_256_default", "error": str(e)})
            if throughput and args.only is None:
                # parity of the throughput schedules against the exact engine, ON THE DEVICE, next to the exact engine's own stream-to-stream
                # spread: 512^2 of the headline's soil at the headline's particle density, measured in this run (the exact engine needs ~4 s per
                # tick at 1024^2 and 82 s at 4096^2 with these counts: the 1024^2 table, 10 ticks, is profiles/r03_p2_envelope_1024.json)
                try:
                    parity = parity_block(cfg, soil, 512, int(250 * 4), int(250 * 4 * cfg.NWIND / max(cfg.NWATER, 1)), 10_000_000, local_rank, 10, 2,
                                          [("relaxed", capi.ENGINE_RELAXED), ("batched", capi.ENGINE_BATCHED)])
                    parity["larger_maps"] = "profiles/r03_p2_envelope_1024.json (tools/p2_envelope.py on the MI355X)"
                except Exception as e:
                    parity = {"error": str(e)}
                # ... and against THE REFERENCE (oracle/_ref/soil_ref_lean on this box's host cores), eight control streams, z-scores: 1024^2 at the
                # headline's particle density, ticks 1-10 (~25 s; the record at the headline's own size and counts: profiles/r05_p2_reference_4096.json)
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import p2_reference
                    pref = p2_reference.run(soil, 1024, 10, 16, eng_name, device=local_rank, verbose=False,
                                            variants=[{"name": "run", "generations": r.get("water_generations"), "stagger": r.get("water_stagger") or 0, "strips": 1, "stream": 0}])
                    if "variants" in pref:                          # (one configuration: the block of round 5's shape)
                        pref.update(pref.pop("variants")["run"])
                    pref["headline_size_record"] = headline_parity_record(r.get("water_generations"), 1, r.get("water_stagger") or 0)
                except Exception as e:
                    pref = {"error": str(e)}
        if args.only is not None:
            print(json.dumps(lines[0] if lines else {"error": f"no line named {args.only}"}), flush=True)
            if dist is not None:
                dist.barrier(); dist.destroy_process_group()
            return
        lbm = None
        if not args.no_lines and world == 1:
            try:
                lbm = measure_lbm(local_rank)
            except Exception as e:                                  # (an extra line must not take the headline down with it)
                lbm = {"error": str(e)}
        cpu = None
        if not args.no_cpu_baseline and world == 1:             # (rank 0 at N=1 only: the other ranks would wait at the barrier for it)
            # the SAME tick the timed window starts with, from the SAME state (the checkpoint measure() wrote after the warm-up), at the SAME
            # counts: ~25 s of one core at 4096^2 / 64 000 + 16 000. Fallback (no checkpoint): a bounded sample from the initial terrain.
            if ckpt:
                pt0 = (r.get("per_tick") or [None])[0]
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=2) as ex:   # (two samples side by side: the first tick of the window and the tick right after it -- the tick grows with the lakes)
                    ncpu_ = os.cpu_count() or 1
                    f_end = ex.submit(cpu_baseline, soil, n, nwater, nwind, 1, pool, False, ckpt2, args.warmup + args.steps, r.get("post_tick"), (ncpu_ // 2 - 1) if ncpu_ >= 72 else max(ncpu_ - 1, 0)) if ckpt2 else None   # (a core the all-cores processes -- cores 1..32 -- do not use)
                    cpu = cpu_baseline(soil, n, nwater, nwind, args.cpu_ticks, pool, all_cores=True, load=ckpt, tick0=args.warmup, gpu_same_tick=pt0)
                    ce = f_end.result() if f_end else None
                if cpu and ce:
                    cpu["after_window"] = {"value": ce["value"], "ms_per_tick": ce["ms_per_tick"], "sample": ce["sample"][:160], "gpu_same_tick": ce.get("gpu_same_tick")}
            else:
                cw, cd = (max(nwater // 8, 1000), max(nwind // 8, 250 if nwind else 0)) if nwater > 8000 else (nwater, nwind)
                cpu = cpu_baseline(soil, n, cw, cd, args.cpu_ticks, pool, all_cores=True)
        p2 = "P2: deterministic (== CPU restatement bit for bit), statistical vs the reference (parity_gate, parity_vs_reference); never claimed bit-exact"
        exact_rec = None                                        # the bit-exact engine on this workload: a RECORD of an earlier run, read from its file (never a literal)
        if throughput and n == 4096 and nwater == 64000 and soil == SOIL:
            for pf in ("r05_exact_area_scaled_4096.json", "r04_exact_area_scaled_4096.json"):
                try:
                    e = json.load(open(os.path.join(ROOT, "profiles", pf)))
                    exact_rec = {"engine": "speculative (bit-identical to the reference)", "value": e["value"], "unit": "particle-steps/s", "ms_per_tick": e["ms_per_tick"],
                                 "measured_in_this_run": False, "source": f"profiles/{pf} (bench.py --only exact_area_scaled on the MI355X)"}
                    break
                except (OSError, ValueError, KeyError):
                    pass
        hrec = headline_parity_record(r.get("water_generations"), world if strips else 1, r.get("water_stagger") or 0) if (throughput and n == 4096 and nwater // max(world if (strips and not args.strong) else 1, 1) == 64000 and soil == SOIL) else None
        # parity_gate: the rule of tests/test_gpu_relaxed.py (nothing at |z| >= 3, at most one figure in [2, 3)) applied to THIS run's own check against the
        # reference (1024^2, 16 control streams) and to the committed record at the headline's size; "fail" if either breaks it
        g_run = (pref or {}).get("gate") if isinstance(pref, dict) else None
        g_rec = (hrec or {}).get("gate")
        if not throughput:
            parity_gate = {"verdict": "exact", "note": "bit-identical to the reference (tests/golden, tests/test_gpu_parity.py)"}
        else:
            verdict = "unmeasured" if (g_run is None and g_rec is None) else ("fail" if ((g_run and not g_run["pass"]) or (g_rec and not g_rec["pass"])) else "pass")
            parity_gate = {"verdict": verdict, "rule": "no figure at |z| >= 3, at most one in [2, 3) (eight figures, z against the reference's own rand()-stream spread)",
                           "this_run_1024": g_run, "headline_size_record": ({"file": hrec["file"], "variant": hrec["variant"], "gate": g_rec, "outliers": hrec["outliers"]} if hrec else None)}
        config = {"workload": workload, "engine": eng_name, "water_generations": r.get("water_generations") if throughput else None,
                  "water_stagger": r.get("water_stagger") if throughput else None,
                  "parallelism": (f"{world} column strips of one map, halo exchange over {'RCCL (ncclSend / ncclRecv on device buffers)' if dist.get_backend() == 'nccl' else 'host callbacks on torch.distributed ' + dist.get_backend()}" if strips else "replicas") if world > 1 else "single",
                  "parity": p2 if throughput else head["parity"]}
        detail = {
            "metric": "particle-steps/s", "value": total_steps / tmax, "unit": "particle-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * tmax / args.steps, "higher_is_better": True,
            "scaling": "strong" if (strips and args.strong) else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": dict(config, scaling_note=(None if world == 1 else ("--strong: the N=1 map split into N strips" if (strips and args.strong) else
                                                 "weak: one (N*size) x size map, every GPU owns a size x size strip and 1/N of N x the particles "
                                                 "(per-GPU work = the N=1 line)") if strips else "independent replicas (exact engines do not shard)"),
                           bit_exact_on_this_workload=exact_rec),
            "value_top_level": head["value_top_level"] if world == 1 else None,
            "steps_definition": "value: move()==true + interact(), top-level AND nested particles; value_top_level: top-level only "
                                "(what cpu_baseline can count)",
            "ms_per_tick": 1e3 * tmax / args.steps, "phase_ms_per_tick": head["phase_ms_per_tick"],
            "per_tick": r.get("per_tick"),
            "steps_per_tick": head["steps_per_tick"], "engine_stats": head["engine_stats"], "digest": head["digest"],
            "roofline": roofline, "cpu_baseline": cpu, "parity_vs_exact": parity, "parity_vs_reference": pref, "parity_gate": parity_gate, "lines": lines, "lbm_wind": lbm,
            # nothing was dropped inside the timed region (nested particles that did not fit the children buffer, sediment
            # that found no pool node): a line with work skipped would not be a measurement
            "work_dropped": {"children_lost": int(head["engine_stats"].get("children_lost", 0)), "pool_overflow": int(head.get("pool_overflow") or 0)},
        }
        # The detailed record (per-tick entries, extra lines, parity blocks, LBM) goes to a SIDE FILE; stdout carries ONE compact strict-JSON
        # line, the last one (round 4's single 25 KB line was not parsed by the driver: BENCH_r04.json parsed = null).
        detail_path = None
        for d_ in (os.environ.get("SMX_BENCH_DETAIL_DIR"), os.path.join(ROOT, "gpurun_out"), os.environ.get("TMPDIR", "/tmp")):
            if not d_:
                continue
            try:
                os.makedirs(d_, exist_ok=True)
                detail_path = os.path.join(d_, f"bench_detail_n{world}.json")
                with open(detail_path, "w") as f:
                    json.dump(detail, f, indent=1, default=float)
                break
            except OSError:
                detail_path = None
        final_line = json.dumps(compact_line(detail, detail_path), default=float, separators=(",", ":"), allow_nan=False)
        if dist is None:
            print(final_line, flush=True)
    for f_ in ("ckpt", "ckpt2"):
        try:
            if not strips and locals().get(f_):
                os.remove(locals()[f_])
        except (OSError, NameError):
            pass
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        if locals().get("final_line"):
            # Several ranks: RCCL writes a version banner ("RCCL version : ... Librccl path : ...") to the C library's stdout, which is a pipe here and therefore
            # buffered until the process ends -- it would land AFTER a line printed earlier (seen on the MI355X box: profiles/r06_bench_multirank_path_one_rank_nccl.json).
            # The contract is ONE JSON line, the last one: flush what C has buffered, print the line, and leave without giving exit handlers another turn at stdout.
            import ctypes
            sys.stdout.flush(); sys.stderr.flush()
            try:
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            print(final_line, flush=True)
            os._exit(0)


if __name__ == "__main__":
    main()
