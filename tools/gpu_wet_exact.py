#!/usr/bin/env python3
"""Exact engines in the wet regime, from a committed-size CPU-made start state (256^2 default.soil after T0 ticks):
per-phase ms/tick over the next N ticks for the serial and the speculative engine, each in its own process, and the state
digest against the oracle's (made by --make on the CPU).
  --make T0 N   (CPU, container)  writes build/wet/start.snap + build/wet/expect.json
  --run         (GPU)             prints one JSON line per variant"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
W = os.path.join(ROOT, "build", "wet")
SOIL = os.path.join(ROOT, "soilmachine_amd", "soils", "default.soil")


def make(t0, n, size=256, nwater=250):
    from oracle_lib import Oracle
    from soilmachine_amd.soilfile import loadsoil
    from soilmachine_amd.snapshot import write_snapshot
    cfg = loadsoil(SOIL)
    o = Oracle(cfg, size, size, seed=0); o.initialize()
    for _ in range(t0): o.tick(nwater, 0, True, False)
    os.makedirs(W, exist_ok=True)
    write_snapshot(os.path.join(W, "start.snap"), o.snapshot())
    t = time.time()
    for _ in range(n): o.tick(nwater, 0, True, False)
    dt = time.time() - t
    s = o.snapshot()
    json.dump(dict(size=size, nwater=nwater, t0=t0, n=n, digest=s.digest(), rand_calls=s.rand_calls, oracle_ms_tick=1e3 * dt / n),
              open(os.path.join(W, "expect.json"), "w"))
    print(open(os.path.join(W, "expect.json")).read())


def one(engine):
    from soilmachine_amd import capi
    from soilmachine_amd.machine import SoilMachine
    from soilmachine_amd.soilfile import loadsoil
    from soilmachine_amd.snapshot import read_snapshot
    ex = json.load(open(os.path.join(W, "expect.json")))
    if os.environ.get("SMX_WET_N"):                          # (profiling passes: fewer ticks, no digest to compare with)
        ex["n"] = int(os.environ["SMX_WET_N"]); ex["digest"] = {"nsec": -1, "typehash": "", "sumh": 0.0}
    cfg = loadsoil(SOIL)
    eng = {"serial": capi.ENGINE_SERIAL, "spec": capi.ENGINE_SPECULATIVE}[engine]
    sm = SoilMachine(cfg, ex["size"], seed=0, nwater=ex["nwater"], nwind=0, dowind=False, engine=eng, initialize=False)
    sm.map.load(read_snapshot(os.path.join(W, "start.snap")), rand_seed=0)
    sm.map.timing_reset(); t = time.perf_counter()
    sm.tick(ex["n"], sync=True)
    dt = time.perf_counter() - t
    tm = sm.map.timing(); d = sm.map.digest()
    ok = d["nsec"] == ex["digest"]["nsec"] and d["typehash"] == ex["digest"]["typehash"] and d["sumh"] == ex["digest"]["sumh"] and d["rand_calls"] == ex["rand_calls"]
    print(json.dumps(dict(engine=engine, ticks=[ex["t0"], ex["t0"] + ex["n"]], ms_tick=1e3 * dt / ex["n"],
                          ms_water=tm["ms_water"] / ex["n"], ms_grid=tm["ms_grid"] / ex["n"], bit_identical_to_oracle=ok, oracle_ms_tick=ex["oracle_ms_tick"])), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--make": make(int(sys.argv[2]), int(sys.argv[3]))
    elif sys.argv[1] == "--one": one(sys.argv[2])
    else:
        for eng in ("serial", "spec"):
            subprocess.call([sys.executable, __file__, "--one", eng])
