#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/ by RUNNING THE REFERENCE ITSELF.

oracle/_ref/soil_ref is the reference's own hot-path headers (source/layermap.h, source/particle/*.h,
source/io.h) compiled headless by oracle/Makefile from /root/reference. This script runs it on the cases
below and stores (a) full state snapshots of small maps (`<case>.t<tick>.npz`) and (b) digests of larger
runs (`digests.json`: sum of heights, section count, type hash, rand() draws -- the digest of SURVEY.md
Appendix E). The reference has no tests or golden vectors of its own; these are its outputs.

Only runnable where /root/reference exists (the build container); the fixtures travel with the repo.
"""
import json, os, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import run_ref
from soilmachine_amd.snapshot import read_snapshot

REFSOIL = os.path.join(os.environ.get("REF_ROOT", "/root/reference"), "soil")

SNAP_CASES = [
    # name, soil, kwargs, dump ticks
    ("default64", "default.soil", dict(seed=0, size=64, wind=False), [0, 1, 5, 20]),
    ("default64s7", "default.soil", dict(seed=7, size=64, wind=False, nwater=150), [40]),
    ("rgps64", "rockgravelpebblessand.soil", dict(seed=0, size=64, nwater=100, nwind=50), [0, 3, 10]),
    ("rocksand48x80", "rocksand.soil", dict(seed=3, sizex=48, sizey=80, nwater=60, nwind=40), [0, 5]),
    ("painted64", "painted.soil", dict(seed=1, size=64, nwater=100, nwind=50), [0, 5]),
]
DIGEST_CASES = [
    ("default256_t20_wind", "default.soil", dict(seed=0, ticks=20)),
    ("default256_t20", "default.soil", dict(seed=0, ticks=20, wind=False)),
    ("default256_t100", "default.soil", dict(seed=0, ticks=100, wind=False)),
    ("rgps256_t10", "rockgravelpebblessand.soil", dict(seed=0, size=256, ticks=10)),
    ("rgps512_t20", "rockgravelpebblessand.soil", dict(seed=0, ticks=20)),
    ("rgpbig256_t10", "rockgravelpebbles_big.soil", dict(seed=0, size=256, ticks=10)),
    ("sand256_t10", "sand.soil", dict(seed=0, ticks=10)),
    ("bigbutte2_128_t10", "bigbutte2.soil", dict(seed=5, size=128, ticks=10)),
    ("default1024_t5", "default.soil", dict(seed=0, size=1024, ticks=5, wind=False)),
    # round 2: the remaining rows of SURVEY.md Appendix E and the BASELINE.json configs at depth
    ("default256_t300", "default.soil", dict(seed=0, ticks=300, wind=False)),
    ("default256_t1000", "default.soil", dict(seed=0, ticks=1000, wind=False)),
    ("rgps1024_t3", "rockgravelpebblessand.soil", dict(seed=0, size=1024, ticks=3)),
    ("rgpbig512_t10", "rockgravelpebbles_big.soil", dict(seed=0, ticks=10)),
    ("default1024_t3_wind", "default.soil", dict(seed=0, size=1024, ticks=3)),          # BASELINE configs[2]
    ("rgps4096_t5", "rockgravelpebblessand.soil", dict(seed=0, size=4096, ticks=5, pool=3 * 4096 * 4096)),   # configs[3]
    ("rgpbig8192_t1", "rockgravelpebbles_big.soil", dict(seed=0, size=8192, ticks=1, pool=2 * 8192 * 8192)), # configs[4]
]


def main():
    # `--only a,b,c`: (re)generate just these digest cases and merge them into the existing digests.json
    only = None
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    path = os.path.join(HERE, "digests.json")
    digests = json.load(open(path)) if only is not None and os.path.exists(path) else {}
    for name, soil, kw, ticks in ([] if only is not None else SNAP_CASES):
        with tempfile.TemporaryDirectory() as td:
            j = run_ref(os.path.join(REFSOIL, soil), ticks=max(ticks), dump_prefix=os.path.join(td, "r"), dump_at=ticks, **kw)
            for t in ticks:
                s = read_snapshot(os.path.join(td, f"r.t{t}.snap"))
                np.savez_compressed(os.path.join(HERE, f"{name}.t{t}.npz"), dimx=s.dimx, dimy=s.dimy, scale=s.scale,
                                    nsoils=s.nsoils, rand_calls=s.rand_calls, count=s.count, type=s.type, size=s.size,
                                    floor=s.floor, sat=s.sat, wfreq=s.wfreq, wtrack=s.wtrack, windfreq=s.windfreq)
        digests[name] = dict(soil=soil, kw=kw, ticks=max(ticks), nwater=j["nwater"], nwind=j["nwind"],
                             sumh=j["sumh"], nsec=j["nsec"], typehash=j["typehash"], rand_calls=j["rand_calls"],
                             steps_water_top=j["steps_water_top"], steps_wind=j["steps_wind"])
        print(name, digests[name]["nsec"], digests[name]["typehash"], flush=True)
    for name, soil, kw in DIGEST_CASES:
        if only is not None and name not in only:
            continue
        j = run_ref(os.path.join(REFSOIL, soil), **kw)
        kw2 = dict(kw); t = kw2.pop("ticks")
        digests[name] = dict(soil=soil, kw=kw2, ticks=t, nwater=j["nwater"], nwind=j["nwind"], dimx=j["dimx"], dimy=j["dimy"],
                             sumh=j["sumh"], nsec=j["nsec"], typehash=j["typehash"], rand_calls=j["rand_calls"],
                             steps_water_top=j["steps_water_top"], steps_wind=j["steps_wind"])
        print(name, j["nsec"], j["typehash"], flush=True)
        json.dump(digests, open(path, "w"), indent=1)
    json.dump(digests, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
