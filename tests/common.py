"""Shared helpers of the test-suite (fixtures on disk, configs)."""
from __future__ import annotations

import json
import os

import numpy as np

from soilmachine_amd.snapshot import Snapshot
from soilmachine_amd.soilfile import loadsoil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
SOILS = os.path.join(ROOT, "soilmachine_amd", "soils")


def soil_path(name: str) -> str:
    return os.path.join(SOILS, name)


def load_cfg(name: str):
    return loadsoil(soil_path(name))


def golden_snapshot(case: str, tick: int) -> Snapshot:
    z = np.load(os.path.join(GOLD, f"{case}.t{tick}.npz"))
    return Snapshot(int(z["dimx"]), int(z["dimy"]), int(z["scale"]), int(z["nsoils"]), int(z["rand_calls"]), 0,
                    z["count"], z["type"], z["size"], z["floor"], z["sat"], z["wfreq"], z["wtrack"], z["windfreq"])


def digests() -> dict:
    return json.load(open(os.path.join(GOLD, "digests.json")))


def case_dims(d: dict, cfg) -> tuple:
    kw = d["kw"]
    if "dimx" in d:
        return d["dimx"], d["dimy"]
    if kw.get("sizex"):
        return kw["sizex"], kw["sizey"]
    n = kw.get("size") or 0
    return (n or cfg.SIZEX, n or cfg.SIZEY)


# golden snapshot cases: name -> (soil file, seed, dowind, dump ticks)
SNAP_CASES = {
    "default64": ("default.soil", 0, False, [0, 1, 5, 20]),
    "default64s7": ("default.soil", 7, False, [40]),
    "rgps64": ("rockgravelpebblessand.soil", 0, True, [0, 3, 10]),
    "rocksand48x80": ("rocksand.soil", 3, True, [0, 5]),
    "painted64": ("painted.soil", 1, True, [0, 5]),
}
