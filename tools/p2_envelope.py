#!/usr/bin/env python3
"""Device-side parity envelope of the throughput schedules: relaxed and batched against the EXACT engine from
the same terrain, next to the exact engine with other rand() streams (the reference's own chaos). One JSON document on stdout.
  p2_envelope.py --size 1024 --ticks 20 --streams 4 [--soil rockgravelpebblessand.soil] [--counts area|file]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from soilmachine_amd import capi
from soilmachine_amd.soilfile import loadsoil

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024); ap.add_argument("--ticks", type=int, default=20); ap.add_argument("--streams", type=int, default=4)
ap.add_argument("--soil", default="rockgravelpebblessand.soil"); ap.add_argument("--counts", default="area", choices=["area", "file"])
ap.add_argument("--engines", default="relaxed,batched")
a = ap.parse_args()
cfg = loadsoil(os.path.join(ROOT, "soilmachine_amd", "soils", a.soil))
area = (a.size / 256.0) ** 2
nw, nd = (int(250 * area), int(250 * area * cfg.NWIND / max(cfg.NWATER, 1))) if a.counts == "area" else (cfg.NWATER, cfg.NWIND)
eng = {"relaxed": capi.ENGINE_RELAXED, "batched": capi.ENGINE_BATCHED}
out = bench.parity_block(cfg, a.soil, a.size, nw, nd, max(10_000_000, 3 * a.size * a.size), 0, a.ticks, a.streams, [(e, eng[e]) for e in a.engines.split(",") if e])
ctl = out["control_exact_other_rand_stream"]
for e in a.engines.split(","):
    if e and ctl:
        out[e]["inside_control_envelope"] = {k: abs(out[e][k]) <= 1.25 * max(abs(c[k]) for c in ctl) for k in ("height_rmse", "nsec_rel", "sumh_rel", "steps_water_rel")}
print(json.dumps(out, indent=1))
