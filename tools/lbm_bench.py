#!/usr/bin/env python3
"""LBM wind: cell updates/s and HBM roofline fraction of the fused collide+stream kernel (csrc/soil_lbm.h) at the reference's
lattice (64x40x64, lbmwind.h:18-21) and at lattices that do not fit the caches. Algorithmic bytes per cell and step:
19 reads + 19 writes of 4 B + mask 4 B + rho 4 B + v 16 B = 176 B."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from soilmachine_amd.lbm import LbmWind
PEAK = 8000.0
rows = []
for dims, steps in (((64, 40, 64), 2000), ((256, 160, 256), 200), ((512, 320, 512), 60), ((768, 480, 768), 20)):
    nx, ny, nz = dims
    g = LbmWind(nx, ny, nz)
    b = np.zeros((nx, ny, nz), np.float32); b[:, :ny // 8, :] = 1.0
    g.set_boundary(b); g.initialize(); g.step(5); g.timing(reset=True)
    t0 = time.perf_counter(); g.step(steps); wall = time.perf_counter() - t0
    tm = g.timing()
    us = 1e3 * tm["ms_steps"] / tm["steps"]
    gbs = g.n * 176 / (us * 1e-6) / 1e9
    rows.append({"lattice": f"{nx}x{ny}x{nz}", "cells": g.n, "us_per_step": us, "cell_updates_per_s": g.n / (us * 1e-6), "GB/s": gbs, "frac_of_8TBs": gbs / PEAK,
                 "wall_us_per_step": 1e6 * wall / steps})
    print(json.dumps(rows[-1]), flush=True)
    g.close()
print("RESULT " + json.dumps(rows))
