#!/usr/bin/env python3
"""Per-launch durations of the batched engine's epoch kernels for ONE tick from SEED 0 (4096^2 rockgravelpebblessand,
64000 + 16000 particles), to set against the host simulation's per-epoch statistics of the same (deterministic) run.
  run:   rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/epoch_trace.py run
  post:  python tools/epoch_trace.py post DIR/…_kernel_trace.csv OUT.npz"""
import csv, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    from soilmachine_amd import capi
    from soilmachine_amd.machine import SoilMachine
    from soilmachine_amd.soilfile import loadsoil
    cfg = loadsoil(os.path.join(ROOT, "soilmachine_amd", "soils", "rockgravelpebblessand.soil"))
    sm = SoilMachine(cfg, 4096, seed=0, nwater=64000, nwind=16000, engine=capi.ENGINE_BATCHED, pool=4 * 4096 * 4096)
    sm.tick(1, sync=True)
    print(sm.map.digest(), sm.map.batch_stats())
else:
    rows = []
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if "k_batch_epoch" in n or "k_batch_grid" in n:
                kind = 2 if "k_batch_grid" in n else (1 if ("ILb1" in n or "<true>" in n) else 0)
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
    rows.sort()
    a = np.array(rows, dtype=np.int64)
    np.savez_compressed(sys.argv[3], start=(a[:, 0] - a[0, 0]).astype(np.int64), dur=(a[:, 1] - a[:, 0]).astype(np.int32), kind=a[:, 2].astype(np.uint8))
    for k, nm in ((0, "water"), (1, "wind"), (2, "grid")):
        d = a[a[:, 2] == k]
        if len(d): print(nm, len(d), "launches, mean us", (d[:, 1] - d[:, 0]).mean() / 1e3, "sum ms", (d[:, 1] - d[:, 0]).sum() / 1e6)
