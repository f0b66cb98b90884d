#!/usr/bin/env python3
"""Golden vectors of the LBM wind, made by RUNNING THE REFERENCE'S OWN SHADERS (oracle/_ref/liblbm_ref.so = the unmodified
statements of source/include/lbmwind/shader/LBM/*.cs and shader/move.cs compiled as C++, see oracle/lbm_ref.cpp).

The reference has no test or golden vector for this path; these are its outputs. Only runnable where /root/reference
exists (the build container); the fixture tests/golden/lbm_ref.npz travels with the repo and pins oracle/lbm_oracle.cpp --
and through it the HIP kernels -- on the GPU box."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import LbmRef
from test_lbm import terrain_boundary

CASES = {"a": ((12, 8, 10), 6, 1), "b": ((7, 5, 33), 9, 4)}     # dims, steps, boundary seed
out = {}
for k, (dims, steps, seed) in CASES.items():
    r = LbmRef(*dims)
    r.set_boundary(terrain_boundary(*dims, seed=seed)); r.initialize()
    rho0, v0, f0 = r.read()
    r.step(steps)
    rho, v, f = r.read()
    rng = np.random.default_rng(seed)
    p = (rng.random((200, 4)) * np.array([dims[0] - 2, dims[1] - 2, dims[2] - 2, 1]) + np.array([0.5, 0.5, 0.5, 0])).astype(np.float32)
    out.update({f"{k}_dims": np.array(dims), f"{k}_steps": steps, f"{k}_seed": seed, f"{k}_f0": f0, f"{k}_rho": rho, f"{k}_v": v, f"{k}_f": f,
                f"{k}_p": p, f"{k}_moved": r.move(p)})
np.savez_compressed(os.path.join(HERE, "lbm_ref.npz"), **out)
print("wrote lbm_ref.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})
