"""LBM wind through the C-ABI (include/soilmx.h smx_lbm_*): the host-side mirror of the reference's `lbmw` namespace
(source/include/lbmwind/lbmwind.h:16-72) -- initialize / update / the retrievable buffers -- on hand-written HIP kernels
(csrc/soil_lbm.h). Visual only in the reference; SURVEY.md 8 row f4."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

Q = 19
# the reference's lattice (lbmwind.h:18-21)
NX, NY, NZ = 64, 40, 64


class LbmWind:
    def __init__(self, nx: int = NX, ny: int = NY, nz: int = NZ, device: int = 0):
        self.L = capi.load()
        self.nx, self.ny, self.nz, self.n = nx, ny, nz, nx * ny * nz
        h = C.c_void_p()
        rc = self.L.smx_lbm_create(nx, ny, nz, device, C.byref(h))
        self.h = h
        if rc != 0:
            msg = self.L.smx_lbm_last_error(h).decode() if h else "smx_lbm_create failed"
            if h:
                self.L.smx_lbm_destroy(h)
            self.h = None
            raise RuntimeError(msg)

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.smx_lbm_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.smx_lbm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_boundary(self, b: np.ndarray):
        """lbmw::b->fill (SoilMachine.cpp:239): float per cell, > 0 = solid; index (x*NY + y)*NZ + z"""
        b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1)
        assert b.size == self.n
        self._chk(self.L.smx_lbm_set_boundary(self.h, capi.ptr(b)))

    def boundary_from_map(self, layermap, sx: float = 1.0, sy: float = 1.0, sz: float = 1.0):
        """SoilMachine.cpp:235-238 on the device: solid where map.height(ivec2(sx*x, sz*z)) > (sy*y)/SCALE"""
        self._chk(self.L.smx_lbm_boundary_from_map(self.h, layermap.h, sx, sy, sz))

    def initialize(self):
        self._chk(self.L.smx_lbm_initialize(self.h))

    def step(self, n: int = 1):
        """lbmw::update's collide + stream (lbmwind.h:176-188), n times"""
        self._chk(self.L.smx_lbm_step(self.h, n))

    def read(self, rho=True, v=True, f=False):
        r = np.zeros(self.n, np.float32) if rho else None
        vv = np.zeros((self.n, 4), np.float32) if v else None
        ff = np.zeros((self.n, Q), np.float32) if f else None
        self._chk(self.L.smx_lbm_read(self.h, capi.ptr(r) if rho else None, capi.ptr(vv) if v else None, capi.ptr(ff) if f else None))
        return r, vv, ff

    def write_f(self, f: np.ndarray):
        f = np.ascontiguousarray(f, dtype=np.float32).reshape(-1)
        assert f.size == self.n * Q
        self._chk(self.L.smx_lbm_write_f(self.h, capi.ptr(f)))

    def move(self, pos4: np.ndarray) -> np.ndarray:
        """move.cs on tracer positions (n x vec4), returns the advected positions"""
        p = np.ascontiguousarray(pos4, dtype=np.float32).copy()
        self._chk(self.L.smx_lbm_move(self.h, capi.ptr(p), p.shape[0]))
        return p

    def timing(self, reset: bool = False):
        ms, st = C.c_double(), C.c_uint64()
        self._chk(self.L.smx_lbm_get_timing(self.h, C.byref(ms), C.byref(st), int(reset)))
        return {"ms_steps": ms.value, "steps": int(st.value)}
