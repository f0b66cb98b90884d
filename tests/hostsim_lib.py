"""ctypes wrapper of tests/hostsim (the device sources compiled for the host -- TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from soilmachine_amd.snapshot import Snapshot
from soilmachine_amd.machine import default_pool
from soilmachine_amd.soilfile import soils_array

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libhostsim.so")
SRC = [os.path.join(HERE, "hostsim.cpp")] + [os.path.join(ROOT, "soilmachine_amd", "csrc", f)
                                              for f in ("soil_core.h", "soil_serial.h", "soil_spec.h", "soil_coop.h", "soil_batch.h", "soil_relax.h")]
_lib = None


def lib():
    global _lib
    if _lib is None:
        srcs = [p for p in SRC if os.path.exists(p)]
        if not os.path.exists(LIB) or any(os.path.getmtime(p) > os.path.getmtime(LIB) for p in srcs):
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-pthread", "-o", LIB, SRC[0]])
        L = C.CDLL(LIB)
        vp = C.c_void_p
        L.hs_create.restype = vp
        L.hs_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64]
        L.hs_destroy.argtypes = [vp]
        L.hs_set_soils.argtypes = [vp, vp, C.c_int]
        L.hs_srand.argtypes = [vp, C.c_uint]
        L.hs_rand_advance.argtypes = [vp, C.c_uint64]
        L.hs_import.argtypes = [vp] + [vp] * 8
        L.hs_num_sections.argtypes = [vp]; L.hs_num_sections.restype = C.c_uint64
        L.hs_export.argtypes = [vp] + [vp] * 8
        L.hs_counters.argtypes = [vp, vp, vp]
        for f in ("hs_water", "hs_wind"):
            getattr(L, f).argtypes = [vp, C.c_int]
        for f in ("hs_grid", "hs_freq"):
            getattr(L, f).argtypes = [vp]
        L.hs_set_water_generations.argtypes = [C.c_int]
        L.hs_set_water_stagger.argtypes = [C.c_uint32, C.c_uint32]
        L.hs_tick.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.hs_add.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_uint32]
        L.hs_remove.argtypes = [vp, C.c_int, C.c_int, C.c_double]; L.hs_remove.restype = C.c_double
        L.hs_spec_tick.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, vp]
        L.hs_spec_water.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_uint, vp]
        L.hs_spec_wind.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_uint, vp]
        L.hs_set_boundary_threads.argtypes = [C.c_int]
        L.hs_batched_water.argtypes = [vp, C.c_int, C.c_int, C.c_uint]
        L.hs_batched_wind.argtypes = [vp, C.c_int, C.c_int, C.c_uint]
        L.hs_batched_grid.argtypes = [vp, C.c_int, C.c_uint]
        L.hs_batched_tick.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint]
        L.hs_batch_stats.argtypes = [vp]
        L.hs_set_relaxed.argtypes = [C.c_int]
        L.hs_set_relax_wind.argtypes = [C.c_uint32, C.c_int]
        L.hs_set_relax_water.argtypes = [C.c_int]
        L.hs_set_batch_strips.argtypes = [C.c_int] * 3
        L.hs_d_set_own.argtypes = [vp, C.c_int, C.c_int]
        L.hs_d_gen_begin.argtypes = [vp, C.c_int, C.c_uint32, vp, vp]
        L.hs_d_gen_spawn.argtypes = [vp, vp, vp]
        L.hs_d_next_phase.argtypes = [vp]
        L.hs_d_chunk.argtypes = [vp, C.c_uint32, vp]
        L.hs_d_gen_end.argtypes = [vp, vp, C.c_uint32, vp]
        for f in ("hs_d_pack_columns", "hs_d_pack_particles"):
            getattr(L, f).argtypes = [vp, C.c_int, C.c_int, vp, C.c_uint64]; getattr(L, f).restype = C.c_uint64
        for f in ("hs_d_unpack_columns", "hs_d_unpack_particles"):
            getattr(L, f).argtypes = [vp, vp, C.c_uint64]
        L.hs_d_grid_begin.argtypes = [vp]
        L.hs_d_grid_sweep.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


CTR = ["steps_water_top", "steps_water_all", "steps_wind", "nested_particles", "floods", "cascade_calls",
       "cascade_transfers", "wcascade_calls", "grid_active_cells", "pool_overflow", "live_sections",
       "flood_nested_steps", "grid_nested_steps"]      # (the last two: device kernels only -- soil_core.h Ctr; 0 on host threads)


class HostSim:
    def __init__(self, cfg, dimx, dimy, pool=None, seed=0):
        self.L = lib()
        self.cfg, self.dimx, self.dimy = cfg, dimx, dimy
        self.pool = int(pool if pool is not None else default_pool(dimx, dimy))
        self.h = self.L.hs_create(dimx, dimy, cfg.SCALE, self.pool)
        self._soils = soils_array(cfg)
        self.L.hs_set_soils(self.h, _p(self._soils), len(self._soils))
        self.L.hs_srand(self.h, seed)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hs_destroy(self.h); self.h = None

    def load(self, s: Snapshot, advance_rand=True):
        self.L.hs_import(self.h, _p(s.count), _p(s.type), _p(s.size), _p(s.floor), _p(s.sat), _p(s.wfreq), _p(s.wtrack), _p(s.windfreq))
        if advance_rand:
            self.L.hs_rand_advance(self.h, s.rand_calls)

    def counters(self):
        a = np.zeros(len(CTR), np.uint64); rc = C.c_uint64()
        self.L.hs_counters(self.h, _p(a), C.byref(rc))
        d = {k: int(v) for k, v in zip(CTR, a)}
        d["rand_calls"] = int(rc.value)
        d["pool_free"] = self.pool - d["live_sections"]
        return d

    def snapshot(self) -> Snapshot:
        nc, ns = self.dimx * self.dimy, int(self.L.hs_num_sections(self.h))
        count = np.zeros(nc, np.uint32); ty = np.zeros(ns, np.uint32)
        size = np.zeros(ns); floor = np.zeros(ns); sat = np.zeros(ns)
        wf = np.zeros(nc, np.float32); wt = np.zeros(nc, np.float32); wi = np.zeros(nc, np.float32)
        self.L.hs_export(self.h, _p(count), _p(ty), _p(size), _p(floor), _p(sat), _p(wf), _p(wt), _p(wi))
        c = self.counters()
        return Snapshot(self.dimx, self.dimy, self.cfg.SCALE, len(self.cfg.soils), c["rand_calls"], c["pool_free"],
                        count, ty, size, floor, sat, wf, wt, wi)

    def tick(self, nwater, nwind, dowater=True, dowind=True):
        self.L.hs_tick(self.h, nwater, nwind, int(dowater), int(dowind))

    def batched_tick(self, nwater, nwind, dowater=True, dowind=True, nthreads=1, shuffle=0):
        """One tick of the BATCHED engine (soil_batch.h) emulated on the host: `nthreads` threads share every "kernel",
        `shuffle` != 0 permutes the order in which the particles of a kernel are run (the result must not depend on it)."""
        self.L.hs_batched_tick(self.h, nwater, nwind, int(dowater), int(dowind), nthreads, shuffle)

    def relaxed_tick(self, nwater, nwind, dowater=True, dowind=True, nthreads=1, shuffle=0, wind_min=0xFFFFFFFF, wind_steps=4, water_steps=1):
        """One tick of the RELAXED schedule (soil_relax.h) emulated on the host; wind runs relaxed (up to `wind_steps` steps per
        epoch) while more than `wind_min` particles are running (smx_set_relax_wind), then goes on under the exclusive schedule."""
        self.L.hs_set_relaxed(1); self.L.hs_set_relax_wind(int(wind_min), int(wind_steps)); self.L.hs_set_relax_water(int(water_steps))
        try:
            self.L.hs_batched_tick(self.h, nwater, nwind, int(dowater), int(dowind), nthreads, shuffle)
        finally:
            self.L.hs_set_relaxed(0); self.L.hs_set_relax_wind(0xFFFFFFFF, 4); self.L.hs_set_relax_water(1)

    def batch_stats(self):
        st = np.zeros(8, np.uint32)
        self.L.hs_batch_stats(_p(st))
        return {"epochs": int(st[0]), "generations": int(st[1]), "children_lost": int(st[2])}

    @staticmethod
    def set_boundary_threads(n: int):
        """Team size of the round boundary (1 = scalar; >1 = threads + barrier, the shape the device kernel runs)."""
        lib().hs_set_boundary_threads(int(n))

    def spec_tick(self, nwater, nwind, dowater=True, dowind=True, nthreads=8, scout=1 | (3 << 4), shuffle=0):
        """One tick with the speculative engine emulated by `nthreads` host threads.
        `scout`: bit0 = read-only trajectory scout, bits 4.. = engine mode (1 blocked-filter, 2 suspend/resume,
        4 = run the particles through the wave-loop structure of soil_coop.h).
        Returns stats: [rounds, aborts, kills, fell_back, executed, critical-path steps] for water then wind."""
        st = np.zeros(12, np.uint32)
        self.L.hs_spec_tick(self.h, nwater, nwind, int(dowater), int(dowind), nthreads, scout, shuffle, _p(st))
        return st


class HostSimStripBackend:
    """soilmachine_amd.strips backend on the host-sim (one HostSim = one strip's context)."""

    def __init__(self, hs: HostSim):
        self.hs, self.L, self.h = hs, hs.L, hs.h
        self._buf = np.zeros(1 << 24, np.uint8)

    def set_strips(self, n, a, b): self.L.hs_set_batch_strips(n, a, b)     # (process-global in the host-sim)
    def set_own(self, x0, x1): self.L.hs_d_set_own(self.h, x0, x1)

    def gen_begin(self, wind, nslots, kids):
        from soilmachine_amd.strips import CHILD_DT
        if kids is None:
            self.L.hs_d_gen_begin(self.h, int(wind), nslots, None, None)
            return None
        kids = np.ascontiguousarray(kids, dtype=CHILD_DT)
        out = np.zeros(nslots, np.uint32)
        self.L.hs_d_gen_begin(self.h, int(wind), nslots, _p(kids), _p(out))
        return out

    def gen_spawn(self, contains):
        n = C.c_uint32()
        c = None if contains is None else np.ascontiguousarray(contains, np.uint32)
        self.L.hs_d_gen_spawn(self.h, None if c is None else _p(c), C.byref(n))
        return int(n.value)

    def next_phase(self): return int(self.L.hs_d_next_phase(self.h))

    def chunk(self, nlive_global):
        n = C.c_uint32()
        self.L.hs_d_chunk(self.h, nlive_global, C.byref(n))
        return int(n.value)

    def gen_end(self):
        from soilmachine_amd.strips import CHILD_DT
        out = np.zeros(1 << 16, CHILD_DT); n = C.c_uint32()
        self.L.hs_d_gen_end(self.h, _p(out), len(out), C.byref(n))
        return out[:int(n.value)].copy()

    def _pack(self, fn, x0, x1):
        k = int(fn(self.h, x0, x1, _p(self._buf), self._buf.size))
        if k == 2**64 - 1:
            raise RuntimeError("strip pack buffer too small")
        return self._buf[:k].tobytes()

    def pack_columns(self, x0, x1): return self._pack(self.L.hs_d_pack_columns, x0, x1)
    def pack_particles(self, x0, x1): return self._pack(self.L.hs_d_pack_particles, x0, x1)

    def unpack_columns(self, b):
        a = np.frombuffer(b, np.uint8); self.L.hs_d_unpack_columns(self.h, _p(a), len(b))

    def unpack_particles(self, b):
        a = np.frombuffer(b, np.uint8); self.L.hs_d_unpack_particles(self.h, _p(a), len(b))

    def grid_begin(self): self.L.hs_d_grid_begin(self.h)
    def grid_sweep(self, phase, lo, hi): self.L.hs_d_grid_sweep(self.h, phase, lo, hi)
    def map_frequency(self): self.L.hs_freq(self.h)
    def reset_frequency(self): pass                                       # (hs_freq does both)
