"""Host-side mirror of the reference's interface for the particle-transport hot path.

``Layermap`` mirrors the surface of ``class Layermap`` (source/layermap.h:127-228) that the tick driver
and the particles use, ``SoilMachine`` mirrors the tick loop of SoilMachine.cpp:283-329 with rendering
removed. All compute is in the HIP library behind include/soilmx.h; nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .snapshot import Snapshot
from .soilfile import SoilConfig, soils_array, layers_array


class SoilmxError(RuntimeError):
    pass


POOLSIZE = 10_000_000      # SoilMachine.cpp:16


def default_pool(dimx: int, dimy: int) -> int:
    """The reference's POOLSIZE (10 M sections, SoilMachine.cpp:16). Deviation, stated: maps of 2048^2 cells and
    more do not fit it (the reference silently drops terrain there), so large maps get
    4 sections per cell instead."""
    return max(POOLSIZE, 4 * int(dimx) * int(dimy))


class Layermap:
    """Device-resident layermap (cells + section pool + frequency planes) behind the C-ABI."""

    def __init__(self, cfg: SoilConfig, dimx: int | None = None, dimy: int | None = None, *, seed: int = 0,
                 pool: int | None = None, device: int = 0, engine: int = capi.ENGINE_SERIAL, initialize: bool = True,
                 x_range: tuple | None = None):
        self.L = capi.load()
        self.cfg = cfg
        self.dimx = int(dimx if dimx is not None else cfg.SIZEX)
        self.dimy = int(dimy if dimy is not None else cfg.SIZEY)
        self.seed = seed
        self.pool = int(pool if pool is not None else default_pool(self.dimx, self.dimy))
        c = capi.Config(self.dimx, self.dimy, cfg.SCALE, device, self.pool, engine, 0)
        h = C.c_void_p()
        self.x_range = x_range                                    # (lo, hi): a STRIP context holding only these columns (smx_create_strip)
        rc = self.L.smx_create(C.byref(c), C.byref(h)) if x_range is None else self.L.smx_create_strip(C.byref(c), int(x_range[0]), int(x_range[1]), C.byref(h))
        self.h = h
        if rc != 0:
            msg = self.L.smx_last_error(h).decode() if h else "smx_create failed"
            if h:
                self.L.smx_destroy(h)
            self.h = None
            raise SoilmxError(f"smx_create: {msg} (rc={rc})")
        self._soils = soils_array(cfg)
        self._chk(self.L.smx_set_soils(self.h, capi.ptr(self._soils), len(self._soils)))
        self._chk(self.L.smx_srand(self.h, seed))                 # srand(SEED) SoilMachine.cpp:41
        if initialize:
            self.initialize(seed)

    # -- plumbing --
    def _chk(self, rc: int):
        if rc != 0:
            raise SoilmxError(self.L.smx_last_error(self.h).decode() + f" (rc={rc})")

    def close(self):
        if getattr(self, "h", None):
            self.L.smx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- Layermap API --
    def initialize(self, seed: int | None = None):
        """Layermap::initialize (layermap.h:163-216) on the device."""
        lay = layers_array(self.cfg)
        self._chk(self.L.smx_initialize(self.h, self.seed if seed is None else seed, capi.ptr(lay), len(lay)))

    def load(self, s: Snapshot, rand_seed: int | None = None):
        """Import columns / frequency planes; re-seed and fast-forward the rand stream to s.rand_calls."""
        self._chk(self.L.smx_import_columns(self.h, capi.ptr(s.count), capi.ptr(s.type), capi.ptr(s.size),
                                            capi.ptr(s.floor), capi.ptr(s.sat)))
        self._chk(self.L.smx_import_frequency(self.h, capi.ptr(s.wfreq), capi.ptr(s.wtrack), capi.ptr(s.windfreq)))
        if rand_seed is not None:
            self._chk(self.L.smx_srand(self.h, rand_seed))
            self._chk(self.L.smx_rand_advance(self.h, s.rand_calls))

    def snapshot(self) -> Snapshot:
        ns = C.c_uint64()
        self._chk(self.L.smx_num_sections(self.h, C.byref(ns)))
        nc, ns = self.dimx * self.dimy, int(ns.value)
        count = np.zeros(nc, np.uint32); ty = np.zeros(ns, np.uint32)
        size = np.zeros(ns); floor = np.zeros(ns); sat = np.zeros(ns)
        wf = np.zeros(nc, np.float32); wt = np.zeros(nc, np.float32); wi = np.zeros(nc, np.float32)
        self._chk(self.L.smx_export_columns(self.h, capi.ptr(count), capi.ptr(ty), capi.ptr(size), capi.ptr(floor), capi.ptr(sat)))
        self._chk(self.L.smx_read_frequency(self.h, capi.ptr(wf), capi.ptr(wt), capi.ptr(wi)))
        c = self.counters()
        return Snapshot(self.dimx, self.dimy, self.cfg.SCALE, len(self.cfg.soils), c["rand_calls"], c["pool_free"],
                        count, ty, size, floor, sat, wf, wt, wi)

    def heights(self) -> np.ndarray:
        out = np.zeros(self.dimx * self.dimy)
        self._chk(self.L.smx_read_heights(self.h, capi.ptr(out)))
        return out

    def surface(self) -> np.ndarray:
        out = np.zeros(self.dimx * self.dimy, np.uint32)
        self._chk(self.L.smx_read_surface(self.h, capi.ptr(out)))
        return out

    def normals(self) -> np.ndarray:
        out = np.zeros((self.dimx * self.dimy, 3), np.float32)
        self._chk(self.L.smx_normals(self.h, capi.ptr(out)))
        return out

    def vertices(self, colors: np.ndarray, cut=None, mode: str = "update") -> np.ndarray:
        """Layermap::update(Vertexpool&) (layermap.h:475-555) for the whole map in one device pass: (cells, 11) float32 =
        position[3], normal[3], color[4], index (the reference's 44-byte Vertex); `colors` = (nsoils, 4) RGBA."""
        colors = np.ascontiguousarray(colors, np.float32).reshape(-1, 4)
        out = np.zeros((self.dimx * self.dimy, 11), np.float32)
        if cut is None:
            self._chk(self.L.smx_fill_vertices(self.h, capi.ptr(colors), colors.shape[0], capi.ptr(out)))
        else:   # mode "update": the global SLICE of Layermap::update (layermap.h:477-510); "slice": Layermap::slice(s) (:557-613)
            self._chk(self.L.smx_fill_vertices_cut(self.h, capi.ptr(colors), colors.shape[0], 0 if mode == "update" else 1, float(cut), capi.ptr(out)))
        return out

    def vertex(self, x: int, y: int, colors: np.ndarray, cut=None, mode: str = "update") -> np.ndarray:
        """ONE column's vertex under the same rules: Layermap::update(ivec2, Vertexpool&) (layermap.h:475-549) -> (11,) float32."""
        colors = np.ascontiguousarray(colors, np.float32).reshape(-1, 4)
        out = np.zeros(11, np.float32)
        m = -1 if cut is None else (0 if mode == "update" else 1)
        self._chk(self.L.smx_fill_vertex_cut(self.h, capi.ptr(colors), colors.shape[0], m, float(cut or 0.0), int(x), int(y), capi.ptr(out)))
        return out

    def heights_bilinear(self, pos: np.ndarray) -> np.ndarray:
        pos = np.ascontiguousarray(pos, np.float32)
        out = np.zeros(pos.shape[0])
        self._chk(self.L.smx_heights_bilinear(self.h, capi.ptr(pos), pos.shape[0], capi.ptr(out)))
        return out

    def add(self, x: int, y: int, size: float, type_: int):
        self._chk(self.L.smx_add(self.h, x, y, size, type_))

    def remove(self, x: int, y: int, h: float) -> float:
        r = C.c_double()
        self._chk(self.L.smx_remove(self.h, x, y, h, C.byref(r)))
        return r.value

    def save(self, path: str):
        """Checkpoint on disk (columns, frequency planes, rand() generator state): smx_save."""
        self._chk(self.L.smx_save(self.h, path.encode()))

    def restore(self, path: str):
        """Resume from a checkpoint written by save() (or a bare snapshot: then the rand stream must be re-seeded)."""
        rc = self.L.smx_load(self.h, path.encode())
        if rc not in (0, 1):
            self._chk(rc)
        return rc == 0

    def digest(self) -> dict:
        """The Appendix-E state digest (sum of heights, section count, type hash) + rand() draws consumed."""
        sh, ns, th = C.c_double(), C.c_uint64(), C.c_uint64()
        self._chk(self.L.smx_digest(self.h, C.byref(sh), C.byref(ns), C.byref(th)))
        return {"sumh": sh.value, "nsec": int(ns.value), "typehash": f"{th.value:016x}",
                "rand_calls": self.counters()["rand_calls"]}

    def counters(self) -> dict:
        c = capi.Counters()
        self._chk(self.L.smx_get_counters_sized(self.h, C.byref(c), C.sizeof(c)))
        return c.as_dict()

    def set_batch_dilate(self, tiles: int):
        self._chk(self.L.smx_set_batch_dilate(self.h, int(tiles)))

    def set_relax_wind(self, min_running: int = 0xFFFFFFFF, steps_per_epoch: int = 4):
        """relaxed schedule: wind steps relaxed (up to `steps_per_epoch` steps per epoch) while more than `min_running` particles run"""
        self._chk(self.L.smx_set_relax_wind(self.h, int(min_running), int(steps_per_epoch)))

    def set_relax_launch(self, persistent: int = -1, tail_at: int = -1):
        """relaxed water epochs: launch shape only, never the result (smx_set_relax_launch): persistent 1 = one cooperative launch per chunk,
        0 = five launches per epoch; tail_at = running particles from which one workgroup runs whole epochs (0..256); -1 = default"""
        self._chk(self.L.smx_set_relax_launch(self.h, int(persistent), int(tail_at)))

    def set_water_generations(self, k: int):
        """throughput engines: the water phase's particles as k consecutive generations of n/k (smx_set_water_generations)"""
        self._chk(self.L.smx_set_water_generations(self.h, int(k)))

    def set_water_stagger(self, gap_epochs: int):
        """relaxed engine: the k generations of a water phase born `gap_epochs` apart inside ONE phase (smx_set_water_stagger; 0 = one after the other)"""
        self._chk(self.L.smx_set_water_stagger(self.h, int(gap_epochs)))

    def water_stagger(self) -> int:
        g = C.c_int32()
        self._chk(self.L.smx_get_water_stagger(self.h, C.byref(g)))
        return int(g.value)

    def water_generations(self) -> int:
        k = C.c_int32()
        self._chk(self.L.smx_get_water_generations(self.h, C.byref(k)))
        return int(k.value)

    def set_batch_strips(self, nstrips: int, inset: int = 16, seam_halfwidth: int = 48):
        self._chk(self.L.smx_set_batch_strips(self.h, int(nstrips), int(inset), int(seam_halfwidth)))

    def batch_stats(self) -> dict:
        e, g, l = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self.L.smx_get_batch_stats(self.h, C.byref(e), C.byref(g), C.byref(l)))
        return {"epochs": int(e.value), "generations": int(g.value), "children_lost": int(l.value)}

    def timing(self) -> dict:
        t = capi.Timing()
        self._chk(self.L.smx_get_timing_sized(self.h, C.byref(t), C.sizeof(t)))
        return t.as_dict()

    def timing_reset(self):
        self._chk(self.L.smx_timing_reset(self.h))

    def sync(self):
        self._chk(self.L.smx_sync(self.h))


class SoilMachine:
    """The tick driver of SoilMachine.cpp:283-329 (rendering removed) on top of a device ``Layermap``."""

    def __init__(self, cfg: SoilConfig, size: int | None = None, *, dimx: int | None = None, dimy: int | None = None,
                 seed: int = 0, nwater: int | None = None, nwind: int | None = None, dowater: bool = True,
                 dowind: bool = True, **kw):
        self.cfg = cfg
        self.map = Layermap(cfg, dimx if dimx is not None else size, dimy if dimy is not None else size, seed=seed, **kw)
        self.nwater = cfg.NWATER if nwater is None else nwater
        self.nwind = cfg.NWIND if nwind is None else nwind
        self.dowater, self.dowind = dowater, dowind

    def tick(self, n: int = 1, sync: bool = False):
        m = self.map
        for _ in range(n):
            m._chk(m.L.smx_tick(m.h, self.nwater, self.nwind, int(self.dowater), int(self.dowind)))
        if sync:
            m.sync()
            self._check_pool()

    def _check_pool(self):
        """The reference prints "Memory Pool Out-Of-Elements" and drops the section (layermap.h:92-95); here the
        counter is surfaced as a warning whenever it has grown since the last synchronised tick."""
        ov = self.map.counters()["pool_overflow"]
        if ov > getattr(self, "_pool_overflow_seen", 0):
            import warnings
            warnings.warn(f"soilmx: section pool exhausted ({ov} pool.get() failures so far, capacity {self.map.pool}): "
                          f"sections are being dropped exactly as the reference's POOLSIZE overflow does", RuntimeWarning)
            self._pool_overflow_seen = ov

    # phase-by-phase access, as the reference's host loop spells it out
    def water(self, n=None): self.map._chk(self.map.L.smx_tick_water(self.map.h, self.nwater if n is None else n))
    def grid_pass(self): self.map._chk(self.map.L.smx_grid_pass(self.map.h))
    def wind(self, n=None): self.map._chk(self.map.L.smx_tick_wind(self.map.h, self.nwind if n is None else n))
    def map_frequency(self): self.map._chk(self.map.L.smx_map_frequency(self.map.h))
    def reset_frequency(self): self.map._chk(self.map.L.smx_reset_frequency(self.map.h))
