"""Python restatement of the column-strip driver (TEST INFRASTRUCTURE): StripRank runs the tick of SoilMachine.cpp:283-329 on one
strip over the step-wise smx_d_* / hs_d_* entry points -- what the library's C++ driver (csrc/soil_strips_host.h, smx_strips_tick,
reached through soilmachine_amd.strips.LibStripRank) is compared with, bit for bit. Moved out of the product package in round 4."""
from __future__ import annotations

import threading

import numpy as np

from soilmachine_amd.strips import CHILD_DT, INTERIOR, SEAM, StripGeometry, ThreadComm, TorchComm, LibStripRank, merge_owned  # noqa: F401


class DeviceStripBackend:
    """One strip's context in libsoilmx (include/soilmx.h smx_d_*), wrapped for StripRank. `lm` = machine.Layermap created
    with engine=capi.ENGINE_BATCHED (full-size map; every rank initialises or imports the same state)."""

    def __init__(self, lm, buf_mb: int = 64):
        import ctypes as C
        from soilmachine_amd import capi
        self.C, self.capi, self.lm, self.L, self.h = C, capi, lm, lm.L, lm.h
        self._buf = np.zeros(buf_mb << 20, np.uint8)

    def _chk(self, rc): self.lm._chk(rc)
    def set_strips(self, n, a, b): self._chk(self.L.smx_set_batch_strips(self.h, n, a, b))
    def set_own(self, x0, x1): self._chk(self.L.smx_d_set_own(self.h, x0, x1))

    def gen_begin(self, wind, nslots, kids):
        if kids is None:
            self._chk(self.L.smx_d_gen_begin(self.h, int(wind), nslots, None, None))
            return None
        kids = np.ascontiguousarray(kids, dtype=CHILD_DT)
        out = np.zeros(max(nslots, 1), np.uint32)
        self._chk(self.L.smx_d_gen_begin(self.h, int(wind), nslots, self.capi.ptr(kids), self.capi.ptr(out)))
        return out[:nslots]

    def gen_spawn(self, contains):
        n = self.C.c_uint32()
        c = None if contains is None else np.ascontiguousarray(contains, np.uint32)
        self._chk(self.L.smx_d_gen_spawn(self.h, None if c is None else self.capi.ptr(c), self.C.byref(n)))
        return int(n.value)

    def next_phase(self): return int(self.L.smx_d_next_phase(self.h))

    def chunk(self, nlive_global):
        n = self.C.c_uint32()
        self._chk(self.L.smx_d_chunk(self.h, int(nlive_global), self.C.byref(n)))
        return int(n.value)

    def gen_end(self):
        out = np.zeros(1 << 16, CHILD_DT); n = self.C.c_uint32()
        rc = self.L.smx_d_gen_end(self.h, self.capi.ptr(out), len(out), self.C.byref(n))
        if rc != 0:
            if int(n.value) <= len(out):
                self._chk(rc)
            out = np.zeros(int(n.value), CHILD_DT)                # "buffer too small": *n says how many records wait (the call is repeatable)
            self._chk(self.L.smx_d_gen_end(self.h, self.capi.ptr(out), len(out), self.C.byref(n)))
        return out[:int(n.value)].copy()

    def _pack(self, fn, x0, x1):
        k = self.C.c_uint64()
        self._chk(fn(self.h, x0, x1, self.capi.ptr(self._buf), self._buf.size, self.C.byref(k)))
        return self._buf[:int(k.value)].tobytes()

    def pack_columns(self, x0, x1): return self._pack(self.L.smx_d_pack_columns, x0, x1)
    def pack_particles(self, x0, x1): return self._pack(self.L.smx_d_pack_particles, x0, x1)

    def unpack_columns(self, b):
        a = np.frombuffer(b, np.uint8); self._chk(self.L.smx_d_unpack_columns(self.h, self.capi.ptr(a), len(b)))

    def unpack_particles(self, b):
        a = np.frombuffer(b, np.uint8); self._chk(self.L.smx_d_unpack_particles(self.h, self.capi.ptr(a), len(b)))

    def grid_begin(self): self._chk(self.L.smx_d_grid_begin(self.h))
    def grid_sweep(self, phase, lo, hi): self._chk(self.L.smx_d_grid_sweep_cols(self.h, phase, lo, hi))
    def map_frequency(self): self._chk(self.L.smx_map_frequency(self.h))
    def reset_frequency(self): self._chk(self.L.smx_reset_frequency(self.h))


class StripRank:
    """The tick of SoilMachine.cpp:283-329 on ONE strip. `backend` = DeviceStripBackend (libsoilmx) or the host-sim's."""

    def __init__(self, backend, comm, geom: StripGeometry):
        self.be, self.comm, self.g = backend, comm, geom
        assert comm.size == geom.n
        r = comm.rank
        self.x0, self.x1 = geom.x[r], geom.x[r + 1]
        backend.set_strips(geom.n, geom.a, geom.b)
        backend.set_own(self.x0, self.x1)
        self.stats = {"chunks": 0, "seam_chunks": 0, "generations": 0}
        self.water_generations = 8                                # smx_set_water_generations' default (csrc/soil_batch.h SMX_WATER_GENERATIONS)
        self.water_stagger = 0                                    # smx_set_water_stagger's default (SMX_WATER_STAGGER); > 0 and a relaxed backend: ONE staggered phase

    # -- halo traffic (nearest neighbours only). A rank holds, besides its strip, the b columns right of its right boundary.
    # one message per direction: [u64 length of the column part][columns][particles]
    @staticmethod
    def _bundle(cols: bytes, parts: bytes) -> bytes:
        return len(cols).to_bytes(8, "little") + cols + parts

    def _apply_bundle(self, b: bytes, particles: bool):
        n = int.from_bytes(b[:8], "little")
        self.be.unpack_columns(b[8:8 + n])
        if particles:
            self.be.unpack_particles(b[8 + n:])

    def _send_band_left(self, particles: bool):
        """band [x0, x0 + b) of this strip -> the rank on the left (which runs that seam)"""
        r, g = self.comm.rank, self.g
        if r > 0:
            self.comm.send(r - 1, self._bundle(self.be.pack_columns(self.x0, self.x0 + g.b),
                                               self.be.pack_particles(self.x0, self.x0 + g.b) if particles else b""))

    def _recv_band_from_right(self, particles: bool):
        r, g = self.comm.rank, self.g
        if r < g.n - 1:
            self._apply_bundle(self.comm.recv(r + 1), particles)

    def _send_band_right(self, particles: bool):
        """the halo [x1, x1 + b) goes home to the rank on the right, with every particle that sits beyond x1"""
        r, g = self.comm.rank, self.g
        if r < g.n - 1:
            self.comm.send(r + 1, self._bundle(self.be.pack_columns(self.x1, self.x1 + g.b),
                                               self.be.pack_particles(self.x1, g.dimx) if particles else b""))

    def _recv_band_from_left(self, particles: bool):
        r = self.comm.rank
        if r > 0:
            self._apply_bundle(self.comm.recv(r - 1), particles)

    def sync_halos(self):
        self._send_band_left(False)
        self._recv_band_from_right(False)

    # -- one particle phase: generations of chunks
    def _phase(self, wind: bool, n: int, kids=None):
        be, comm = self.be, self.comm
        nslots = n if kids is None else len(kids)
        while nslots > 0:
            contains = be.gen_begin(wind, nslots, kids)
            if kids is not None:
                contains = comm.allreduce_min_u32(contains)          # each child's `contains` comes from the owner of its random cell
            nlive = comm.allreduce_sum(be.gen_spawn(contains if kids is not None else None))
            while nlive > 0:
                seam = self.g.n > 1 and be.next_phase() == SEAM
                if seam:
                    self._send_band_left(True)
                    self._recv_band_from_right(True)
                local = be.chunk(nlive)
                if seam:
                    self._send_band_right(True)
                    self._recv_band_from_left(True)
                    self.stats["seam_chunks"] += 1
                self.stats["chunks"] += 1
                nlive = comm.allreduce_sum(local)
            mine = be.gen_end()
            self.stats["generations"] += 1
            if wind:
                break
            parts = [np.frombuffer(b, dtype=CHILD_DT) for b in comm.allgather_bytes(mine.tobytes())]
            kids = np.concatenate(parts) if parts else np.zeros(0, CHILD_DT)
            kids = kids[np.argsort(kids["key"], kind="stable")]
            nslots = len(kids)
        if self.g.n > 1:
            self.sync_halos()                                        # the halo is current again (frequency planes, next phase)

    def _grid(self):
        be, g, r = self.be, self.g, self.comm.rank
        be.grid_begin()
        be.grid_sweep(INTERIOR, self.x0, self.x1)
        if g.n > 1:
            self.sync_halos()
            if r < g.n - 1:
                be.grid_sweep(SEAM, self.x1 - g.b, self.x1 + g.b)
            self._send_band_right(False)
            self._recv_band_from_left(False)
        mine = be.gen_end()
        parts = [np.frombuffer(b, dtype=CHILD_DT) for b in self.comm.allgather_bytes(mine.tobytes())]
        kids = np.concatenate(parts) if parts else np.zeros(0, CHILD_DT)
        kids = kids[np.argsort(kids["key"], kind="stable")]
        if len(kids):
            self._phase(False, 0, kids)
        elif g.n > 1:
            self.sync_halos()

    def tick(self, nwater: int, nwind: int, dowater: bool = True, dowind: bool = True):
        if dowater:                                               # smx_strips_tick: k consecutive top-level generations of n/k (csrc/soil_batch.h SMX_WATER_GENERATIONS)
            left, k = nwater, self.water_generations
            if self.water_stagger > 0 and k > 1 and left > 0:      # (the backend -- device context or hostsim -- carries the same stagger setting: its gen_begin marks the phase)
                self._phase(False, left); left = 0
            for q in range(k):
                nq = left // (k - q); left -= nq
                if nq:
                    self._phase(False, nq)
            self._grid()
        if dowind:
            self._phase(True, nwind)
        if dowater:
            self.be.map_frequency()
            self.be.reset_frequency()


def run_threads(ranks, fn):
    """loop-back: run fn(rank_object) for every StripRank in its own thread (the transports block on each other)"""
    errs = []

    def wrap(rk):
        try:
            fn(rk)
        except BaseException as e:   # noqa: BLE001 -- re-raised below
            errs.append(e)
            try:
                rk.comm.sh.barrier.abort()
            except Exception:
                pass

    th = [threading.Thread(target=wrap, args=(rk,)) for rk in ranks]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]

