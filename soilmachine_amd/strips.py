"""Column strips: ONE layermap sharded over several devices (batched engine only; DESIGN.md "Multi-GPU").

The map is cut into x-strips (Layermap index is x*dimy + y, layermap.h:151: an x-strip is one contiguous block). The
batched engine's schedule alternates two kinds of 32-epoch chunks (csrc/soil_batch.h BStrips):

  INTERIOR  a particle acts iff its whole reservation lies `a` cells inside one strip  -> strips touch disjoint cells,
            every rank runs its own, nothing is exchanged;
  SEAM      a particle acts iff its reservation lies inside one seam zone [x_b - b, x_b + b) around a strip boundary
            -> the rank LEFT of the boundary runs the seam on a halo of b columns -- or b cells inside a strip, out of
            every seam zone's reach (the particles that are nowhere near a seam do not idle).

Per chunk the ranks agree on the number of running particles (one all-reduce of 4 bytes: the schedule depends on it);
around every SEAM chunk neighbours exchange, point to point, (i) the halo columns that changed (dirty 4x4 tiles only)
and (ii) the particles that sit in the halo -- one message per direction. Nested particles are all-gathered once per generation (they are sorted by
key: slot numbers = priorities must be global). The result is a function of (nstrips, a, b) only: running the strips on
one device with smx_set_batch_strips() gives the same bits (tests/test_strips.py, loop-back), as does any transport.
The protocol itself runs in C++ (csrc/soil_strips_host.h, smx_strips_*); LibStripRank only attaches a transport. The Python
restatement of the driver that the C++ driver is tested against lives with the tests (tests/strips_ref.py).

Every rank allocates the full-size arrays and keeps only its strip + halo current (addresses stay global; at 288 GB per
device the unused part costs nothing that matters). Each rank initialises the whole terrain itself (3 ms at 4096^2):
no initial scatter.

Transports: ThreadComm (ranks = threads of one process; loop-back on one device or on the host-sim), TorchComm
(torch.distributed: gloo on the CPU, nccl = RCCL over xGMI on the GPUs).
"""
from __future__ import annotations

import queue
import threading

import numpy as np

# csrc/soil_batch.h BChild: a nested particle suspended at the edge of its parent's region, as it stood (round 5)
CHILD_DT = np.dtype([("key", "<u8"), ("pos", "<f4", (2,)), ("speed", "<f4", (2,)), ("volume", "<f8"), ("sediment", "<f8"), ("spill", "<i4"),
                     ("contains", "<u4"), ("ended", "<u4"), ("pad", "<u4")])
assert CHILD_DT.itemsize == 56
INTERIOR, SEAM = 0, 1


class StripGeometry:
    """n equal strips with boundaries on multiples of 16 cells (== csrc/soil_batch.h batch_make_strips)."""

    def __init__(self, dimx: int, n: int, a: int = 16, b: int = 48):
        if b - a < 32:
            raise ValueError("seam half-width - inset must be >= 32 cells (the widest reservation)")
        if n > 1 and dimx // n < 2 * b + 16:
            raise ValueError("strips narrower than two seam zones")
        self.n, self.a, self.b, self.dimx = n, a, b, dimx
        self.x = [dimx if g == n else (dimx * g // n) // 16 * 16 for g in range(n + 1)]

    def held(self, rank: int, dimy: int):
        """the columns a STRIP context of `rank` must hold (Layermap(x_range=...), smx_create_strip): its strip, the seam zone right of
        it and 16 columns of slack -- or None (hold everything) when that range cannot start on a 64-cell boundary"""
        lo = max(0, self.x[rank] - 16) if rank > 0 else 0
        hi = min(self.dimx, self.x[rank + 1] + self.b + 16) if rank < self.n - 1 else self.dimx
        return (lo, hi) if (lo * dimy) % 64 == 0 else None


# ------------------------------------------------------------------------------------------------ transports
class ThreadComm:
    """Ranks are threads of this process (loop-back transport). world = ThreadComm.world(n) -> one comm per rank."""

    class _Shared:
        def __init__(self, n):
            self.n = n
            self.barrier = threading.Barrier(n)
            self.slots = [None] * n
            self.q = {(s, d): queue.Queue() for s in range(n) for d in range(n)}

    def __init__(self, shared, rank):
        self.sh, self.rank, self.size = shared, rank, shared.n
        self.bytes_sent = 0
        self.messages = 0

    @staticmethod
    def world(n):
        sh = ThreadComm._Shared(n)
        return [ThreadComm(sh, r) for r in range(n)]

    def _gather(self, v):
        self.sh.slots[self.rank] = v
        self.sh.barrier.wait()
        out = list(self.sh.slots)
        self.sh.barrier.wait()
        return out

    def allreduce_sum(self, v: int) -> int:
        return int(sum(self._gather(int(v))))

    def allreduce_min_u32(self, a: np.ndarray) -> np.ndarray:
        return np.minimum.reduce(self._gather(a))

    def allgather_bytes(self, b: bytes):
        return self._gather(b)

    def send(self, dst: int, b: bytes):
        self.bytes_sent += len(b); self.messages += 1
        self.sh.q[(self.rank, dst)].put(b)

    def recv(self, src: int) -> bytes:
        return self.sh.q[(src, self.rank)].get()


class TorchComm:
    """torch.distributed transport: backend "gloo" (CPU tensors) or "nccl" (= RCCL; tensors on this rank's GPU)."""

    def __init__(self, dist, device=None):
        import torch
        self.t, self.dist = torch, dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self.dev = device if device is not None else torch.device("cpu")
        self.bytes_sent = 0
        self.messages = 0

    def allreduce_sum(self, v: int) -> int:
        x = self.t.tensor([int(v)], dtype=self.t.int64, device=self.dev)
        self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM)
        return int(x.item())

    def allreduce_min_u32(self, a: np.ndarray) -> np.ndarray:
        x = self.t.from_numpy(a.astype(np.int64)).to(self.dev)
        self.dist.all_reduce(x, op=self.dist.ReduceOp.MIN)
        return x.cpu().numpy().astype(np.uint32)

    def allgather_bytes(self, b: bytes):
        n = self.t.tensor([len(b)], dtype=self.t.int64, device=self.dev)
        sizes = [self.t.zeros(1, dtype=self.t.int64, device=self.dev) for _ in range(self.size)]
        self.dist.all_gather(sizes, n)
        m = max(int(s.item()) for s in sizes)
        buf = self.t.zeros(max(m, 1), dtype=self.t.uint8, device=self.dev)
        if len(b):
            buf[:len(b)] = self.t.frombuffer(bytearray(b), dtype=self.t.uint8).to(self.dev)
        outs = [self.t.zeros(max(m, 1), dtype=self.t.uint8, device=self.dev) for _ in range(self.size)]
        self.dist.all_gather(outs, buf)
        return [bytes(o[:int(s.item())].cpu().numpy().tobytes()) for o, s in zip(outs, sizes)]

    def send(self, dst: int, b: bytes):
        self.bytes_sent += len(b); self.messages += 1
        n = self.t.tensor([len(b)], dtype=self.t.int64, device=self.dev)
        self.dist.send(n, dst)
        if len(b):
            self.dist.send(self.t.frombuffer(bytearray(b), dtype=self.t.uint8).to(self.dev), dst)

    def recv(self, src: int) -> bytes:
        n = self.t.zeros(1, dtype=self.t.int64, device=self.dev)
        self.dist.recv(n, src)
        k = int(n.item())
        if k == 0:
            return b""
        buf = self.t.zeros(k, dtype=self.t.uint8, device=self.dev)
        self.dist.recv(buf, src)
        return bytes(buf.cpu().numpy().tobytes())

# ------------------------------------------------------------------------------------------------ the library's driver
class LibStripRank:
    """The strip driver of the library (csrc/soil_strips_host.h, smx_strips_*): C++ runs the protocol; this class only attaches
    a transport. transport="rccl": ncclSend/ncclRecv on device buffers (`comm` = a TorchComm or anything with rank/size/
    allgather_bytes to hand the unique id around). Otherwise `comm` (ThreadComm, TorchComm on gloo) is wrapped in callbacks that
    move HOST buffers -- the protocol above the transport is the same C++ either way."""

    def __init__(self, lm, comm, geom: StripGeometry, transport: str = "host"):
        import ctypes as C
        from . import capi
        self.C, self.capi, self.lm, self.L, self.h, self.comm, self.g = C, capi, lm, lm.L, lm.h, comm, geom
        assert comm.size == geom.n
        self.x0, self.x1 = geom.x[comm.rank], geom.x[comm.rank + 1]
        if transport == "rccl":
            uid = (C.c_uint8 * 128)()
            if comm.rank == 0:
                lm._chk(self.L.smx_strips_rccl_unique_id(uid))
            uid_b = comm.allgather_bytes(bytes(uid))[0]              # rank 0's id reaches everybody
            buf = (C.c_uint8 * 128).from_buffer_copy(uid_b)
            lm._chk(self.L.smx_strips_attach_rccl(self.h, buf, comm.rank, comm.size, geom.a, geom.b))
        else:
            self._pending = {}                                        # src -> a message the library's buffer was too small for
            self._t = self._make_transport()
            lm._chk(self.L.smx_strips_attach(self.h, C.byref(self._t), comm.rank, comm.size, geom.a, geom.b))

    def _make_transport(self):
        C, capi, comm = self.C, self.capi, self.comm

        def send(user, dst, buf, n):
            comm.send(int(dst), C.string_at(buf, n)); return 0

        def recv(user, src, buf, cap, out):
            b = self._pending.pop(int(src), None)
            if b is None:
                b = comm.recv(int(src))
            out[0] = len(b)
            if len(b) > cap:
                self._pending[int(src)] = b; return 1
            C.memmove(buf, b, len(b)); return 0

        def red_sum(user, v):
            v[0] = comm.allreduce_sum(int(v[0])); return 0

        def red_max(user, v):
            v[0] = max(int.from_bytes(b, "little") for b in comm.allgather_bytes(int(v[0]).to_bytes(8, "little"))); return 0

        def red_min(user, a, n):
            arr = np.ctypeslib.as_array(a, shape=(int(n),))
            arr[:] = comm.allreduce_min_u32(arr.copy()); return 0

        def gather(user, mine, n, allp, each, sizes):
            parts = comm.allgather_bytes(C.string_at(mine, n) if n else b"")
            for r, b in enumerate(parts):
                sizes[r] = len(b)
                if len(b):
                    C.memmove(allp + r * each, b, len(b))
            return 0
        self._cbs = (capi.SEND_FN(send), capi.RECV_FN(recv), capi.RED64_FN(red_sum), capi.RED64_FN(red_max), capi.MIN32_FN(red_min), capi.GATHER_FN(gather))
        return capi.Transport(None, *self._cbs)

    def tick(self, nwater: int, nwind: int, dowater: bool = True, dowind: bool = True):
        self.lm._chk(self.L.smx_strips_tick(self.h, int(nwater), int(nwind), int(dowater), int(dowind)))

    @property
    def stats(self):
        C = self.C
        v = [C.c_uint64() for _ in range(5)]
        self.lm._chk(self.L.smx_strips_stats(self.h, *[C.byref(x) for x in v]))
        d = dict(zip(("chunks", "seam_chunks", "generations", "bytes_sent", "messages"), (int(x.value) for x in v)))
        w = [C.c_uint64(), C.c_uint64()]                            # what the transport itself cost (round 6): stream synchronisations, grouped operations
        self.lm._chk(self.L.smx_strips_sync_stats(self.h, C.byref(w[0]), C.byref(w[1])))
        d["transport_host_syncs"], d["transport_groups"] = int(w[0].value), int(w[1].value)
        if d["seam_chunks"]:
            d["transport_host_syncs_per_seam_chunk"] = d["transport_host_syncs"] / d["seam_chunks"]
        return d

    def selfcheck(self, nrec: int, nbytes: int, npart: int):
        """One synthetic seam message from this rank to itself through the attached transport's exchange (smx_strips_selfcheck); raises when it comes back changed."""
        self.lm._chk(self.L.smx_strips_selfcheck(self.h, int(nrec), int(nbytes), int(npart)))

    def detach(self):
        self.lm._chk(self.L.smx_strips_detach(self.h))


# ------------------------------------------------------------------------------------------------ one rank
def merge_owned(geom: StripGeometry, snaps, dimy: int):
    """One global Snapshot from per-rank snapshots: every column from the rank that owns it."""
    from .snapshot import Snapshot
    s0 = snaps[0]
    owner = np.zeros(geom.dimx * dimy, np.int32)
    for r in range(geom.n):
        owner[geom.x[r] * dimy: geom.x[r + 1] * dimy] = r
    count = np.zeros_like(s0.count)
    cols = {k: [] for k in ("type", "size", "floor", "sat")}
    starts = [np.concatenate([[0], np.cumsum(s.count.astype(np.int64))]) for s in snaps]
    for r in range(geom.n):
        c0, c1 = geom.x[r] * dimy, geom.x[r + 1] * dimy
        count[c0:c1] = snaps[r].count[c0:c1]
        a, b = starts[r][c0], starts[r][c1]
        for k in cols:
            cols[k].append(getattr(snaps[r], k)[a:b])
    planes = {}
    for k in ("wfreq", "wtrack", "windfreq"):       # frequency order: index y*dimx + x
        out = np.zeros_like(getattr(s0, k)).reshape(dimy, geom.dimx)
        for r in range(geom.n):
            out[:, geom.x[r]:geom.x[r + 1]] = getattr(snaps[r], k).reshape(dimy, geom.dimx)[:, geom.x[r]:geom.x[r + 1]]
        planes[k] = out.reshape(-1)
    return Snapshot(s0.dimx, s0.dimy, s0.scale, s0.nsoils, s0.rand_calls, s0.pool_free, count,
                    np.concatenate(cols["type"]), np.concatenate(cols["size"]), np.concatenate(cols["floor"]),
                    np.concatenate(cols["sat"]), planes["wfreq"], planes["wtrack"], planes["windfreq"])
