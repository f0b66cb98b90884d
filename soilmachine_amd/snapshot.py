"""Binary state snapshot (layout: oracle/snapshot.h) -- interchange between the oracle binaries,
the C-ABI import/export (include/soilmx.h) and the tests.  The reference has no on-disk state
format (SURVEY.md §5 "Checkpoint / resume"); this one stores, per column in Layermap order
``x*dimy + y`` (layermap.h:151), the run-length sections bottom -> top."""
from __future__ import annotations

from dataclasses import dataclass
import numpy as np

MAGIC = b"SMXSNAP1"
_HDR = np.dtype([("magic", "S8"), ("dimx", "<i4"), ("dimy", "<i4"), ("scale", "<i4"), ("nsoils", "<i4"),
                 ("ncells", "<u8"), ("nsec", "<u8"), ("rand_calls", "<u8"), ("pool_free", "<u8"), ("reserved", "<u8")])
assert _HDR.itemsize == 64


@dataclass
class Snapshot:
    dimx: int
    dimy: int
    scale: int
    nsoils: int
    rand_calls: int
    pool_free: int
    count: np.ndarray      # u32 [ncells]
    type: np.ndarray       # u32 [nsec]
    size: np.ndarray       # f64 [nsec]
    floor: np.ndarray      # f64 [nsec]
    sat: np.ndarray        # f64 [nsec]
    wfreq: np.ndarray      # f32 [ncells]  index y*dimx+x
    wtrack: np.ndarray
    windfreq: np.ndarray

    @property
    def ncells(self) -> int:
        return self.dimx * self.dimy

    @property
    def nsec(self) -> int:
        return int(self.type.shape[0])

    def heights(self) -> np.ndarray:
        """height(ivec2) per cell (layermap.h:422-425): floor+size of the top section, 0 if empty."""
        end = np.cumsum(self.count.astype(np.int64))
        h = np.zeros(self.ncells, dtype=np.float64)
        nz = self.count > 0
        topi = end[nz] - 1
        h[nz] = self.floor[topi] + self.size[topi]
        return h

    def digest(self) -> dict:
        """SURVEY.md Appendix E digest: sum of heights in cell order, section count, type hash (top->bottom)."""
        end = np.cumsum(self.count.astype(np.int64))
        sumh = 0.0
        for v in self.heights():          # sequential double accumulation, x outer / y inner
            sumh += float(v)
        hsh = 1469598103934665603
        mask = (1 << 64) - 1
        start = end - self.count
        ty = self.type
        for c in range(self.ncells):
            for k in range(int(end[c]) - 1, int(start[c]) - 1, -1):
                hsh = ((hsh ^ int(ty[k])) * 1099511628211) & mask
        return {"sumh": sumh, "nsec": self.nsec, "typehash": f"{hsh:016x}"}


def read_snapshot(path: str) -> Snapshot:
    raw = np.fromfile(path, dtype=np.uint8)
    h = raw[:64].view(_HDR)[0]
    if bytes(h["magic"]) != MAGIC:
        raise ValueError(f"{path}: not a soilmx snapshot")
    nc, ns = int(h["ncells"]), int(h["nsec"])
    off = 64

    def take(dt, n):
        nonlocal off
        a = raw[off:off + n * np.dtype(dt).itemsize].view(dt).copy()
        off += n * np.dtype(dt).itemsize
        return a
    count = take("<u4", nc); ty = take("<u4", ns)
    size = take("<f8", ns); floor = take("<f8", ns); sat = take("<f8", ns)
    wfreq = take("<f4", nc); wtrack = take("<f4", nc); windfreq = take("<f4", nc)
    return Snapshot(int(h["dimx"]), int(h["dimy"]), int(h["scale"]), int(h["nsoils"]), int(h["rand_calls"]),
                    int(h["pool_free"]), count, ty, size, floor, sat, wfreq, wtrack, windfreq)


def write_snapshot(path: str, s: Snapshot) -> None:
    h = np.zeros(1, dtype=_HDR)
    h["magic"] = MAGIC
    h["dimx"], h["dimy"], h["scale"], h["nsoils"] = s.dimx, s.dimy, s.scale, s.nsoils
    h["ncells"], h["nsec"], h["rand_calls"], h["pool_free"] = s.ncells, s.nsec, s.rand_calls, s.pool_free
    with open(path, "wb") as f:
        f.write(h.tobytes())
        for a, dt in ((s.count, "<u4"), (s.type, "<u4"), (s.size, "<f8"), (s.floor, "<f8"), (s.sat, "<f8"),
                      (s.wfreq, "<f4"), (s.wtrack, "<f4"), (s.windfreq, "<f4")):
            f.write(np.ascontiguousarray(a, dtype=dt).tobytes())


def compare(a: Snapshot, b: Snapshot) -> list:
    """Bitwise comparison; returns a list of human-readable mismatches (empty == identical state)."""
    bad = []
    if (a.dimx, a.dimy) != (b.dimx, b.dimy):
        return [f"dim {a.dimx}x{a.dimy} vs {b.dimx}x{b.dimy}"]
    if not np.array_equal(a.count, b.count):
        d = np.nonzero(a.count != b.count)[0]
        bad.append(f"section counts differ in {d.size} cells (first cell {int(d[0])}: {int(a.count[d[0]])} vs {int(b.count[d[0]])})")
        return bad
    for name in ("type", "size", "floor", "sat", "wfreq", "wtrack", "windfreq"):
        x, y = getattr(a, name), getattr(b, name)
        if x.dtype.kind == "f":
            same = x.view(f"u{x.dtype.itemsize}") == y.view(f"u{y.dtype.itemsize}")
        else:
            same = x == y
        if not same.all():
            d = np.nonzero(~same)[0]
            bad.append(f"{name}: {d.size} elements differ (first idx {int(d[0])}: {x[d[0]]!r} vs {y[d[0]]!r})")
    if a.rand_calls != b.rand_calls:
        bad.append(f"rand_calls {a.rand_calls} vs {b.rand_calls}")
    return bad
