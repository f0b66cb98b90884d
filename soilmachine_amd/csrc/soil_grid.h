// soil_grid.h -- the EXACT grid pass (WaterParticle::seep(map), water.h:335-343) with its working set in LDS.
//
// The exact pass is one walker in the reference's x-major order (a later cell sees what every earlier cell did, and the
// nested particles of water.h:246-264 draw from the one rand() stream in that order), so its speed is the latency of a
// chain of dependent accesses: ~5 WaterParticle::cascade frames per active cell, each reading two cell records per
// neighbour. Everything one active cell's seep + cascade(.., 3) can touch lies within WIN_R = 4 cells of it
// (one level of recursion per unit of spill, water.h:278-279), and the walk visits neighbours one after the other: a
// window of 9 rows x 64 columns of 32-byte cell records (18 KB) is loaded into LDS by the whole wavefront in ONE memory round
// trip and serves the next ~56 active cells of the row from there.
//
// Coherence is by construction: a cell record is authoritative in the window iff the window covers it (WinSerialPolicy::cref),
// in HBM otherwise -- so a nested particle that leaves the window simply reads and writes the map. The window moves only
// between two active cells (write everything back, load around the next cell). Results are bit-identical to k_grid_serial:
// the same Sim code runs, only where a record lives differs (tests/test_hostsim.py, tests/test_gpu_depth.py).
#pragma once
#include "soil_serial.h"

namespace smx {

constexpr int WIN_R = 4;                        // reach of one active cell's seep + cascade(.., 3)
constexpr int WIN_ROWS = 2 * WIN_R + 1, WIN_WY = 64;

struct WinSerialPolicy : SerialPolicy {
  static constexpr bool WINDOWED = true;
  Sec* win;                                     // WIN_ROWS x WIN_WY records (LDS on the device)
  size_t base = 0, span = 0;                    // cell index of (x0, 0); nrows * dimy
  uint32_t y0 = 0, wy = 0, dimy;
  unsigned long long magic;                     // ceil(2^40 / dimy): u / dimy == (u * magic) >> 40 for u < 2^20
  int x0 = 0, nrows = 0;
  uint32_t topv;                                // freelist[nfree - 1], requested one allocation ahead (node_alloc never waits)
  SMX_D WinSerialPolicy(const DevState& s, Sec* w)
      : SerialPolicy(s), win(w), dimy((uint32_t)s.dimy), magic(((1ull << 40) + (unsigned long long)s.dimy - 1ull) / (unsigned long long)s.dimy) {
    topv = nfree ? s.freelist[nfree - 1] : NIL;
  }
  SMX_D Sec& cref(const DevState& s, size_t c) const {
    const size_t u = c - base;                  // (wraps for c < base)
    if (u < span) {
      const uint32_t r = (uint32_t)(((unsigned long long)u * magic) >> 40);
      const uint32_t col = (uint32_t)u - r * dimy - y0;   // (wraps for y < y0)
      if (col < wy) return win[r * (uint32_t)WIN_WY + col];
    }
    return s.cells[c];
  }
  // does the window hold everything the active cell (x, y) can touch?
  SMX_D bool covers(const DevState& s, int x, int y) const {
    const int xl = x - WIN_R < 0 ? 0 : x - WIN_R, xh = x + WIN_R > s.dimx - 1 ? s.dimx - 1 : x + WIN_R;
    const int yl = y - WIN_R < 0 ? 0 : y - WIN_R, yh = y + WIN_R > s.dimy - 1 ? s.dimy - 1 : y + WIN_R;
    return nrows > 0 && xl >= x0 && xh < x0 + nrows && yl >= (int)y0 && yh < (int)(y0 + wy);
  }
  SMX_D void place(const DevState& s, int x, int y) {
    x0 = x - WIN_R < 0 ? 0 : x - WIN_R;
    nrows = s.dimx - x0 < WIN_ROWS ? s.dimx - x0 : WIN_ROWS;
    const int yy = y - WIN_R < 0 ? 0 : y - WIN_R;
    y0 = (uint32_t)yy;
    wy = (uint32_t)(s.dimy - yy < WIN_WY ? s.dimy - yy : WIN_WY);
    base = (size_t)x0 * s.dimy; span = (size_t)nrows * s.dimy;
  }
  SMX_D void drop() { nrows = 0; span = 0; }
  // the whole team (lane of nlanes) moves records between the map and the window
  SMX_D void load(const DevState& s, uint32_t lane, uint32_t nlanes) const {
    const uint32_t cnt = (uint32_t)nrows * wy;
    for (uint32_t i = lane; i < cnt; i += nlanes) {
      const uint32_t r = i / wy, col = i - r * wy;
      win[r * (uint32_t)WIN_WY + col] = s.cells[base + (size_t)r * dimy + y0 + col];
    }
  }
  SMX_D void flush(const DevState& s, uint32_t lane, uint32_t nlanes) const {
    const uint32_t cnt = (uint32_t)nrows * wy;
    for (uint32_t i = lane; i < cnt; i += nlanes) {
      const uint32_t r = i / wy, col = i - r * wy;
      s.cells[base + (size_t)r * dimy + y0 + col] = win[r * (uint32_t)WIN_WY + col];
    }
  }
  SMX_D uint32_t node_alloc(const DevState& s) {
    if (nfree == 0) return NIL;
    const uint32_t r = topv;
    nfree--;
    topv = nfree ? s.freelist[nfree - 1] : NIL;
    return r;
  }
  SMX_D void node_free(const DevState& s, uint32_t e) { s.freelist[nfree] = e; nfree++; topv = e; }
};

// The walk itself, run by a team of `nlanes` (one wavefront on the device, one thread on the host): lane 0 is the walker,
// the others only move the window. `sync` orders the team's accesses to the window, `bcast` hands lane 0's value to all.
template <class SYNC, class BCAST>
SMX_D void grid_walk_windowed(const DevState& s, const SoilP* soils, Sec* win, uint32_t lane, uint32_t nlanes, SYNC sync, BCAST bcast) {
  WinSerialPolicy pol(s, win);
  Sim<WinSerialPolicy> sim(s, soils, pol);
  sim.grid_mode = true;
  Frame st[MAX_FRAMES];
  int depth = 0;
  const size_t n = (size_t)s.dimx * s.dimy;
  unsigned long long visited = 0;
  Water dummy;
  dummy.pos = {0.f, 0.f}; dummy.speed = {0.f, 0.f}; dummy.volume = 0.0; dummy.sediment = 0.0; dummy.evaprate = 0.0;
  dummy.spill = 0; dummy.ix = dummy.iy = 0; dummy.friction = 0.f; dummy.surface = dummy.contains = 0;
  size_t c = lane == 0 ? sim.next_active(0) : 0;
  c = bcast(c);
  while (c < n) {
    const int x = (int)(c / (size_t)s.dimy), y = (int)(c - (size_t)x * (size_t)s.dimy);
    if (!pol.covers(s, x, y)) {                              // (uniform: every lane holds the same window geometry)
      pol.flush(s, lane, nlanes);
      pol.place(s, x, y);
      sync();
      pol.load(s, lane, nlanes);
      sync();
    }
    if (lane == 0) {
      sim.seep(x, y);                                        // water.h:339
      sim.push_frame(st, depth, x, y, 3);                    // water.h:340 WaterParticle::cascade(ivec2(x,y), .., 3)
      sim.water_drive(dummy, false, false, st, depth);
      visited++;
      c = sim.next_active(c + 1);
    }
    sync();
    c = bcast(c);
  }
  pol.flush(s, lane, nlanes);
  sync();
  if (lane == 0) {
    sim.n_wcasc += n - visited;                              // every non-active cell still "calls" WaterParticle::cascade once (counter parity)
    sim.flush_counters();
    pol.add_counter(s, C_GRID_ACTIVE, visited);
    pol.finish(s);
  }
}

}  // namespace smx
