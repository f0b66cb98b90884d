// soil_core.h -- device-side state layout and the exact (reference-order) particle transport code.
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   cells[x*dimy+y]  : one 32-byte Sec record per column = the column's TOP section stored INLINE
//                      {size, floor, sat : f64; type : u32; prev : u32 index of the section below}
//                      -> Layermap::height()/surface()/top() (layermap.h:150,417-425) are ONE aligned
//                         32-byte load, no pointer chase; an empty column has type == EMPTY and
//                         size = floor = 0 (height 0.0, surface 0 as the reference returns for NULL).
//   pool[i]          : the same 32-byte record for BURIED sections (reached through prev), plus a
//                      free-index stack. Replaces sec / secpool (layermap.h:37-119); node identity is
//                      never observable in the reference, so this allocator is parity-safe.
//   flags[cell]      : u8, bit0 = top section is water ("Air"), bit1 = column holds a non-zero
//                      saturation somewhere (sticky) -> input of the grid-pass classification kernel.
//   wfreq/wtrack/windfreq[y*dimx+x] : f32 planes (water.h:22-23,345-346; wind.h:25,48).
//
// Arithmetic follows SURVEY.md Appendix A exactly (mixed float/double promotions, no FMA: this file
// must be compiled with -ffp-contract=off). Every function cites the reference lines it implements.
#pragma once
#include <stdint.h>
#include <math.h>

#ifndef SMX_D
#define SMX_D __device__ __forceinline__
#define SMX_DN __device__ __noinline__
#endif

namespace smx {

constexpr uint32_t NIL = 0xFFFFFFFFu;
constexpr uint32_t EMPTY = 0xFFFFFFFFu;   // cells[c].type of an empty column (dat[c] == NULL)
constexpr uint32_t AIR = 0;               // soilmap["Air"] (surface.h:53-57); doubles as water
constexpr uint8_t F_AIR = 1, F_SAT = 2;

struct alignas(32) Sec {
  double size, floor, sat;
  uint32_t type, prev;
};

struct SoilP {   // == smx_soil
  uint32_t transports, erodes, cascades, abrades;
  float density, porosity, solubility, equrate, friction, erosionrate, maxdiff, settling, suspension, abrasion;
};

struct RandState {   // glibc TYPE_3 additive feedback generator: r[i] = r[i-31] + r[i-3]
  uint32_t ring[31];
  uint32_t idx;
  uint64_t calls;
};

enum Ctr { C_STEPS_WATER_TOP = 0, C_STEPS_WATER_ALL, C_STEPS_WIND, C_NESTED, C_FLOODS, C_CASCADE_CALLS,
           C_CASCADE_TRANSFERS, C_WCASCADE_CALLS, C_GRID_ACTIVE, C_POOL_OVERFLOW, C_LIVE_SECTIONS, C_COUNT };

struct DevState {
  int dimx, dimy, scale;
  uint32_t nsoils;
  Sec* cells;
  Sec* pool;
  uint32_t* freelist;        // stack of free pool indices
  uint32_t* free_count;      // number of entries in freelist
  uint8_t* flags;
  float* wfreq; float* wtrack; float* windfreq;
  const SoilP* soils;
  RandState* rnd;
  unsigned long long* ctr;   // Ctr
  unsigned long long* active;// grid-pass bitmap, one bit per cell in cell order (level 0)
  unsigned long long* active1;// level 1: bit j set => level-0 word j may be non-zero
  unsigned long long* active2;// level 2: bit j set => level-1 word j may be non-zero
  uint64_t pool_capacity;    // max live sections (POOLSIZE semantics)
};

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct Pt { int x, y; double h; };

SMX_D f3 sub3(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
SMX_D f3 cross3(f3 x, f3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
SMX_D float dot3(f3 a, f3 b) { float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }
SMX_D float dot2(f2 a, f2 b) { float tx = a.x * b.x, ty = a.y * b.y; return tx + ty; }
SMX_D f3 normalize3(f3 v) { float s = 1.0f / sqrtf(dot3(v, v)); return {v.x * s, v.y * s, v.z * s}; }
SMX_D f2 normalize2(f2 v) { float s = 1.0f / sqrtf(dot2(v, v)); return {v.x * s, v.y * s}; }
SMX_D float mixd(float x, float y, double a) { return (float)((double)x * (1.0 - a) + (double)y * a); }

// std::sort on <= 16 elements == insertion sort, stable for the strict comparator (SURVEY.md Appx D#5)
SMX_D void sort_desc(Pt* sn, int num) {
  for (int i = 1; i < num; i++) {
    Pt v = sn[i];
    if (v.h > sn[0].h) {
      for (int j = i; j > 0; j--) sn[j] = sn[j - 1];
      sn[0] = v;
    } else {
      int j = i;
      while (v.h > sn[j - 1].h) { sn[j] = sn[j - 1]; j--; }
      sn[j] = v;
    }
  }
}

struct Water {            // WaterParticle (water.h:9-41)
  f2 pos, speed;
  double volume, sediment, evaprate;
  int spill, ix, iy;
  f3 n;
  SoilP param;
  uint32_t surface, contains;
};
struct Wind {             // WindParticle (wind.h:11-46)
  f2 pos;
  f3 speed;
  double sediment, height, sheight;
  int ix, iy;
  f3 n;
  uint32_t surface, contains;
  SoilP param;
};

// The exact engine: every method is the sequential reference semantics executed by ONE lane.
struct Sim {
  DevState s;
  bool grid_mode;          // inside the grid pass: newly wet cells must be added to the active bitmap

  SMX_D Sim(const DevState& st) : s(st), grid_mode(false) {}

  SMX_D size_t cidx(int x, int y) const { return (size_t)x * s.dimy + y; }   // layermap.h:151
  SMX_D size_t fidx(int x, int y) const { return (size_t)y * s.dimx + x; }   // water.h:53,349
  SMX_D void count(int which, unsigned long long v = 1) { s.ctr[which] += v; }

  // ---------------- pool: secpool::get / unget (layermap.h:89-111) ----------------
  // The reference takes a node for EVERY pool.get() (even when add() merges and returns it at once), so
  // get() fails exactly when live sections == capacity. We keep that rule with a live-section counter.
  SMX_D bool can_get() {
    if (s.ctr[C_LIVE_SECTIONS] >= s.pool_capacity) { count(C_POOL_OVERFLOW); return false; }
    return true;
  }
  SMX_D uint32_t node_alloc() {
    uint32_t n = *s.free_count;
    if (n == 0) { count(C_POOL_OVERFLOW); return NIL; }   // cannot happen while can_get() guards callers
    *s.free_count = n - 1;
    return s.freelist[n - 1];
  }
  SMX_D void node_free(uint32_t e) {
    uint32_t n = *s.free_count;
    s.freelist[n] = e;
    *s.free_count = n + 1;
  }

  // ---------------- queries ----------------
  SMX_D double height(int x, int y) const {                 // Layermap::height(ivec2) layermap.h:422-425
    const Sec& c = s.cells[cidx(x, y)];
    return c.floor + c.size;
  }
  SMX_D uint32_t surface(int x, int y) const {              // Layermap::surface layermap.h:417-420
    uint32_t t = s.cells[cidx(x, y)].type;
    return t == EMPTY ? 0u : t;
  }
  SMX_D double height_bilinear(f2 pos) const {              // Layermap::height(vec2) layermap.h:427-439
    double h = 0.0;
    float fx = floorf(pos.x), fy = floorf(pos.y);
    int px = (int)fx, py = (int)fy;
    float wx = pos.x - fx, wy = pos.y - fy;
    h += (1.0 - wx) * (1.0 - wy) * height(px, py);
    h += (1.0 - wx) * wy * height(px + 1, py);              // crossed weights as in the reference (:434-435)
    h += wx * (1.0 - wy) * height(px, py + 1);
    h += wx * wy * height(px + 1, py + 1);                  // wx*wy is a FLOAT product (:436)
    return h;
  }
  SMX_D f3 normal(int x, int y) const {                     // Layermap::normal(ivec2) layermap.h:341-377
    const int SC = s.scale;
    f3 n = {0.f, 0.f, 0.f};
    f3 p = {(float)x, (float)(SC * height(x, y)), (float)y};
    int k = 0;
    const bool xm = x > 0, xp = x < s.dimx - 1, ym = y > 0, yp = y < s.dimy - 1;
    f3 bm = {0, 0, 0}, bp = {0, 0, 0}, cm = {0, 0, 0}, cp = {0, 0, 0};
    if (xm) bm = {(float)(x - 1), (float)(SC * height(x - 1, y)), (float)y};
    if (xp) bp = {(float)(x + 1), (float)(SC * height(x + 1, y)), (float)y};
    if (ym) cm = {(float)x, (float)(SC * height(x, y - 1)), (float)(y - 1)};
    if (yp) cp = {(float)x, (float)(SC * height(x, y + 1)), (float)(y + 1)};
    if (xm && ym) { f3 v = cross3(sub3(cm, p), sub3(bm, p)); n.x += v.x; n.y += v.y; n.z += v.z; k++; }
    if (xm && yp) { f3 v = cross3(sub3(cp, p), sub3(bm, p)); n.x -= v.x; n.y -= v.y; n.z -= v.z; k++; }
    if (xp && ym) { f3 v = cross3(sub3(cm, p), sub3(bp, p)); n.x -= v.x; n.y -= v.y; n.z -= v.z; k++; }
    if (xp && yp) { f3 v = cross3(sub3(cp, p), sub3(bp, p)); n.x += v.x; n.y += v.y; n.z += v.z; k++; }
    float kf = (float)k;
    f3 q = {n.x / kf, n.y / kf, n.z / kf};
    return normalize3(q);
  }

  // ---------------- wet bookkeeping for the grid pass ----------------
  SMX_D void mark_active(size_t c) {
    s.active[c >> 6] |= (1ull << (c & 63));
    s.active1[c >> 12] |= (1ull << ((c >> 6) & 63));
    s.active2[c >> 18] |= (1ull << ((c >> 12) & 63));
  }
  // next set bit >= c in the 3-level bitmap, or ncells (summary levels are conservative)
  SMX_D size_t next_active(size_t c) const {
    const size_t n = (size_t)s.dimx * s.dimy;
    const size_t nw0 = (n + 63) >> 6, nw1 = (nw0 + 63) >> 6, nw2 = (nw1 + 63) >> 6;
    while (c < n) {
      const size_t w0 = c >> 6;
      unsigned long long bits = s.active[w0] & (~0ull << (c & 63));
      if (bits) { size_t r = (w0 << 6) + (size_t)__builtin_ctzll(bits); return r < n ? r : n; }
      const size_t w = w0 + 1;
      if (w >= nw0) return n;
      const size_t w1 = w >> 6;
      bits = s.active1[w1] & (~0ull << (w & 63));
      if (bits) { c = ((w1 << 6) + (size_t)__builtin_ctzll(bits)) << 6; continue; }
      size_t v = w1 + 1;
      if (v >= nw1) return n;
      size_t w2 = v >> 6;
      unsigned long long mask = ~0ull << (v & 63);
      bool found = false;
      for (; w2 < nw2; w2++, mask = ~0ull) {
        bits = s.active2[w2] & mask;
        if (bits) { found = true; break; }
      }
      if (!found) return n;
      c = ((w2 << 6) + (size_t)__builtin_ctzll(bits)) << 12;
    }
    return n;
  }
  SMX_D void on_air_top(int x, int y) {                     // column (x,y) now has a water top
    size_t c = cidx(x, y);
    s.flags[c] |= F_AIR;
    if (grid_mode)
      for (int dx = -1; dx <= 1; dx++)
        for (int dy = -1; dy <= 1; dy++) {
          int nx = x + dx, ny = y + dy;
          if (nx < 0 || ny < 0 || nx >= s.dimx || ny >= s.dimy) continue;
          mark_active(cidx(nx, ny));
        }
  }
  SMX_D void on_sat(int x, int y) {                         // a saturation in column (x,y) was written
    size_t c = cidx(x, y);
    s.flags[c] |= F_SAT;
    if (grid_mode) mark_active(c);
  }
  SMX_D void sync_air_flag(int x, int y) {                  // after any change of the top section's type
    size_t c = cidx(x, y);
    if (s.cells[c].type == AIR) on_air_top(x, y);
    else s.flags[c] &= (uint8_t)~F_AIR;
  }

  // ---------------- mutation primitives ----------------
  SMX_D void pop_top(Sec& cell) {                           // dat = E->prev; pool.unget(E)
    uint32_t pv = cell.prev;
    if (pv == NIL) {
      cell.size = 0.0; cell.floor = 0.0; cell.sat = 0.0; cell.type = EMPTY; cell.prev = NIL;
    } else {
      cell = s.pool[pv];
      node_free(pv);
    }
    s.ctr[C_LIVE_SECTIONS]--;
  }
  // add the section {size, type, sat} on top of column (x,y): Layermap::add (layermap.h:230-307).
  // `fresh` == the section came from pool.get() (counts against the pool); the water re-add of the swap
  // path re-uses the popped water section.
  SMX_D void add_sec(int x, int y, double size, uint32_t type, double sat, bool fresh) {
    if (fresh) {
      if (!can_get()) return;                               // pool.get() == NULL -> add(NULL) ignored (:233)
      s.ctr[C_LIVE_SECTIONS]++;                             // the node E now exists
    }
    if (size <= 0) { s.ctr[C_LIVE_SECTIONS]--; return; }    // :237-240 unget(E)
    Sec& cell = s.cells[cidx(x, y)];
    if (cell.type == EMPTY) {                               // :243-246
      cell.size = size; cell.floor = 0.0; cell.sat = sat; cell.type = type; cell.prev = NIL;
      sync_air_flag(x, y);
      if (sat != 0.0) on_sat(x, y);
      return;
    }
    if (cell.type == type) {                                // :249-253 elongate, unget(E)
      cell.size += size;
      s.ctr[C_LIVE_SECTIONS]--;
      return;
    }
    if (cell.type == AIR) {                                 // :258-275 water stays on top
      double wsize = cell.size, wsat = cell.sat;
      uint32_t pv = cell.prev;                              // dat = top->prev (water section kept aside)
      if (pv == NIL) { cell.size = 0.0; cell.floor = 0.0; cell.sat = 0.0; cell.type = EMPTY; cell.prev = NIL; }
      else { cell = s.pool[pv]; node_free(pv); }
      add_sec_nofresh(x, y, size, type, sat);               // add(pos, E)
      add_sec_nofresh(x, y, wsize, AIR, wsat);              // add(pos, top)
      return;
    }
    // :302-305 push on top: the old top becomes a buried pool node
    uint32_t nd = node_alloc();
    if (nd == NIL) { s.ctr[C_LIVE_SECTIONS]--; return; }
    double h = cell.floor + cell.size;
    s.pool[nd] = cell;
    cell.size = size; cell.floor = h; cell.sat = sat; cell.type = type; cell.prev = nd;
    sync_air_flag(x, y);
    if (sat != 0.0) on_sat(x, y);
  }
  // same as add_sec(..., fresh=false) but never recurses into the swap path twice (E != Air there)
  SMX_D void add_sec_nofresh(int x, int y, double size, uint32_t type, double sat) {
    if (size <= 0) { s.ctr[C_LIVE_SECTIONS]--; return; }
    Sec& cell = s.cells[cidx(x, y)];
    if (cell.type == EMPTY) {
      cell.size = size; cell.floor = 0.0; cell.sat = sat; cell.type = type; cell.prev = NIL;
      sync_air_flag(x, y);
      if (sat != 0.0) on_sat(x, y);
      return;
    }
    if (cell.type == type) { cell.size += size; s.ctr[C_LIVE_SECTIONS]--; return; }
    uint32_t nd = node_alloc();
    if (nd == NIL) { s.ctr[C_LIVE_SECTIONS]--; return; }
    double h = cell.floor + cell.size;
    s.pool[nd] = cell;
    cell.size = size; cell.floor = h; cell.sat = sat; cell.type = type; cell.prev = nd;
    sync_air_flag(x, y);
    if (sat != 0.0) on_sat(x, y);
  }
  SMX_D void add(int x, int y, double size, uint32_t type) { add_sec(x, y, size, type, 0.0, true); }

  // Layermap::remove (layermap.h:310-339); *popped (optional) tells the caller that the top section went away
  SMX_D double remove(int x, int y, double h, bool* popped = nullptr) {
    Sec& cell = s.cells[cidx(x, y)];
    if (popped) *popped = false;
    if (cell.type == EMPTY) return 0.0;
    if (cell.size <= 0.0) { pop_top(cell); sync_air_flag(x, y); if (popped) *popped = true; return 0.0; }
    if (h <= 0.0) return 0.0;
    double diff = h - cell.size;
    cell.size -= h;
    if (diff >= 0.0) { pop_top(cell); sync_air_flag(x, y); if (popped) *popped = true; return diff; }
    return 0.0;
  }

  // ---------------- glibc rand() ----------------
  SMX_D int rand_() {
    RandState& r = *s.rnd;
    uint32_t i = r.idx;
    uint32_t v = r.ring[i % 31] + r.ring[(i - 3) % 31];
    r.ring[i % 31] = v;
    r.idx = i + 1;
    r.calls++;
    return (int)(v >> 1);
  }

  // ---------------- Particle::cascade particle.h:24-101 ----------------
  SMX_DN void particle_cascade(f2 pos, int transferloop) {
    count(C_CASCADE_CALLS);
    const int ix = (int)roundf(pos.x), iy = (int)roundf(pos.y);
    Pt sn[8];
    int num = 0;
    for (int dx = -1; dx <= 1; dx++)
      for (int dy = -1; dy <= 1; dy++) {
        if (dx == 0 && dy == 0) continue;
        int nx = ix + dx, ny = iy + dy;
        if (nx >= s.dimx || ny >= s.dimy || nx < 0 || ny < 0) continue;
        sn[num].x = nx; sn[num].y = ny; sn[num].h = height(nx, ny); num++;
      }
    sort_desc(sn, num);
    const double sc = (double)(float)s.scale, d80 = (double)80.0f;
    for (int i = 0; i < num; i++) {
      const int nx = sn[i].x, ny = sn[i].y;
      float diff = (float)((height(ix, iy) - height(nx, ny)) * sc / d80);
      if (diff == 0) continue;
      const int tx = (diff > 0) ? ix : nx, ty = (diff > 0) ? iy : ny;
      const int bx = (diff > 0) ? nx : ix, by = (diff > 0) ? ny : iy;
      const uint32_t type = surface(tx, ty);
      const SoilP& param = s.soils[type];
      float excess = fabsf(diff) - param.maxdiff;
      if (excess <= 0) continue;
      float transfer = param.settling * excess / 2.0f;
      bool recascade = false;
      const double tsize = s.cells[cidx(tx, ty)].size;
      if ((double)transfer > tsize) transfer = (float)tsize;
      const uint32_t casc = param.cascades;
      if (remove(tx, ty, (double)transfer) != 0) recascade = true;
      add(bx, by, (double)transfer, casc);
      count(C_CASCADE_TRANSFERS);
      if (recascade && transferloop > 0) particle_cascade(f2{(float)nx, (float)ny}, --transferloop);
    }
  }

  // ---------------- WaterParticle water.h:9-368 ----------------
  SMX_D void water_ctor(Water& p) {                          // water.h:11-19 (g++: 1st draw -> y, 2nd -> x)
    int ry = rand_() % s.dimy;
    int rx = rand_() % s.dimx;
    p.pos = {(float)rx, (float)ry};
    p.speed = {0.f, 0.f};
    p.volume = 1.0; p.sediment = 0.0; p.evaprate = 0.001; p.spill = 3;
    p.ix = (int)roundf(p.pos.x); p.iy = (int)roundf(p.pos.y);
    p.n = {0.f, 0.f, 0.f};
    p.surface = surface(p.ix, p.iy);
    p.param = s.soils[p.surface];
    p.contains = p.param.transports;
  }
  SMX_D bool water_move(Water& p) {                          // water.h:43-73
    p.ix = (int)roundf(p.pos.x); p.iy = (int)roundf(p.pos.y);
    p.n = normal(p.ix, p.iy);
    p.surface = surface(p.ix, p.iy);
    p.param = s.soils[p.surface];
    p.evaprate = 0.01;
    const size_t fi = fidx(p.ix, p.iy);
    s.wtrack[fi] = (float)((double)s.wtrack[fi] + p.volume);            // updatefrequency :348-351
    const float fr = s.wfreq[fi];
    p.param.friction = p.param.friction * (1.0f - fr);
    p.evaprate = p.evaprate * (double)(1.0f - 0.2f * fr);
    const f2 nxz = {p.n.x, p.n.z};
    const f2 scaled = {nxz.x * p.param.friction, nxz.y * p.param.friction};
    if ((double)sqrtf(dot2(scaled, scaled)) < 1E-5) return false;
    const float a = p.param.friction;                                    // mix(nxz, speed, friction)
    p.speed = {nxz.x * (1.0f - a) + p.speed.x * a, nxz.y * (1.0f - a) + p.speed.y * a};
    const f2 nn = normalize2(p.speed);
    const float s2 = sqrtf(2.0f);
    p.speed = {s2 * nn.x, s2 * nn.y};
    p.pos.x += p.speed.x; p.pos.y += p.speed.y;
    if (!(p.pos.x >= 0.0f && p.pos.y >= 0.0f) ||
        !(p.pos.x < (float)s.dimx - 1.0f && p.pos.y < (float)s.dimy - 1.0f)) {
      p.volume = 0.0;
      return false;
    }
    return true;
  }
  SMX_D bool water_interact(Water& p) {                      // water.h:75-121
    double c_eq = p.param.solubility * (height(p.ix, p.iy) - height_bilinear(p.pos)) * (double)s.scale / 80.0;
    if (c_eq < 0.0) c_eq = 0.0;
    if (c_eq > 1.0) c_eq = 1.0;
    const size_t fi = fidx(p.ix, p.iy);
    if ((double)(s.soils[p.contains].erosionrate) < s.wfreq[fi]) p.contains = s.soils[p.contains].erodes;
    const double cdiff = c_eq - p.sediment;
    if (cdiff > 0) {
      p.sediment += p.param.equrate * cdiff;
      p.contains = s.soils[surface(p.ix, p.iy)].transports;
      double diff = remove(p.ix, p.iy, p.param.equrate * cdiff * p.volume);
      while (fabs(diff) > 1E-8) diff = remove(p.ix, p.iy, diff);
    } else if (cdiff < 0) {
      p.sediment += s.soils[p.contains].equrate * cdiff;
      add(p.ix, p.iy, -s.soils[p.contains].equrate * cdiff * p.volume, p.contains);
    }
    particle_cascade(p.pos, 0);
    p.sediment /= (1.0 - p.evaprate);
    if (p.sediment > 1.0) p.sediment = 1.0;
    p.volume *= (1.0 - p.evaprate);
    return (p.volume > 0.01);
  }
  SMX_DN bool water_flood(Water& p) {                        // water.h:123-145 (always returns false)
    if (p.volume < 0.01 || p.spill-- <= 0) return false;
    count(C_FLOODS);
    p.ix = (int)p.pos.x; p.iy = (int)p.pos.y;               // truncation (:128)
    add(p.ix, p.iy, p.sediment * s.soils[p.contains].equrate, p.contains);
    particle_cascade(p.pos, 0);
    add(p.ix, p.iy, p.volume * 0.015, AIR);                 // volumeFactor = 0.015 (:370)
    seep(p.ix, p.iy);
    water_cascade(p.ix, p.iy, p.spill);
    return false;
  }
  SMX_DN void water_run(Water& p, bool toplevel) {           // SoilMachine.cpp:292-296 / water.h:258-262
    while (true) {
      while (true) {
        if (!water_move(p)) break;
        count(C_STEPS_WATER_ALL);
        if (toplevel) count(C_STEPS_WATER_TOP);
        if (!water_interact(p)) break;
      }
      if (!water_flood(p)) break;
    }
  }
  SMX_DN void water_cascade(int ix, int iy, int spill) {     // WaterParticle::cascade water.h:151-283
    count(C_WCASCADE_CALLS);
    Pt sn[8];
    int num = 0;
    for (int dx = -1; dx <= 1; dx++)
      for (int dy = -1; dy <= 1; dy++) {
        if (dx == 0 && dy == 0) continue;
        int nx = ix + dx, ny = iy + dy;
        if (nx >= s.dimx || ny >= s.dimy || nx < 0 || ny < 0) continue;
        sn[num].x = nx; sn[num].y = ny; sn[num].h = height(nx, ny); num++;
      }
    sort_desc(sn, num);
    for (int i = 0; i < num; i++) {
      const int nx = sn[i].x, ny = sn[i].y;
      const Sec& A = s.cells[cidx(ix, iy)];
      const Sec& B = s.cells[cidx(nx, ny)];
      const double whA = A.size, whB = B.size, fA = A.floor, fB = B.floor;   // 0 for empty columns
      const double diff = (fA + whA - fB - whB) * (double)s.scale / 80.0;
      if (diff == 0) continue;
      const Sec& T = (diff > 0) ? A : B;
      const int tx = (diff > 0) ? ix : nx, ty = (diff > 0) ? iy : ny;
      const int bx = (diff > 0) ? nx : ix, by = (diff > 0) ? ny : iy;
      if (T.type != AIR) continue;                          // (EMPTY != AIR: the reference cannot get here with NULL)
      double transfer = fabs(diff) / 2.0;
      const double wh = T.size;
      transfer = (wh < transfer) ? wh : transfer;
      if (transfer <= 0) continue;
      bool recascade = false;
      if (transfer == wh) {
        remove(tx, ty, transfer);
        count(C_NESTED);
        Water q;
        water_ctor(q);
        const f2 d = {(float)bx - (float)tx, (float)by - (float)ty};
        const f2 dn = normalize2(d);
        const float s2 = sqrtf(2.0f);
        q.speed = {s2 * dn.x, s2 * dn.y};
        q.pos = {(float)tx, (float)ty};
        q.spill = spill;
        q.volume = transfer / 0.015;
        water_run(q, false);
      } else {
        if (remove(tx, ty, transfer) != 0) recascade = true;
        if (transfer > 0) recascade = true;
        add(bx, by, transfer, AIR);
        Sec& R = s.cells[cidx(bx, by)];
        if (R.type != EMPTY) { R.sat = 1.0f; on_sat(bx, by); }
      }
      if (recascade && spill > 0) water_cascade(nx, ny, --spill);
    }
  }
  SMX_DN void seep(int ix, int iy) {                         // WaterParticle::seep(vec2,..) water.h:285-333
    Sec& cell = s.cells[cidx(ix, iy)];
    if (cell.type == EMPTY) return;
    // walk top -> bottom; `t`/`pv` are either the inline cell record or pool nodes
    Sec* t = &cell;
    while (t != nullptr && t->prev != NIL) {
      uint32_t pvi = t->prev;
      Sec* pv = &s.pool[pvi];
      const SoilP& param = s.soils[t->type];
      const SoilP& nparam = s.soils[pv->type];
      const double vol = t->size * t->sat * param.porosity;
      const double nevol = pv->size * (1.0 - pv->sat) * nparam.porosity;
      const double seepage = 1.0;
      const double transfer = (vol < nevol) ? vol : nevol;
      if (transfer > 0) {
        if (t->type == AIR) {
          // map.remove(ipos, ..) acts on the COLUMN top; if it pops, the node below the top moves into
          // the inline cell record: redirect pv when that node is the one we hold.
          const uint32_t below_top = cell.prev;
          bool popped = false;
          remove(ix, iy, seepage * transfer, &popped);
          if (popped && below_top == pvi) pv = &cell;
        } else {
          t->sat -= (seepage * transfer) / (t->size * param.porosity);
        }
        pv->sat += (seepage * transfer) / (pv->size * nparam.porosity);
        on_sat(ix, iy);
      }
      t = pv;
    }
  }

  // ---------------- WindParticle wind.h:11-138 ----------------
  SMX_D void wind_ctor(Wind& p) {                            // wind.h:13-22
    int ry = rand_() % s.dimy;
    int rx = rand_() % s.dimx;
    p.pos = {(float)rx, (float)ry};
    p.speed = {-2.f, 0.f, 1.f};
    p.sediment = 0.0; p.height = 0.0; p.sheight = 0.0;
    p.ix = (int)roundf(p.pos.x); p.iy = (int)roundf(p.pos.y);
    p.n = {0.f, 0.f, 0.f};
    p.surface = surface(p.ix, p.iy);
    p.param = s.soils[p.surface];
    p.contains = p.param.transports;
  }
  SMX_D bool wind_move(Wind& p) {                            // wind.h:54-92
    const f3 pspeed = {-2.f, 0.f, 1.f};
    const double gravity = 0.25, winddominance = 0.2, windfriction = 0.8;
    if (s.soils[p.contains].suspension == 0.0) return false;
    p.ix = (int)roundf(p.pos.x); p.iy = (int)roundf(p.pos.y);
    p.n = normal(p.ix, p.iy);
    p.surface = surface(p.ix, p.iy);
    p.param = s.soils[p.surface];
    const size_t fi = fidx(p.ix, p.iy);
    s.windfreq[fi] = (float)(0.5 * s.windfreq[fi] + 0.5f);              // wind.h:49-52
    p.sheight = height(p.ix, p.iy) * (float)s.scale / 80.0f;
    if (p.height < p.sheight) p.height = p.sheight;
    if (p.height > p.sheight) {
      p.speed.y = (float)(p.speed.y - gravity);
    } else {
      const f3 c = cross3(cross3(p.speed, p.n), p.n);
      p.speed = {mixd(p.speed.x, c.x, windfriction), mixd(p.speed.y, c.y, windfriction), mixd(p.speed.z, c.z, windfriction)};
    }
    p.speed = {mixd(p.speed.x, pspeed.x, winddominance), mixd(p.speed.y, pspeed.y, winddominance),
               mixd(p.speed.z, pspeed.z, winddominance)};
    p.pos.x += p.speed.x; p.pos.y += p.speed.z;
    p.height += p.speed.y;
    if (!(p.pos.x >= 0.0f && p.pos.y >= 0.0f) || !((int)p.pos.x < s.dimx - 1 && (int)p.pos.y < s.dimy - 1)) return false;
    if (sqrtf(dot3(p.speed, p.speed)) < 0.01) return false;
    return true;
  }
  SMX_D bool wind_interact(Wind& p) {                        // wind.h:94-136
    const int nx = (int)roundf(p.pos.x), ny = (int)roundf(p.pos.y);
    if (p.height <= height_bilinear(p.pos) * (float)s.scale / 80.0f) {
      if (p.param.transports == p.contains) {
        const double force = sqrtf(dot3(p.speed, p.speed)) * (height(nx, ny) - p.height) * (float)s.scale / 80.0f * (1.0f - p.sediment);
        const double diff = remove(p.ix, p.iy, p.param.suspension * force);
        p.sediment += (p.param.suspension * force - diff);
        particle_cascade(f2{(float)p.ix, (float)p.iy}, 1);
      }
    } else if (p.param.suspension > 0.0) {
      p.sediment -= s.soils[p.contains].suspension * p.sediment;
      add(nx, ny, 0.5f * s.soils[p.contains].suspension * p.sediment, p.contains);
      add(p.ix, p.iy, 0.5f * s.soils[p.contains].suspension * p.sediment, p.contains);
      particle_cascade(f2{(float)p.ix, (float)p.iy}, 1);
      particle_cascade(f2{(float)nx, (float)ny}, 1);
    }
    return true;
  }
  SMX_D void wind_run(Wind& p) {
    while (true) {
      if (!wind_move(p)) break;
      count(C_STEPS_WIND);
      if (!wind_interact(p)) break;
    }
  }
};

}  // namespace smx
