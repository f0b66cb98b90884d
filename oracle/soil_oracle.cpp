// CPU ORACLE -- a restatement of the SoilMachine particle-transport hot path (TEST INFRASTRUCTURE ONLY).
//
// This file is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may load it. It is never on the product path (soilmachine_amd/ + include/soilmx.h).
//
// What it restates (every function cites the reference file:line it follows, /root/reference/...):
//   source/layermap.h   sec / secpool / Layermap::{height, normal, surface, add, remove, initialize}
//   source/surface.h    SurfParam tables, SurfLayer::get
//   source/particle/particle.h  Particle::cascade
//   source/particle/water.h     WaterParticle::{ctor, move, interact, flood, cascade, seep, *frequency}
//   source/particle/wind.h      WindParticle::{ctor, move, interact, updatefrequency}
//   SoilMachine.cpp:283-329     tick order
// Third-party pieces that are NOT under /root/reference and are restated from their published
// algorithms: glibc rand()/srand() (TYPE_3 additive feedback generator, glibc 2.35 stdlib/random_r.c),
// glm 0.9.9 vector semantics. FastNoiseLite v1.0.1 (vendored by the reference,
// source/include/FastNoiseLite.h) is restated for the OpenSimplex2 + FBm 3-D path only.
//
// Pinned against: oracle/_ref/soil_ref (the reference's own headers compiled by g++), bit-for-bit on
// full state snapshots -- tests/test_oracle_vs_ref.py and the committed fixtures in tests/golden/.
//
// Data layout differs on purpose from the reference (index-based SoA pool instead of pointer lists):
// node identity is never observable in the reference, so any allocator is parity-safe.
#include "soil_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr uint32_t NIL = 0xFFFFFFFFu;
constexpr uint32_t AIR = 0;   // soilmap["Air"] == 0 (surface.h:53-57)

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct i2 { int x, y; };

// ---- glm restatements ----
inline f3 sub(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline f3 cross(f3 x, f3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
inline float dot3(f3 a, f3 b) { float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }
inline float dot2(f2 a, f2 b) { float tx = a.x * b.x, ty = a.y * b.y; return tx + ty; }
inline f3 normalize3(f3 v) { float s = 1.0f / std::sqrt(dot3(v, v)); return {v.x * s, v.y * s, v.z * s}; }
inline f2 normalize2(f2 v) { float s = 1.0f / std::sqrt(dot2(v, v)); return {v.x * s, v.y * s}; }
inline float length2(f2 v) { return std::sqrt(dot2(v, v)); }
inline float length3(f3 v) { return std::sqrt(dot3(v, v)); }
inline float mixd(float x, float y, double a) { return (float)((double)x * (1.0 - a) + (double)y * a); }

struct Point { i2 pos; double h; };

// libstdc++ std::sort on <= 16 elements is a plain insertion sort (bits/stl_algo.h __insertion_sort):
// stable for the strict comparator a.h > b.h.
inline void sort_desc(Point* sn, int num) {
  for (int i = 1; i < num; i++) {
    Point v = sn[i];
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

const i2 NEIGH[8] = {{-1, -1}, {-1, 0}, {-1, 1}, {0, -1}, {0, 1}, {1, -1}, {1, 0}, {1, 1}};

// ---- FastNoiseLite v1.0.1, OpenSimplex2 3-D + FBm (FastNoiseLite.h:321-345,473-552,686-727,865-885,1053-1150) ----
struct Noise {
  int seed = 1337;          // never changed by the reference (FastNoiseLite.h:114-116)
  float frequency = 0.01f;
  int octaves = 3;
  float lacunarity = 2.0f, gain = 0.5f, weighted = 0.0f, bounding = 1 / 1.75f;
  float grad[256];

  Noise() {
    static const int base[12][3] = {{0, 1, 1}, {0, -1, 1}, {0, 1, -1}, {0, -1, -1}, {1, 0, 1}, {-1, 0, 1},
                                    {1, 0, -1}, {-1, 0, -1}, {1, 1, 0}, {-1, 1, 0}, {1, -1, 0}, {-1, -1, 0}};
    static const int tail[4][3] = {{1, 1, 0}, {0, -1, 1}, {-1, 1, 0}, {0, -1, -1}};
    for (int e = 0; e < 64; e++) {
      const int* g = e < 60 ? base[e % 12] : tail[e - 60];
      grad[4 * e] = (float)g[0]; grad[4 * e + 1] = (float)g[1]; grad[4 * e + 2] = (float)g[2]; grad[4 * e + 3] = 0.0f;
    }
  }
  void bound() {            // CalculateFractalBounding (FastNoiseLite.h:473-484)
    float g = gain < 0 ? -gain : gain;
    float amp = g, ampFractal = 1.0f;
    for (int i = 1; i < octaves; i++) { ampFractal += amp; amp *= g; }
    bounding = 1 / ampFractal;
  }
  static int fastround(float f) { return f >= 0 ? (int)(f + 0.5f) : (int)(f - 0.5f); }
  float gradcoord(int sd, int xp, int yp, int zp, float xd, float yd, float zd) const {
    uint32_t h = (uint32_t)sd ^ (uint32_t)xp ^ (uint32_t)yp ^ (uint32_t)zp;
    h *= 0x27d4eb2du;
    int hash = (int)h;
    hash ^= hash >> 15;
    hash &= 63 << 2;
    float xg = grad[hash], yg = grad[hash | 1], zg = grad[hash | 2];
    return xd * xg + yd * yg + zd * zg;
  }
  float single(int sd, float x, float y, float z) const {   // SingleOpenSimplex2 (FastNoiseLite.h:1053-1150)
    const int PX = 501125321, PY = 1136930381, PZ = 1720413743;
    int i = fastround(x), j = fastround(y), k = fastround(z);
    float x0 = (float)(x - i), y0 = (float)(y - j), z0 = (float)(z - k);
    int xs = (int)(-1.0f - x0) | 1, ys = (int)(-1.0f - y0) | 1, zs = (int)(-1.0f - z0) | 1;
    float ax0 = xs * -x0, ay0 = ys * -y0, az0 = zs * -z0;
    i = (int)((uint32_t)i * (uint32_t)PX); j = (int)((uint32_t)j * (uint32_t)PY); k = (int)((uint32_t)k * (uint32_t)PZ);
    float value = 0;
    float a = (0.6f - x0 * x0) - (y0 * y0 + z0 * z0);
    for (int l = 0;; l++) {
      if (a > 0) value += (a * a) * (a * a) * gradcoord(sd, i, j, k, x0, y0, z0);
      float b = a + 1;
      int i1 = i, j1 = j, k1 = k;
      float x1 = x0, y1 = y0, z1 = z0;
      if (ax0 >= ay0 && ax0 >= az0) {
        x1 += xs; b -= xs * 2 * x1; i1 = (int)((uint32_t)i1 - (uint32_t)(xs * PX));
      } else if (ay0 > ax0 && ay0 >= az0) {
        y1 += ys; b -= ys * 2 * y1; j1 = (int)((uint32_t)j1 - (uint32_t)(ys * PY));
      } else {
        z1 += zs; b -= zs * 2 * z1; k1 = (int)((uint32_t)k1 - (uint32_t)(zs * PZ));
      }
      if (b > 0) value += (b * b) * (b * b) * gradcoord(sd, i1, j1, k1, x1, y1, z1);
      if (l == 1) break;
      ax0 = 0.5f - ax0; ay0 = 0.5f - ay0; az0 = 0.5f - az0;
      x0 = xs * ax0; y0 = ys * ay0; z0 = zs * az0;
      a += (0.75f - ax0) - (ay0 + az0);
      i = (int)((uint32_t)i + (uint32_t)((xs >> 1) & PX));
      j = (int)((uint32_t)j + (uint32_t)((ys >> 1) & PY));
      k = (int)((uint32_t)k + (uint32_t)((zs >> 1) & PZ));
      xs = -xs; ys = -ys; zs = -zs;
      sd = ~sd;
    }
    return value * 32.69428253173828125f;
  }
  float get(float x, float y, float z) const {   // GetNoise 3-D (FastNoiseLite.h:321-345) with FractalType_FBm
    x *= frequency; y *= frequency; z *= frequency;
    const float R3 = (float)(2.0 / 3.0);          // TransformType3D_DefaultOpenSimplex2 (:712-720)
    float r = (x + y + z) * R3;
    x = r - x; y = r - y; z = r - z;
    int sd = seed;
    float sum = 0, amp = bounding;
    for (int o = 0; o < octaves; o++) {           // GenFractalFBm (:865-885)
      float n = single(sd++, x, y, z);
      sum += n * amp;
      amp *= 1.0f + weighted * (((n + 1) * 0.5f) - 1.0f);   // Lerp(1, (n+1)*0.5, weighted)
      x *= lacunarity; y *= lacunarity; z *= lacunarity;
      amp *= gain;
    }
    return sum;
  }
};

}  // namespace

struct so_ctx {
  int dimx, dimy, SCALE;
  std::vector<so_soil> soils;
  // column tops + section pool (SoA)
  std::vector<uint32_t> top;
  std::vector<double> size, floor, sat;
  std::vector<uint32_t> type, prev;
  std::vector<uint32_t> freelist;
  uint64_t capacity;
  std::vector<float> wfreq, wtrack, windfreq;
  // batched grid pass only: "this column has held a non-zero saturation" (sticky, like the device's F_SAT flag byte)
  std::vector<uint8_t> satflag;
  // glibc TYPE_3 rand state
  uint32_t ring[31];
  uint32_t ridx;
  so_counters ctr;
  int nest_depth = 0;

  // ---- optional dependency analysis (DESIGN.md "how much exact parallelism is there"): per cell, the
  // level (whole-particle DAG depth) and time (step-granular critical path) of the last writer/reader.
  bool dep_on = false;
  mutable std::vector<uint32_t> plog_r, plog_w, slog_r, slog_w;
  std::vector<uint32_t> wlev, rlev, wlev_mv, wtime, rtime;
  uint64_t dep_tasks = 0, dep_maxlev = 0, dep_maxlev_mv = 0, dep_free = 0, dep_crit = 0, dep_steps = 0, dep_sumlev = 0;
  uint32_t dep_tcur = 0;
  bool dep_rr = false;   // analysis variant: read-read sharing also counts as a conflict (single-stamp protocols)
  inline void tr(size_t c) const { if (b_guard) b_check(c); if (dep_on) (dep_rr ? slog_w : slog_r).push_back((uint32_t)c); }
  inline void tw(size_t c) const { if (b_guard) b_check(c); if (dep_on) slog_w.push_back((uint32_t)c); }
  void dep_reset() {
    if (!dep_on) return;
    size_t n = (size_t)dimx * dimy;
    wlev.assign(n, 0); rlev.assign(n, 0); wlev_mv.assign(n, 0); wtime.assign(n, 0); rtime.assign(n, 0);
    plog_r.clear(); plog_w.clear(); slog_r.clear(); slog_w.clear();
  }
  void dep_task_begin() { if (dep_on) dep_tcur = 0; }
  void dep_step_end() {          // one particle step (or one flood / grid-cell visit) finished
    if (!dep_on) return;
    uint32_t t = dep_tcur;
    for (uint32_t c : slog_r) if (wtime[c] > t) t = wtime[c];
    for (uint32_t c : slog_w) { if (wtime[c] > t) t = wtime[c]; if (rtime[c] > t) t = rtime[c]; }
    t += 1;
    for (uint32_t c : slog_r) if (rtime[c] < t) rtime[c] = t;
    for (uint32_t c : slog_w) wtime[c] = t;
    dep_tcur = t;
    if (t > dep_crit) dep_crit = t;
    dep_steps++;
    plog_r.insert(plog_r.end(), slog_r.begin(), slog_r.end());
    plog_w.insert(plog_w.end(), slog_w.begin(), slog_w.end());
    slog_r.clear(); slog_w.clear();
  }
  void dep_task_end() {          // one top-level particle (with everything it spawned) finished
    if (!dep_on) return;
    dep_step_end();
    uint32_t l = 0, lmv = 0;
    for (uint32_t c : plog_r) { if (wlev[c] > l) l = wlev[c]; if (wlev_mv[c] > lmv) lmv = wlev_mv[c]; }
    for (uint32_t c : plog_w) { if (wlev[c] > l) l = wlev[c]; if (rlev[c] > l) l = rlev[c]; if (wlev_mv[c] > lmv) lmv = wlev_mv[c]; }
    l += 1; lmv += 1;
    for (uint32_t c : plog_r) if (rlev[c] < l) rlev[c] = l;
    for (uint32_t c : plog_w) { if (wlev[c] < l) wlev[c] = l; if (wlev_mv[c] < lmv) wlev_mv[c] = lmv; }
    dep_tasks++; dep_sumlev += l;
    if (l == 1) dep_free++;
    if (l > dep_maxlev) dep_maxlev = l;
    if (lmv > dep_maxlev_mv) dep_maxlev_mv = lmv;
    plog_r.clear(); plog_w.clear();
  }

  // batched schedule: is cell c inside the tiles the acting particle holds? (counted, checked by the tests)
  void b_check(size_t c) const;

  // ---------------- pool (layermap.h:64-119) ----------------
  uint32_t get(double s, uint32_t t) {             // secpool::get + sec(double, SurfType) (:48-51,:89-103)
    if (freelist.empty()) { ctr.pool_overflow++; return NIL; }
    uint32_t e = freelist.back(); freelist.pop_back();
    size[e] = s; type[e] = t; floor[e] = 0.0; sat[e] = 0.0; prev[e] = NIL;
    return e;
  }
  void unget(uint32_t e) { if (e == NIL) return; freelist.push_back(e); }   // secpool::unget (:106-111)

  inline size_t cidx(int x, int y) const { return (size_t)x * dimy + y; }   // layermap.h:151
  inline size_t fidx(int x, int y) const { return (size_t)y * dimx + x; }   // water.h:53,349 (transposed)

  // ---------------- queries ----------------
  double height(int x, int y) const {              // Layermap::height(ivec2) layermap.h:422-425
    tr(cidx(x, y));
    uint32_t t = top[cidx(x, y)];
    if (t == NIL) return 0.0;
    return floor[t] + size[t];
  }
  uint32_t surface(int x, int y) const {           // Layermap::surface layermap.h:417-420
    tr(cidx(x, y));
    uint32_t t = top[cidx(x, y)];
    if (t == NIL) return 0;
    return type[t];
  }
  double height_bilinear(f2 pos) const {           // Layermap::height(vec2) layermap.h:427-439 (crossed weights kept)
    double h = 0.0f;
    float fx = std::floor(pos.x), fy = std::floor(pos.y);
    int px = (int)fx, py = (int)fy;
    float wx = pos.x - fx, wy = pos.y - fy;
    h += (1.0 - wx) * (1.0 - wy) * height(px, py);
    h += (1.0 - wx) * wy * height(px + 1, py);
    h += wx * (1.0 - wy) * height(px, py + 1);
    h += wx * wy * height(px + 1, py + 1);         // wx*wy is a FLOAT product (layermap.h:436)
    return h;
  }
  f3 normal(int x, int y) const {                  // Layermap::normal(ivec2) layermap.h:341-377
    f3 n = {0, 0, 0};
    f3 p = {(float)x, (float)(SCALE * height(x, y)), (float)y};
    int k = 0;
    if (x > 0 && y > 0) {
      f3 b = {(float)(x - 1), (float)(SCALE * height(x - 1, y)), (float)y};
      f3 c = {(float)x, (float)(SCALE * height(x, y - 1)), (float)(y - 1)};
      f3 v = cross(sub(c, p), sub(b, p)); n.x += v.x; n.y += v.y; n.z += v.z; k++;
    }
    if (x > 0 && y < dimy - 1) {
      f3 b = {(float)(x - 1), (float)(SCALE * height(x - 1, y)), (float)y};
      f3 c = {(float)x, (float)(SCALE * height(x, y + 1)), (float)(y + 1)};
      f3 v = cross(sub(c, p), sub(b, p)); n.x -= v.x; n.y -= v.y; n.z -= v.z; k++;
    }
    if (x < dimx - 1 && y > 0) {
      f3 b = {(float)(x + 1), (float)(SCALE * height(x + 1, y)), (float)y};
      f3 c = {(float)x, (float)(SCALE * height(x, y - 1)), (float)(y - 1)};
      f3 v = cross(sub(c, p), sub(b, p)); n.x -= v.x; n.y -= v.y; n.z -= v.z; k++;
    }
    if (x < dimx - 1 && y < dimy - 1) {
      f3 b = {(float)(x + 1), (float)(SCALE * height(x + 1, y)), (float)y};
      f3 c = {(float)x, (float)(SCALE * height(x, y + 1)), (float)(y + 1)};
      f3 v = cross(sub(c, p), sub(b, p)); n.x += v.x; n.y += v.y; n.z += v.z; k++;
    }
    float kf = (float)k;
    f3 q = {n.x / kf, n.y / kf, n.z / kf};
    return normalize3(q);
  }

  // ---------------- mutation primitives ----------------
  void add(int x, int y, uint32_t e) {             // Layermap::add layermap.h:230-307
    if (e == NIL) return;
    if (size[e] <= 0) { unget(e); return; }
    size_t c = cidx(x, y);
    tw(c);
    uint32_t t = top[c];
    if (t == NIL) { top[c] = e; return; }
    if (type[t] == type[e]) { size[t] += size[e]; unget(e); return; }
    if (type[t] == AIR) {                          // water swap :258-275
      top[c] = prev[t];
      add(x, y, e);
      add(x, y, t);
      return;
    }
    prev[e] = t;
    floor[e] = height(x, y);
    top[c] = e;
  }
  double remove(int x, int y, double h) {          // Layermap::remove layermap.h:310-339
    size_t c = cidx(x, y);
    tw(c);
    uint32_t t = top[c];
    if (t == NIL) return 0.0;
    if (size[t] <= 0.0) { top[c] = prev[t]; unget(t); return 0.0; }
    if (h <= 0.0) return 0.0;
    double diff = h - size[t];
    size[t] -= h;
    if (diff >= 0.0) { top[c] = prev[t]; unget(t); return diff; }
    return 0.0;
  }

  // ---------------- glibc rand (stdlib/random_r.c, TYPE_3: r[i] = r[i-31] + r[i-3]) ----------------
  uint32_t rnext() {
    uint32_t v = ring[ridx % 31] + ring[(ridx - 3) % 31];
    ring[ridx % 31] = v;
    ridx++;
    return v;
  }
  void srand_(unsigned seed) {
    if (seed == 0) seed = 1;
    int32_t word = (int32_t)seed;
    ring[0] = (uint32_t)word;
    for (int i = 1; i < 31; i++) {
      long hi = word / 127773, lo = word % 127773;
      word = (int32_t)(16807 * lo - 2836 * hi);
      if (word < 0) word += 2147483647;
      ring[i] = (uint32_t)word;
    }
    ridx = 34;
    for (int i = 0; i < 310; i++) rnext();
    ctr.rand_calls = 0;
  }
  int rand_() { ctr.rand_calls++; return (int)(rnext() >> 1); }

  // ---------------- Particle::cascade particle.h:24-101 ----------------
  void particle_cascade(f2 pos, int transferloop) {
    ctr.cascade_calls++;
    int ix = (int)std::round(pos.x), iy = (int)std::round(pos.y);
    Point sn[8];
    int num = 0;
    for (const i2& nn : NEIGH) {
      int nx = ix + nn.x, ny = iy + nn.y;
      if (nx >= dimx || ny >= dimy || nx < 0 || ny < 0) continue;
      sn[num++] = {{nx, ny}, height(nx, ny)};
    }
    sort_desc(sn, num);
    for (int i = 0; i < num; i++) {
      i2 npos = sn[i].pos;
      float diff = (float)((height(ix, iy) - height(npos.x, npos.y)) * (double)(float)SCALE / (double)80.0f);
      if (diff == 0) continue;
      i2 tpos = (diff > 0) ? i2{ix, iy} : npos;
      i2 bpos = (diff > 0) ? npos : i2{ix, iy};
      uint32_t ty = surface(tpos.x, tpos.y);
      const so_soil& param = soils[ty];
      float excess = std::fabs(diff) - param.maxdiff;
      if (excess <= 0) continue;
      float transfer = param.settling * excess / 2.0f;
      bool recascade = false;
      uint32_t tt = top[cidx(tpos.x, tpos.y)];
      if ((double)transfer > size[tt]) transfer = (float)size[tt];
      if (remove(tpos.x, tpos.y, (double)transfer) != 0) recascade = true;
      add(bpos.x, bpos.y, get((double)transfer, param.cascades));
      ctr.cascade_transfers++;
      if (recascade && transferloop > 0) particle_cascade(f2{(float)npos.x, (float)npos.y}, --transferloop);
    }
  }

  // ---------------- WaterParticle water.h:9-368 ----------------
  struct Water {
    f2 pos, speed = {0, 0};
    double volume = 1.0, sediment = 0.0, evaprate = 0.001;
    int spill = 3;
    int ix, iy;
    f3 n;
    so_soil param;
    uint32_t surface, contains;
  };
  static constexpr double minvol = 0.01;
  static constexpr double volumeFactor = 0.015;   // water.h:370

  void water_ctor(Water& p) {                      // water.h:11-19; g++ evaluates ctor args right-to-left:
    int ry = rand_() % dimy;                       //   1st draw -> y
    int rx = rand_() % dimx;                       //   2nd draw -> x   
    p.pos = {(float)rx, (float)ry};
    p.ix = (int)std::round(p.pos.x); p.iy = (int)std::round(p.pos.y);
    p.surface = surface(p.ix, p.iy);
    p.param = soils[p.surface];
    p.contains = p.param.transports;
  }
  bool water_move(Water& p, bool pure = false) {   // water.h:43-73 (pure: the batched schedule's look-ahead, no track write)
    p.ix = (int)std::round(p.pos.x); p.iy = (int)std::round(p.pos.y);
    p.n = normal(p.ix, p.iy);
    p.surface = surface(p.ix, p.iy);
    p.param = soils[p.surface];
    p.evaprate = 0.01;
    size_t fi = fidx(p.ix, p.iy);
    tw(cidx(p.ix, p.iy));
    if (!pure) wtrack[fi] = (float)((double)wtrack[fi] + p.volume);      // updatefrequency water.h:348-351
    float fr = wfreq[fi];
    p.param.friction = p.param.friction * (1.0f - fr);
    p.evaprate = p.evaprate * (double)(1.0f - 0.2f * fr);
    f2 nxz = {p.n.x, p.n.z};
    f2 scaled = {nxz.x * p.param.friction, nxz.y * p.param.friction};
    if ((double)length2(scaled) < 1E-5) return false;
    float a = p.param.friction;                                           // mix(nxz, speed, friction)
    p.speed = {nxz.x * (1.0f - a) + p.speed.x * a, nxz.y * (1.0f - a) + p.speed.y * a};
    f2 nn = normalize2(p.speed);
    float s2 = std::sqrt(2.0f);
    p.speed = {s2 * nn.x, s2 * nn.y};
    p.pos.x += p.speed.x; p.pos.y += p.speed.y;
    if (!(p.pos.x >= 0.0f && p.pos.y >= 0.0f) ||
        !(p.pos.x < (float)dimx - 1.0f && p.pos.y < (float)dimy - 1.0f)) {
      p.volume = 0.0;
      return false;
    }
    return true;
  }
  bool water_interact(Water& p) {                  // water.h:75-121
    double c_eq = p.param.solubility * (height(p.ix, p.iy) - height_bilinear(p.pos)) * (double)SCALE / 80.0;
    if (c_eq < 0.0) c_eq = 0.0;
    if (c_eq > 1.0) c_eq = 1.0;
    size_t fi = fidx(p.ix, p.iy);
    if ((double)(soils[p.contains].erosionrate) < wfreq[fi]) p.contains = soils[p.contains].erodes;
    double cdiff = c_eq - p.sediment;
    if (cdiff > 0) {
      p.sediment += p.param.equrate * cdiff;
      p.contains = soils[surface(p.ix, p.iy)].transports;
      double diff = remove(p.ix, p.iy, p.param.equrate * cdiff * p.volume);
      while (std::fabs(diff) > 1E-8) diff = remove(p.ix, p.iy, diff);
    } else if (cdiff < 0) {
      p.sediment += soils[p.contains].equrate * cdiff;
      add(p.ix, p.iy, get(-soils[p.contains].equrate * cdiff * p.volume, p.contains));
    }
    particle_cascade(p.pos, 0);
    p.sediment /= (1.0 - p.evaprate);
    if (p.sediment > 1.0) p.sediment = 1.0;
    p.volume *= (1.0 - p.evaprate);
    return (p.volume > minvol);
  }
  bool water_flood(Water& p) {                     // water.h:123-145 (always returns false)
    if (p.volume < minvol || p.spill-- <= 0) return false;
    ctr.floods++;
    p.ix = (int)p.pos.x; p.iy = (int)p.pos.y;      // truncation (water.h:128)
    add(p.ix, p.iy, get(p.sediment * soils[p.contains].equrate, p.contains));
    particle_cascade(p.pos, 0);
    add(p.ix, p.iy, get(p.volume * volumeFactor, AIR));
    seep(p.ix, p.iy);
    water_cascade(p.ix, p.iy, p.spill);
    return false;
  }
  void water_run(Water& p, bool toplevel) {        // SoilMachine.cpp:292-296 / water.h:258-262
    while (true) {
      while (true) {
        if (!water_move(p)) break;
        ctr.steps_water_all++;
        if (toplevel) ctr.steps_water_top++;
        bool alive = water_interact(p);
        dep_step_end();
        if (!alive) break;
      }
      if (!water_flood(p)) break;
    }
  }
  void water_cascade(int ix, int iy, int spill) {  // WaterParticle::cascade water.h:151-283
    ctr.wcascade_calls++;
    Point sn[8];
    int num = 0;
    for (const i2& nn : NEIGH) {
      int nx = ix + nn.x, ny = iy + nn.y;
      if (nx >= dimx || ny >= dimy || nx < 0 || ny < 0) continue;
      sn[num++] = {{nx, ny}, height(nx, ny)};
    }
    sort_desc(sn, num);
    for (int i = 0; i < num; i++) {
      i2 npos = sn[i].pos;
      tr(cidx(ix, iy)); tr(cidx(npos.x, npos.y));
      uint32_t secA = top[cidx(ix, iy)], secB = top[cidx(npos.x, npos.y)];
      double whA = 0, whB = 0;
      if (secA != NIL) whA = size[secA];
      if (secB != NIL) whB = size[secB];
      double fA = 0.0, fB = 0.0;
      if (secA != NIL) fA = floor[secA];
      if (secB != NIL) fB = floor[secB];
      double diff = (fA + whA - fB - whB) * (double)SCALE / 80.0;
      if (diff == 0) continue;
      uint32_t tsec = (diff > 0) ? secA : secB;
      i2 tpos = (diff > 0) ? i2{ix, iy} : npos;
      i2 bpos = (diff > 0) ? npos : i2{ix, iy};
      if (tsec == NIL) continue;                   // reference would dereference NULL; unreachable (heights >= 0)
      if (type[tsec] != AIR) continue;
      double transfer = std::fabs(diff) / 2.0;
      double wh = size[tsec];
      transfer = (wh < transfer) ? wh : transfer;
      if (transfer <= 0) continue;
      bool recascade = false;
      if (transfer == wh) {
        if (b_defer) {                             // throughput schedules: the nested particle runs at once INSIDE the region its parent owns
          if (b_seq >= B_MAXCHILD) continue;       // no record left, should it have to be suspended: the water stays
          remove(tpos.x, tpos.y, transfer);
          ctr.nested_particles++;
          Water q;
          q.pos = {(float)tpos.x, (float)tpos.y};
          q.ix = tpos.x; q.iy = tpos.y;
          q.surface = surface(q.ix, q.iy); q.param = soils[q.surface];
          q.contains = q.param.transports;         // (from the cell it starts on, not a random one: soil_core.h water_drive says why that is unobservable)
          f2 d = {(float)bpos.x - (float)tpos.x, (float)bpos.y - (float)tpos.y};
          f2 dn = normalize2(d);
          float s2 = std::sqrt(2.0f);
          q.speed = {s2 * dn.x, s2 * dn.y};
          q.spill = spill;
          q.volume = transfer / volumeFactor;
          b_nested(q);
          continue;
        }
        remove(tpos.x, tpos.y, transfer);
        ctr.nested_particles++;
        Water q;
        water_ctor(q);
        f2 d = {(float)bpos.x - (float)tpos.x, (float)bpos.y - (float)tpos.y};
        f2 dn = normalize2(d);
        float s2 = std::sqrt(2.0f);
        q.speed = {s2 * dn.x, s2 * dn.y};
        q.pos = {(float)tpos.x, (float)tpos.y};
        q.spill = spill;
        q.volume = transfer / volumeFactor;
        water_run(q, false);
      } else {
        if (remove(tpos.x, tpos.y, transfer) != 0) recascade = true;
        if (transfer > 0) recascade = true;
        add(bpos.x, bpos.y, get(transfer, AIR));
        uint32_t bt = top[cidx(bpos.x, bpos.y)];
        if (bt != NIL) { sat[bt] = 1.0f; satflag[cidx(bpos.x, bpos.y)] = 1; }
        tw(cidx(bpos.x, bpos.y));
      }
      if (recascade && spill > 0) water_cascade(npos.x, npos.y, --spill);
    }
  }
  void seep(int ix, int iy) {                      // WaterParticle::seep(vec2,...) water.h:285-333
    uint32_t t = top[cidx(ix, iy)];
    tr(cidx(ix, iy));
    if (t == NIL) return;
    while (t != NIL && prev[t] != NIL) {
      uint32_t pv = prev[t];
      const so_soil& param = soils[type[t]];
      const so_soil& nparam = soils[type[pv]];
      double vol = size[t] * sat[t] * param.porosity;
      double nevol = size[pv] * (1.0 - sat[pv]) * nparam.porosity;
      double seepage = 1.0;
      double transfer = (vol < nevol) ? vol : nevol;
      if (transfer > 0) {
        tw(cidx(ix, iy));
        if (type[t] == AIR) remove(ix, iy, seepage * transfer);
        else sat[t] -= (seepage * transfer) / (size[t] * param.porosity);
        sat[pv] += (seepage * transfer) / (size[pv] * nparam.porosity);
        satflag[cidx(ix, iy)] = 1;
      }
      t = pv;
    }
  }
  void grid_pass() {                               // WaterParticle::seep(Layermap&,...) water.h:335-343
    for (int x = 0; x < dimx; x++)
      for (int y = 0; y < dimy; y++) {
        seep(x, y);
        water_cascade(x, y, 3);
      }
  }
  void map_frequency() {                           // water.h:358-365
    const float lrate = 0.01f, K = 50.0f;
    size_t n = (size_t)dimx * dimy;
    for (size_t i = 0; i < n; i++)
      wfreq[i] = (1.0f - lrate) * wfreq[i] + lrate * K * wtrack[i] / (1.0f + K * wtrack[i]);
  }
  void reset_frequency() {                         // water.h:353-356
    size_t n = (size_t)dimx * dimy;
    for (size_t i = 0; i < n; i++) wtrack[i] = 0.0f;
  }

  // ---------------- WindParticle wind.h:11-138 ----------------
  struct Wind {
    f2 pos;
    f3 speed = {-2, 0, 1};
    double sediment = 0.0, height = 0.0, sheight = 0.0;
    int ix, iy;
    f3 n;
    uint32_t surface, contains;
    so_soil param;
  };
  void wind_ctor(Wind& p) {                        // wind.h:13-22
    int ry = rand_() % dimy;
    int rx = rand_() % dimx;
    p.pos = {(float)rx, (float)ry};
    p.ix = (int)std::round(p.pos.x); p.iy = (int)std::round(p.pos.y);
    p.surface = surface(p.ix, p.iy);
    p.param = soils[p.surface];
    p.contains = p.param.transports;
  }
  bool wind_move(Wind& p, bool pure = false) {     // wind.h:54-92
    const f3 pspeed = {-2, 0, 1};
    const double gravity = 0.25, winddominance = 0.2, windfriction = 0.8;
    if (!pure && soils[p.contains].suspension == 0.0) return false;   // (the look-ahead of a particle that has not been constructed yet has no `contains`)
    p.ix = (int)std::round(p.pos.x); p.iy = (int)std::round(p.pos.y);
    p.n = normal(p.ix, p.iy);
    p.surface = surface(p.ix, p.iy);
    p.param = soils[p.surface];
    size_t fi = fidx(p.ix, p.iy);
    tw(cidx(p.ix, p.iy));
    if (!pure) windfreq[fi] = (float)(0.5 * windfreq[fi] + 0.5f);        // wind.h:49-52
    p.sheight = height(p.ix, p.iy) * (float)SCALE / 80.0f;
    if (p.height < p.sheight) p.height = p.sheight;
    if (p.height > p.sheight) {
      p.speed.y = (float)(p.speed.y - gravity);
    } else {
      f3 c = cross(cross(p.speed, p.n), p.n);
      p.speed = {mixd(p.speed.x, c.x, windfriction), mixd(p.speed.y, c.y, windfriction), mixd(p.speed.z, c.z, windfriction)};
    }
    p.speed = {mixd(p.speed.x, pspeed.x, winddominance), mixd(p.speed.y, pspeed.y, winddominance),
               mixd(p.speed.z, pspeed.z, winddominance)};
    p.pos.x += p.speed.x; p.pos.y += p.speed.z;
    p.height += p.speed.y;
    if (!(p.pos.x >= 0.0f && p.pos.y >= 0.0f) || !((int)p.pos.x < dimx - 1 && (int)p.pos.y < dimy - 1)) return false;
    if (length3(p.speed) < 0.01) return false;
    return true;
  }
  bool wind_interact(Wind& p) {                    // wind.h:94-136
    int nx = (int)std::round(p.pos.x), ny = (int)std::round(p.pos.y);
    if (p.height <= height_bilinear(p.pos) * (float)SCALE / 80.0f) {
      if (p.param.transports == p.contains) {
        double force = length3(p.speed) * (height(nx, ny) - p.height) * (float)SCALE / 80.0f * (1.0f - p.sediment);
        double diff = remove(p.ix, p.iy, p.param.suspension * force);
        p.sediment += (p.param.suspension * force - diff);
        particle_cascade(f2{(float)p.ix, (float)p.iy}, 1);
      }
    } else if (p.param.suspension > 0.0) {
      p.sediment -= soils[p.contains].suspension * p.sediment;
      add(nx, ny, get(0.5f * soils[p.contains].suspension * p.sediment, p.contains));
      add(p.ix, p.iy, get(0.5f * soils[p.contains].suspension * p.sediment, p.contains));
      particle_cascade(f2{(float)p.ix, (float)p.iy}, 1);
      particle_cascade(f2{(float)nx, (float)ny}, 1);
    }
    return true;
  }


  // =====================================================================================================
  // BATCHED SCHEDULE -- CPU restatement of soilmachine_amd/csrc/soil_batch.h (the throughput engine).
  // Same per-step functions as above (= the reference's move/interact/flood/cascade/seep); what is restated
  // here is the engine's own definition of WHO acts WHEN:
  //   * 4x4-cell claim tiles; a particle reserves the bounding tile rectangle of what its next action can
  //     touch (+ a margin of `dilate` tiles); the lowest slot wins a tile; a particle acts in an epoch iff it
  //     holds every tile of its reservation, and may go on (<= maxsteps actions) while the next action stays
  //     inside the tiles it holds; (dilate, maxsteps) follow from the number of running particles at the start
  //     of every 32-epoch chunk (b_schedule);
  //   * nested particles run at once inside the tiles their parent owns (b_nested); one that leaves them is suspended as it stands: the
  //     records sorted by (parent slot << 8 | sequence) form the next generation (no rand() draws: a record carries its state);
  //   * grid pass: 8x8-cell tiles colour by colour (tx mod 3, ty mod 3), cells of a tile x-major, active set fixed at
  //     the start of the pass.
  // Acting order inside an epoch is irrelevant by construction (winners own disjoint tiles); this
  // restatement walks slots in ascending order.
  // =====================================================================================================
  static constexpr int B_GRID = 3, B_MAXCHILD = 250, B_CHUNK = 32;   // grid pass: 8x8-cell tiles
  int B_TILE = 2;                                   // claim granularity of the running chunk (b_schedule): tiles of 2^B_TILE cells
  struct BRect { int tx0, tx1, ty0, ty1; bool empty() const { return tx1 < tx0; } };
  struct BChild { uint64_t key; f2 pos, speed; double volume, sediment; int spill; uint32_t contains; bool ended; };   // a suspended nested particle (soil_batch.h BChild)
  bool b_defer = false;
  uint64_t b_parent = 0;
  int b_seq = 0, b_base_dilate = 0;
  std::vector<BChild> b_children;
  uint64_t b_epochs = 0, b_ngen = 0;
  const char* b_ctx = "";
  const BRect* b_guard = nullptr;                 // while set: every cell the step functions touch must lie in these tiles
  uint64_t b_guard_violations = 0;

  int b_ntx() const { return (dimx + (1 << B_TILE) - 1) >> B_TILE; }
  int b_nty() const { return (dimy + (1 << B_TILE) - 1) >> B_TILE; }
  BRect b_cells(int x0, int x1, int y0, int y1) const {
    x0 = std::max(x0, 0); y0 = std::max(y0, 0); x1 = std::min(x1, dimx - 1); y1 = std::min(y1, dimy - 1);
    return {x0 >> B_TILE, x1 >> B_TILE, y0 >> B_TILE, y1 >> B_TILE};
  }
  BRect b_dilate(BRect r, int d) const {
    if (r.empty() || d <= 0) return r;
    BRect o;
    o.tx0 = r.tx0 - d < 0 ? 0 : r.tx0 - d; o.tx1 = r.tx1 + d > b_ntx() - 1 ? b_ntx() - 1 : r.tx1 + d;
    o.ty0 = r.ty0 - d < 0 ? 0 : r.ty0 - d; o.ty1 = r.ty1 + d > b_nty() - 1 ? b_nty() - 1 : r.ty1 + d;
    return o;
  }
  static bool b_inside(const BRect& a, const BRect& b) {   // b inside a
    return !a.empty() && b.tx0 >= a.tx0 && b.tx1 <= a.tx1 && b.ty0 >= a.ty0 && b.ty1 <= a.ty1;
  }
  uint32_t b_cell_above = 0xFFFFFFFFu;            // water reserves single cells while more particles than this run (tests lower it)
  void b_schedule(bool wind, uint32_t nlive, int base, int& tshift, int& dilate, int& maxsteps) const {
    if (wind) {
      tshift = 2; dilate = 1; maxsteps = 32;                  // (round 4: one schedule at every particle count, soil_batch.h batch_schedule)
    } else {
      tshift = 2; dilate = 0; maxsteps = 4;
      if (nlive > b_cell_above) { tshift = 0; maxsteps = 1; }
    }
    if (base > dilate) dilate = base;
  }
  // the footprint of the next action, in tiles (soil_batch.h *_intent)
  BRect b_water_need_run(const Water& p) {
    Water t = p;
    const int ix = (int)std::round(p.pos.x), iy = (int)std::round(p.pos.y);
    if (!water_move(t, true)) return b_cells(ix - 1, ix + 1, iy - 1, iy + 1);   // the decision read the plus-stencil
    const int nx = (int)std::round(t.pos.x), ny = (int)std::round(t.pos.y);
    return b_cells(std::min(ix, nx) - 1, std::max(ix, nx) + 1, std::min(iy, ny) - 1, std::max(iy, ny) + 1);
  }
  static bool b_water_floods(const Water& p) { return !(p.volume < minvol || p.spill <= 0); }
  BRect b_water_need_flood(const Water& p) const {
    const int R = p.spill > 2 ? p.spill : 2, fx = (int)p.pos.x, fy = (int)p.pos.y;
    return b_cells(fx - R, fx + R, fy - R, fy + R);
  }
  BRect b_wind_need_run(const Wind& p) {
    Wind t = p;
    const int ix = (int)std::round(p.pos.x), iy = (int)std::round(p.pos.y);
    if (!wind_move(t, true)) return b_cells(ix - 1, ix + 1, iy - 1, iy + 1);
    const int nx = (int)std::round(t.pos.x), ny = (int)std::round(t.pos.y);
    return b_cells(std::min(ix, nx) - 2, std::max(ix, nx) + 2, std::min(iy, ny) - 2, std::max(iy, ny) + 2);
  }

  // A nested particle (water.h:246-264) inside the action of its parent -- a flood behind its claim tiles, a grid tile with the ring of tiles
  // around it: `*b_guard`, the tiles the parent owns. while(move && interact) for as long as the next step's footprint (b_water_need_run) lies
  // inside them, then flood() if its reach does; whatever does not fit is suspended AS IT STANDS into a record of the next generation.
  void b_nested(Water& q) {
    const BRect region = *b_guard;
    bool suspended = false;
    while (true) {
      const int ix = (int)std::round(q.pos.x), iy = (int)std::round(q.pos.y);
      if (!b_inside(region, b_cells(ix - 1, ix + 1, iy - 1, iy + 1)) || !b_inside(region, b_water_need_run(q))) { suspended = true; break; }
      if (!water_move(q)) break;
      ctr.steps_water_all++;
      if (!water_interact(q)) break;
    }
    bool ended = false;
    if (!suspended) {
      if (!b_water_floods(q)) return;                         // water.h:125
      if (b_inside(region, b_water_need_flood(q))) { water_flood(q); return; }   // (b_defer stays set: its own nested particles run the same way)
      ended = true;
    }
    b_children.push_back({(b_parent << 8) | (uint64_t)b_seq, q.pos, q.speed, q.volume, q.sediment, q.spill, q.contains, ended});
    b_seq++;
  }

  enum { B_RUN = 0, B_ENDED = 1, B_DONE = 2, B_NEW = 3 };   // B_NEW: a top-level particle that has not acted yet
  struct BSlot { Water w; Wind d; int state; BRect rect; bool toplevel; bool claiming; };
  // column strips (soil_batch.h BStrips): INTERIOR and SEAM chunks alternate; only eligible particles reserve and act
  int bs_n = 1, bs_a = 16, bs_b = 48, bs_phase = 0;
  int bs_x(int g) const { return g >= bs_n ? dimx : (int)(((long long)dimx * g / bs_n) / 16 * 16); }
  bool b_eligible(const BRect& r) const {
    if (bs_n <= 1) return true;
    if (r.empty()) return false;
    return b_eligible_x(r.tx0 << B_TILE, std::min(((r.tx1 + 1) << B_TILE) - 1, dimx - 1));
  }
  bool b_eligible_x(int cx0, int cx1) const {               // may an action that touches the columns cx0 .. cx1 run in this kind of chunk?
    if (bs_n <= 1) return true;
    cx0 = std::max(cx0, 0); cx1 = std::min(cx1, dimx - 1);
    if (bs_phase == 0) {
      for (int g = 0; g < bs_n; g++) { const int lo = bs_x(g) + (g ? bs_a : 0), hi = bs_x(g + 1) - (g < bs_n - 1 ? bs_a : 0); if (cx0 >= lo && cx1 < hi) return true; }
    } else {
      for (int g = 0; g + 1 < bs_n; g++) if (cx0 >= bs_x(g + 1) - bs_b && cx1 < bs_x(g + 1) + bs_b) return true;
      for (int g = 0; g < bs_n; g++) {                        // deep inside a strip: out of reach of every seam zone -> goes on in SEAM chunks too
        const int lo = bs_x(g) + (g ? bs_b : 0), hi = bs_x(g + 1) - (g < bs_n - 1 ? bs_b : 0);
        if (cx0 >= lo && cx1 < hi) return true;
      }
    }
    return false;
  }

  // winners of an epoch: slot i holds its reservation iff it is the lowest claimant of every tile in it
  void b_winners(const std::vector<BSlot>& sl, std::vector<uint32_t>& tilemin, std::vector<char>& win) {
    const int nty = b_nty();
    std::vector<size_t> touched;
    for (uint32_t i = 0; i < sl.size(); i++) {
      if (sl[i].state == B_DONE || !sl[i].claiming) continue;
      const BRect& r = sl[i].rect;
      for (int tx = r.tx0; tx <= r.tx1; tx++)
        for (int ty = r.ty0; ty <= r.ty1; ty++) {
          const size_t t = (size_t)tx * nty + ty;
          if (tilemin[t] == 0xFFFFFFFFu) touched.push_back(t);
          if (i < tilemin[t]) tilemin[t] = i;
        }
    }
    for (uint32_t i = 0; i < sl.size(); i++) {
      win[i] = 0;
      if (sl[i].state == B_DONE || !sl[i].claiming) continue;
      const BRect& r = sl[i].rect;
      bool ok = true;
      for (int tx = r.tx0; tx <= r.tx1 && ok; tx++)
        for (int ty = r.ty0; ty <= r.ty1; ty++) if (tilemin[(size_t)tx * nty + ty] != i) { ok = false; break; }
      win[i] = ok ? 1 : 0;
    }
    for (size_t t : touched) tilemin[t] = 0xFFFFFFFFu;
  }

  void b_generations(bool wind, uint32_t n, std::vector<BChild> kids) {
    bool first = kids.empty();
    uint32_t nslots = first ? n : (uint32_t)kids.size();
    std::vector<uint32_t> tilemin;
    while (nslots > 0) {
      std::vector<BSlot> sl(nslots);
      std::vector<int> draws(first ? 2 * (size_t)nslots : 0);
      for (auto& v : draws) v = rand_();                    // a top-level generation's draws, in slot order: 2i -> y, 2i+1 -> x (suspended particles carry their state)
      int dilate = b_base_dilate, maxsteps = 1;
      (void)dilate;
      for (uint32_t i = 0; i < nslots; i++) {               // spawn on the map as it is now; first reservations
        BSlot& q = sl[i];
        q.toplevel = first; q.state = first ? B_NEW : (kids[i].ended ? B_ENDED : B_RUN);   // top-level: surface/contains are read at the first action
        q.rect = {0, -1, 0, -1};
        if (wind) {
          const int ry = draws[2 * i] % dimy, rx = draws[2 * i + 1] % dimx;
          Wind& p = q.d;
          p.pos = {(float)rx, (float)ry};
          p.ix = (int)std::round(p.pos.x); p.iy = (int)std::round(p.pos.y);
          p.surface = surface(p.ix, p.iy); p.param = soils[p.surface]; p.contains = p.param.transports;
        } else if (first) {
          const int ry = draws[2 * i] % dimy, rx = draws[2 * i + 1] % dimx;
          Water& p = q.w;
          p.pos = {(float)rx, (float)ry};
          p.ix = (int)std::round(p.pos.x); p.iy = (int)std::round(p.pos.y);
          p.surface = surface(p.ix, p.iy); p.param = soils[p.surface]; p.contains = p.param.transports;
        } else {                                              // a nested particle, as it stood when it left its parent's region
          const BChild& c = kids[i];
          Water& p = q.w;
          p.pos = c.pos; p.speed = c.speed; p.volume = c.volume; p.sediment = c.sediment; p.spill = c.spill; p.contains = c.contains;
          p.ix = (int)std::round(p.pos.x); p.iy = (int)std::round(p.pos.y);
          p.surface = 0; p.param = soils[0];
        }
      }
      std::vector<char> win(nslots);
      uint32_t nlive = nslots, epoch = 0;
      b_children.clear();
      int grain = -1;
      uint32_t chunk = 0;
      if (!wind && b_relaxed) { r_water_generation(sl); nlive = 0; }   // SMX_ENGINE_RELAXED: the water generations follow soil_relax.h
      while (nlive > 0) {
        int tshift;
        b_schedule(wind, nlive, b_base_dilate, tshift, dilate, maxsteps);
        bs_phase = bs_n > 1 ? (int)(chunk & 1u) : 0;          // column strips: INTERIOR and SEAM chunks alternate
        chunk++;
        if (tshift != grain || bs_n > 1) {                    // first chunk / new claim granularity / (strips) every chunk:
          if (tshift != grain) { grain = tshift; B_TILE = tshift; tilemin.assign((size_t)b_ntx() * b_nty(), 0xFFFFFFFFu); }
          for (BSlot& q : sl) {                               // every running particle re-issues its reservation from its current state
            if (q.state == B_DONE) continue;
            if (wind) q.rect = b_dilate(b_wind_need_run(q.d), dilate);
            else if (q.state == B_ENDED) { if (!b_water_floods(q.w)) { q.state = B_DONE; q.rect = {0, -1, 0, -1}; } else q.rect = b_water_need_flood(q.w); }
            else q.rect = b_dilate(b_water_need_run(q.w), dilate);
            q.claiming = q.state != B_DONE && b_eligible(q.rect);
          }
        }
        for (int k = 0; k < B_CHUNK; k++, epoch++) {
          b_winners(sl, tilemin, win);
          for (uint32_t i = 0; i < nslots; i++) {
            if (!win[i]) continue;
            BSlot& q = sl[i];
            const BRect held = q.rect;
            int acted = 0;
            while (true) {
              BRect need;
              if (wind) need = b_wind_need_run(q.d);
              else if (q.state == B_RUN || q.state == B_NEW) need = b_water_need_run(q.w);
              else { if (!b_water_floods(q.w)) { q.state = B_DONE; break; } need = b_water_need_flood(q.w); }
              if (acted >= maxsteps || !b_inside(held, need)) { q.rect = (!wind && q.state == B_ENDED) ? need : b_dilate(need, dilate); q.claiming = b_eligible(q.rect); break; }   // no margin for a flood
              acted++;
              b_guard = &held; b_ctx = wind ? "wind" : (q.state != B_ENDED ? "water-step" : "flood");
              if (q.state == B_NEW) {                         // the particle's constructor reads its spawn cell NOW (water.h:15-17, wind.h:19-21)
                q.state = B_RUN;
                if (wind) {
                  Wind& p = q.d;
                  p.surface = surface(p.ix, p.iy); p.param = soils[p.surface]; p.contains = p.param.transports;
                  if (soils[p.contains].suspension == 0.0) { q.state = B_DONE; b_guard = nullptr; break; }   // wind.h:56-57
                } else {
                  Water& p = q.w;
                  p.surface = surface(p.ix, p.iy); p.param = soils[p.surface]; p.contains = p.param.transports;
                }
              }
              if (wind) {
                if (wind_move(q.d)) { ctr.steps_wind++; wind_interact(q.d); }
                else q.state = B_DONE;
              } else if (q.state == B_RUN) {
                if (water_move(q.w)) {
                  ctr.steps_water_all++; if (q.toplevel) ctr.steps_water_top++;
                  if (!water_interact(q.w)) q.state = B_ENDED;
                } else q.state = B_ENDED;
              } else {
                b_defer = true; b_parent = i; b_seq = 0;
                water_flood(q.w);
                b_defer = false;
                q.state = B_DONE;
              }
              b_guard = nullptr;
              if (q.state == B_DONE) break;
            }
            if (q.state == B_DONE) q.rect = {0, -1, 0, -1};
          }
        }
        b_epochs += B_CHUNK;
        nlive = 0;
        for (const BSlot& q : sl) if (q.state != B_DONE) nlive++;
      }
      b_ngen++;
      kids = b_children;
      std::stable_sort(kids.begin(), kids.end(), [](const BChild& a, const BChild& b) { return a.key < b.key; });
      b_children.clear();
      first = false;
      nslots = wind ? 0u : (uint32_t)kids.size();
    }
  }
  // =====================================================================================================
  // RELAXED SCHEDULE -- CPU restatement of soilmachine_amd/csrc/soil_relax.h (SMX_ENGINE_RELAXED), written from its
  // definition, not from its code: one generation of water particles (wind keeps the exclusive schedule above; the
  // grid pass is batched_grid(), whose nested particles come back here as generations). Per epoch:
  //   step     every running particle, in any order, takes ONE step of the reference -- move() (water.h:43-73) and the
  //            arithmetic of interact() (water.h:75-121) -- on the map as the epoch found it; what it would have written
  //            (the visit of the track plane, the erosion or the deposit at the cell it leaves, the cascade at the cell
  //            it arrives at) becomes a request;
  //   apply    the requests, per cell in slot order (cells are independent, so: all of them in slot order), with the
  //            reference's Layermap::remove / add;
  //   cascade  every cell that was asked for, ONCE, if the engine's conservative pre-check (soil_core.h cascade_precheck:
  //            some neighbour within 0.01 % of its slope limit) holds on the map after the edits -- colour by colour
  //            (x mod 3, y mod 3), Particle::cascade(.., 0);
  //   floods   particles that ended: behind 4x4-cell claim tiles without margin, lowest slot wins, a claim issued in one flood
  //            epoch is good for the next epoch only; flood epochs are every fourth epoch and every epoch in which at least
  //            1/64 as many particles wait as ran (relax_floods_due); nested particles -> children of this generation.
  //   strips   (so_batched_set_strips) INTERIOR / SEAM chunks of 32 epochs: a particle steps iff the columns its step touches are eligible
  //            in the running kind of chunk, a flood reservation is issued iff eligible, every chunk starts with fresh reservations of
  //            the waiting particles, every epoch is a flood epoch (the rule above counts one device's particles: not partition-invariant).
  // =====================================================================================================
  bool b_relaxed = false;
  int b_water_generations = 8;                                // so_set_water_generations (smx_set_water_generations; the engine's default)
  int b_water_stagger = 0;                                    // (== csrc/soil_batch.h SMX_WATER_STAGGER, the engine's default) so_set_water_stagger (smx_set_water_stagger): > 0 = a relaxed water phase is ONE phase whose k batches of ceil(n / k)
                                                              // particles (by slot) take their first step in epoch batch * gap
  struct RReq { int x, y; double vol, amount; uint32_t type; int kind; int cx, cy; };   // kind 1: erode, 2: deposit
  uint32_t r_surf(int x, int y) const { const uint32_t t = top[cidx(x, y)]; return t == NIL ? 0u : type[t]; }
  bool r_precheck(int x, int y) const {
    const double sc = (double)(float)SCALE, hC = height(x, y);
    bool any = false;
    for (const i2& nn : NEIGH) {
      const int nx = x + nn.x, ny = y + nn.y;
      if (nx >= dimx || ny >= dimy || nx < 0 || ny < 0) continue;
      const double dh = hC - height(nx, ny);
      const uint32_t ty = dh > 0 ? r_surf(x, y) : r_surf(nx, ny);
      const double lim = (double)soils[ty].maxdiff * 0.9999;
      if (!(std::fabs(dh) * sc * 0.0125 < lim)) any = true;
    }
    return any;
  }
  void r_water_generation(std::vector<BSlot>& sl) {
    const uint32_t nslots = (uint32_t)sl.size();
    B_TILE = 2;                                               // floods claim 4x4-cell tiles (b_schedule, water)
    const int nty = b_nty();
    std::vector<uint32_t> tilemin((size_t)b_ntx() * nty, 0xFFFFFFFFu), claim_for(nslots, 0xFFFFFFFFu), waiting;
    std::vector<RReq> reqs;
    std::vector<std::pair<int, int>> lists[9];
    std::vector<uint32_t> stamp((size_t)dimx * dimy, 0u);     // epoch + 1 of the last cascade request of a cell
    uint32_t epoch = 0, chunk = 0;
    uint32_t birth_n = 0, birth_gap = 0;                      // smx_set_water_stagger: a top-level generation of a staggered phase is the whole phase
    if (b_water_stagger > 0 && b_water_generations > 1 && nslots && sl[0].toplevel) { birth_n = (nslots + (uint32_t)b_water_generations - 1u) / (uint32_t)b_water_generations; birth_gap = (uint32_t)b_water_stagger; }
    for (uint32_t i = 0; i < nslots; i++)                    // records whose flood is what did not fit their parent's region: reserved for epoch 0
      if (sl[i].state == B_ENDED && bs_n <= 1) { sl[i].rect = b_water_need_flood(sl[i].w); claim_for[i] = 0u; waiting.push_back(i); }
    auto running = [&] { for (const BSlot& q : sl) if (q.state != B_DONE) return true; return false; };
    while (running()) {
     // column strips: INTERIOR and SEAM chunks of 32 epochs alternate; an epoch number is skipped between two chunks, and every chunk
     // starts with fresh reservations of the particles that wait for their flood (without strips chunking has no meaning: one epoch per turn)
     bs_phase = bs_n > 1 ? (int)(chunk & 1u) : 0;
     if (bs_n > 1 && chunk > 0) epoch++;
     chunk++;
     if (bs_n > 1) {
       waiting.clear();
       for (uint32_t i = 0; i < nslots; i++) {
         if (sl[i].state != B_ENDED) continue;
         sl[i].rect = b_water_need_flood(sl[i].w);
         claim_for[i] = b_eligible(sl[i].rect) ? epoch : 0xFFFFFFFFu;
         waiting.push_back(i);
       }
     }
     for (int k = 0; k < (bs_n > 1 ? B_CHUNK : 1); k++) {
      // ---- step
      reqs.clear();
      uint32_t nrun = 0;
      for (uint32_t i = 0; i < nslots; i++) {
        BSlot& q = sl[i];
        if (q.state != B_RUN && q.state != B_NEW) continue;
        if (q.state == B_NEW && q.toplevel && birth_gap && epoch < (i / birth_n) * birth_gap) continue;   // staggered generations: not born yet (it waits as it was spawned)
        nrun++;
        Water t = q.w;
        const double vol0 = t.volume;
        const bool mv = water_move(t, true);                  // sets t.ix/t.iy = the cell it starts from; no side effects
        const int ix = t.ix, iy = t.iy;
        if (bs_n > 1) {                                       // the step must lie inside what this kind of chunk may touch: dormant otherwise
          const int ex = mv ? (int)std::round(t.pos.x) : ix;
          if (!b_eligible_x(std::min(ix, ex) - 1, std::max(ix, ex) + 1)) { nrun--; continue; }
        }
        if (q.state == B_NEW) { t.contains = soils[r_surf(ix, iy)].transports; q.state = B_RUN; }   // WaterParticle(map), water.h:15-17
        RReq r{ix, iy, vol0, 0.0, 0u, 0, -1, -1};
        if (mv) {
          ctr.steps_water_all++; if (q.toplevel) ctr.steps_water_top++;
          const int nx = (int)std::round(t.pos.x), ny = (int)std::round(t.pos.y);
          double c_eq = t.param.solubility * (height(ix, iy) - height_bilinear(t.pos)) * (double)SCALE / 80.0;
          if (c_eq < 0.0) c_eq = 0.0;
          if (c_eq > 1.0) c_eq = 1.0;
          if ((double)(soils[t.contains].erosionrate) < wfreq[fidx(ix, iy)]) t.contains = soils[t.contains].erodes;
          const double cdiff = c_eq - t.sediment;
          if (cdiff > 0) {
            t.sediment += t.param.equrate * cdiff;
            t.contains = soils[r_surf(ix, iy)].transports;
            r.amount = t.param.equrate * cdiff * t.volume; r.kind = 1;
          } else if (cdiff < 0) {
            t.sediment += soils[t.contains].equrate * cdiff;
            r.amount = -soils[t.contains].equrate * cdiff * t.volume; r.type = t.contains; r.kind = 2;
          }
          r.cx = nx; r.cy = ny;
          t.sediment /= (1.0 - t.evaprate);
          if (t.sediment > 1.0) t.sediment = 1.0;
          t.volume *= (1.0 - t.evaprate);
          t.ix = nx; t.iy = ny;
          if (!(t.volume > minvol)) q.state = B_ENDED;
        } else q.state = B_ENDED;
        if (q.state == B_ENDED) {
          q.rect = {0, -1, 0, -1};
          if (!b_water_floods(t)) q.state = B_DONE; else waiting.push_back(i);
        }
        q.w = t;
        reqs.push_back(r);
      }
      // ---- apply (slot order)
      for (const RReq& r : reqs) {
        const size_t fi = fidx(r.x, r.y);
        wtrack[fi] = (float)((double)wtrack[fi] + r.vol);     // water.h:53,348-351
        if (r.kind == 1) { double diff = remove(r.x, r.y, r.amount); while (std::fabs(diff) > 1E-8) diff = remove(r.x, r.y, diff); }
        else if (r.kind == 2) add(r.x, r.y, get(r.amount, r.type));
      }
      // ---- which requested cells can cascade at all? (on the map after the edits, before any cascade)
      for (auto& l : lists) l.clear();
      for (const RReq& r : reqs) {
        if (r.cx < 0) continue;
        uint32_t& st = stamp[cidx(r.cx, r.cy)];
        if (st == epoch + 1u) continue;                       // once per cell and epoch
        st = epoch + 1u;
        if (r_precheck(r.cx, r.cy)) lists[(r.cx % 3) * 3 + (r.cy % 3)].push_back({r.cx, r.cy});
        else ctr.cascade_calls++;                             // (the engine counts the dismissed cell as a call, as Particle::cascade would be)
      }
      for (auto& l : lists)
        for (const auto& c : l) particle_cascade(f2{(float)c.first, (float)c.second}, 0);
      // ---- floods
      const uint32_t nflood = (uint32_t)waiting.size();
      if (bs_n > 1 || (epoch & 3u) == 3u || (unsigned long long)nflood * 64ull >= (unsigned long long)nrun) {   // (strips: every epoch; 1/64 since round 4)
        std::vector<uint32_t> cand, win, rest;
        for (uint32_t i : waiting) if (claim_for[i] == epoch) cand.push_back(i);
        for (uint32_t i : cand) { const BRect& r = sl[i].rect; for (int tx = r.tx0; tx <= r.tx1; tx++) for (int ty = r.ty0; ty <= r.ty1; ty++) { uint32_t& m = tilemin[(size_t)tx * nty + ty]; if (i < m) m = i; } }
        std::vector<char> won(nslots, 0);
        for (uint32_t i : cand) {
          const BRect& r = sl[i].rect; bool ok = true;
          for (int tx = r.tx0; tx <= r.tx1 && ok; tx++) for (int ty = r.ty0; ty <= r.ty1; ty++) if (tilemin[(size_t)tx * nty + ty] != i) { ok = false; break; }
          if (ok) { won[i] = 1; win.push_back(i); }
        }
        for (uint32_t i : cand) { const BRect& r = sl[i].rect; for (int tx = r.tx0; tx <= r.tx1; tx++) for (int ty = r.ty0; ty <= r.ty1; ty++) tilemin[(size_t)tx * nty + ty] = 0xFFFFFFFFu; }
        std::sort(win.begin(), win.end());
        for (uint32_t i : win) {                              // winners own disjoint tiles: any order
          BSlot& q = sl[i];
          const BRect held = q.rect;
          b_guard = &held; b_ctx = "relaxed-flood";
          b_defer = true; b_parent = i; b_seq = 0;
          water_flood(q.w);
          b_defer = false; b_guard = nullptr;
          q.state = B_DONE; q.rect = {0, -1, 0, -1};
        }
        for (uint32_t i : waiting) if (!won[i]) { sl[i].rect = b_water_need_flood(sl[i].w); claim_for[i] = b_eligible(sl[i].rect) ? epoch + 1u : 0xFFFFFFFFu; rest.push_back(i); }
        waiting.swap(rest);
      }
      epoch++;
     }
    }
    b_epochs += epoch;
  }

  void batched_grid() {
    const size_t n = (size_t)dimx * dimy;
    std::vector<char> act(n, 0);                              // fixed at the start of the pass (soilmx.hip k_grid_classify)
    for (int x = 0; x < dimx; x++)
      for (int y = 0; y < dimy; y++) {
        bool a = satflag[cidx(x, y)] != 0;
        for (int dx = -1; dx <= 1 && !a; dx++)
          for (int dy = -1; dy <= 1 && !a; dy++) {
            int nx = x + dx, ny = y + dy;
            if (nx < 0 || ny < 0 || nx >= dimx || ny >= dimy) continue;
            uint32_t t = top[cidx(nx, ny)];
            a = (t != NIL && type[t] == AIR);
          }
        act[cidx(x, y)] = a ? 1 : 0;
      }
    const int ntx = (dimx + (1 << B_GRID) - 1) >> B_GRID, nty = (dimy + (1 << B_GRID) - 1) >> B_GRID;
    B_TILE = B_GRID;                                          // (the access guard below speaks in grid tiles)
    uint64_t visited = 0;
    b_children.clear();
    // 8x8-cell tiles, colour by colour (tx mod 3, ty mod 3), x-major inside a tile. A tile's REGION -- the tile and the ring of tiles around it,
    // 24 x 24 cells -- is what its sweep may touch: seep + cascade(.., 3) reach 4 cells, and its nested particles run inside (b_nested) or are
    // suspended at the edge. Regions of equal colour are disjoint (any order: ascending here); the engine orders tiles of overlapping regions
    // by colour (soilmx.hip k_grid_tiles_flow).
    for (int ph = 0; ph < (bs_n > 1 ? 2 : 1); ph++)
    for (int col = 0; col < 9; col++)
      for (int tx = col / 3; tx < ntx; tx += 3) {
        if (bs_n > 1) {                                       // INTERIOR sweep: tiles whose region lies inside one strip; SEAM sweep: the rest
          const int cx0 = std::max(tx - 1, 0) << B_GRID, cx1 = std::min(((tx + 2) << B_GRID) - 1, dimx - 1);
          bool interior = false;
          for (int g = 0; g < bs_n; g++) { const int lo = bs_x(g) + (g ? bs_a : 0), hi = bs_x(g + 1) - (g < bs_n - 1 ? bs_a : 0); if (cx0 >= lo && cx1 < hi) interior = true; }
          if ((ph == 0) != interior) continue;
        }
        for (int ty = col % 3; ty < nty; ty += 3) {
          b_defer = true; b_parent = ((uint64_t)ph * 9 + (uint64_t)col) * ntx * nty + (uint64_t)tx * nty + ty; b_seq = 0;
          const BRect g = {tx - 1 < 0 ? 0 : tx - 1, tx + 1 > ntx - 1 ? ntx - 1 : tx + 1, ty - 1 < 0 ? 0 : ty - 1, ty + 1 > nty - 1 ? nty - 1 : ty + 1};
          b_guard = &g; b_ctx = "grid";
          for (int x = tx << B_GRID; x < ((tx + 1) << B_GRID) && x < dimx; x++)
            for (int y = ty << B_GRID; y < ((ty + 1) << B_GRID) && y < dimy; y++) {
              if (!act[cidx(x, y)]) continue;
              seep(x, y);
              water_cascade(x, y, 3);
              visited++;
            }
          b_guard = nullptr; b_defer = false;
        }
      }
    ctr.wcascade_calls += n - visited;                        // the reference "calls" cascade on every cell (water.h:340)
    ctr.grid_active_cells += visited;
    std::vector<BChild> kids = b_children;
    b_children.clear();
    std::stable_sort(kids.begin(), kids.end(), [](const BChild& a, const BChild& b) { return a.key < b.key; });
    if (!kids.empty()) b_generations(false, 0, kids);         // the nested particles that left their tile's region
  }

  void water_particles(int n) {
    dep_reset();
    for (int i = 0; i < n; i++) { dep_task_begin(); Water p; water_ctor(p); water_run(p, true); dep_task_end(); }
  }
  void wind_particles(int n) {
    dep_reset();
    for (int i = 0; i < n; i++) {
      dep_task_begin();
      Wind p; wind_ctor(p);
      while (true) {
        if (!wind_move(p)) break;
        ctr.steps_wind++;
        bool alive = wind_interact(p);
        dep_step_end();
        if (!alive) break;
      }
      dep_task_end();
    }
  }
};

void so_ctx::b_check(size_t c) const {
  const int x = (int)(c / (size_t)dimy), y = (int)(c % (size_t)dimy);
  const int tx = x >> B_TILE, ty = y >> B_TILE;
  if (tx < b_guard->tx0 || tx > b_guard->tx1 || ty < b_guard->ty0 || ty > b_guard->ty1) {
    const_cast<so_ctx*>(this)->b_guard_violations++;
    if (getenv("SO_GUARD_TRACE")) fprintf(stderr, "[oracle] guard: cell (%d,%d) tile (%d,%d) outside [%d..%d]x[%d..%d] ctx %s\n", x, y, tx, ty, b_guard->tx0, b_guard->tx1, b_guard->ty0, b_guard->ty1, b_ctx);
  }
}

extern "C" {

so_ctx* so_create(int dimx, int dimy, int scale, uint64_t pool_capacity) {
  so_ctx* c = new so_ctx();
  c->dimx = dimx; c->dimy = dimy; c->SCALE = scale; c->capacity = pool_capacity;
  size_t n = (size_t)dimx * dimy;
  c->top.assign(n, NIL);
  c->size.resize(pool_capacity); c->floor.resize(pool_capacity); c->sat.resize(pool_capacity);
  c->type.resize(pool_capacity); c->prev.resize(pool_capacity);
  c->freelist.resize(pool_capacity);
  for (uint64_t i = 0; i < pool_capacity; i++) c->freelist[i] = (uint32_t)(pool_capacity - 1 - i);
  c->wfreq.assign(n, 0.0f); c->wtrack.assign(n, 0.0f); c->windfreq.assign(n, 0.0f); c->satflag.assign(n, 0);
  memset(&c->ctr, 0, sizeof(c->ctr));
  c->soils.push_back(so_soil{0, 0, 0, 0, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f});  // "Air" surface.h:43-49
  c->srand_(1);
  return c;
}
void so_destroy(so_ctx* c) { delete c; }
void so_set_soils(so_ctx* c, const so_soil* s, int n) { c->soils.assign(s, s + n); }
void so_set_scale(so_ctx* c, int scale) { c->SCALE = scale; }
void so_srand(so_ctx* c, unsigned seed) { c->srand_(seed); }
int so_rand(so_ctx* c) { return c->rand_(); }

void so_initialize(so_ctx* c, int seed, const so_layer* layers, int nlayers) {   // layermap.h:163-216
  size_t n = (size_t)c->dimx * c->dimy;
  c->top.assign(n, NIL); c->satflag.assign(n, 0);
  c->freelist.resize(c->capacity);
  for (uint64_t i = 0; i < c->capacity; i++) c->freelist[i] = (uint32_t)(c->capacity - 1 - i);
  Noise noise;                                       // static FastNoiseLite SurfLayer::noise (surface.h:76,103)
  const int MAXSEED = 10000;
  for (int l = 0; l < nlayers; l++) {
    const float f = (float)l / (float)nlayers;
    const int Z = (int)((float)seed + f * (float)MAXSEED);
    noise.octaves = (int)layers[l].octaves; noise.bound();        // SurfLayer::init surface.h:82-89
    noise.lacunarity = layers[l].lacunarity;
    noise.gain = layers[l].gain; noise.bound();
    noise.frequency = layers[l].frequency;
    float pz = (float)(Z % MAXSEED) / 1.0f;
    for (int i = 0; i < c->dimx; i++)
      for (int j = 0; j < c->dimy; j++) {
        float px = (float)i / (float)c->dimx, py = (float)j / (float)c->dimy;
        float val = layers[l].bias + layers[l].scale * noise.get(px, py, pz);   // SurfLayer::get surface.h:95-99
        if (val < layers[l].min) val = layers[l].min;
        double h = val;
        c->add(i, j, c->get(h, layers[l].type));
      }
  }
}

int so_import(so_ctx* c, const uint32_t* count, const uint32_t* type, const double* size, const double* floor,
              const double* sat, const float* wfreq, const float* wtrack, const float* windfreq, uint64_t rand_calls) {
  size_t n = (size_t)c->dimx * c->dimy;
  c->top.assign(n, NIL); c->satflag.assign(n, 0);
  c->freelist.resize(c->capacity);
  for (uint64_t i = 0; i < c->capacity; i++) c->freelist[i] = (uint32_t)(c->capacity - 1 - i);
  size_t off = 0;
  for (size_t cell = 0; cell < n; cell++) {
    uint32_t pv = NIL;
    for (uint32_t k = 0; k < count[cell]; k++, off++) {
      if (c->freelist.empty()) return -1;
      uint32_t e = c->freelist.back(); c->freelist.pop_back();
      c->type[e] = type[off]; c->size[e] = size[off]; c->floor[e] = floor[off]; c->sat[e] = sat[off]; c->prev[e] = pv;
      if (sat[off] != 0.0) c->satflag[cell] = 1;
      pv = e;
    }
    c->top[cell] = pv;
  }
  if (wfreq) c->wfreq.assign(wfreq, wfreq + n);
  if (wtrack) c->wtrack.assign(wtrack, wtrack + n);
  if (windfreq) c->windfreq.assign(windfreq, windfreq + n);
  // advance the rand stream to the given number of consumed draws (stream must have been seeded first)
  uint64_t have = c->ctr.rand_calls;
  for (uint64_t i = have; i < rand_calls; i++) c->rand_();
  return 0;
}
uint64_t so_num_sections(so_ctx* c) { return c->capacity - c->freelist.size(); }
void so_export(so_ctx* c, uint32_t* count, uint32_t* type, double* size, double* floor, double* sat,
               float* wfreq, float* wtrack, float* windfreq) {
  size_t n = (size_t)c->dimx * c->dimy;
  size_t off = 0;
  std::vector<uint32_t> col;
  for (size_t cell = 0; cell < n; cell++) {
    col.clear();
    for (uint32_t t = c->top[cell]; t != NIL; t = c->prev[t]) col.push_back(t);
    count[cell] = (uint32_t)col.size();
    for (size_t k = col.size(); k-- > 0; off++) {
      uint32_t e = col[k];
      type[off] = c->type[e]; size[off] = c->size[e]; floor[off] = c->floor[e]; sat[off] = c->sat[e];
    }
  }
  if (wfreq) memcpy(wfreq, c->wfreq.data(), n * 4);
  if (wtrack) memcpy(wtrack, c->wtrack.data(), n * 4);
  if (windfreq) memcpy(windfreq, c->windfreq.data(), n * 4);
}
void so_heights(so_ctx* c, double* out) {
  for (int x = 0; x < c->dimx; x++)
    for (int y = 0; y < c->dimy; y++) out[c->cidx(x, y)] = c->height(x, y);
}
void so_tick(so_ctx* c, int nwater, int nwind, int dowater, int dowind) {   // SoilMachine.cpp:283-329
  if (dowater) c->water_particles(nwater);
  if (dowater) c->grid_pass();
  if (dowind) c->wind_particles(nwind);
  if (dowater) { c->map_frequency(); c->reset_frequency(); }
}
void so_water_particles(so_ctx* c, int n) { c->water_particles(n); }
void so_grid_pass(so_ctx* c) { c->grid_pass(); }
void so_wind_particles(so_ctx* c, int n) { c->wind_particles(n); }
void so_map_frequency(so_ctx* c) { c->map_frequency(); }
void so_reset_frequency(so_ctx* c) { c->reset_frequency(); }
void so_get_counters(so_ctx* c, so_counters* out) {
  c->ctr.pool_free = c->freelist.size();
  *out = c->ctr;
}
double so_height(so_ctx* c, int x, int y) { return c->height(x, y); }
double so_height_bilinear(so_ctx* c, float px, float py) { return c->height_bilinear(f2{px, py}); }
void so_normal(so_ctx* c, int x, int y, float* o) { f3 n = c->normal(x, y); o[0] = n.x; o[1] = n.y; o[2] = n.z; }
uint32_t so_surface(so_ctx* c, int x, int y) { return c->surface(x, y); }
void so_add(so_ctx* c, int x, int y, double size, uint32_t type) { c->add(x, y, c->get(size, type)); }
double so_remove(so_ctx* c, int x, int y, double h) { return c->remove(x, y, h); }
void so_particle_cascade(so_ctx* c, float px, float py, int tl) { c->particle_cascade(f2{px, py}, tl); }
void so_water_cascade(so_ctx* c, int x, int y, int spill) { c->water_cascade(x, y, spill); }
void so_seep(so_ctx* c, int x, int y) { c->seep(x, y); }
void so_batched_water(so_ctx* c, int n) { c->b_generations(false, (uint32_t)(n > 0 ? n : 0), {}); }
void so_batched_wind(so_ctx* c, int n) { c->b_generations(true, (uint32_t)(n > 0 ? n : 0), {}); }
void so_batched_grid(so_ctx* c) { c->batched_grid(); }
void so_set_water_generations(so_ctx* c, int k) { c->b_water_generations = k < 1 ? 1 : k; }
void so_set_water_stagger(so_ctx* c, int gap) { c->b_water_stagger = gap < 0 ? 0 : gap; }
void so_batched_tick(so_ctx* c, int nwater, int nwind, int dowater, int dowind) {
  if (dowater) {                                              // smx_tick_water: k consecutive top-level generations of n/k (soil_batch.h SMX_WATER_GENERATIONS)
    uint32_t left = nwater > 0 ? (uint32_t)nwater : 0u;
    if (c->b_relaxed && c->b_water_stagger > 0 && c->b_water_generations > 1 && left > 0) { so_batched_water(c, (int)left); left = 0; }   // ONE staggered phase (smx_set_water_stagger)
    for (uint32_t q = 0, k = (uint32_t)c->b_water_generations; q < k && left > 0; q++) { const uint32_t nq = left / (k - q); left -= nq; if (nq) so_batched_water(c, (int)nq); }
    so_batched_grid(c);
  }
  if (dowind) so_batched_wind(c, nwind);
  if (dowater) { c->map_frequency(); c->reset_frequency(); }
}
void so_set_relaxed(so_ctx* c, int on) { c->b_relaxed = on != 0; }   // water generations of so_batched_* follow the relaxed schedule
void so_batched_set_dilate(so_ctx* c, int tiles) { c->b_base_dilate = tiles < 0 ? 0 : tiles; }
void so_batched_set_cell_above(so_ctx* c, uint32_t n) { c->b_cell_above = n; }
void so_batched_set_strips(so_ctx* c, int n, int a, int b) { c->bs_n = n < 1 ? 1 : n; c->bs_a = a; c->bs_b = b; }
void so_batched_stats(so_ctx* c, uint64_t* out3) { out3[0] = c->b_epochs; out3[1] = c->b_ngen; out3[2] = c->b_guard_violations; }
void so_dep_enable(so_ctx* c, int on) {
  c->dep_on = on != 0;
  c->dep_rr = on == 2;
  c->dep_tasks = c->dep_maxlev = c->dep_maxlev_mv = c->dep_free = c->dep_crit = c->dep_steps = c->dep_sumlev = 0;
}
void so_dep_stats(so_ctx* c, uint64_t* out7) {
  out7[0] = c->dep_tasks; out7[1] = c->dep_maxlev; out7[2] = c->dep_maxlev_mv; out7[3] = c->dep_free;
  out7[4] = c->dep_crit; out7[5] = c->dep_steps; out7[6] = c->dep_sumlev;
}
float so_noise_fbm(float x, float y, float z, int octaves, float lacunarity, float gain, float frequency) {
  Noise n;
  n.octaves = octaves; n.bound(); n.lacunarity = lacunarity; n.gain = gain; n.bound(); n.frequency = frequency;
  return n.get(x, y, z);
}

}  // extern "C"
