// soil_core.h -- device-side state layout and the exact (reference-order) particle transport code.
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   cells[x*dimy+y]  : one 32-byte Sec record per column = the column's TOP section stored INLINE
//                      {size, floor, sat : f64; type : u32; prev : u32 index of the section below}
//                      -> Layermap::height()/surface()/top() (layermap.h:150,417-425) are ONE aligned
//                         32-byte record, no pointer chase; an empty column has type == EMPTY and
//                         size = floor = 0 (height 0.0, surface 0 as the reference returns for NULL).
//   pool[i]          : the same 32-byte record for BURIED sections (reached through prev), plus a
//                      free-index stack. Replaces sec / secpool (layermap.h:37-119); node identity is
//                      never observable in the reference, so this allocator is parity-safe.
//   flags[cell]      : u8, bit0 = top section is water ("Air"), bit1 = column holds a non-zero
//                      saturation somewhere (sticky) -> input of the grid-pass classification kernel.
//   wfreq/wtrack/windfreq[y*dimx+x] : f32 planes (water.h:22-23,345-346; wind.h:25,48).
//
// Code structure: NO recursion and no out-of-line calls on the particle path. The reference's mutual
// recursion WaterParticle::cascade -> nested WaterParticle -> flood -> cascade (water.h:123-283) is an
// explicit frame stack (`Frame`, depth bounded by the spill budget, water.h:34,125,255,279), so a particle
// lives in registers for its whole life. All state access goes through an access policy `P`:
//   SerialPolicy       one walker, reference order, direct access                       (soil_serial.h)
//   SpecPolicy         optimistic parallel execution with per-cell stamps + undo log    (soil_spec.h)
//
// Arithmetic follows SURVEY.md Appendix A exactly (mixed float/double promotions, no FMA: this file
// must be compiled with -ffp-contract=off). Every function cites the reference lines it implements.
#pragma once
#include <stdint.h>
#include <math.h>

#ifndef SMX_D
#define SMX_D __device__ __forceinline__
#endif
// Fire-and-forget read-modify-writes (no return value -> the lane never waits for the memory round trip) and the
// coherent load that goes with them. The exact engines are chains of DEPENDENT accesses on one lane: a plain
// `word |= bit` costs a full round trip each (load, wait, or, store); the grid pass did 27 of them per water transfer.
#ifdef SMX_HOSTSIM
#define SMX_OR64_ASYNC(p, v) (void)(*(p) |= (v))
#define SMX_OR32_ASYNC(p, v) (void)(*(p) |= (v))
#define SMX_AND32_ASYNC(p, v) (void)(*(p) &= (v))
#define SMX_LD64_COHERENT(p) (*(p))
#else
#define SMX_OR64_ASYNC(p, v) (void)__hip_atomic_fetch_or((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SMX_OR32_ASYNC(p, v) (void)__hip_atomic_fetch_or((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SMX_AND32_ASYNC(p, v) (void)__hip_atomic_fetch_and((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SMX_LD64_COHERENT(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

#if defined(SMX_PROF_FLOOD) && !defined(SMX_HOSTSIM)
// (experiment build, -DSMX_PROF_FLOOD: where does one lane's flood / grid tile go? 100 MHz ticks and counts per section, accumulated in g_sect[];
//  printed by smx_destroy -- profiles/r06_floods.md)
extern __device__ unsigned long long g_sect[32];
#define SMX_SECT_T0() const unsigned long long sect_t0_ = wall_clock64()
#define SMX_SECT(k) do { atomicAdd(&g_sect[2 * (k)], wall_clock64() - sect_t0_); atomicAdd(&g_sect[2 * (k) + 1], 1ull); } while (0)
#else
#define SMX_SECT_T0() do {} while (0)
#define SMX_SECT(k) do {} while (0)
#endif
namespace smx {

constexpr uint32_t NIL = 0xFFFFFFFFu;
constexpr uint32_t EMPTY = 0xFFFFFFFFu;   // cells[c].type of an empty column (dat[c] == NULL)
constexpr uint32_t AIR = 0;               // soilmap["Air"] (surface.h:53-57); doubles as water
constexpr uint8_t F_AIR = 1, F_SAT = 2;
constexpr int MAX_SOILS = 256;

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
           C_CASCADE_TRANSFERS, C_WCASCADE_CALLS, C_GRID_ACTIVE, C_POOL_OVERFLOW, C_LIVE_SECTIONS,
           C_FLOOD_NESTED_STEPS, C_GRID_NESTED_STEPS /* steps of nested particles run INSIDE floods / grid tiles (device kernels only: the bench's algorithmic bytes) */, C_COUNT };

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
  // Columns x_lo .. x_hi-1 are HELD by this context (a column strip + its halo; the whole map by default). cells / flags and the
  // per-cell planes of the engines are allocated for that range only and addressed with the map's global index x*dimy + y (the
  // pointers are biased by x_lo*dimy): nothing outside the range may be touched.
  int x_lo, x_hi;
};

struct f2 { float x, y; };
struct f3 { float x, y, z; };

SMX_D f3 sub3(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
SMX_D f3 cross3(f3 x, f3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
SMX_D float dot3(f3 a, f3 b) { float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }
SMX_D float dot2(f2 a, f2 b) { float tx = a.x * b.x, ty = a.y * b.y; return tx + ty; }
SMX_D f3 normalize3(f3 v) { float s = 1.0f / sqrtf(dot3(v, v)); return {v.x * s, v.y * s, v.z * s}; }
SMX_D f2 normalize2(f2 v) { float s = 1.0f / sqrtf(dot2(v, v)); return {v.x * s, v.y * s}; }
SMX_D float mixd(float x, float y, double a) { return (float)((double)x * (1.0 - a) + (double)y * a); }

// The 8 neighbours in the order of the reference's table (particle.h:30-39, water.h:155-164):
// code k -> (dx, dy) = (-1,-1),(-1,0),(-1,1),(0,-1),(0,1),(1,-1),(1,0),(1,1)
SMX_D int nb_dx(int k) { return (k < 3) ? -1 : (k < 5 ? 0 : 1); }
SMX_D int nb_dy(int k) { return (k < 3) ? (k - 1) : (k < 5 ? (k == 3 ? -1 : 1) : (k - 6)); }

// In-bounds neighbours of (ix,iy) sorted by height, highest first. std::sort on <= 16 elements is an
// insertion sort and therefore stable for the strict comparator a.h > b.h.
// Result: number of neighbours, `codes` = 4-bit neighbour codes, first in the low nibble.
struct Sorted { int num; uint32_t codes; };

struct Water {            // WaterParticle (water.h:9-41); `param` is re-read from the soil table via `surface`
  f2 pos, speed;
  double volume, sediment, evaprate;
  int spill, ix, iy;
  float friction;         // param.friction after the frequency scaling of move() (water.h:53)
  uint32_t surface, contains;
};
struct Wind {             // WindParticle (wind.h:11-46)
  f2 pos;
  f3 speed;
  double sediment, height, sheight;
  int ix, iy;
  uint32_t surface, contains;
};

// (size, floor, type) of a column's top section, and the 3x3 block of such views a particle carries from step
// to step (see Sim "the carried 3x3 block"). Namespace scope: the speculative engine checkpoints them.
struct View { double size, floor; uint32_t type; };
#define SMX_FOR9(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#define SMX_FOR8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
struct Blk {
  View v0, v1, v2, v3, v4, v5, v6, v7, v8;                   // v8 = centre, vk = neighbour code k
  size_t c0; uint32_t inb; float fr, tr;                     // inb: bit k = neighbour code k is in bounds
};

// One suspended WaterParticle::cascade invocation (water.h:151-283): centre, remaining spill budget,
// the sorted neighbour list and the loop index.
struct Frame { int ix, iy, spill, num, i; uint32_t codes, cand, stamp; };   // cand / stamp: see Sim::push_frame
constexpr int MAX_FRAMES = 24;

// The wind step's one level of cascade recursion (particle.h:96-97), OUT of line and BY VALUE for policies that ask
// for it (P::OUTLINE_RECURSION): inlined four times into the wave kernel (two cascades x the A/B ping-pong of the
// carried blocks) the map-based cascade made k_spec_exec_coop<wind> 492 KB of code with 514 spilled VGPRs.
template <class P> struct RecRes { P pol; Blk b; bool ab, moved; };
#ifdef SMX_HOSTSIM
#define SMX_NOINLINE __attribute__((noinline))
#else
#define SMX_NOINLINE __device__ __attribute__((noinline))
#endif
template <class P> SMX_NOINLINE RecRes<P> wind_recurse(DevState s, const SoilP* soils, P pol, float cx, float cy, int ix, int iy);
// policies whose thread owns every cell it touches for as long as it acts (serial walker, batched / relaxed engines): nothing changes under it
template <class P, class = void> struct exclusive_access { static constexpr bool value = false; };
template <class P> struct exclusive_access<P, decltype((void)P::EXCLUSIVE)> { static constexpr bool value = P::EXCLUSIVE; };
template <class P, class = void> struct outline_recursion { static constexpr bool value = false; };
template <class P> struct outline_recursion<P, decltype((void)P::OUTLINE_RECURSION)> { static constexpr bool value = P::OUTLINE_RECURSION; };

template <class P>
struct Sim {
  const DevState& s;
  const SoilP* soils;      // soil table (LDS copy on the device)
  P& pol;
  bool grid_mode = false;  // inside the grid pass: newly wet cells go to the active bitmap
  bool ab = false;         // speculative execution only: this particle has been aborted
  bool susp_ok = false;    // speculative execution only: the main loop of this particle may be suspended ...
  int susp = 0;            // ... and has been (2 = before interact()'s block load; the block is checkpointed)
  uint32_t n_mod = 0;      // writes of (size, floor, type) this Sim has made to the map so far (push_frame's candidate mask is valid while it stands still)
  // counters live in registers and are flushed once per kernel
  unsigned long long n_steps_w_top = 0, n_steps_w_all = 0, n_steps_wind = 0, n_nested = 0, n_floods = 0,
                     n_casc = 0, n_casc_tr = 0, n_wcasc = 0, n_overflow = 0;

  SMX_D Sim(const DevState& st, const SoilP* so, P& p) : s(st), soils(so), pol(p) {}

  SMX_D size_t cidx(int x, int y) const { return (size_t)x * s.dimy + y; }   // layermap.h:151
  SMX_D size_t fidx(int x, int y) const { return (size_t)y * s.dimx + x; }   // water.h:53,349

  SMX_D void flush_counters() {
    pol.add_counter(s, C_STEPS_WATER_TOP, n_steps_w_top); pol.add_counter(s, C_STEPS_WATER_ALL, n_steps_w_all);
    pol.add_counter(s, C_STEPS_WIND, n_steps_wind); pol.add_counter(s, C_NESTED, n_nested);
    pol.add_counter(s, C_FLOODS, n_floods); pol.add_counter(s, C_CASCADE_CALLS, n_casc);
    pol.add_counter(s, C_CASCADE_TRANSFERS, n_casc_tr); pol.add_counter(s, C_WCASCADE_CALLS, n_wcasc);
    pol.add_counter(s, C_POOL_OVERFLOW, n_overflow);
    n_steps_w_top = n_steps_w_all = n_steps_wind = n_nested = n_floods = n_casc = n_casc_tr = n_wcasc = n_overflow = 0;
  }

  // ---------------- cell access (every access announces itself to the policy first) ----------------
  SMX_D Sec& cell(int x, int y) {
    const size_t c = cidx(x, y);
    if (!pol.touch(s, c, fidx(x, y))) ab = true;
    return s.cells[c];
  }
  SMX_D double height(int x, int y) {                       // Layermap::height(ivec2) layermap.h:422-425
    const Sec& c = cell(x, y);
    return c.floor + c.size;
  }
  SMX_D uint32_t surface(int x, int y) {                    // Layermap::surface layermap.h:417-420
    const uint32_t t = cell(x, y).type;
    return t == EMPTY ? 0u : t;
  }
  // ---- batched access: announce N cells to the policy at once, then load their (size, floor, type) views with
  // all loads in flight together. One memory round trip per batch instead of one per cell: the exact engines
  // are latency-bound (DESIGN.md "Per-step latency"), so the step is organised as three batches.
  using View = smx::View;
  using Blk = smx::Blk;
  SMX_D static double vh(const View& v) { return v.floor + v.size; }
  SMX_D static uint32_t vsurf(const View& v) { return v.type == EMPTY ? 0u : v.type; }
  SMX_D View load_view(size_t c) const {
    const Sec& r = s.cells[c];
    View v; v.size = r.size; v.floor = r.floor; v.type = r.type;
    return v;
  }
  // c[i] = cell-order index, f[i] = frequency-order index of the same column (water.h:53)
  template <int N>
  SMX_D void batch(const size_t (&c)[N], const size_t (&f)[N], View (&v)[N]) {
    typename P::template Pre<N> pf;
    pol.template prefetch<N>(s, c, f, pf);                  // stamps, claims, undo words: loads only, no waiting
    Sec r[N];
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = s.cells[c[i]];       // whole 32-byte records: the views AND the undo snapshots
    if (!pol.template acquire_log<N>(s, c, pf, r)) ab = true;
#pragma unroll
    for (int i = 0; i < N; i++) { v[i].size = r[i].size; v[i].floor = r[i].floor; v[i].type = r[i].type; }
  }

  SMX_D double height_bilinear(f2 pos) {                    // Layermap::height(vec2) layermap.h:427-439
    double h = 0.0;
    const float fx = floorf(pos.x), fy = floorf(pos.y);
    const int px = (int)fx, py = (int)fy;
    const float wx = pos.x - fx, wy = pos.y - fy;
    const size_t c00 = cidx(px, py), f00 = fidx(px, py);
    const size_t cc[4] = {c00, c00 + (size_t)s.dimy, c00 + 1, c00 + (size_t)s.dimy + 1};
    const size_t ff[4] = {f00, f00 + 1, f00 + (size_t)s.dimx, f00 + (size_t)s.dimx + 1};
    View v[4];
    batch<4>(cc, ff, v);
    const double h00 = vh(v[0]), h10 = vh(v[1]), h01 = vh(v[2]), h11 = vh(v[3]);
    h += (1.0 - wx) * (1.0 - wy) * h00;
    h += (1.0 - wx) * wy * h10;                             // crossed weights as in the reference (:434-435)
    h += wx * (1.0 - wy) * h01;
    h += wx * wy * h11;                                     // wx*wy is a FLOAT product (:436)
    return h;
  }
  // Layermap::normal(ivec2) layermap.h:341-377; also hands back the centre cell's view (height / surface of ipos)
  SMX_D f3 normal(int x, int y, View& centre) {
    const int SC = s.scale;
    const bool xm = x > 0, xp = x < s.dimx - 1, ym = y > 0, yp = y < s.dimy - 1;
    const size_t c0 = cidx(x, y);
    const size_t f0 = fidx(x, y);
    const size_t cc[5] = {c0, xm ? c0 - (size_t)s.dimy : c0, xp ? c0 + (size_t)s.dimy : c0, ym ? c0 - 1 : c0, yp ? c0 + 1 : c0};
    const size_t ff[5] = {f0, xm ? f0 - 1 : f0, xp ? f0 + 1 : f0, ym ? f0 - (size_t)s.dimx : f0, yp ? f0 + (size_t)s.dimx : f0};
    View v[5];
    batch<5>(cc, ff, v);
    centre = v[0];
    const double h0 = vh(v[0]);
    const double hxm = xm ? vh(v[1]) : 0.0, hxp = xp ? vh(v[2]) : 0.0;
    const double hym = ym ? vh(v[3]) : 0.0, hyp = yp ? vh(v[4]) : 0.0;
    f3 n = {0.f, 0.f, 0.f};
    const f3 p = {(float)x, (float)(SC * h0), (float)y};
    const f3 bm = {(float)(x - 1), (float)(SC * hxm), (float)y}, bp = {(float)(x + 1), (float)(SC * hxp), (float)y};
    const f3 cm = {(float)x, (float)(SC * hym), (float)(y - 1)}, cp = {(float)x, (float)(SC * hyp), (float)(y + 1)};
    int k = 0;
    if (xm && ym) { f3 w = cross3(sub3(cm, p), sub3(bm, p)); n.x += w.x; n.y += w.y; n.z += w.z; k++; }
    if (xm && yp) { f3 w = cross3(sub3(cp, p), sub3(bm, p)); n.x -= w.x; n.y -= w.y; n.z -= w.z; k++; }
    if (xp && ym) { f3 w = cross3(sub3(cm, p), sub3(bp, p)); n.x -= w.x; n.y -= w.y; n.z -= w.z; k++; }
    if (xp && yp) { f3 w = cross3(sub3(cp, p), sub3(bp, p)); n.x += w.x; n.y += w.y; n.z += w.z; k++; }
    const float kf = (float)k;
    const f3 q = {n.x / kf, n.y / kf, n.z / kf};
    return normalize3(q);
  }
  SMX_D f3 normal(int x, int y) { View c; return normal(x, y, c); }

  // Stable descending sort of the <= 8 in-bounds neighbours by height == the reference's std::sort (an
  // insertion sort for n <= 16): ties keep neighbour-table order, i.e. the order is
  // the total order (h desc, code asc). Done as a 19-comparator sorting network on registers; out-of-bounds
  // slots carry h = -inf and sort to the end (heights are >= 0).
  SMX_D static Sorted sort8(double (&h)[8]) {
    int code[8]; int num = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { code[k] = k; num += (h[k] > -INFINITY) ? 1 : 0; }
#define SMX_CE(i, j)                                                                             \
    {                                                                                            \
      const bool sw = (h[j] > h[i]) || (h[j] == h[i] && code[j] < code[i]);                      \
      const double hi = sw ? h[j] : h[i], hj = sw ? h[i] : h[j];                                 \
      const int ci = sw ? code[j] : code[i], cj = sw ? code[i] : code[j];                        \
      h[i] = hi; h[j] = hj; code[i] = ci; code[j] = cj;                                          \
    }
    SMX_CE(0, 1) SMX_CE(2, 3) SMX_CE(4, 5) SMX_CE(6, 7)
    SMX_CE(0, 2) SMX_CE(1, 3) SMX_CE(4, 6) SMX_CE(5, 7)
    SMX_CE(1, 2) SMX_CE(5, 6) SMX_CE(0, 4) SMX_CE(3, 7)
    SMX_CE(1, 5) SMX_CE(2, 6)
    SMX_CE(1, 4) SMX_CE(3, 6)
    SMX_CE(2, 4) SMX_CE(3, 5)
    SMX_CE(3, 4)
#undef SMX_CE
    Sorted r; r.num = num; r.codes = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.codes |= (uint32_t)code[i] << (4 * i);
    return r;
  }
  // the 3x3 block around (ix,iy): cells[k] / views[k] for neighbour code k (out-of-bounds -> the centre cell,
  // flagged by inb[k] = false), index 8 = centre
  SMX_D void block3x3(int ix, int iy, size_t (&cc)[9], bool (&inb)[8], View (&v)[9]) {
    const size_t c0 = cidx(ix, iy), f0 = fidx(ix, iy);
    size_t ff[9];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int nx = ix + nb_dx(k), ny = iy + nb_dy(k);
      inb[k] = !(nx >= s.dimx || ny >= s.dimy || nx < 0 || ny < 0);
      cc[k] = inb[k] ? cidx(nx, ny) : c0;
      ff[k] = inb[k] ? fidx(nx, ny) : f0;
    }
    cc[8] = c0; ff[8] = f0;
    batch<9>(cc, ff, v);
  }
  SMX_D Sorted sorted_neighbours(int ix, int iy) {          // particle.h:43-60 == water.h:168-185
    size_t cc[9]; bool inb[8]; View v[9];
    block3x3(ix, iy, cc, inb, v);
    double h[8];
#pragma unroll
    for (int k = 0; k < 8; k++) h[k] = inb[k] ? vh(v[k]) : -INFINITY;   // heights are >= 0: out-of-bounds slots sort last
    return sort8(h);
  }

  // ---------------- wet bookkeeping for the grid pass ----------------
  SMX_D void mark_active(size_t c) {
    SMX_OR64_ASYNC(&s.active[c >> 6], 1ull << (c & 63));
    SMX_OR64_ASYNC(&s.active1[c >> 12], 1ull << ((c >> 6) & 63));
    SMX_OR64_ASYNC(&s.active2[c >> 18], 1ull << ((c >> 12) & 63));
  }
  // next set bit >= c in the 3-level bitmap, or ncells (summary levels are conservative)
  SMX_D size_t next_active(size_t c) const {
    const size_t n = (size_t)s.dimx * s.dimy;
    const size_t nw0 = (n + 63) >> 6, nw1 = (nw0 + 63) >> 6, nw2 = (nw1 + 63) >> 6;
    while (c < n) {
      const size_t w0 = c >> 6;
      unsigned long long bits = SMX_LD64_COHERENT(&s.active[w0]) & (~0ull << (c & 63));
      if (bits) { size_t r = (w0 << 6) + (size_t)__builtin_ctzll(bits); return r < n ? r : n; }
      const size_t w = w0 + 1;
      if (w >= nw0) return n;
      const size_t w1 = w >> 6;
      bits = SMX_LD64_COHERENT(&s.active1[w1]) & (~0ull << (w & 63));
      if (bits) { c = ((w1 << 6) + (size_t)__builtin_ctzll(bits)) << 6; continue; }
      size_t v = w1 + 1;
      if (v >= nw1) return n;
      size_t w2 = v >> 6;
      unsigned long long mask = ~0ull << (v & 63);
      bool found = false;
      for (; w2 < nw2; w2++, mask = ~0ull) {
        bits = SMX_LD64_COHERENT(&s.active2[w2]) & mask;
        if (bits) { found = true; break; }
      }
      if (!found) return n;
      c = ((w2 << 6) + (size_t)__builtin_ctzll(bits)) << 12;
    }
    return n;
  }
  SMX_D void on_air_top(int x, int y) {                     // column (x,y) now has a water top
    pol.set_flag(s, cidx(x, y), F_AIR);
    if (grid_mode)
      for (int dx = -1; dx <= 1; dx++)
        for (int dy = -1; dy <= 1; dy++) {
          const int nx = x + dx, ny = y + dy;
          if (nx < 0 || ny < 0 || nx >= s.dimx || ny >= s.dimy) continue;
          mark_active(cidx(nx, ny));
        }
  }
  SMX_D void on_sat(int x, int y) {                         // a saturation in column (x,y) was written
    pol.set_flag(s, cidx(x, y), F_SAT);
    if (grid_mode) mark_active(cidx(x, y));
  }
  SMX_D void sync_air_flag(int x, int y, uint32_t type) {   // after any change of the top section's type
    if (type == AIR) on_air_top(x, y);
    else pol.clear_flag(s, cidx(x, y), F_AIR);
  }

  // ---------------- mutation primitives ----------------
  SMX_D void pop_top(Sec& c) {                              // dat = E->prev; pool.unget(E)
    n_mod++;
    const uint32_t pv = c.prev;
    if (pv == NIL) {
      c.size = 0.0; c.floor = 0.0; c.sat = 0.0; c.type = EMPTY; c.prev = NIL;
    } else {
      c = s.pool[pv];
      pol.node_free(s, pv);
    }
    pol.live_add(s, -1);
  }
  // the tail of Layermap::add once the incoming section is known to be kept apart from a water top:
  // empty column -> place (:243-246); same type -> elongate (:249-253); else push (:302-305)
  SMX_D void place(int x, int y, Sec& c, double size, uint32_t type, double sat) {
    n_mod++;
    if (c.type == EMPTY) {
      c.size = size; c.floor = 0.0; c.sat = sat; c.type = type; c.prev = NIL;
    } else if (c.type == type) {
      c.size += size;
      pol.live_add(s, -1);
      return;
    } else {
      const uint32_t nd = pol.node_alloc(s);
      if (nd == NIL) { n_overflow++; pol.live_add(s, -1); return; }
      const double h = c.floor + c.size;                    // E->floor = height(pos)
      s.pool[nd] = c;
      c.size = size; c.floor = h; c.sat = sat; c.type = type; c.prev = nd;
    }
    sync_air_flag(x, y, type);
    if (sat != 0.0) on_sat(x, y);
  }
  // Layermap::add(pos, pool.get(size, type)) (layermap.h:230-307 + :89-103)
  // `owned`: the caller has already announced this cell to the policy in the current step
  SMX_D void add(int x, int y, double size, uint32_t type, bool owned = false) {
    if (ab) return;
    if constexpr (P::READ_ONLY) { if (!owned) (void)cell(x, y); return; }   // scout: footprint only, never writes
    Sec& c = owned ? s.cells[cidx(x, y)] : cell(x, y);
    if (ab) return;
    // pool.get(): the reference takes a node for EVERY get (even if add() merges and returns it at once),
    // so get() fails exactly when live sections == capacity; add(NULL) is ignored (:233)
    if (!pol.can_get(s)) { n_overflow++; return; }
    pol.live_add(s, +1);
    if (size <= 0) { pol.live_add(s, -1); return; }         // :237-240 unget(E)
    if (c.type != EMPTY && c.type != type && c.type == AIR) {   // :258-275 water stays on top
      n_mod++;
      const double wsize = c.size, wsat = c.sat;
      const uint32_t pv = c.prev;                           // dat = top->prev (the water section is kept aside)
      if (pv == NIL) { c.size = 0.0; c.floor = 0.0; c.sat = 0.0; c.type = EMPTY; c.prev = NIL; }
      else { c = s.pool[pv]; pol.node_free(s, pv); }
      place(x, y, c, size, type, 0.0);                      // add(pos, E)      (E != Air here)
      if (wsize <= 0) { pol.live_add(s, -1); sync_air_flag(x, y, c.type); return; }
      place(x, y, c, wsize, AIR, wsat);                     // add(pos, top)
      return;
    }
    place(x, y, c, size, type, 0.0);
  }
  // Layermap::remove (layermap.h:310-339); *popped tells the caller that the top section went away
  SMX_D double remove(int x, int y, double h, bool* popped = nullptr, bool owned = false) {
    if (popped) *popped = false;
    if (ab) return 0.0;
    if constexpr (P::READ_ONLY) { if (!owned) (void)cell(x, y); return 0.0; }
    Sec& c = owned ? s.cells[cidx(x, y)] : cell(x, y);
    if (ab) return 0.0;
    if (c.type == EMPTY) return 0.0;
    if (c.size <= 0.0) { pop_top(c); sync_air_flag(x, y, c.type); if (popped) *popped = true; return 0.0; }
    if (h <= 0.0) return 0.0;
    const double diff = h - c.size;
    c.size -= h; n_mod++;
    if (diff >= 0.0) { pop_top(c); sync_air_flag(x, y, c.type); if (popped) *popped = true; return diff; }
    return 0.0;
  }

  // ---- Layermap::remove / add on a cell whose view is already in registers (and which is owned). The common
  // cases touch only the `size` field: one 8-byte store, no load. Anything structural (pop, push, water swap,
  // empty column) goes through the generic primitives and the view is re-loaded. Bit-identical by construction:
  // the fast branches are exactly the generic code paths with the loads removed.
  SMX_D double remove_v(int x, int y, size_t c, View& v, double h) {
    if (ab) return 0.0;
    if constexpr (P::READ_ONLY) return 0.0;
    if (v.type != EMPTY && v.size > 0.0) {
      if (h <= 0.0) return 0.0;                             // layermap.h:325
      if (h < v.size) {                                     // :328-337 with diff < 0: shrink, nothing pops
        v.size -= h;
        s.cells[c].size = v.size; n_mod++;
        return 0.0;
      }
    }
    const double r = remove(x, y, h, nullptr, true);
    v = load_view(c);
    return r;
  }
  SMX_D void add_v(int x, int y, size_t c, View& v, double size, uint32_t type) {
    if (ab) return;
    if constexpr (P::READ_ONLY) return;
    if (size <= 0) { if (!pol.can_get(s)) n_overflow++; return; }   // pool.get + unget of an empty section (layermap.h:237-240)
    if (v.type == type && v.type != EMPTY && size > 0 && pol.can_get(s)) {   // layermap.h:249-253 elongate (pool.get + unget: net 0)
      v.size += size;
      s.cells[c].size = v.size; n_mod++;
      return;
    }
    add(x, y, size, type, true);
    v = load_view(c);
  }

  // ---------------- the carried 3x3 block: ONE memory round trip per particle step ----------------
  // A step of either particle needs: the plus-stencil at ipos (normal), the 2x2 block under the new position
  // (bilinear height) and the 3x3 block around npos = round(new position) (cascade). floor(p)+{0,1} is always
  // inside round(p)+{-1,0,1}, and the NEXT step's ipos is this step's npos -- so the 3x3 block around npos,
  // fetched once per step as a single batch (9 stamps + 9 views + the frequency/track words of its centre),
  // serves all three, and is carried into the next step in registers.
  // NOTE: the nine views are NAMED fields, not an array: a select chain over array elements gets folded back into
  // a variably-indexed load by the compiler, which forces the whole block into scratch memory (measured: 38
  // scratch stores per step). Named scalars cannot be indexed, so the block provably stays in registers.
  SMX_D static int code_rel(int dx, int dy) { const int lin = (dx + 1) * 3 + (dy + 1); return lin < 4 ? lin : (lin == 4 ? 8 : lin - 1); }
  SMX_D static View blk_get(const Blk& b, int k) {
    View r = b.v0;
#define X(i)                                                                                   \
    {                                                                                          \
      const bool hit = (k == i);                                                               \
      r.size = hit ? b.v##i.size : r.size;                                                     \
      r.floor = hit ? b.v##i.floor : r.floor;                                                  \
      r.type = hit ? b.v##i.type : r.type;                                                     \
    }
    SMX_FOR9(X)
#undef X
    return r;
  }
  SMX_D static void blk_put(Blk& b, int k, const View& w) {
#define X(i)                                                                                   \
    {                                                                                          \
      const bool hit = (k == i);                                                               \
      b.v##i.size = hit ? w.size : b.v##i.size;                                                \
      b.v##i.floor = hit ? w.floor : b.v##i.floor;                                             \
      b.v##i.type = hit ? w.type : b.v##i.type;                                                \
    }
    SMX_FOR9(X)
#undef X
  }
  template <int MODE>                                         // 0: + water frequency/track, 1: + wind frequency, 2: cells only
  SMX_D void load_block(int cx, int cy, Blk& b) {
    const size_t c0 = cidx(cx, cy), f0 = fidx(cx, cy);
    size_t cc[9], ff[9];
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int nx = cx + nb_dx(k), ny = cy + nb_dy(k);
      const bool in = !(nx >= s.dimx || ny >= s.dimy || nx < 0 || ny < 0);
      m |= in ? (1u << k) : 0u;
      cc[k] = in ? (size_t)((long long)c0 + (long long)nb_dx(k) * s.dimy + nb_dy(k)) : c0;
      ff[k] = in ? (size_t)((long long)f0 + (long long)nb_dy(k) * s.dimx + nb_dx(k)) : f0;
    }
    cc[8] = c0; ff[8] = f0;
    b.c0 = c0; b.inb = m;
    View t[9];
    batch<9>(cc, ff, t);
#define X(i) b.v##i = t[i];
    SMX_FOR9(X)
#undef X
    const size_t fi = fidx(cx, cy);
    if constexpr (MODE == 1) { b.fr = 0.f; b.tr = s.windfreq[fi]; }
    else if constexpr (MODE == 0) { b.fr = s.wfreq[fi]; b.tr = s.wtrack[fi]; }
    else { b.fr = 0.f; b.tr = 0.f; }
  }
  SMX_D f3 normal_blk(int x, int y, const Blk& b) const {   // Layermap::normal(ivec2) layermap.h:341-377
    const int SC = s.scale;
    const bool xm = (b.inb >> 1) & 1, xp = (b.inb >> 6) & 1, ym = (b.inb >> 3) & 1, yp = (b.inb >> 4) & 1;   // (-1,0) (1,0) (0,-1) (0,1)
    const double h0 = vh(b.v8);
    const double hxm = xm ? vh(b.v1) : 0.0, hxp = xp ? vh(b.v6) : 0.0;
    const double hym = ym ? vh(b.v3) : 0.0, hyp = yp ? vh(b.v4) : 0.0;
    f3 n = {0.f, 0.f, 0.f};
    const f3 p = {(float)x, (float)(SC * h0), (float)y};
    const f3 bm = {(float)(x - 1), (float)(SC * hxm), (float)y}, bp = {(float)(x + 1), (float)(SC * hxp), (float)y};
    const f3 cm = {(float)x, (float)(SC * hym), (float)(y - 1)}, cp = {(float)x, (float)(SC * hyp), (float)(y + 1)};
    int k = 0;
    if (xm && ym) { f3 w = cross3(sub3(cm, p), sub3(bm, p)); n.x += w.x; n.y += w.y; n.z += w.z; k++; }
    if (xm && yp) { f3 w = cross3(sub3(cp, p), sub3(bm, p)); n.x -= w.x; n.y -= w.y; n.z -= w.z; k++; }
    if (xp && ym) { f3 w = cross3(sub3(cm, p), sub3(bp, p)); n.x -= w.x; n.y -= w.y; n.z -= w.z; k++; }
    if (xp && yp) { f3 w = cross3(sub3(cp, p), sub3(bp, p)); n.x += w.x; n.y += w.y; n.z += w.z; k++; }
    // k = (xm + xp) * (ym + yp) is 1, 2 or 4 on any map at least two cells wide: dividing by it is exactly a
    // multiplication by its (exact) reciprocal -- three correctly-rounded float divisions less per step
    const float kf = (float)k;
    f3 q;
    if (k == 4 || k == 2 || k == 1) { const float r = (k == 4) ? 0.25f : (k == 2 ? 0.5f : 1.0f); q = {n.x * r, n.y * r, n.z * r}; }
    else q = {n.x / kf, n.y / kf, n.z / kf};
    return normalize3(q);
  }
  SMX_D double bilinear_blk(f2 pos, int cx, int cy, const Blk& b) const {   // Layermap::height(vec2) layermap.h:427-439
    const float fx = floorf(pos.x), fy = floorf(pos.y);
    const int dx = (int)fx - cx, dy = (int)fy - cy;          // in {-1, 0}: the 2x2 block lies inside the 3x3 block
    const float wx = pos.x - fx, wy = pos.y - fy;
    // the four corners are picked from the nine column heights with ten two-way selects (first the column pair by
    // dx, then the row pair by dy) -- a generic blk_get() per corner costs four nine-way select chains per step
    const bool lx = dx < 0, ly = dy < 0;
    const double hmm = vh(b.v0), hm0 = vh(b.v1), hmp = vh(b.v2);   // x-1: y-1, y, y+1
    const double h0m = vh(b.v3), h00c = vh(b.v8), h0p = vh(b.v4);  // x
    const double hpm = vh(b.v5), hp0 = vh(b.v6), hpp = vh(b.v7);   // x+1
    const double am = lx ? hmm : h0m, a0 = lx ? hm0 : h00c, ap = lx ? hmp : h0p;   // column floor(x)
    const double bm = lx ? h0m : hpm, b0 = lx ? h00c : hp0, bp = lx ? h0p : hpp;   // column floor(x)+1
    const double h00 = ly ? am : a0, h01 = ly ? a0 : ap;
    const double h10 = ly ? bm : b0, h11 = ly ? b0 : bp;
    double h = 0.0;
    h += (1.0 - wx) * (1.0 - wy) * h00;
    h += (1.0 - wx) * wy * h10;                             // crossed weights as in the reference (:434-435)
    h += wx * (1.0 - wy) * h01;
    h += wx * wy * h11;                                     // wx*wy is a FLOAT product (:436)
    return h;
  }
  // ---- Particle::cascade (particle.h:24-101) on a block that is already in registers.
  // cascade_precheck: cheap, conservative test -- |dh|*scale/80 is at least 0.01 % below maxdiff for every
  // neighbour, so the exact float excess (particle.h:78) is negative everywhere and the loop would do nothing.
  SMX_D bool cascade_precheck(const Blk& b) const {
    const double sc = (double)(float)s.scale;
    const View& ctr = b.v8;
    const double hC = vh(ctr);
    // One surface type in the whole block (the usual case): every neighbour has the same limit, and because
    // x -> x*sc*0.0125 is monotonic the eight comparisons collapse into one on the largest |dh| -- the same
    // answer as the loop below, bit for bit. (Out-of-bounds slots hold copies of a centre record: |dh| = 0, or, if
    // the centre has changed since, a spurious "maybe" that the exact loop of cascade_loop() then dismisses.)
    bool same = true;
    double dmax = 0.0;
#define X(k) same &= (b.v##k.type == ctr.type); dmax = fmax(dmax, fabs(hC - vh(b.v##k)));
    SMX_FOR8(X)
#undef X
    if (same) {
      const double lim = (double)soils[vsurf(ctr)].maxdiff * 0.9999;
      return !(dmax * sc * 0.0125 < lim);
    }
    bool any = false;
#define X(k)                                                                                   \
    if ((b.inb >> k) & 1) {                                                                    \
      const double dh = hC - vh(b.v##k);                                                       \
      const uint32_t ty = dh > 0 ? vsurf(ctr) : vsurf(b.v##k);                                 \
      const double lim = (double)soils[ty].maxdiff * 0.9999;                                   \
      if (!(fabs(dh) * sc * 0.0125 < lim)) any = true;                                         \
    }
    SMX_FOR8(X)
#undef X
    return any;
  }
  SMX_D Sorted cascade_sort(const Blk& b) const {           // particle.h:43-60
    double hs[8];
#define X(k) hs[k] = ((b.inb >> k) & 1) ? vh(b.v##k) : -INFINITY;
    SMX_FOR8(X)
#undef X
    return sort8(hs);
  }
  // Which neighbours can possibly move material against the CURRENT centre: bit k is clear only if neighbour k is
  // provably below its limit (the conservative test of cascade_precheck, per neighbour), i.e. the loop body would
  // `continue` on it (particle.h:76-79) without touching anything.
  SMX_D uint32_t cascade_candidates(const Blk& b) const {
    const double sc = (double)(float)s.scale;
    const View& ctr = b.v8;
    const double hC = vh(ctr);
    bool same = true;
#define X(k) same &= (b.v##k.type == ctr.type);
    SMX_FOR8(X)
#undef X
    uint32_t m = 0;
    if (same) {
      const double lim = (double)soils[vsurf(ctr)].maxdiff * 0.9999;
#define X(k) m |= (!(fabs(hC - vh(b.v##k)) * sc * 0.0125 < lim)) ? (1u << k) : 0u;
      SMX_FOR8(X)
#undef X
      return m & b.inb;
    }
#define X(k)                                                                                   \
    if ((b.inb >> k) & 1) {                                                                    \
      const double dh = hC - vh(b.v##k);                                                       \
      const uint32_t ty = dh > 0 ? vsurf(ctr) : vsurf(b.v##k);                                 \
      const double lim = (double)soils[ty].maxdiff * 0.9999;                                   \
      if (!(fabs(dh) * sc * 0.0125 < lim)) m |= 1u << k;                                       \
    }
    SMX_FOR8(X)
#undef X
    return m;
  }
  // one neighbour of the loop (particle.h:62-99); the block is kept exact (views written back).
  // 0: nothing moved, 1: material moved, 2: moved and a section was exhausted (the caller may recurse there)
  SMX_D int cascade_visit(int ix, int iy, Blk& b, View& ctr, int k, bool& moved) {
    const double sc = (double)(float)s.scale, d80 = (double)80.0f;
    View nb = blk_get(b, k);
    const int nx = ix + nb_dx(k), ny = iy + nb_dy(k);
    const size_t cn = cidx(nx, ny);
    const float diff = (float)((vh(ctr) - vh(nb)) * sc / d80);
    if (diff == 0) return 0;
    const bool down = diff > 0;                             // the centre is the higher column
    const View& T = down ? ctr : nb;
    const uint32_t type = vsurf(T);
    const float excess = fabsf(diff) - soils[type].maxdiff;
    if (excess <= 0) return 0;
    float transfer = soils[type].settling * excess / 2.0f;
    const double tsize = T.size;
    if ((double)transfer > tsize) transfer = (float)tsize;
    const uint32_t casc = soils[type].cascades;
    bool recascade = false;
    if (down) { if (remove_v(ix, iy, b.c0, ctr, (double)transfer) != 0) recascade = true; add_v(nx, ny, cn, nb, (double)transfer, casc); }
    else { if (remove_v(nx, ny, cn, nb, (double)transfer) != 0) recascade = true; add_v(ix, iy, b.c0, ctr, (double)transfer, casc); }
    n_casc_tr++;
    moved = true;
    blk_put(b, k, nb);
    b.v8 = ctr;
    return recascade ? 2 : 1;
  }
  // the neighbour loop from sorted index i on. Neighbours that are provably below their limit are skipped without
  // being looked at (the candidate mask is refreshed whenever the centre has changed).
  // Stops early (returning the neighbour code) when a section was exhausted and the caller may recurse there.
  SMX_D int cascade_loop(int ix, int iy, Blk& b, const Sorted& sn, int& i, bool may_recurse, bool& moved) {
    View ctr = b.v8;
    uint32_t cand = cascade_candidates(b);
    for (; i < sn.num; i++) {
      const int k = (sn.codes >> (4 * i)) & 15;
      if (!((cand >> k) & 1)) continue;
      const int r = cascade_visit(ix, iy, b, ctr, k, moved);
      if (ab) return -1;
      if (r) cand = cascade_candidates(b);
      if (r == 2 && may_recurse) { i++; return k; }
    }
    return -1;
  }
  // Particle::cascade(pos, .., 0) on a loaded block -- the water step's cascade. No sort: the visiting order
  // (entry-time height descending, neighbour code ascending: particle.h:43-60, a stable sort of the code-ordered
  // list) only matters among the neighbours that can move anything, so the loop repeatedly picks, among the
  // current candidates, the first one AFTER the neighbour visited last. Typical cost: one visit, two mask scans.
  SMX_D void cascade_blk(int ix, int iy, Blk& b) {
    n_casc++;
    if (!cascade_precheck(b)) return;
#define X(k) const double e##k = vh(b.v##k);              /* entry-time heights */
    SMX_FOR8(X)
#undef X
    double lh = INFINITY; int lc = -1;                       // the neighbour visited last
    View ctr = b.v8;
    bool moved = false;
    while (true) {
      uint32_t cand = cascade_candidates(b);
      if (!cand) return;
      double bh = -INFINITY; int bk = -1;
#define X(k) if (((cand >> k) & 1) && (e##k < lh || (e##k == lh && k > lc)) && e##k > bh) { bh = e##k; bk = k; }
      SMX_FOR8(X)
#undef X
      if (bk < 0) return;
      lh = bh; lc = bk;
      cascade_visit(ix, iy, b, ctr, bk, moved);
      if (ab) return;
    }
  }
  // Particle::cascade(pos, .., LOOP) (particle.h:24-101), LOOP = recursion budget (transferloop): water 0, wind 1
  // (wind.h:112,126,129). The one level of recursion is an explicit parent frame, so the neighbour loop exists
  // once in the code. Returns true if any material moved.
  template <int LOOP>
  SMX_D bool particle_cascade(f2 pos) {
    static_assert(LOOP <= 1, "one level of recursion is all the reference uses");
    int ix = (int)roundf(pos.x), iy = (int)roundf(pos.y);
    int budget = LOOP;
    bool moved = false, in_child = false;
    int pix = 0, piy = 0, pi = 0; Sorted psn; psn.num = 0; psn.codes = 0;
    Blk b;
    Sorted sn; int i = 0;
    bool enter = true;
    while (true) {
      if (enter) {
        n_casc++;
        load_block<2>(ix, iy, b);
        if (ab) return moved;
        i = 0; sn.num = 0; sn.codes = 0;
        if (cascade_precheck(b)) sn = cascade_sort(b);
      }
      const int k = cascade_loop(ix, iy, b, sn, i, budget > 0 && !in_child, moved);
      if (ab) return moved;
      if (k >= 0) {                                         // cascade(npos, .., --transferloop) (particle.h:96-97)
        --budget;
        pix = ix; piy = iy; pi = i; psn = sn; in_child = true;
        ix = ix + nb_dx(k); iy = iy + nb_dy(k);
        enter = true;
        continue;
      }
      if (!in_child) return moved;
      in_child = false;                                     // back in the parent: its block may have changed
      ix = pix; iy = piy; i = pi; sn = psn;
      load_block<2>(ix, iy, b);
      if (ab) return moved;
      enter = false;
    }
  }

  // Particle::cascade(pos, .., 1) -- the wind step's cascades (wind.h:112,126,129) -- on a block held in registers,
  // in the sort-free form of cascade_blk. `o` is the other carried block, around (ox, oy): every column changed
  // here is patched there too, so both stay exact images of the map. The one level of recursion the budget
  // allows (a section was exhausted: particle.h:96-97) runs on the map itself (particle_cascade<0>); the parent
  // then re-reads its block and goes on. Returns false if that happened: `b` is still exact, `o` no longer is.
  SMX_D bool cascade_blk1(int ix, int iy, Blk& b, int ox, int oy, Blk& o, bool& moved) {
    n_casc++;
#define X(k) const double e##k = vh(b.v##k);              /* entry-time heights */
    SMX_FOR8(X)
#undef X
    double lh = INFINITY; int lc = -1;                       // the neighbour visited last
    View ctr = b.v8;
    int budget = 1;
    bool o_exact = true;
    while (true) {
      const uint32_t cand = cascade_candidates(b);
      if (!cand) return o_exact;
      double bh = -INFINITY; int bk = -1;
#define X(k) if (((cand >> k) & 1) && (e##k < lh || (e##k == lh && k > lc)) && e##k > bh) { bh = e##k; bk = k; }
      SMX_FOR8(X)
#undef X
      if (bk < 0) return o_exact;
      lh = bh; lc = bk;
      const int r = cascade_visit(ix, iy, b, ctr, bk, moved);
      if (ab) return false;
      if (r) {                                               // the two columns that changed, seen from the other block
        const int kx = ix + nb_dx(bk), ky = iy + nb_dy(bk);
        const int cdx = ix - ox, cdy = iy - oy, ndx = kx - ox, ndy = ky - oy;
        if (cdx >= -1 && cdx <= 1 && cdy >= -1 && cdy <= 1) blk_put(o, code_rel(cdx, cdy), b.v8);
        if (ndx >= -1 && ndx <= 1 && ndy >= -1 && ndy <= 1) blk_put(o, code_rel(ndx, ndy), blk_get(b, bk));
      }
      if (r == 2 && budget > 0) {                            // cascade(npos, .., --transferloop)
        --budget;
        const f2 cp = {(float)(ix + nb_dx(bk)), (float)(iy + nb_dy(bk))};
        const float fr = b.fr, tr = b.tr;
        if constexpr (outline_recursion<P>::value) {
          RecRes<P> r = wind_recurse<P>(s, soils, pol, cp.x, cp.y, ix, iy);
          pol.adopt(r.pol);
          moved |= r.moved;
          if (r.ab) { ab = true; return false; }
          b = r.b;                                           // back in the parent: its block may have changed
        } else {
          moved |= particle_cascade<0>(cp);
          if (ab) return false;
          load_block<2>(ix, iy, b);                          // back in the parent: its block may have changed
        }
        b.fr = fr; b.tr = tr;
        if (ab) return false;
        ctr = b.v8;
        o_exact = false;
      }
    }
  }

  // ---------------- WaterParticle water.h:9-368 ----------------
  SMX_D void water_init(Water& p, int rx, int ry) {          // water.h:11-19 (caller supplies the two draws)
    p.pos = {(float)rx, (float)ry};
    p.speed = {0.f, 0.f};
    p.volume = 1.0; p.sediment = 0.0; p.evaprate = 0.001; p.spill = 3;
    p.ix = (int)roundf(p.pos.x); p.iy = (int)roundf(p.pos.y);
    p.surface = surface(p.ix, p.iy);
    p.friction = soils[p.surface].friction;
    p.contains = soils[p.surface].transports;
  }
  SMX_D void seep(int ix, int iy) {                          // WaterParticle::seep(vec2,..) water.h:285-333
    Sec& top = cell(ix, iy);
    if (ab) return;
    if (top.type == EMPTY) return;
    // walk top -> bottom; `t`/`pv` are either the inline cell record or pool nodes
    Sec* t = &top;
    while (t->prev != NIL) {
      const uint32_t pvi = t->prev;
      Sec* pv = &s.pool[pvi];
      const SoilP& param = soils[t->type];
      const SoilP& nparam = soils[pv->type];
      const double vol = t->size * t->sat * param.porosity;
      const double nevol = pv->size * (1.0 - pv->sat) * nparam.porosity;
      const double seepage = 1.0;
      const double transfer = (vol < nevol) ? vol : nevol;
      if (transfer > 0) {
        pol.pre_write_node(s, pvi);
        if (t != &top) pol.pre_write_node(s, (uint32_t)(t - s.pool));
        if (t->type == AIR) {
          // map.remove(ipos, ..) acts on the COLUMN top; if it pops, the node below the top moves into
          // the inline cell record: redirect pv when that node is the one we hold.
          const uint32_t below_top = top.prev;
          bool popped = false;
          remove(ix, iy, seepage * transfer, &popped);
          if (popped && below_top == pvi) pv = &top;
        } else {
          t->sat -= (seepage * transfer) / (t->size * param.porosity);
        }
        pv->sat += (seepage * transfer) / (pv->size * nparam.porosity);
        on_sat(ix, iy);
      }
      t = pv;
    }
  }
  // water.h:123-145 up to (and excluding) the WaterParticle::cascade call; true = cascade must follow
  SMX_D bool water_flood_prelude(Water& p) {
    if (p.volume < 0.01 || p.spill-- <= 0) return false;
    n_floods++;
    p.ix = (int)p.pos.x; p.iy = (int)p.pos.y;               // truncation (:128)
    { SMX_SECT_T0(); add(p.ix, p.iy, p.sediment * soils[p.contains].equrate, p.contains); SMX_SECT(0); }
    { SMX_SECT_T0(); particle_cascade<0>(p.pos); SMX_SECT(1); }
    { SMX_SECT_T0(); add(p.ix, p.iy, p.volume * 0.015, AIR); SMX_SECT(2); }                 // volumeFactor = 0.015 (:370)
    { SMX_SECT_T0(); seep(p.ix, p.iy); SMX_SECT(3); }
    return !ab;
  }
  SMX_D void push_frame(Frame* st, int& depth, int ix, int iy, int spill) {   // entry of WaterParticle::cascade
    SMX_SECT_T0();
    n_wcasc++;
    size_t cc[9]; bool inb[8]; View v[9];
    block3x3(ix, iy, cc, inb, v);
    double h[8];
#pragma unroll
    for (int k = 0; k < 8; k++) h[k] = inb[k] ? vh(v[k]) : -INFINITY;   // particle.h:43-60 == water.h:168-185
    // Which neighbours can the loop of water.h:187-281 do anything with, on the map as it is NOW? It skips a neighbour whose water table equals
    // the centre's (:216-218), whose higher column has no water on top (:229-230) or no water left (:240-241). While this Sim has not written
    // the map since (n_mod stands still -- and nobody else can: exclusive policies only), water_drive skips those neighbours without loading them
    // again: the same `continue`, two dependent memory round trips less per neighbour (a lake cell runs ~40 of them).
    uint32_t cand = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (!inb[k]) continue;
      const double diff = (v[8].floor + v[8].size - v[k].floor - v[k].size) * (double)s.scale / 80.0;   // the expression of water_drive, operand by operand
      const View& T = diff > 0 ? v[8] : v[k];
      if (diff != 0 && T.type == AIR && T.size > 0) cand |= 1u << k;
    }
    const Sorted sn = sort8(h);
    if (depth >= MAX_FRAMES) { n_overflow++; return; }      // unreachable: depth is bounded by the spill budget
    Frame& f = st[depth++];
    f.ix = ix; f.iy = iy; f.spill = spill; f.num = sn.num; f.i = 0; f.codes = sn.codes; f.cand = cand; f.stamp = n_mod;
    SMX_SECT(4);
  }
  // WaterParticle::move (water.h:43-73) on the carried block `cur` around ipos. false = move() returned false.
  // PURE: the batched engine's look-ahead -- where would this move go? No track write, no counters.
  template <bool PURE = false>
  SMX_D bool water_move(Water& p, bool toplevel, const Blk& cur) {
    const View cv = cur.v8;
    const f3 n = normal_blk(p.ix, p.iy, cur);
    p.surface = vsurf(cv);
    p.evaprate = 0.01;
    const size_t fi = fidx(p.ix, p.iy);
    const float fr = cur.fr;
    if constexpr (!P::READ_ONLY && !PURE)
      s.wtrack[fi] = (float)((double)cur.tr + p.volume);                 // updatefrequency :348-351
    p.friction = soils[p.surface].friction * (1.0f - fr);
    p.evaprate = p.evaprate * (double)(1.0f - 0.2f * fr);
    const f2 nxz = {n.x, n.z};
    const f2 scaled = {nxz.x * p.friction, nxz.y * p.friction};
    if ((double)sqrtf(dot2(scaled, scaled)) < 1E-5) return false;
    const float a = p.friction;                                          // mix(nxz, speed, friction)
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
    if constexpr (!PURE) {
      n_steps_w_all++;
      if (toplevel) n_steps_w_top++;
    }
    return true;
  }
  // the side effects of a move() whose arithmetic was already done by water_move<true> on a copy (batched engine):
  // updatefrequency (water.h:53, :348-351; `before` = the particle as it entered move()) and the step counters
  SMX_D void water_move_commit(const Water& before, bool moved, bool toplevel, const Blk& cur) {
    s.wtrack[fidx(before.ix, before.iy)] = (float)((double)cur.tr + before.volume);
    if (moved) { n_steps_w_all++; if (toplevel) n_steps_w_top++; }
  }
  SMX_D void wind_move_commit(bool moved, const Blk& cur) {   // wind.h:49-52 on the block's centre
    s.windfreq[cur_fidx(cur)] = (float)(0.5 * cur.tr + 0.5f);
    if (moved) n_steps_wind++;
  }
  SMX_D size_t cur_fidx(const Blk& b) const {                // frequency-order index of a block's centre cell
    const size_t x = b.c0 / (size_t)s.dimy, y = b.c0 - x * (size_t)s.dimy;
    return y * (size_t)s.dimx + x;
  }
  // WaterParticle::interact (water.h:75-121): `cur` is the block around ipos, `nb` receives the block around the
  // new position. 1 = interact() returned true, 0 = false, -1 = the new block could not be acquired (speculative
  // engine): nothing of interact() has happened yet, so the particle can be resumed exactly here.
  // `loaded`: the caller has already fetched `nb` (the wave-cooperative loop of soil_coop.h)
  SMX_D int water_interact(Water& p, const Blk& cur, Blk& nb, bool loaded = false) {
    View cv = cur.v8;
    const float fr = cur.fr;
    const int nx = (int)roundf(p.pos.x), ny = (int)roundf(p.pos.y);
    if (!loaded) {
      load_block<0>(nx, ny, nb);                             // THE memory round trip of this step
      if (ab) return -1;
    }
    const SoilP& param = soils[p.surface];
    const double hb = bilinear_blk(p.pos, nx, ny, nb);
    double c_eq = param.solubility * (vh(cv) - hb) * (double)s.scale / 80.0;
    if (c_eq < 0.0) c_eq = 0.0;
    if (c_eq > 1.0) c_eq = 1.0;
    if ((double)(soils[p.contains].erosionrate) < fr) p.contains = soils[p.contains].erodes;
    const double cdiff = c_eq - p.sediment;
    const size_t cc = cur.c0;
    if (cdiff > 0) {
      p.sediment += param.equrate * cdiff;
      p.contains = soils[vsurf(cv)].transports;              // soils[map.surface(ipos)].transports (:94)
      double diff = remove_v(p.ix, p.iy, cc, cv, param.equrate * cdiff * p.volume);
      while (fabs(diff) > 1E-8) diff = remove_v(p.ix, p.iy, cc, cv, diff);
    } else if (cdiff < 0) {
      p.sediment += soils[p.contains].equrate * cdiff;
      add_v(p.ix, p.iy, cc, cv, -soils[p.contains].equrate * cdiff * p.volume, p.contains);
    }
    if (ab) return 0;
    const int rdx = p.ix - nx, rdy = p.iy - ny;              // the column just eroded may lie inside the new block
    if (rdx >= -1 && rdx <= 1 && rdy >= -1 && rdy <= 1) blk_put(nb, code_rel(rdx, rdy), cv);
    cascade_blk(nx, ny, nb);                                 // Particle::cascade(pos, .., 0) (:113)
    if (ab) return 0;
    p.sediment /= (1.0 - p.evaprate);
    if (p.sediment > 1.0) p.sediment = 1.0;
    p.volume *= (1.0 - p.evaprate);
    p.ix = nx; p.iy = ny;                                    // (the reference recomputes round(pos) in the next move())
    return p.volume > 0.01 ? 1 : 0;
  }
  // while(particle.move(..) && particle.interact(..)); (SoilMachine.cpp:293, water.h:259); the two blocks ping-pong.
  // resume = true: continue a suspended particle at its interact() (the block around ipos comes from the checkpoint)
  SMX_D void water_loop(Water& p, bool toplevel, bool resume = false) {
    Blk A, B;
    if (resume) pol.load_blk(A);
    else {
      p.ix = (int)roundf(p.pos.x); p.iy = (int)roundf(p.pos.y);
      load_block<0>(p.ix, p.iy, A);
      if (ab) return;
    }
    while (true) {
      if (!resume && !water_move(p, toplevel, A)) return;
      resume = false;
      int r = water_interact(p, A, B);
      if (r <= 0) { if (r < 0 && susp_ok) { susp = 2; pol.save_blk(A); } return; }
      if (!water_move(p, toplevel, B)) return;
      r = water_interact(p, B, A);
      if (r <= 0) { if (r < 0 && susp_ok) { susp = 2; pol.save_blk(B); } return; }
    }
  }
  // WindParticle::move (wind.h:54-92) on the carried block; false = move() returned false
  template <bool PURE = false>
  SMX_D bool wind_move(Wind& p, const Blk& cur) {
    const f3 pspeed = {-2.f, 0.f, 1.f};
    const double gravity = 0.25, winddominance = 0.2, windfriction = 0.8;
    const View cv = cur.v8;
    const f3 n = normal_blk(p.ix, p.iy, cur);
    p.surface = vsurf(cv);
    const size_t fi = fidx(p.ix, p.iy);
    if constexpr (!P::READ_ONLY && !PURE)
      s.windfreq[fi] = (float)(0.5 * cur.tr + 0.5f);                     // wind.h:49-52
    p.sheight = vh(cv) * (float)s.scale / 80.0f;
    if (p.height < p.sheight) p.height = p.sheight;
    if (p.height > p.sheight) {
      p.speed.y = (float)(p.speed.y - gravity);
    } else {
      const f3 c = cross3(cross3(p.speed, n), n);
      p.speed = {mixd(p.speed.x, c.x, windfriction), mixd(p.speed.y, c.y, windfriction), mixd(p.speed.z, c.z, windfriction)};
    }
    p.speed = {mixd(p.speed.x, pspeed.x, winddominance), mixd(p.speed.y, pspeed.y, winddominance),
               mixd(p.speed.z, pspeed.z, winddominance)};
    p.pos.x += p.speed.x; p.pos.y += p.speed.z;
    p.height += p.speed.y;
    if (!(p.pos.x >= 0.0f && p.pos.y >= 0.0f) || !((int)p.pos.x < s.dimx - 1 && (int)p.pos.y < s.dimy - 1)) return false;
    if (sqrtf(dot3(p.speed, p.speed)) < 0.01) return false;
    if constexpr (!PURE) n_steps_wind++;
    return true;
  }
  // WindParticle::interact (wind.h:94-136). 1 = go on (interact() always returns true), 0 = aborted somewhere
  // inside (speculative engine), -1 = the new block could not be acquired: resumable exactly here.
  SMX_D int wind_interact(Wind& p, Blk& cur, Blk& nb, bool loaded = false) {
    View cv = cur.v8;
    const int nx = (int)roundf(p.pos.x), ny = (int)roundf(p.pos.y);
    if (!loaded) {
      load_block<1>(nx, ny, nb);
      if (ab) return -1;
    }
    const SoilP& param = soils[p.surface];
    const double hb = bilinear_blk(p.pos, nx, ny, nb);
    View nv = nb.v8;
    int ncasc = 0;                                           // Particle::cascade(.., 1) calls that follow (one code site)
    bool changed = false;
    if (p.height <= hb * (float)s.scale / 80.0f) {
      if (param.transports == p.contains) {
        const double force = sqrtf(dot3(p.speed, p.speed)) * (vh(nv) - p.height) * (float)s.scale / 80.0f * (1.0f - p.sediment);
        const double amount = param.suspension * force;
        const double diff = remove_v(p.ix, p.iy, cur.c0, cv, amount);
        p.sediment += (amount - diff);
        changed = amount > 0.0;
        ncasc = 1;                                           // cascade(ipos)
      }
    } else if (param.suspension > 0.0) {
      p.sediment -= soils[p.contains].suspension * p.sediment;
      const double dep = 0.5f * soils[p.contains].suspension * p.sediment;
      add_v(nx, ny, nb.c0, nv, dep, p.contains);
      if (nb.c0 == cur.c0) cv = nv;                          // npos == ipos: the second add sees the first
      add_v(p.ix, p.iy, cur.c0, cv, dep, p.contains);
      changed = dep > 0.0;
      ncasc = 2;                                             // cascade(ipos); cascade(npos)
    }
    // The two columns just changed lie in both carried blocks: patch the register copies so that they stay exact
    // images of the map. The usual outcome of the cascades that follow -- a grain of dust moves nothing -- is then
    // decided on registers, and the step needs no memory round trip beyond the block load it started with.
    const int rdx = p.ix - nx, rdy = p.iy - ny;              // ipos seen from npos
    const bool near = rdx >= -1 && rdx <= 1 && rdy >= -1 && rdy <= 1;
    if (changed) {
      cur.v8 = cv;
      if (ncasc == 2 && !(rdx == 0 && rdy == 0)) {
        nb.v8 = nv;
        if (near) blk_put(cur, code_rel(-rdx, -rdy), nv);
      }
      if (near) blk_put(nb, code_rel(rdx, rdy), cv);        // (npos == ipos: the centre of nb)
    }
    bool nb_exact = true, mv = false;                        // nb is an exact image of the map around npos
    if (ncasc >= 1) {                                        // cascade(ipos)
      if (cascade_precheck(cur)) nb_exact = cascade_blk1(p.ix, p.iy, cur, nx, ny, nb, mv); else n_casc++;
      if (ab) return 0;
    }
    if (ncasc == 2) {                                        // cascade(npos)
      if (!nb_exact) { load_block<1>(nx, ny, nb); if (ab) return 0; nb_exact = true; }
      if (cascade_precheck(nb)) cascade_blk1(nx, ny, nb, p.ix, p.iy, cur, mv); else n_casc++;
      if (ab) return 0;
    }
    p.ix = nx; p.iy = ny;
    if (!nb_exact) load_block<1>(nx, ny, nb);             // refresh the views
    return ab ? 0 : 1;
  }
  SMX_D void wind_loop(Wind& p, bool resume = false) {       // SoilMachine.cpp:306
    Blk A, B;
    if (resume) pol.load_blk(A);
    else {
      if (soils[p.contains].suspension == 0.0) return;       // wind.h:56-57 (contains never changes: checked once)
      p.ix = (int)roundf(p.pos.x); p.iy = (int)roundf(p.pos.y);
      load_block<1>(p.ix, p.iy, A);
      if (ab) return;
    }
    while (true) {
      if (!resume && !wind_move(p, A)) return;
      resume = false;
      int r = wind_interact(p, A, B);
      if (r <= 0) { if (r < 0 && susp_ok) { susp = 2; pol.save_blk(A); } return; }
      if (!wind_move(p, B)) return;
      r = wind_interact(p, B, A);
      if (r <= 0) { if (r < 0 && susp_ok) { susp = 2; pol.save_blk(B); } return; }
    }
  }
  // Runs one WaterParticle to completion INCLUDING everything it spawns (SoilMachine.cpp:292-296, water.h:258-262),
  // then drains the cascade frame stack. `start_with_particle` = false: only drain the frames already pushed.
  // `loop_done`: the particle's move/interact loop has already been run by the caller (soil_coop.h)
  SMX_D void water_drive(Water& p, bool toplevel, bool start_with_particle, Frame* st, int& depth, bool resume = false,
                         bool loop_done = false) {
    bool have = start_with_particle;
    // The TOP frame lives in registers (`f`); the stack array -- dynamically indexed, hence scratch memory on the device -- is written when a frame is
    // pushed over it and read when a frame is popped back to, not on every field access of every neighbour iteration (round 6: ~4 dependent scratch
    // loads per iteration before). `top` = st[depth - 1] is cached in f; every push_frame below is preceded by a write-back.
    Frame f; bool top = false;
    f.ix = f.iy = f.spill = f.num = f.i = 0; f.codes = f.cand = f.stamp = 0u;
#define SMX_FRAME_WRITE_BACK() do { if (top) { st[depth - 1] = f; top = false; } } while (0)
    while (true) {
      if constexpr (P::DEFER_NESTED) {
        // throughput engines: a nested particle (spawned below) runs here, at once, inside the cells this thread owns; what does not
        // fit -- a step that leaves the region, a flood whose reach does -- is suspended into a record (soil_batch.h run_nested)
        if (have) {
          have = false;
          int rn_; { SMX_SECT_T0(); rn_ = pol.run_nested(*this, p); SMX_SECT(8); }
          if (rn_ == 0 && !(p.volume < 0.01 || p.spill <= 0) && pol.flood_fits(*this, p)) {   // water.h:125 gate, then :123-145
            if (water_flood_prelude(p)) { SMX_FRAME_WRITE_BACK(); push_frame(st, depth, p.ix, p.iy, p.spill); }
          }
        }
      } else
      if (have) {
        if (!loop_done) water_loop(p, toplevel, resume);    // while(move && interact);
        loop_done = false;
        resume = false;
        if (ab) return;
        if constexpr (P::READ_ONLY) {                       // scout: claim where a flood would act, then stop
          const int fx = (int)p.pos.x, fy = (int)p.pos.y;
          for (int dx = -2; dx <= 2; dx++)
            for (int dy = -2; dy <= 2; dy++) {
              const int nx = fx + dx, ny = fy + dy;
              if (nx < 0 || ny < 0 || nx >= s.dimx || ny >= s.dimy) continue;
              (void)cell(nx, ny);
            }
          return;
        }
        if (water_flood_prelude(p)) { SMX_FRAME_WRITE_BACK(); push_frame(st, depth, p.ix, p.iy, p.spill); }   // :140
        if (ab) return;
        have = false;
      }
      if (depth == 0) return;
      if (!top) { f = st[depth - 1]; top = true; }
      if (f.i >= f.num) { depth--; top = false; continue; }
      const int k = (f.codes >> (4 * f.i)) & 15;
      f.i++;
      if constexpr (exclusive_access<P>::value) { if (f.stamp == n_mod && !((f.cand >> k) & 1u)) continue; }   // (push_frame: nothing to do there, decided on registers)
      const int ix = f.ix, iy = f.iy;
      const int nx = ix + nb_dx(k), ny = iy + nb_dy(k);
      SMX_SECT_T0();
      const Sec& A = cell(ix, iy);
      const Sec& B = cell(nx, ny);
      if (ab) return;
      const double whA = A.size, whB = B.size, fA = A.floor, fB = B.floor;   // 0 for empty columns (:194-214)
      const uint32_t tyA = A.type, tyB = B.type;            // (with the sizes: ONE memory round trip for everything this neighbour's decision reads)
      const double diff = (fA + whA - fB - whB) * (double)s.scale / 80.0;
      if (diff == 0) { SMX_SECT(5); continue; }
      const int tx = (diff > 0) ? ix : nx, ty = (diff > 0) ? iy : ny;
      const int bx = (diff > 0) ? nx : ix, by = (diff > 0) ? ny : iy;
      if (((diff > 0) ? tyA : tyB) != AIR) { SMX_SECT(5); continue; }   // :229 (EMPTY != AIR; the reference cannot reach here with NULL)
      double transfer = fabs(diff) / 2.0;
      const double wh = (diff > 0) ? whA : whB;
      transfer = (wh < transfer) ? wh : transfer;
      if (transfer <= 0) { SMX_SECT(5); continue; }
      if (transfer == wh) {                                 // :246-264 the whole water section leaves: nested particle
        if constexpr (P::DEFER_NESTED) {
          if (!pol.child_room()) continue;                  // no record left for it, should it have to be suspended: the water stays where it is
        }
        remove(tx, ty, transfer);
        n_nested++;
        int ry, rx;
        if constexpr (P::DEFER_NESTED) {
          // throughput engines: `contains` from the cell the particle starts on instead of a random one (water.h:251 via :15-17). The
          // value never reaches the map: sediment is 0 until the first erosion (water.h:91-95), which overwrites `contains` in the same
          // statement block; a deposit (:103-107) and the flood's soil (:132) need sediment > 0. Reading a random, far cell would race
          // with whoever owns it.
          rx = tx; ry = ty;
        } else {
          pol.rand2(s, ry, rx);                             // 1st draw -> y, 2nd -> x (g++ evaluation order)
          ry %= s.dimy; rx %= s.dimx;
        }
        water_init(p, rx, ry);
        const f2 d = {(float)bx - (float)tx, (float)by - (float)ty};
        const f2 dn = normalize2(d);
        const float s2 = sqrtf(2.0f);
        p.speed = {s2 * dn.x, s2 * dn.y};
        p.pos = {(float)tx, (float)ty};
        p.spill = f.spill;
        p.volume = transfer / 0.015;
        toplevel = false;
        susp_ok = false;                                   // nested particles run inside their parent: not resumable
        have = true;
        SMX_SECT(7);
        if (ab) return;
      } else {                                              // :266-276
        bool recascade = false;
        bool fast = false;
        if constexpr (exclusive_access<P>::value) {
          // Water onto water -- a lake levelling out, 35 of these per active cell of the grid pass in the wet regime (profiles/r06_grid_tiles.md): remove() only
          // shrinks the higher column's top (0 < transfer < its size: nothing pops, :328-337), add() only elongates the other's (same type: pool.get + unget,
          // :249-253; it needs a section to be available all the same), then sat = 1 (:273-276). The generic calls below do exactly this after loading both
          // records again, three dependent round trips; here the sizes are in registers already. Same operations on the same operands: bit-identical.
          if (((diff > 0) ? tyB : tyA) == AIR && !ab && pol.can_get(s)) {
            Sec& Tw = s.cells[cidx(tx, ty)];
            Sec& Rw = s.cells[cidx(bx, by)];
            Tw.size = wh - transfer;                        // remove(): c.size -= h
            Rw.size = ((diff > 0) ? whB : whA) + transfer;  // add() -> place(): c.size += size
            n_mod += 2;
            recascade = true;                               // (transfer > 0)
            Rw.sat = 1.0f;
            pol.set_flag_async(s, cidx(bx, by), F_SAT); if (grid_mode) mark_active(cidx(bx, by));   // on_sat()
            fast = true;
          }
        }
        if (!fast) {
          if (remove(tx, ty, transfer) != 0) recascade = true;
          if (transfer > 0) recascade = true;
          add(bx, by, transfer, AIR);
          Sec& R = cell(bx, by);
          if (ab) return;
          if (R.type != EMPTY) { R.sat = 1.0f; on_sat(bx, by); }
        }
        SMX_SECT(6);
        if (recascade && f.spill > 0) {                     // :278-279 cascade(npos, --spill)
          const int sp = --f.spill;
          SMX_FRAME_WRITE_BACK();
          push_frame(st, depth, nx, ny, sp);
        }
        if (ab) return;
      }
    }
#undef SMX_FRAME_WRITE_BACK
  }

  // ---------------- WindParticle wind.h:11-138 ----------------
  SMX_D void wind_init(Wind& p, int rx, int ry) {            // wind.h:13-22
    p.pos = {(float)rx, (float)ry};
    p.speed = {-2.f, 0.f, 1.f};
    p.sediment = 0.0; p.height = 0.0; p.sheight = 0.0;
    p.ix = (int)roundf(p.pos.x); p.iy = (int)roundf(p.pos.y);
    p.surface = surface(p.ix, p.iy);
    p.contains = soils[p.surface].transports;
  }
  SMX_D void wind_run(Wind& p, bool resume = false) { wind_loop(p, resume); }   // SoilMachine.cpp:305-306
};

template <class P>
SMX_NOINLINE RecRes<P> wind_recurse(DevState s, const SoilP* soils, P pol, float cx, float cy, int ix, int iy) {
  RecRes<P> r{pol, Blk{}, false, false};
  Sim<P> sim(s, soils, r.pol);
  r.moved = sim.template particle_cascade<0>(f2{cx, cy});
  if (!sim.ab) sim.template load_block<2>(ix, iy, r.b);
  r.ab = sim.ab;
  sim.flush_counters();
  return r;
}

}  // namespace smx
