// soil_batch.h -- the BATCHED (throughput) engine: every particle of a phase advances concurrently, one particle per
// LANE, one step per epoch. This is the north-star's "one wavefront per particle-batch" path; it does NOT reproduce
// the reference's particle-after-particle order (SoilMachine.cpp:287-307) and is never claimed bit-exact against the
// reference (DESIGN.md "Batched engine", parity tier P2). It IS deterministic, and bit-exact against its own CPU
// restatement (oracle/soil_oracle.cpp so_batched_*), because every step is executed ATOMICALLY AND IN ISOLATION:
//
//   * a step of the reference (move() + interact(), water.h:43-121 / wind.h:54-136; or flood(), water.h:123-145) is run
//     by exactly the code of soil_core.h, unchanged, on the live map -- what changes is only the interleaving of the
//     steps of different particles.
//   * claim tiles: the map is cut into 4x4-cell tiles. Before a step a particle reserves every tile its step can touch
//     (a bounding rectangle, see *_intent below) with an atomic max of {epoch, ~slot} on a claim plane; in the next
//     kernel it acts only if it holds ALL of them -- the lowest slot wins a contested tile, the others simply try
//     again in the following epoch. Winners of one epoch therefore have disjoint footprints: the epoch's result does
//     not depend on the order in which the hardware runs them, plain loads and stores suffice, and no undo log exists.
//   * two claim planes alternate (epoch parity): kernel e checks plane[e&1] (complete since kernel e-1 ended) and
//     writes the reservations for epoch e+1 into the other one.
//   * one kernel per epoch. A winner recomputes its move from the map (its tiles are exclusively its own), acts, then
//     looks ahead (a side-effect-free move() on the block it already holds) to learn which tiles the NEXT step needs.
//     A loser re-issues its reservation without reading anything.
//   * nested particles (water.h:246-264) run INSIDE their parent -- at once, as in the reference -- for as long as every step stays
//     inside the cells the parent owns exclusively (the tiles a flood holds; a grid tile and the ring of tiles around it): round 5.
//     A nested particle that would leave that region is SUSPENDED: its state becomes a record; the records, sorted by
//     (parent, sequence), run as the next GENERATION of the phase. (Rounds 2-4 deferred every nested particle: they then run
//     over ground the pass has meanwhile dried, live twice as long and leave +30 % cells under water -- profiles/r05_p2_grid_pass.md.)
//     A nested particle's `contains` is read from the cell it starts from instead of a random cell (water.h:251 via :15-17):
//     the value is overwritten (water.h:94) before anything reads it, see BatchPolicy::nested_init.
//   * grid pass (water.h:335-343): a cell's seep + cascade(.., 3) reaches 4 cells = one tile around it. Tiles whose regions
//     (tile + the ring around it) overlap are processed in BLOCK-RASTER order (tx-major), the reference's sweep direction --
//     the sweep ploughs water ahead of itself, and an order that does not (the nine tile colours of rounds 2-4) dries lakes
//     27 % too far; tiles whose regions are disjoint run concurrently: dependency passes (soilmx.hip batch_grid).
#pragma once
#include "soil_core.h"
#include "soil_spec.h"   // SMX_A_* atomics (device / host-sim)

namespace smx {

constexpr int GRID_SHIFT = 3;                 // the grid pass works on 8x8-cell tiles. A tile's REGION -- the cells its thread owns while it sweeps the tile: the reach of seep +
                                              // cascade(.., 3) (4 cells) and the room its nested particles run in -- is the tile and the ring of tiles around it, 24 x 24 cells;
                                              // tiles whose regions overlap have different colours (tx mod 3, ty mod 3) and are swept in colour order. Measured
                                              // (tools/p2_proxy.py, profiles/r05_p2_analysis.md): 4x4-cell tiles dry lakes 14 % too far in colour order (the reference's raster
                                              // sweep ploughs water ahead of itself; inside an 8x8 tile the order IS the raster order) and leave their nested particles 4 cells of
                                              // room (+26 % cells under water when the suspended ones run after the pass); 8x8: every figure within 1 sigma of the reference.
constexpr int DIRTY_SHIFT = 2;                // column strips: the halo exchange tracks changes in 4x4-cell tiles (bs.dirty)
constexpr uint32_t B_RUN = 0, B_ENDED = 1, B_DONE = 2, B_NEW = 3;   // B_NEW: top-level particle that has not acted yet
constexpr uint32_t B_AWAY = 4;               // column strips on several devices: the particle currently lives on another device
constexpr int SMX_WATER_GENERATIONS = 8;      // a tick's water particles run as this many consecutive top-level generations of n/k (smx_set_water_generations): the particles of
                                              // one generation advance together and do not see the lakes their own generation makes -- in the reference particle i sees what
                                              // every particle < i did (SoilMachine.cpp:287-298). Measured against 33 streams of the reference itself at the headline workload
                                              // (4096^2, 64 000 + 16 000, profiles/r06_p2_reference_4096.json): one generation is outside on four of eight figures by 4.8-7.9
                                              // sigma, four (round 5's default) on three by 2.3-3.0 (nested particles -23 %), EIGHT inside |z| < 2 on all eight, also under the
                                              // two-strip schedule and two of three other device streams; sixteen: water volume +2.1
constexpr int SMX_WATER_STAGGER = 0;          // epochs between the births of two consecutive generations inside ONE water phase (smx_set_water_stagger); 0 = the generations run one after
                                              // the other, each to its end
constexpr int B_CHUNK = 32;                   // epochs between two looks at the number of running particles (schedule, compaction)
constexpr int B_MAXCHILD = 250;               // nested particles one flood / one grid tile may emit per pass (the key keeps 8 bits for the sequence). Round 5: 24 -> 250:
                                              // a 4x4 tile of lake cells spawns more than 24 (every sweep position can send off its own film and those of its eight
                                              // neighbours, water.h:246-264), and a refused spawn leaves its water standing -- the cap alone put the grid pass +33 %
                                              // cells under water outside the reference (profiles/r05_p2_child_cap.md)

struct BRect { int tx0, tx1, ty0, ty1; };     // inclusive tile ranges; tx1 < tx0: empty

// c_*: this slot's share of the step / cascade counters (summed into DevState::ctr when the generation ends: an atomic
// per particle and epoch on five shared words would serialise the whole epoch)
// hint/pnx/pny: latency only (never read by the schedule): the particle acted in the previous epoch, and where its look-ahead
// says the next step lands -- so the blocks of that step can be requested together with the reservation words.
struct BWater { Water w; BRect rect; uint32_t state, toplevel; uint32_t c_steps, c_casc, c_tr, hint; int pnx, pny; };
struct BWind { Wind d; BRect rect; uint32_t state, hint; uint32_t c_steps, c_casc, c_tr, pad2; int pnx, pny; };
struct BChild {                               // a nested WaterParticle SUSPENDED at the edge of its parent's region, waiting for the next generation
  unsigned long long key;                     // (parent << 8) | sequence: the deterministic order of the children
  f2 pos, speed;                              // the particle as it stood before the step it could not take (water.h:9-41)
  double volume, sediment;
  int spill;
  uint32_t contains;
  uint32_t ended, pad;                        // 1: its move/interact loop is over, the flood (water.h:123-145) is what did not fit the region
};

// Column strips (DESIGN.md "Multi-GPU"): the map is cut into x-strips [x[g], x[g+1]); chunks of epochs alternate between
//   INTERIOR  a particle may act iff its whole reservation lies at least `a` cells inside ONE strip (map edges excepted), and
//   SEAM      iff it lies inside ONE seam zone [x[g+1]-b, x[g+1]+b) -- or at least `b` cells inside one strip, where no
//             seam zone reaches (so that a SEAM chunk does not idle the particles that are nowhere near a seam).
// b - a is at least the widest reservation (32 cells), so every reservation fits one of the two. In an INTERIOR chunk the
// strips touch disjoint cells -- one device each, nothing to exchange; a SEAM zone is run by the device left of the
// seam, on a halo that is refreshed before and handed back after, while every device goes on deep inside its own strip.
// Ineligible particles do not reserve anything (they are dormant until the other kind of chunk). With n == 1 everything is
// interior: the single-device schedule.
struct BStrips { int n; int x[9]; int a, b; };
constexpr int STRIP_INTERIOR = 0, STRIP_SEAM = 1;

// relaxed schedule, per epoch parity p = epoch & 1: BC_NFLOOD + p = length of the flood list, BC_NFLAG + p = cells flagged for a cascade,
// BC_NCASC + 9 * p + colour = length of a colour's cascade list
// BC_NRUN + p = particles that stepped in the epoch
enum BCtrl { BC_NLIVE = 0, BC_NCHILD, BC_CHILD_LOST, BC_NFREED, BC_FREED_LOST, BC_LIST_LOST, BC_RANGE_REJECT /* reservations that reach past the held columns (bholds) */, BC_CASC_CURSOR /* k_relax_cascade_flow */, BC_NFLOOD = 8, BC_NFLAG = 10, BC_NCASC = 12, BC_NRUN = 30, BC_LISTS_END = 32 /* [BC_NFLOOD, BC_LISTS_END): the per-epoch list lengths */, BC_STALLED = 32 /* k_relax_epochs: a device-wide barrier or a dataflow wait ran out of its spin budget */, BC_COUNT = 40 };
constexpr int R_MAXSTEPS = 8;                 // relaxed schedule: steps one wind particle may take per epoch (2 request entries each)

struct BatchShared {
  unsigned long long* claim[2];               // sized for one word per CELL; indexed [tx * nty + ty] at the current granularity
  int tshift, ntx, nty;                       // claim granularity: tiles of 2^tshift x 2^tshift cells (0 = single cells)
  int txb, txe;                               // tile columns [txb, txe) the planes hold (DevState::x_lo >> tshift ...: a strip context holds its range only)
  BWater* water; BWind* wind;                 // slots of the current generation
  uint32_t* live;                             // indices of the slots that are not DONE (rebuilt by batch_compact)
  BChild* children; uint32_t child_cap;
  uint32_t* freed; uint32_t freed_cap;        // pool nodes released during a kernel (pushed back between kernels)
  uint32_t* ctrl;                             // BCtrl
  const uint32_t* draws;                      // rand() values (already >> 1) of this generation: 2 per slot
  int dilate;                                 // reservations are widened by this many tiles on every side (see bdilate)
  int maxsteps;                               // actions one particle may take in one epoch while it stays inside the tiles it holds
  BStrips strips; int sphase;                 // column strips and the kind of the running chunk
  uint8_t* dirty;                             // [grid tiles] strips only: columns of this 4x4 tile changed since the last halo exchange
  int own_x0, own_x1;                         // strips on several devices: particles spawning outside [own_x0, own_x1) live elsewhere
  uint32_t birth_n, birth_gap;                // STAGGERED generations (smx_set_water_stagger; 0 = off): top-level water slot i is born -- takes its first step -- in epoch
                                              // (i / birth_n) * birth_gap of its phase; until then it waits as B_NEW (relax_water_step)
  // RELAXED schedule (soil_relax.h): running particles step without reservations; their map edits are queued per cell
  int relaxed;                                // 1: the epoch kernel only runs floods (B_ENDED); running particles belong to relax_*_step
  int rstride, rsteps;                        // request entries per slot (2 per step); steps a particle may take per epoch (water 1)
  uint32_t rtag;                              // this epoch's tag for the cascade flags (unique over the context's life, never 0)
  struct RReq* req;                           // [rstride * slots] queued edits of this epoch
  uint32_t* rnext;                            // [rstride * slots] next entry of the same cell's list
  uint32_t* rcount;                           // [slots] entries this slot queued in this epoch
  uint32_t* rhead;                            // [cells] newest entry queued for the cell (NIL: none)
  uint32_t* cflag;                            // [cells] tag of the last epoch that asked for Particle::cascade here
  uint32_t* cstate;                           // [cells] device only (nullptr: unused): 2 * tag = the cell waits in this epoch's colour lists, 2 * tag + 1 = its cascade is done
  uint32_t* flist[2]; uint32_t* flagl[2]; uint32_t* clist[2];   // per epoch parity: flood list (slots), flagged cells, cascade lists [9 colours][clist_cap]
  uint32_t list_cap, flag_cap, clist_cap;
};
struct RReq { double amount, vol; uint32_t cell, type, kind, casc; };   // kind: R_* bits; casc: cell whose cascade this step asks for (NIL: none)

SMX_D unsigned long long btag(uint32_t epoch, uint32_t slot) { return ((unsigned long long)(epoch + 1u) << 32) | (unsigned long long)(0xFFFFFFFFu - slot); }
SMX_D bool brect_empty(const BRect& r) { return r.tx1 < r.tx0; }
SMX_D bool brect_contains(const BRect& a, const BRect& b) {   // b inside a
  return !brect_empty(a) && b.tx0 >= a.tx0 && b.tx1 <= a.tx1 && b.ty0 >= a.ty0 && b.ty1 <= a.ty1;
}
// tiles of the cell rectangle [x0,x1] x [y0,y1], clipped to the map
SMX_D BRect brect_cells(const DevState& s, const BatchShared& bs, int x0, int x1, int y0, int y1) {
  x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0;
  x1 = x1 > s.dimx - 1 ? s.dimx - 1 : x1; y1 = y1 > s.dimy - 1 ? s.dimy - 1 : y1;
  BRect r; r.tx0 = x0 >> bs.tshift; r.tx1 = x1 >> bs.tshift; r.ty0 = y0 >> bs.tshift; r.ty1 = y1 >> bs.tshift;
  return r;
}
// A reservation covers the tiles the action can touch PLUS a margin of `dilate` tiles: two particles act in the same
// epoch only if they are more than 2*dilate tiles apart. Without the margin the engine is still deterministic and
// race-free, but particles that drain into the same pit keep disturbing each other and no single one ever levels its
// spot to the 1e-5 flatness that ends a move() (water.h:56) -- the event that seeds every lake in the reference, where
// a particle has the map to itself. The margin gives the lowest particle of a neighbourhood that quiet.
SMX_D BRect bdilate(const BatchShared& bs, const BRect& r) {
  if (brect_empty(r) || bs.dilate <= 0) return r;
  BRect o;
  o.tx0 = r.tx0 - bs.dilate < 0 ? 0 : r.tx0 - bs.dilate; o.tx1 = r.tx1 + bs.dilate > bs.ntx - 1 ? bs.ntx - 1 : r.tx1 + bs.dilate;
  o.ty0 = r.ty0 - bs.dilate < 0 ? 0 : r.ty0 - bs.dilate; o.ty1 = r.ty1 + bs.dilate > bs.nty - 1 ? bs.nty - 1 : r.ty1 + bs.dilate;
  return o;
}
// may an action that touches the columns cx0 .. cx1 run in the current kind of chunk?
SMX_D bool beligible_x(const DevState& s, const BatchShared& bs, int cx0, int cx1);
// may a particle with this reservation act in the running kind of chunk?
SMX_D bool beligible(const DevState& s, const BatchShared& bs, const BRect& r) {
  if (bs.strips.n <= 1) return true;
  if (brect_empty(r)) return false;
  const int cx0 = r.tx0 << bs.tshift;
  int cx1 = ((r.tx1 + 1) << bs.tshift) - 1; if (cx1 > s.dimx - 1) cx1 = s.dimx - 1;
  return beligible_x(s, bs, cx0, cx1);
}
SMX_D bool beligible_x(const DevState& s, const BatchShared& bs, int cx0, int cx1) {
  const BStrips& st = bs.strips;
  if (st.n <= 1) return true;
  if (cx0 < 0) cx0 = 0;
  if (cx1 > s.dimx - 1) cx1 = s.dimx - 1;
  if (bs.sphase == STRIP_INTERIOR) {
    for (int g = 0; g < st.n; g++) {
      const int lo = st.x[g] + (g ? st.a : 0), hi = st.x[g + 1] - (g < st.n - 1 ? st.a : 0);
      if (cx0 >= lo && cx1 < hi) return true;
    }
  } else {
    for (int g = 0; g + 1 < st.n; g++)
      if (cx0 >= st.x[g + 1] - st.b && cx1 < st.x[g + 1] + st.b) return true;
    // ... and whoever is so deep inside a strip (b cells from its boundaries) that no seam zone can touch what it touches
    // goes on: a SEAM chunk must not idle the 97 % of the particles that are nowhere near a seam
    for (int g = 0; g < st.n; g++) {
      const int lo = st.x[g] + (g ? st.b : 0), hi = st.x[g + 1] - (g < st.n - 1 ? st.b : 0);
      if (cx0 >= lo && cx1 < hi) return true;
    }
  }
  return false;
}
// strips: remember which 4x4 grid tiles a particle that acted may have changed (its whole reservation), for the halo exchange
SMX_D void bmark_dirty(const DevState& s, const BatchShared& bs, const BRect& r) {
  if (bs.strips.n <= 1 || !bs.dirty || brect_empty(r)) return;
  const int gty = (s.dimy + (1 << DIRTY_SHIFT) - 1) >> DIRTY_SHIFT;
  const int cx0 = r.tx0 << bs.tshift, cy0 = r.ty0 << bs.tshift;
  int cx1 = ((r.tx1 + 1) << bs.tshift) - 1, cy1 = ((r.ty1 + 1) << bs.tshift) - 1;
  if (cx1 > s.dimx - 1) cx1 = s.dimx - 1;
  if (cy1 > s.dimy - 1) cy1 = s.dimy - 1;
  for (int gx = cx0 >> DIRTY_SHIFT; gx <= (cx1 >> DIRTY_SHIFT); gx++)
    for (int gy = cy0 >> DIRTY_SHIFT; gy <= (cy1 >> DIRTY_SHIFT); gy++) bs.dirty[(size_t)gx * gty + gy] = 1;
}
SMX_D void bmark_dirty_cells(const DevState& s, const BatchShared& bs, int cx0, int cx1, int cy0, int cy1) {   // the same, for a rectangle of cells
  if (bs.strips.n <= 1 || !bs.dirty) return;
  const int gty = (s.dimy + (1 << DIRTY_SHIFT) - 1) >> DIRTY_SHIFT;
  if (cx0 < 0) cx0 = 0; if (cy0 < 0) cy0 = 0;
  if (cx1 > s.dimx - 1) cx1 = s.dimx - 1;
  if (cy1 > s.dimy - 1) cy1 = s.dimy - 1;
  for (int gx = cx0 >> DIRTY_SHIFT; gx <= (cx1 >> DIRTY_SHIFT); gx++)
    for (int gy = cy0 >> DIRTY_SHIFT; gy <= (cy1 >> DIRTY_SHIFT); gy++) bs.dirty[(size_t)gx * gty + gy] = 1;
}
// (a reservation that reaches past the columns a strip context holds -- a resident particle's old rectangle after the strips were
//  re-cut, a wide dilate -- is neither written nor ever held: the planes end at [txb, txe))
SMX_D void bclaim(const BatchShared& bs, int plane, const BRect& r, unsigned long long tag) {
  for (int tx = r.tx0 < bs.txb ? bs.txb : r.tx0; tx <= r.tx1 && tx < bs.txe; tx++)
    for (int ty = r.ty0; ty <= r.ty1; ty++) (void)SMX_A_MAX64(&bs.claim[plane][(size_t)(tx - bs.txb) * bs.nty + ty], tag);
}
SMX_D bool bholds(const BatchShared& bs, int plane, const BRect& r, unsigned long long tag) {
  if (r.tx1 >= r.tx0 && (r.tx0 < bs.txb || r.tx1 >= bs.txe)) { SMX_A_ADD(&bs.ctrl[BC_RANGE_REJECT], 1u); return false; }   // (counted: the strip driver names it when a phase stalls)
  bool ok = true;
  for (int tx = r.tx0; tx <= r.tx1; tx++)
    for (int ty = r.ty0; ty <= r.ty1; ty++) ok &= (bs.claim[plane][(size_t)(tx - bs.txb) * bs.nty + ty] == tag);
  return ok;
}

// ---------------- access policy: direct access (exclusive by construction), shared pool through atomics ----------------
struct BatchPolicy {
  static constexpr bool READ_ONLY = false;
  static constexpr bool DEFER_NESTED = true;
  static constexpr bool EXCLUSIVE = true;       // this thread owns every cell it touches while it acts (soil_core.h push_frame)
  const BatchShared& bs;
  unsigned long long live0;                   // live sections when the kernel started (+ what this thread added)
  long long live_delta = 0;
  unsigned long long parent = 0;              // key prefix of the children this thread emits
  uint32_t nchild = 0;
  uint32_t a_steps = 0, a_casc = 0, a_tr = 0; // frequent counters: kept per slot, see BWater::c_*
  uint32_t a_nsteps = 0;                      // steps of nested particles run inside this thread's action (never top-level steps)
  bool in_nested = false;
  BRect region = BRect{0, -1, 0, -1};         // the tiles this thread owns exclusively while it acts (nested particles run inside)
  SMX_D BatchPolicy(const DevState& s, const BatchShared& b) : bs(b), live0(s.ctr[C_LIVE_SECTIONS]) {}
  template <int N> struct Pre {};
  SMX_D void save_blk(const Blk&) {}
  SMX_D void load_blk(Blk&) {}
  SMX_D bool touch(const DevState&, size_t, size_t) { return true; }
  template <int N> SMX_D void prefetch(const DevState&, const size_t (&)[N], const size_t (&)[N], Pre<N>&) {}
  template <int N> SMX_D bool acquire_log(const DevState&, const size_t (&)[N], const Pre<N>&, const Sec (&)[N]) { return true; }
  SMX_D void pre_write_node(const DevState&, uint32_t) {}
  // Flag bytes (F_AIR, F_SAT of a column; the classification of the next grid pass reads them). On the device: fire-and-forget OR / AND on the byte's 32-bit word
  // at the L2 (the plane is allocated in whole words; the other three bytes get the neutral element, so whoever owns those cells is not disturbed) -- a thread
  // that first LOADS the byte to see whether it must change waits a memory round trip per call (35 water-onto-water transfers per active cell of the grid pass in
  // the wet regime set F_SAT, profiles/r06_grid_tiles.md). ALL updates of this policy go this way: a plain load after an atomic could be served a stale line by
  // the CU's vector cache. The host simulator keeps the byte-wide read-modify-write (its threads own their cells; a word-wide one would race with the neighbours').
#ifdef SMX_HOSTSIM
  SMX_D void set_flag(const DevState& s, size_t c, uint8_t f) { const uint8_t v = s.flags[c]; if (!(v & f)) s.flags[c] = v | f; }
  SMX_D void clear_flag(const DevState& s, size_t c, uint8_t f) { const uint8_t v = s.flags[c]; if (v & f) s.flags[c] = v & (uint8_t)~f; }
#else
  SMX_D void set_flag(const DevState& s, size_t c, uint8_t f) {
    const uintptr_t a = (uintptr_t)(s.flags + c);
    SMX_OR32_ASYNC(reinterpret_cast<uint32_t*>(a & ~(uintptr_t)3), (uint32_t)f << (8u * (uint32_t)(a & 3u)));
  }
  SMX_D void clear_flag(const DevState& s, size_t c, uint8_t f) {
    const uintptr_t a = (uintptr_t)(s.flags + c);
    SMX_AND32_ASYNC(reinterpret_cast<uint32_t*>(a & ~(uintptr_t)3), ~((uint32_t)f << (8u * (uint32_t)(a & 3u))));
  }
#endif
  SMX_D void set_flag_async(const DevState& s, size_t c, uint8_t f) { set_flag(s, c, f); }
  // secpool::get (layermap.h:89-103) fails when live sections == capacity. Concurrent particles see the count as of
  // the kernel's start plus their own additions: exact while the pool has headroom (the host checks), approximate
  // in the last few sections before exhaustion -- stated in DESIGN.md.
  SMX_D bool can_get(const DevState& s) const { return (long long)live0 + live_delta < (long long)s.pool_capacity; }
  SMX_D void live_add(const DevState&, int d) { live_delta += d; }
  SMX_D uint32_t node_alloc(const DevState& s) {            // pops only during a kernel (frees are deferred): lock-free
    const uint32_t old = SMX_A_SUB(s.free_count, 1u);
    if (old == 0u || old > (uint32_t)s.pool_capacity) { SMX_A_ADD(s.free_count, 1u); return NIL; }
    return s.freelist[old - 1u];
  }
  SMX_D void node_free(const DevState&, uint32_t e) {
    const uint32_t k = SMX_A_ADD(&bs.ctrl[BC_NFREED], 1u);
    if (k < bs.freed_cap) bs.freed[k] = e; else SMX_A_ADD(&bs.ctrl[BC_FREED_LOST], 1u);
  }
  SMX_D void rand2(const DevState&, int& a, int& b) { a = 0; b = 0; }   // (nested particles draw when their generation starts)
  // A nested particle may start only while this thread can still write a record for it (soil_core.h water_drive checks BEFORE the
  // water leaves its column): when the per-thread limit is reached the spawn is refused and the water stays where it is.
  SMX_D bool child_room() const { return nchild < (uint32_t)B_MAXCHILD; }
  // the record of a suspended nested particle (the generation's buffer full: the particle is lost and COUNTED -- bench.py and the
  // drivers treat children_lost != 0 as an invalid run)
  SMX_D void emit_child(const Water& w, bool ended) {
    const uint32_t k = SMX_A_ADD(&bs.ctrl[BC_NCHILD], 1u);   // (may overshoot child_cap: the host clamps; records [0, cap) are all valid)
    if (k >= bs.child_cap) { SMX_A_ADD(&bs.ctrl[BC_CHILD_LOST], 1u); return; }
    BChild& c = bs.children[k];
    c.key = (parent << 8) | (unsigned long long)nchild; nchild++;
    c.pos = w.pos; c.speed = w.speed; c.volume = w.volume; c.sediment = w.sediment; c.spill = w.spill; c.contains = w.contains;
    c.ended = ended ? 1u : 0u; c.pad = 0u;
  }
  // the nested particle of water.h:246-264 run INSIDE this thread's action: definitions after the intents below
  template <class S> SMX_D int run_nested(S& sim, Water& w);       // 0: its loop ended inside the region, 1: suspended (record written)
  template <class S> SMX_D bool flood_fits(S& sim, const Water& w); // may its flood (water.h:123-145) run here? else: record written
  SMX_D void add_counter(const DevState& s, int which, unsigned long long v) {
    if (!v) return;
    if (which == C_STEPS_WATER_ALL && in_nested) a_nsteps += (uint32_t)v;                 // (a nested particle inside a top-level one's flood)
    else if (which == C_STEPS_WATER_ALL || which == C_STEPS_WIND) a_steps += (uint32_t)v; // (top-level steps: by the slot's flag)
    else if (which == C_STEPS_WATER_TOP) {}
    else if (which == C_CASCADE_CALLS) a_casc += (uint32_t)v;
    else if (which == C_CASCADE_TRANSFERS) a_tr += (uint32_t)v;
    else SMX_A_ADD64(&s.ctr[which], v);                                                   // floods, nested, overflow: rare
  }
  SMX_D void finish(const DevState& s) {
    if (live_delta) SMX_A_ADD64(&s.ctr[C_LIVE_SECTIONS], (unsigned long long)live_delta); live_delta = 0;
    if (a_nsteps) SMX_A_ADD64(&s.ctr[C_STEPS_WATER_ALL], (unsigned long long)a_nsteps); a_nsteps = 0;
  }
};

// ---------------- intents: which tiles can the NEXT action of this particle touch? (pure functions of particle + block) ----
// Water, running: a step = move() at ipos (plus-stencil, track word) + interact(): remove/add at ipos, bilinear height and
// Particle::cascade(.., 0) in the 3x3 around npos = round(new pos)  ->  bbox(3x3(ipos) u 3x3(npos)).
// A move() that returns false (stall, out of bounds) has still READ the plus-stencil of ipos to decide that (and
// writes ipos's track word): the 3x3 around ipos stays reserved, or the decision itself would race with a neighbour.
// `t` / `mv`: the particle after that move and whether move() returned true -- the epoch loop commits exactly this
// look-ahead instead of computing the move a second time.
template <class P>
SMX_D BRect water_intent_run(Sim<P>& sim, const BatchShared& bs, const Water& w, const Blk& blk, Water& t, bool& mv) {
  t = w;
  mv = sim.template water_move<true>(t, false, blk);
  if (!mv) return brect_cells(sim.s, bs, w.ix - 1, w.ix + 1, w.iy - 1, w.iy + 1);
  const int nx = (int)roundf(t.pos.x), ny = (int)roundf(t.pos.y);
  const int x0 = (w.ix < nx ? w.ix : nx) - 1, x1 = (w.ix > nx ? w.ix : nx) + 1;
  const int y0 = (w.iy < ny ? w.iy : ny) - 1, y1 = (w.iy > ny ? w.iy : ny) + 1;
  return brect_cells(sim.s, bs, x0, x1, y0, y1);
}
// Water, move/interact loop over: flood() (water.h:123-145) or nothing. The flood acts around trunc(pos): sediment +
// water added there, Particle::cascade in the 3x3 around round(pos), seep, and WaterParticle::cascade whose recursion
// (one level per unit of spill, water.h:278-279) reaches `spill` cells + 1 ... = radius max(spill, 2) in all.
SMX_D bool water_floods(const Water& w) { return !(w.volume < 0.01 || w.spill <= 0); }   // the gate of water.h:125
SMX_D BRect water_intent_flood(const DevState& s, const BatchShared& bs, const Water& w) {
  const int R = w.spill > 2 ? w.spill : 2;
  const int fx = (int)w.pos.x, fy = (int)w.pos.y;
  return brect_cells(s, bs, fx - R, fx + R, fy - R, fy + R);
}
// Wind, running: move() at ipos (plus-stencil, frequency word) + interact(): remove at ipos / add at npos and ipos,
// Particle::cascade(.., 1) around both with one level of recursion  ->  bbox(5x5(ipos) u 5x5(npos)).
template <class P>
SMX_D BRect wind_intent_run(Sim<P>& sim, const BatchShared& bs, const Wind& d, const Blk& blk, Wind& t, bool& mv) {
  t = d;
  mv = sim.template wind_move<true>(t, blk);
  if (!mv) return brect_cells(sim.s, bs, d.ix - 1, d.ix + 1, d.iy - 1, d.iy + 1);
  const int nx = (int)roundf(t.pos.x), ny = (int)roundf(t.pos.y);
  const int x0 = (d.ix < nx ? d.ix : nx) - 2, x1 = (d.ix > nx ? d.ix : nx) + 2;
  const int y0 = (d.iy < ny ? d.iy : ny) - 2, y1 = (d.iy > ny ? d.iy : ny) + 2;
  return brect_cells(sim.s, bs, x0, x1, y0, y1);
}

// ---------------- nested particles inside their parent's action (water.h:246-264) ----------------
// while(particle.move(..) && particle.interact(..)) (water.h:259) for as long as every step -- the 3x3 block it leaves and the 3x3
// block it enters, water_intent_run -- lies inside the tiles this thread owns (pol.region): the same look-ahead / commit pair the
// epoch loop uses. The first step that does not fit suspends the particle as it stands (BChild).
template <class S>
SMX_D int BatchPolicy::run_nested(S& sim, Water& w) {
  sim.flush_counters();
  in_nested = true;
  w.ix = (int)roundf(w.pos.x); w.iy = (int)roundf(w.pos.y);
  int r = 0;
  Blk A, B;
  bool haveA = false;
  while (true) {
    if (!brect_contains(region, brect_cells(sim.s, bs, w.ix - 1, w.ix + 1, w.iy - 1, w.iy + 1))) { r = 1; break; }   // (not even the stencil of move())
    if (!haveA) sim.template load_block<0>(w.ix, w.iy, A);
    Water t; bool mv;
    const BRect need = water_intent_run(sim, bs, w, A, t, mv);
    if (!brect_contains(region, need)) { r = 1; break; }
    sim.water_move_commit(w, mv, false, A);                   // water.h:43-73: the look-ahead becomes the move
    w = t;
    if (!mv) break;
    if (sim.water_interact(w, A, B) == 0) break;               // water.h:75-121
    A = B; haveA = true;
  }
  if (r == 1) emit_child(w, false);
  sim.flush_counters();                                      // (the nested particle's steps are not its parent's: add_counter looks at in_nested)
  in_nested = false;
  return r;
}
template <class S>
SMX_D bool BatchPolicy::flood_fits(S& sim, const Water& w) {
  if (brect_contains(region, water_intent_flood(sim.s, bs, w))) return true;
  emit_child(w, true);
  return false;
}

// ---------------- generation start: construct the particles, publish their first reservations (plane 0, epoch 0) -------
// top-level particle `slot` of a phase: WaterParticle(map) (water.h:11-19) from draws 2*slot (-> y), 2*slot+1 (-> x);
// or (child != nullptr) a nested particle that was suspended at the edge of its parent's region, exactly as it stood (BChild)
template <class P>
SMX_D void batch_water_spawn(Sim<P>& sim, const BatchShared& bs, uint32_t slot, const BChild* child, const uint32_t* contains = nullptr) {
  const DevState& s = sim.s;
  BWater& q = bs.water[slot];
  Water w;
  if (child) {
    w.pos = child->pos; w.speed = child->speed; w.volume = child->volume; w.sediment = child->sediment; w.evaprate = 0.001;
    w.spill = child->spill; w.surface = 0; w.friction = 0.f; w.contains = child->contains;
    (void)contains;
  } else {
    const int ry = (int)(bs.draws[2u * slot] % (uint32_t)s.dimy), rx = (int)(bs.draws[2u * slot + 1u] % (uint32_t)s.dimx);
    if (rx < s.x_lo || rx >= s.x_hi) {                       // a strip context does not hold that column: the particle lives on another device (B_AWAY below)
      w.pos = {(float)rx, (float)ry}; w.speed = {0.f, 0.f};
      w.volume = 1.0; w.sediment = 0.0; w.evaprate = 0.001; w.spill = 3;
      w.surface = 0; w.friction = 0.f; w.contains = 0;
    } else sim.water_init(w, rx, ry);
  }
  q.toplevel = child ? 0u : 1u;
  w.ix = (int)roundf(w.pos.x); w.iy = (int)roundf(w.pos.y);
  // A top-level particle reads what it `contains` from its spawn cell (water.h:15-17) when it first ACTS, not now: the
  // reference constructs a particle right before it runs, so it sees what earlier particles did to that cell.
  q.w = w; q.state = child ? (child->ended ? B_ENDED : B_RUN) : B_NEW;
  if (bs.rcount) bs.rcount[slot] = 0u;
  q.c_steps = 0; q.c_casc = 0; q.c_tr = 0; q.hint = 0; q.pnx = -1; q.pny = -1;
  q.rect = BRect{0, -1, 0, -1};                              // first reservation: batch_water_reclaim
  // who owns a particle: the device that owns the cell it stands on (a flood acts around trunc(pos), a step around round(pos))
  const int ox = (child && child->ended) ? (int)w.pos.x : w.ix;
  if (ox < bs.own_x0 || ox >= bs.own_x1) q.state = B_AWAY;
}
// (rounds 2-4: a nested particle's `contains` came from a random cell, read by the device that owned it; the record carries it now.
//  Kept for the step-wise strip driver's call sequence: every device answers with the record's own value.)
SMX_D uint32_t batch_child_contains(const BChild* kids, uint32_t slot) { return kids[slot].contains; }
template <class P>
SMX_D void batch_wind_spawn(Sim<P>& sim, const BatchShared& bs, uint32_t slot) {
  const DevState& s = sim.s;
  BWind& q = bs.wind[slot];
  const int ry = (int)(bs.draws[2u * slot] % (uint32_t)s.dimy), rx = (int)(bs.draws[2u * slot + 1u] % (uint32_t)s.dimx);
  Wind d;
  if (rx < s.x_lo || rx >= s.x_hi) {                         // (not held by this strip context: lives on another device)
    d.pos = {(float)rx, (float)ry}; d.speed = {-2.f, 0.f, 1.f}; d.sediment = 0.0; d.height = 0.0; d.sheight = 0.0;
    d.ix = rx; d.iy = ry; d.surface = 0; d.contains = 0;
  } else sim.wind_init(d, rx, ry);
  if (bs.rcount) bs.rcount[slot] = 0u;
  q.d = d; q.hint = 0; q.c_steps = 0; q.c_casc = 0; q.c_tr = 0; q.pad2 = 0; q.pnx = -1; q.pny = -1;
  q.state = B_NEW;                                           // `contains` (wind.h:19-21) is read when the particle first acts
  q.rect = BRect{0, -1, 0, -1};                              // first reservation: batch_wind_reclaim
  if (d.ix < bs.own_x0 || d.ix >= bs.own_x1) q.state = B_AWAY;
}

// (Re)issue the reservation of a running particle for epoch `epoch` from its CURRENT state: at the start of a
// generation, and whenever the schedule changes the claim granularity (the planes are cleared then). Runs in a kernel
// of its own: nothing is written to the map meanwhile, so the look-ahead may read freely.
template <class P>
SMX_D void batch_water_reclaim(Sim<P>& sim, const BatchShared& bs, uint32_t slot, uint32_t epoch) {
  BWater& q = bs.water[slot];
  if (q.state == B_DONE || q.state == B_AWAY) return;
  if (bs.relaxed && (q.state == B_RUN || q.state == B_NEW)) { q.rect = BRect{0, -1, 0, -1}; return; }   // (steps without reservations)
  BRect r;
  if (q.state == B_ENDED) {
    if (!water_floods(q.w)) { q.state = B_DONE; q.rect = BRect{0, -1, 0, -1}; return; }
    r = water_intent_flood(sim.s, bs, q.w);
  } else {
    Blk A;
    sim.template load_block<0>(q.w.ix, q.w.iy, A);
    Water t; bool mv;
    r = bdilate(bs, water_intent_run(sim, bs, q.w, A, t, mv));
  }
  q.rect = r;
  if (beligible(sim.s, bs, r)) bclaim(bs, (int)(epoch & 1u), r, btag(epoch, slot));
}
template <class P>
SMX_D void batch_wind_reclaim(Sim<P>& sim, const BatchShared& bs, uint32_t slot, uint32_t epoch) {
  BWind& q = bs.wind[slot];
  if (q.state == B_DONE || q.state == B_AWAY) return;
  if (bs.relaxed) { q.rect = BRect{0, -1, 0, -1}; return; }  // (steps without reservations)
  Blk A;
  sim.template load_block<1>(q.d.ix, q.d.iy, A);
  Wind t; bool mv;
  q.rect = bdilate(bs, wind_intent_run(sim, bs, q.d, A, t, mv));
  if (beligible(sim.s, bs, q.rect)) bclaim(bs, (int)(epoch & 1u), q.rect, btag(epoch, slot));
}

// ---------------- one epoch of one particle ----------------
// A particle that holds its reservation acts; while the NEXT action still lies inside the tiles it holds (they are
// exclusively its own until this kernel ends) it may go on, up to bs.maxsteps actions per epoch -- with a margin
// (bs.dilate) a lone particle advances several steps per kernel, which is what keeps the long tail of a phase (a few
// far-travelling wind particles, a queue of floods in one pit) from costing one kernel launch per step.
template <class P>
SMX_D void batch_water_epoch(Sim<P>& sim, const BatchShared& bs, uint32_t slot, uint32_t epoch, Frame* st) {
  const DevState& s = sim.s;
  BWater& q = bs.water[slot];
  uint32_t state = q.state;
  if (state == B_DONE || state == B_AWAY) return;
  if (bs.relaxed && (state == B_RUN || state == B_NEW)) return;   // relaxed schedule: this kernel only runs the floods
  BRect rect = q.rect;
  const int cur = (int)(epoch & 1u), nxt = cur ^ 1;
  // A particle that acted in the previous epoch is, as a rule, the head of its queue and holds again: request the two
  // blocks of its next step NOW, together with the reservation words, instead of after them (two dependent memory round
  // trips less on the chain that bounds a phase). If it does hold, nobody else writes those cells in this kernel, so what
  // was read early is current; if not, the values are dropped.
  Blk A, B;
  bool haveA = false, haveB = false;
  const int pbx = q.pnx, pby = q.pny;
  if (q.hint && (state == B_RUN || state == B_NEW)) {
    sim.template load_block<0>(q.w.ix, q.w.iy, A); haveA = true;
    if (pbx >= 0) { sim.template load_block<0>(pbx, pby, B); haveB = true; }
  }
  uint32_t hint = 0; int pnx = -1, pny = -1;
  if (bholds(bs, cur, rect, btag(epoch, slot))) {
    const BRect held = rect;
    Water w = q.w;
    int acted = 0;
    while (true) {
      BRect need;
      Water t; bool mv = false;
      if (state == B_RUN || state == B_NEW) {
        if (!haveA) sim.template load_block<0>(w.ix, w.iy, A);
        need = water_intent_run(sim, bs, w, A, t, mv);
      } else {                                               // B_ENDED: flood() or nothing (water.h:125)
        if (!water_floods(w)) { state = B_DONE; break; }
        need = water_intent_flood(s, bs, w);
      }
      // reserve it for the next epoch. The margin is for particles that TRAVEL (it buys them several steps per epoch);
      // a flood is a particle's last action, and floods crowd together on lake shores: no margin there.
      if (acted >= bs.maxsteps || !brect_contains(held, need)) {
        rect = (state == B_ENDED) ? need : bdilate(bs, need);
        if (state != B_ENDED && mv) { pnx = (int)roundf(t.pos.x); pny = (int)roundf(t.pos.y); }
        break;
      }
      acted++;
      if (state == B_NEW) {                                  // WaterParticle(map) water.h:15-17, on the map as it is NOW
        t.contains = sim.soils[Sim<P>::vsurf(A.v8)].transports;   // (move() does not look at `contains`: the look-ahead stands)
        state = B_RUN;
      }
      if (state == B_RUN) {
        sim.water_move_commit(w, mv, q.toplevel != 0u, A);   // water.h:43-73: the look-ahead becomes the move (track word, counters)
        w = t;
        if (mv) {
          const bool pre = haveB && (int)roundf(w.pos.x) == pbx && (int)roundf(w.pos.y) == pby;   // the block requested early is the one needed
          haveB = false;
          const int r = sim.water_interact(w, A, B, pre);    // water.h:75-121
          if (r == 0) state = B_ENDED;
          else { A = B; haveA = true; }                      // B = the exact block around the new ipos
        } else state = B_ENDED;
      } else {                                               // flood (water.h:123-145); nested particles -> children
        sim.pol.parent = (unsigned long long)slot; sim.pol.nchild = 0;
        sim.pol.region = held;                               // nested particles run inside the tiles this flood holds
        int depth = 0;
        if (sim.water_flood_prelude(w)) sim.push_frame(st, depth, w.ix, w.iy, w.spill);
        sim.water_drive(w, false, false, st, depth);
        state = B_DONE;
        break;
      }
    }
    q.w = w; q.state = state;
    if (acted) bmark_dirty(s, bs, held);
    if (state == B_DONE) rect = BRect{0, -1, 0, -1};
    q.rect = rect;
    hint = (acted && state != B_DONE) ? 1u : 0u;
    sim.flush_counters();
    q.c_steps += sim.pol.a_steps; q.c_casc += sim.pol.a_casc; q.c_tr += sim.pol.a_tr;
    sim.pol.a_steps = sim.pol.a_casc = sim.pol.a_tr = 0;
  }
  if (q.hint != hint) q.hint = hint;
  if (hint) { q.pnx = pnx; q.pny = pny; }
  if (state != B_DONE && beligible(s, bs, rect)) bclaim(bs, nxt, rect, btag(epoch + 1u, slot));
}

template <class P>
SMX_D void batch_wind_epoch(Sim<P>& sim, const BatchShared& bs, uint32_t slot, uint32_t epoch) {
  BWind& q = bs.wind[slot];
  uint32_t state = q.state;
  if (state == B_DONE || state == B_AWAY) return;
  BRect rect = q.rect;
  const int cur = (int)(epoch & 1u), nxt = cur ^ 1;
  Blk A, B;
  bool haveA = false;
  if (q.hint && (state == B_RUN || state == B_NEW)) { sim.template load_block<1>(q.d.ix, q.d.iy, A); haveA = true; }   // (see batch_water_epoch)
  uint32_t hint = 0;
  if (bholds(bs, cur, rect, btag(epoch, slot))) {
    const BRect held = rect;
    Wind d = q.d;
    int acted = 0;
    while (true) {
      if (!haveA) sim.template load_block<1>(d.ix, d.iy, A);
      Wind t; bool mv;
      const BRect need = wind_intent_run(sim, bs, d, A, t, mv);
      if (acted >= bs.maxsteps || !brect_contains(held, need)) { rect = bdilate(bs, need); break; }
      acted++;
      if (state == B_NEW) {                                  // WindParticle(map) wind.h:19-21 + the gate of wind.h:56-57
        t.contains = sim.soils[Sim<P>::vsurf(A.v8)].transports;   // (move() does not look at `contains`)
        state = B_RUN;
        if (sim.soils[t.contains].suspension == 0.0) { d.contains = t.contains; state = B_DONE; break; }
      }
      sim.wind_move_commit(mv, A);                           // wind.h:54-92: the look-ahead becomes the move (frequency word, counter)
      d = t;
      if (mv) {
        sim.wind_interact(d, A, B);                          // wind.h:94-136 (always true); B = exact block around the new ipos
        A = B; haveA = true;
      } else { state = B_DONE; break; }
    }
    q.d = d; q.state = state;
    if (acted) bmark_dirty(sim.s, bs, held);
    if (state == B_DONE) rect = BRect{0, -1, 0, -1};
    q.rect = rect;
    hint = (acted && state != B_DONE) ? 1u : 0u;
    sim.flush_counters();
    q.c_steps += sim.pol.a_steps; q.c_casc += sim.pol.a_casc; q.c_tr += sim.pol.a_tr;
    sim.pol.a_steps = sim.pol.a_casc = sim.pol.a_tr = 0;
  }
  if (q.hint != hint) q.hint = hint;
  if (state != B_DONE && beligible(sim.s, bs, rect)) bclaim(bs, nxt, rect, btag(epoch + 1u, slot));
}

// The schedule of a generation (part of the engine's definition: the CPU restatement follows the same rule): claim
// granularity, margin and actions per epoch as a function of how many particles are still running when a chunk of
// epochs starts. Measured on the MI355X (profiles/r02_batched_schedules.md):
//   water  what limits a phase is particles EXCLUDING each other -- they drain into the same pits and queue there, one
//          particle per epoch and neighbourhood -- so reservations get no margin; 4x4-cell tiles leave enough slack for
//          the particle at the head of a queue to take up to 4 steps per launch without widening anybody's exclusion.
//   wind   particles travel far (up to ~13 000 steps, 2-3 cells each) and rarely meet: wide tiles with a margin buy a
//          lone particle up to 32 steps per kernel launch.
// `fine_above`: water reserves single cells while more particles than this are running (default: never; the tests lower it).
struct BSched { int tshift, dilate, maxsteps; };

#ifdef SMX_HOSTSIM
#define SMX_HD inline
#else
#define SMX_HD __host__ __device__ inline
#endif
SMX_HD BSched batch_schedule(bool wind, uint32_t nlive, int base_dilate, uint32_t fine_above = 0xFFFFFFFFu) {
  BSched k;
  if (wind) {
    // round 4 (profiles/r04_wind_schedule.md, swept on the device): 4x4-cell tiles with one tile of margin and up to 32 actions per epoch at
    // EVERY particle count. Round 2's schedule (8 actions, and 8x8-cell tiles / 32 actions below 4 096 running particles) cost 206-220 ms of
    // wind phase on the headline workload, this one 139 ms: larger tiles and margins only add contention; how many actions fit an epoch
    // is bounded by the geometry (a wind particle leaves a 12-cell reservation after ~6 steps), not by the limit.
    k.tshift = 2; k.dilate = 1; k.maxsteps = 32;
    (void)nlive;
  } else {
    k.tshift = 2; k.dilate = 0; k.maxsteps = 4;
    if (nlive > fine_above) { k.tshift = 0; k.maxsteps = 1; }
  }
  if (base_dilate > k.dilate) k.dilate = base_dilate;
  return k;
}
// n equal strips, boundaries on multiples of 16 cells (the widest claim tile is 8 cells, the grid tiles are 4)
SMX_HD BStrips batch_make_strips(int dimx, int n, int a, int b) {
  BStrips st; st.n = n < 1 ? 1 : (n > 8 ? 8 : n); st.a = a; st.b = b;
  for (int g = 0; g <= st.n; g++) st.x[g] = g == st.n ? dimx : (int)(((long long)dimx * g / st.n) / 16 * 16);
  for (int g = st.n + 1; g < 9; g++) st.x[g] = dimx;
  return st;
}
SMX_HD void batch_apply_schedule(BatchShared& bs, const DevState& s, const BSched& k) {
  bs.tshift = k.tshift; bs.dilate = k.dilate; bs.maxsteps = k.maxsteps;
  bs.ntx = (s.dimx + (1 << k.tshift) - 1) >> k.tshift; bs.nty = (s.dimy + (1 << k.tshift) - 1) >> k.tshift;
  bs.txb = s.x_lo >> k.tshift; bs.txe = ((s.x_hi - 1) >> k.tshift) + 1;
}

// ---------------- grid pass: one colour, one tile ----------------
// WaterParticle::seep(map) (water.h:335-343) for the ACTIVE cells (classification bitmap, soilmx.hip k_grid_classify)
// of tile (tx, ty), x-major inside the tile; nested particles become children keyed by (colour pass, tile, sequence).
template <class P>
SMX_D unsigned long long batch_grid_tile(Sim<P>& sim, int tx, int ty, unsigned long long key, Frame* st) {
  const DevState& s = sim.s;
  sim.pol.parent = key; sim.pol.nchild = 0;
  {                                                          // the tile and the ring of tiles around it are this thread's own (its seep + cascade reach them)
    const int gtx = (s.dimx + (1 << GRID_SHIFT) - 1) >> GRID_SHIFT, gty = (s.dimy + (1 << GRID_SHIFT) - 1) >> GRID_SHIFT;
    const int R = 1;
    sim.pol.region = BRect{tx - R < 0 ? 0 : tx - R, tx + R > gtx - 1 ? gtx - 1 : tx + R, ty - R < 0 ? 0 : ty - R, ty + R > gty - 1 ? gty - 1 : ty + R};
  }
  Water dummy;
  dummy.pos = {0.f, 0.f}; dummy.speed = {0.f, 0.f}; dummy.volume = 0.0; dummy.sediment = 0.0; dummy.evaprate = 0.0;
  dummy.spill = 0; dummy.ix = dummy.iy = 0; dummy.friction = 0.f; dummy.surface = dummy.contains = 0;
  unsigned long long visited = 0;
  const int x0 = tx << GRID_SHIFT, y0 = ty << GRID_SHIFT;
  for (int x = x0; x < x0 + (1 << GRID_SHIFT) && x < s.dimx; x++)
    for (int y = y0; y < y0 + (1 << GRID_SHIFT) && y < s.dimy; y++) {
      const size_t c = (size_t)x * s.dimy + y;
      if (!((s.active[c >> 6] >> (c & 63)) & 1ull)) continue;
      int depth = 0;
      { SMX_SECT_T0(); sim.seep(x, y); SMX_SECT(9); }        // water.h:339
      sim.push_frame(st, depth, x, y, 3);                    // water.h:340
      sim.water_drive(dummy, false, false, st, depth);
      visited++;
    }
  if (visited) bmark_dirty_cells(s, sim.pol.bs, x0 - (1 << GRID_SHIFT), x0 + 2 * (1 << GRID_SHIFT) - 1, y0 - (1 << GRID_SHIFT), y0 + 2 * (1 << GRID_SHIFT) - 1);   // strips: the region may have changed
  sim.flush_counters();                                      // (cascade counters of the grid pass go straight to DevState::ctr)
  if (sim.pol.a_casc) SMX_A_ADD64(&s.ctr[C_CASCADE_CALLS], (unsigned long long)sim.pol.a_casc);
  if (sim.pol.a_tr) SMX_A_ADD64(&s.ctr[C_CASCADE_TRANSFERS], (unsigned long long)sim.pol.a_tr);
  sim.pol.a_casc = sim.pol.a_tr = 0;
  return visited;
}

// grid pass under column strips: a tile's seep + cascade reaches one tile around it. It is processed in the INTERIOR
// sweep if that neighbourhood lies `a` cells inside one strip, else in the SEAM sweep (exactly once either way).
SMX_D bool batch_grid_tile_phase(const DevState& s, const BatchShared& bs, int tx) {
  const BStrips& st = bs.strips;
  if (st.n <= 1) return bs.sphase == STRIP_INTERIOR;
  const int cx0 = (tx - 1 < 0 ? 0 : tx - 1) << GRID_SHIFT;   // (the columns of the tile's region: one ring under strips, grid_ring)
  int cx1 = ((tx + 2) << GRID_SHIFT) - 1; if (cx1 > s.dimx - 1) cx1 = s.dimx - 1;
  bool interior = false;
  for (int g = 0; g < st.n; g++) {
    const int lo = st.x[g] + (g ? st.a : 0), hi = st.x[g + 1] - (g < st.n - 1 ? st.a : 0);
    if (cx0 >= lo && cx1 < hi) interior = true;
  }
  return (bs.sphase == STRIP_INTERIOR) == interior;
}

// generation end: the per-slot counters of slot `i` (device: wave-reduced before the atomics)
SMX_D void batch_slot_counters(const BatchShared& bs, bool wind, uint32_t i, unsigned long long (&acc)[4]) {
  if (wind) { const BWind& q = bs.wind[i]; if (q.state == B_AWAY) return; acc[0] += q.c_steps; acc[2] += q.c_casc; acc[3] += q.c_tr; }
  else { const BWater& q = bs.water[i]; if (q.state == B_AWAY) return; acc[0] += q.c_steps; if (q.toplevel) acc[1] += q.c_steps; acc[2] += q.c_casc; acc[3] += q.c_tr; }
}
SMX_D void batch_commit_counters(const DevState& s, bool wind, const unsigned long long (&acc)[4]) {
  if (acc[0]) SMX_A_ADD64(&s.ctr[wind ? C_STEPS_WIND : C_STEPS_WATER_ALL], acc[0]);
  if (!wind && acc[1]) SMX_A_ADD64(&s.ctr[C_STEPS_WATER_TOP], acc[1]);
  if (acc[2]) SMX_A_ADD64(&s.ctr[C_CASCADE_CALLS], acc[2]);
  if (acc[3]) SMX_A_ADD64(&s.ctr[C_CASCADE_TRANSFERS], acc[3]);
}

}  // namespace smx
