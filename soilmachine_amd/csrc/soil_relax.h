// soil_relax.h -- the RELAXED schedule of the batched engine (SMX_ENGINE_RELAXED): what north_star calls "particles advanced
// one wavefront per particle-batch with atomic deposit and a ... cascade pass" (SURVEY.md 7 step 6), made DETERMINISTIC.
//
// The batched engine (soil_batch.h) keeps every step of the reference atomic AND isolated: steps in one 3x3 neighbourhood
// exclude each other, so particles that drain into the same pit queue there, one per kernel launch -- the chain that bounds
// a phase (profiles/r02_epoch_latency.md). The relaxed schedule gives the isolation up, not the determinism:
//
//   step    (relax_water_step / relax_wind_step)  EVERY running particle takes one step of the reference (move() + the arithmetic of
//           interact(), water.h:43-121 / wind.h:54-136) on the map as the previous epoch left it. It does not write the map: the
//           erosion / deposit (water.h:96-107, wind.h:109-124) and the visit of the track / frequency plane (water.h:53, wind.h:49-52)
//           become REQUESTS queued on the cell's list, Particle::cascade at the position (water.h:113, wind.h:112,126,129) a flag on
//           that cell (the first particle to ask in an epoch enters it in the epoch's list of flagged cells).
//   apply   (relax_apply)    per cell: the queued requests in ENTRY order (slot, then step, then the reference's order inside a step), each
//           with exactly Layermap::remove / add (layermap.h:230-339): column invariants are exact.
//   filter  (relax_filter)   the flagged cells that can move anything at all (the conservative test Particle::cascade starts with,
//           on the map after the epoch's edits) enter the cascade list of their colour (x mod 3, y mod 3) -- except the isolated ones
//           (no other flagged cell within two cells), which cascade at once: same result, eight launches' worth of latency less;
//   cascade (relax_cascade)  the listed cells, ONCE each, with the reference's Particle::cascade(.., 0) -- in colour order (nine passes; on the
//           device one dataflow launch that keeps that order between cells whose blocks overlap: soilmx.hip k_relax_cascade_flow)
//           (the cascades of one pass touch disjoint 3x3 blocks, so they commute).
//   floods  water.h:123-145 reach 4 cells and spawn nested particles: they stay behind claim tiles (batch_water_epoch, run over the
//           epoch's list of ended particles), nested particles form the next generation as before.
//
// What is given up against the reference, beyond the order of particles (which the batched engine already gives up): the steps of
// one epoch all see the epoch's start state (an edit becomes visible one step later; a wind particle may take up to bs.rsteps steps
// in one epoch), a cell that several particles step onto in one epoch cascades once, a wind lift's yield (wind.h:110) is what
// Layermap::remove would return on the top section as the step SAW it, and wind cascades run without their one level of recursion
// (particle.h:96-97 with transferloop 1: it only triggers when a transfer exhausts a section).
// Nothing depends on which lane or in which order the hardware runs anything: the result is a function of the inputs
// (tests/test_relaxed.py: host threads in any interleaving == device, tests/test_gpu_relaxed.py).
// Parity: tier P2 like the batched engine, gated on the device against the exact engine (bench.py parity_vs_exact, tools/p2_envelope.py).
#pragma once
#include "soil_batch.h"

namespace smx {

constexpr uint32_t R_TRACK = 1, R_ERODE = 2, R_DEPOSIT = 4, R_WINDFREQ = 8, R_LIFT = 16;

// Append to a list whose length is an atomic counter. On the device the lanes of a wavefront that append to the SAME counter at the
// same time combine into one atomic (64 000 single atomics on one word serialise in the L2: the step kernel took 180 us with them, r03d);
// where an entry lands in the list is the hardware's business either way -- no result depends on list order.
#ifdef SMX_HOSTSIM
SMX_D uint32_t relax_append(uint32_t* counter, bool want) { return want ? SMX_A_ADD(counter, 1u) : NIL; }
#else
SMX_D uint32_t relax_append(uint32_t* counter, bool want) {
  const unsigned long long m = __ballot(want);
  if (!want) return NIL;
  const int leader = __ffsll((long long)m) - 1, lane = (int)(threadIdx.x & 63u);
  uint32_t base = 0u;
  if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
  base = (uint32_t)__shfl((int)base, leader);
  return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}
#endif
SMX_D uint32_t relax_colour(const DevState& s, uint32_t c) {
  const uint32_t x = c / (uint32_t)s.dimy, y = c - x * (uint32_t)s.dimy;
  return (x % 3u) * 3u + (y % 3u);
}
// queue one request entry on its cell's list; ask for the cascade of `casc` (once per cell and epoch)
SMX_D void relax_push(const DevState& s, const BatchShared& bs, uint32_t e, const RReq& r, uint32_t epoch) {
  bs.req[e] = r;
  if (r.kind) bs.rnext[e] = SMX_A_XCHG(&bs.rhead[r.cell], e);
  const uint32_t par = epoch & 1u;
  const bool flag = r.casc != NIL && SMX_A_XCHG(&bs.cflag[r.casc], bs.rtag) != bs.rtag;   // the first one to ask in this epoch
  const uint32_t k = relax_append(&bs.ctrl[BC_NFLAG + par], flag);
  if (flag) { if (k < bs.flag_cap) bs.flagl[par][k] = r.casc; else SMX_A_ADD(&bs.ctrl[BC_LIST_LOST], 1u); }
  (void)s;
}
SMX_D void relax_flood_later(const BatchShared& bs, uint32_t slot, uint32_t parity, bool want = true) {   // the slot's flood joins that epoch's flood list
  const uint32_t k = relax_append(&bs.ctrl[BC_NFLOOD + parity], want);
  if (want) { if (k < bs.list_cap) bs.flist[parity][k] = slot; else SMX_A_ADD(&bs.ctrl[BC_LIST_LOST], 1u); }
}
// Floods are long, branchy paths (~150 us each on a lane); while nearly everybody is still running, the few particles that ended
// early wait for every fourth epoch. Once the ended ones are 1/64 of the runners (or nobody runs), every epoch floods.
// A function of the epoch's own counts: not of chunking, not of how the host drives the epochs.
// (Under column strips the counts are one device's share, so every epoch floods there: the result must not depend on the partition.)
// Round 4: the threshold is 1/64 (round 3 shipped 1/16). A claim issued in one flood epoch is only good for the next epoch, so floods
// progress when two flood epochs follow each other, i.e. once the threshold holds; with 1/16 the established wet regime ran +11 % water
// steps against the reference (control <= +5 %), with 1/64 +2 % at equal height RMSE (profiles/r03_flood_rule_cpu.md, r04_p2_*).
constexpr unsigned long long RELAX_FLOOD_SHARE = 64ull;
SMX_D bool relax_floods_due(const BatchShared& bs, uint32_t epoch, uint32_t nflood, uint32_t nrun) {
  return bs.strips.n > 1 || (epoch & 3u) == 3u || (unsigned long long)nflood * RELAX_FLOOD_SHARE >= (unsigned long long)nrun;
}
// after the epoch's edits are applied: which flagged cells can move anything at all? (Particle::cascade starts with the same
// conservative test, Sim::cascade_precheck: the others are done) -- the rest go to their colour's cascade list.
// ISOLATED cells cascade right here (round 4): a flagged cell with no other flagged cell within two cells of it (Chebyshev) shares
// no column of its 3x3 block with any other flagged cell's block -- neither with a block this pass is still testing, nor with one a
// colour pass will work on. Cascades on disjoint blocks commute and each tests only its own block, so running it now, in the filter's
// own launch, gives exactly the state the nine colour passes would leave (the restatement, oracle/soil_oracle.cpp r_water_generation,
// still runs every cell through its colour: tests/test_relaxed.py, test_gpu_relaxed.py compare full states). What is left for the
// colour passes is the residue: cells in pits and channels where particles crowd (measured: profiles/r04_relaxed.md).
template <class P>
SMX_D void relax_filter(Sim<P>& sim, const BatchShared& bs, uint32_t c, bool valid, uint32_t epoch) {
  const DevState& s = sim.s;
  bool need = false; uint32_t col = 0u;
  if (valid) {
    const int x = (int)(c / (uint32_t)s.dimy), y = (int)(c - (uint32_t)x * (uint32_t)s.dimy);
    Blk b;
    sim.template load_block<2>(x, y, b);
    need = sim.cascade_precheck(b);
    if (!need) sim.n_casc++;                                  // (a listed cell counts in cascade_blk)
    else {
      bool alone = true;                                      // (flags outside the columns this context holds cannot be set by it)
      const int xa = x - 2 < s.x_lo ? s.x_lo : x - 2, xb = x + 2 >= s.x_hi ? s.x_hi - 1 : x + 2;
      const int ya = y - 2 < 0 ? 0 : y - 2, yb = y + 2 >= s.dimy ? s.dimy - 1 : y + 2;
      for (int xx = xa; xx <= xb; xx++)
        for (int yy = ya; yy <= yb; yy++)
          if ((xx != x || yy != y) && bs.cflag[(size_t)xx * (size_t)s.dimy + (size_t)yy] == bs.rtag) alone = false;
      if (alone) { sim.cascade_blk(x, y, b); need = false; }
    }
    col = (uint32_t)((x % 3) * 3 + (y % 3));
  }
  const uint32_t par = epoch & 1u;
#pragma unroll 1
  for (uint32_t q = 0; q < 9u; q++) {
    const bool mine = need && col == q;
    const uint32_t k = relax_append(&bs.ctrl[BC_NCASC + 9u * par + q], mine);
    if (mine) { if (k < bs.clist_cap) bs.clist[par][(size_t)q * bs.clist_cap + k] = c; else SMX_A_ADD(&bs.ctrl[BC_LIST_LOST], 1u); }
    if (mine && k < bs.clist_cap && bs.cstate) bs.cstate[c] = 2u * bs.rtag;       // (the colour lists as ONE dataflow launch: soilmx.hip k_relax_cascade_flow)
  }
}

template <class P>
SMX_D void relax_water_step(Sim<P>& sim, const BatchShared& bs, uint32_t slot, uint32_t epoch) {
  BWater& q = bs.water[slot];
  uint32_t state = q.state;
  if (state != B_RUN && state != B_NEW) { bs.rcount[slot] = 0u; return; }
  if (state == B_NEW && bs.birth_gap && q.toplevel && epoch < (slot / bs.birth_n) * bs.birth_gap) { bs.rcount[slot] = 0u; return; }   // staggered generations: not born yet
  const DevState& s = sim.s;
  const SoilP* soils = sim.soils;
  Water w = q.w;
  uint32_t nent = 0;
  const uint32_t base = (uint32_t)bs.rstride * slot;
  const int nsteps = bs.rsteps < 1 ? 1 : bs.rsteps;         // steps per epoch (smx_set_relax_water; 1 = every step sees the previous one's edits applied)
  for (int step = 0; step < nsteps && (state == B_RUN || state == B_NEW); step++) {
    Blk A;
    sim.template load_block<0>(w.ix, w.iy, A);
    Water t = w;
    const bool mv = sim.template water_move<true>(t, false, A);                                 // water.h:43-73, no side effects
    if (bs.strips.n > 1) {                                    // column strips: the step (3x3 around ipos and around the new position) must lie
      const int nx = mv ? (int)roundf(t.pos.x) : w.ix, ny = mv ? (int)roundf(t.pos.y) : w.iy; // inside what this kind of chunk may touch
      const int x0 = (w.ix < nx ? w.ix : nx) - 1, x1 = (w.ix > nx ? w.ix : nx) + 1;
      if (!beligible_x(s, bs, x0, x1)) break;                 // dormant until the other kind of chunk (this step has changed nothing yet)
      bmark_dirty_cells(s, bs, x0, x1, (w.iy < ny ? w.iy : ny) - 1, (w.iy > ny ? w.iy : ny) + 1);
    }
    if (step == 0) (void)relax_append(&bs.ctrl[BC_NRUN + (epoch & 1u)], true);                 // how many step in this epoch (relax_floods_due)
    if (state == B_NEW) { t.contains = soils[Sim<P>::vsurf(A.v8)].transports; state = B_RUN; } // WaterParticle(map) water.h:15-17 (move() ignores it)
    RReq r;
    r.cell = (uint32_t)A.c0; r.vol = w.volume; r.amount = 0.0; r.type = 0u; r.kind = R_TRACK; r.casc = NIL;   // water.h:53 track[ipos] += volume
    w = t;
    if (mv) {
      q.c_steps++;
      const View cv = A.v8;
      const float fr = A.fr;
      const int nx = (int)roundf(w.pos.x), ny = (int)roundf(w.pos.y);
      Blk B;
      sim.template load_block<2>(nx, ny, B);
      // the arithmetic of WaterParticle::interact (water.h:75-121), statement by statement as in Sim::water_interact
      const SoilP& param = soils[w.surface];
      const double hb = sim.bilinear_blk(w.pos, nx, ny, B);
      double c_eq = param.solubility * (Sim<P>::vh(cv) - hb) * (double)s.scale / 80.0;
      if (c_eq < 0.0) c_eq = 0.0;
      if (c_eq > 1.0) c_eq = 1.0;
      if ((double)(soils[w.contains].erosionrate) < fr) w.contains = soils[w.contains].erodes;
      const double cdiff = c_eq - w.sediment;
      if (cdiff > 0) {
        w.sediment += param.equrate * cdiff;
        w.contains = soils[Sim<P>::vsurf(cv)].transports;
        r.amount = param.equrate * cdiff * w.volume; r.kind |= R_ERODE;
      } else if (cdiff < 0) {
        w.sediment += soils[w.contains].equrate * cdiff;
        r.amount = -soils[w.contains].equrate * cdiff * w.volume; r.type = w.contains; r.kind |= R_DEPOSIT;
      }
      r.casc = (uint32_t)B.c0;                                // Particle::cascade(pos, .., 0) (:113)
      w.sediment /= (1.0 - w.evaprate);
      if (w.sediment > 1.0) w.sediment = 1.0;
      w.volume *= (1.0 - w.evaprate);
      w.ix = nx; w.iy = ny;
      if (!(w.volume > 0.01)) state = B_ENDED;
    } else state = B_ENDED;
    relax_push(s, bs, base + nent, r, epoch);
    nent++;
  }
  if (state == B_ENDED) {                                     // flood() or nothing (water.h:125): behind claim tiles, from this epoch on
    q.rect = BRect{0, -1, 0, -1}; q.hint = 0;
    if (!water_floods(w)) state = B_DONE;
  }
  relax_flood_later(bs, slot, epoch & 1u, state == B_ENDED);
  q.w = w; q.state = state;
  bs.rcount[slot] = nent;
}

// WindParticle (wind.h:54-136), the same way; up to bs.rsteps steps per epoch (a wind particle flies for thousands of steps and
// touches the ground now and then). A wind step can edit TWO cells (half of what falls out lands at npos, half at ipos,
// wind.h:120-124) and asks for up to two cascades: its first entry carries the npos deposit (the reference adds there first) and the
// cascade of ipos, its second the ipos edit (lift or deposit), the visit of the frequency plane and the cascade of npos.
template <class P>
SMX_D void relax_wind_step(Sim<P>& sim, const BatchShared& bs, uint32_t slot, uint32_t epoch) {
  BWind& q = bs.wind[slot];
  uint32_t state = q.state;
  if (state != B_RUN && state != B_NEW) { bs.rcount[slot] = 0u; return; }
  const DevState& s = sim.s;
  const SoilP* soils = sim.soils;
  Wind d = q.d;
  uint32_t nent = 0;
  const uint32_t base = (uint32_t)bs.rstride * slot;
  for (int step = 0; step < bs.rsteps && state != B_DONE; step++) {
    Blk A;
    sim.template load_block<1>(d.ix, d.iy, A);
    RReq r0, r1;
    r0.cell = 0u; r0.vol = 0.0; r0.amount = 0.0; r0.type = 0u; r0.kind = 0u; r0.casc = NIL;
    r1 = r0;
    if (state == B_NEW) {                                     // WindParticle(map) wind.h:19-21 + the gate of wind.h:56-57
      d.contains = soils[Sim<P>::vsurf(A.v8)].transports;
      state = B_RUN;
      if (soils[d.contains].suspension == 0.0) { state = B_DONE; break; }
    }
    r1.cell = (uint32_t)A.c0; r1.kind = R_WINDFREQ;           // wind.h:49-52 frequency[ipos] = 0.5*frequency + 0.5
    Wind t = d;
    const bool mv = sim.template wind_move<true>(t, A);       // wind.h:54-92, no side effects
    if (bs.strips.n > 1) {                                    // column strips: see relax_water_step
      const int ex = mv ? (int)roundf(t.pos.x) : d.ix, ey = mv ? (int)roundf(t.pos.y) : d.iy;
      const int x0 = (d.ix < ex ? d.ix : ex) - 1, x1 = (d.ix > ex ? d.ix : ex) + 1;
      if (!beligible_x(s, bs, x0, x1)) break;                 // dormant until the other kind of chunk (this step has changed nothing yet)
      bmark_dirty_cells(s, bs, x0, x1, (d.iy < ey ? d.iy : ey) - 1, (d.iy > ey ? d.iy : ey) + 1);
    }
    d = t;
    if (mv) {
      q.c_steps++;
      const View cv = A.v8;
      const int nx = (int)roundf(d.pos.x), ny = (int)roundf(d.pos.y);
      Blk B;
      sim.template load_block<2>(nx, ny, B);
      const SoilP& param = soils[d.surface];
      const double hb = sim.bilinear_blk(d.pos, nx, ny, B);
      const View nv = B.v8;
      if (d.height <= hb * (float)s.scale / 80.0f) {          // wind.h:102-113: on the surface -- lift
        if (param.transports == d.contains) {
          const double force = sqrtf(dot3(d.speed, d.speed)) * (Sim<P>::vh(nv) - d.height) * (float)s.scale / 80.0f * (1.0f - d.sediment);
          const double amount = param.suspension * force;
          double diff = 0.0;                                  // Layermap::remove(ipos, amount) on the section as seen (layermap.h:310-339)
          if (cv.type != EMPTY && cv.size > 0.0 && amount > 0.0 && amount - cv.size >= 0.0) diff = amount - cv.size;
          d.sediment += (amount - diff);
          r1.amount = amount; r1.kind |= R_LIFT;
          r0.casc = (uint32_t)A.c0;                           // cascade(ipos)
        }
      } else if (param.suspension > 0.0) {                    // wind.h:115-131: in the air -- half falls at npos, half at ipos
        d.sediment -= soils[d.contains].suspension * d.sediment;
        const double dep = 0.5f * soils[d.contains].suspension * d.sediment;
        r0.cell = (uint32_t)B.c0; r0.amount = dep; r0.type = d.contains; r0.kind = R_DEPOSIT; r0.casc = (uint32_t)A.c0;
        r1.amount = dep; r1.type = d.contains; r1.kind |= R_DEPOSIT; r1.casc = (uint32_t)B.c0;
      }
      d.ix = nx; d.iy = ny;
    } else state = B_DONE;
    relax_push(s, bs, base + nent, r0, epoch);
    relax_push(s, bs, base + nent + 1u, r1, epoch);
    nent += 2u;
  }
  if (state == B_DONE) q.rect = BRect{0, -1, 0, -1};
  q.d = d; q.state = state;
  bs.rcount[slot] = nent;
}

// One cell's queued requests, applied by the thread whose entry is the list's head, in ascending entry order (the list itself
// is in arrival order, which is the hardware's business): repeated selection of the smallest entry above the last one applied
// -- lists are one to three entries long except in pits, and need no storage this way.
template <class P>
SMX_D void relax_apply(Sim<P>& sim, const BatchShared& bs, uint32_t entry) {
  const DevState& s = sim.s;
  if (bs.req[entry].kind == 0u) return;                     // (nothing queued under this entry)
  const uint32_t c = bs.req[entry].cell;
  if (bs.rhead[c] != entry) return;
  const int x = (int)(c / (uint32_t)s.dimy), y = (int)(c - (uint32_t)x * (uint32_t)s.dimy);
  const size_t fi = (size_t)y * s.dimx + x;
  uint32_t last = 0u; bool first = true;
  while (true) {
    uint32_t best = NIL;
    for (uint32_t e = entry; e != NIL; e = bs.rnext[e]) if ((first || e > last) && e < best) best = e;
    if (best == NIL) break;
    first = false; last = best;
    const RReq r = bs.req[best];
    if (r.kind & R_TRACK) s.wtrack[fi] = (float)((double)s.wtrack[fi] + r.vol);                 // updatefrequency water.h:348-351
    if (r.kind & R_WINDFREQ) s.windfreq[fi] = (float)(0.5 * s.windfreq[fi] + 0.5f);             // wind.h:49-52
    if (r.kind & R_LIFT) (void)sim.remove(x, y, r.amount);                                      // wind.h:109 (one call: the top section only)
    if (r.kind & R_ERODE) {                                                                     // water.h:96-99
      double diff = sim.remove(x, y, r.amount);
      while (fabs(diff) > 1E-8) diff = sim.remove(x, y, diff);
    } else if (r.kind & R_DEPOSIT) sim.add(x, y, r.amount, r.type);                             // water.h:105-106, wind.h:120-124
  }
  bs.rhead[c] = NIL;
}
// all entries one slot queued in this epoch
template <class P>
SMX_D void relax_apply_slot(Sim<P>& sim, const BatchShared& bs, uint32_t slot) {
  const uint32_t n = bs.rcount[slot], base = (uint32_t)bs.rstride * slot;
  for (uint32_t k = 0; k < n; k++) relax_apply(sim, bs, base + k);
}

// Particle::cascade(pos, .., 0) (particle.h:24-101) at one listed cell (exclusive: the cells of one colour list are >= 3 apart)
template <class P>
SMX_D void relax_cascade(Sim<P>& sim, uint32_t c) {
  const DevState& s = sim.s;
  const int x = (int)(c / (uint32_t)s.dimy), y = (int)(c - (uint32_t)x * (uint32_t)s.dimy);
  Blk b;
  sim.template load_block<2>(x, y, b);
  sim.cascade_blk(x, y, b);
}

}  // namespace smx
