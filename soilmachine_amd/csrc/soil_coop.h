// soil_coop.h -- wave-cooperative block access (device only).
//
// One particle per wavefront. The particle's physics is a scalar dependent chain and runs on lane 0; but the one
// memory round trip of a step -- the 3x3 block around the new position (soil_core.h "the carried 3x3 block") -- is
// nine independent cells, and so is everything the speculative engine does per cell (stamp check, claim check,
// owner status, CAS, undo-log record). Here lanes 0..8 each take ONE cell: address arithmetic, loads, the
// acquisition protocol of soil_spec.h and the 64-byte log write happen as single wave instructions instead of
// nine scalar copies; the nine (size, floor, type) views are then handed to lane 0 with v_readlane. Control flow
// stays wave-uniform: decisions are taken by lane 0 and broadcast with v_readfirstlane.
// (PMC before: 2 300 instructions and 67 loads per speculative water step, of which ~1 100 are the physics.)
#pragma once
#include "soil_core.h"
#include "soil_serial.h"
#include "soil_spec.h"

namespace smx {

#ifdef SMX_HOSTSIM
// host emulation (tests/hostsim): a "wave" of one lane -- exercises the loop structure, not the lane protocol
SMX_D int coop_lane() { return 0; }
SMX_D int rfl(int v) { return v; }
template <class P, int MODE>
SMX_D bool coop_load_block(Sim<P>& sim, int cx, int cy, Blk& b, bool = false, int = 0, int = 0) { sim.template load_block<MODE>(cx, cy, b); return !sim.ab; }
#else
SMX_D int coop_lane() { return (int)(threadIdx.x & 63u); }
SMX_D int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <int K> SMX_D uint32_t rl32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, K); }
template <int K> SMX_D float rlf(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), K)); }
template <int K> SMX_D double rl64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), K), hi = __builtin_amdgcn_readlane((int)(b >> 32), K);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

SMX_D uint32_t sh_cap(const ScoutPolicy& p) { return p.sh.scout_cap; }

// Fetch (and, for the speculative engine, acquire + log) the 3x3 block around (cx, cy). Called by ALL lanes with
// wave-uniform arguments; `b` is valid in every lane afterwards (lane 0 is the one that uses it). Returns false
// (uniform) if the block could not be acquired -- then nothing of it is held by this particle.
template <class P, int MODE>
SMX_D bool coop_load_block(Sim<P>& sim, int cx, int cy, Blk& b, bool have_prev = false, int pcx = 0, int pcy = 0) {
  const DevState& s = sim.s;
  const int lane = coop_lane();
  const int k = lane < 8 ? lane : 8;
  const int dx = lane < 8 ? nb_dx(k) : 0, dy = lane < 8 ? nb_dy(k) : 0;
  const int nx = cx + dx, ny = cy + dy;
  const bool in = !(nx >= s.dimx || ny >= s.dimy || nx < 0 || ny < 0);
  const bool valid = lane < 9 && in;                         // lanes 9.. shadow the centre and take no part
  const size_t c0 = (size_t)cx * s.dimy + cy, f0 = (size_t)cy * s.dimx + cx;
  const size_t c = valid ? (size_t)nx * s.dimy + ny : c0;
  const size_t fi = valid ? (size_t)ny * s.dimx + nx : f0;
  const Sec r = s.cells[c];
  float fr = 0.f, tr = 0.f;
  if constexpr (MODE == 0) { fr = s.wfreq[f0]; tr = s.wtrack[f0]; }
  else if constexpr (MODE == 1) { tr = s.windfreq[f0]; }
  bool ok = true;
  if constexpr (std::is_same<P, SpecPolicy>::value) {
    // Acquisition in ONE memory round trip for the usual cell. Cells of the previous block (have_prev: the block
    // around (pcx, pcy) is held entirely by this particle) need nothing but their record. For every other cell
    // the record, the claim, the undo words AND a compare-and-swap of the stamp from 0 ("untouched in this
    // phase") to `me` are issued together:
    //   old == 0     the cell is ours (a stamp never returns to 0 inside a phase, so the record read alongside is
    //                the untouched one) -- unless a lower pending particle claims it: then it is handed back
    //   old == me    ours already (revisited)
    //   otherwise    the owner decides, as in SpecPolicy::acquire_log: later particle / poisoned -> fail; finished
    //                lower particle -> status load + second CAS (two more trips; rare outside the late rounds)
    SpecPolicy& pol = sim.pol;
    const SpecShared& sh = pol.sh;
    const uint32_t me = pol.me, pid = pol.pid, cbase = sh.ctrl[SC_BASE];
    const bool mine = valid && have_prev && (uint32_t)(nx - pcx + 1) <= 2u && (uint32_t)(ny - pcy + 1) <= 2u;
    const bool need = valid && !mine;
    uint32_t cl = NOCLAIM, fl = 0u, old = me;
    float utr = 0.f, uwf = 0.f;
    if (need) {
      cl = sh.claim[c]; fl = s.flags[c]; utr = s.wtrack[fi]; uwf = s.windfreq[fi];
      old = SMX_A_CAS(&sh.stamp[c], 0u, me);
    }
    const bool fresh = need && old == 0u;
    const bool other = need && old != 0u && old != me;
    const bool failA = other && (old == POISON || old > me);
    if (failA && old != POISON) SMX_A_STORE(&sh.part[old - 1].kill, 1u);   // a later particle got here first: undo it
    const bool failCl = (fresh || (other && !failA)) && cl < pid;           // yield to a lower pending particle
    if (fresh && failCl) SMX_A_STORE(&sh.stamp[c], 0u);
    const bool takeover = other && !failA && !failCl;
    // owners below the committed prefix (ctrl[SC_BASE], fixed while this kernel runs) need no status load
    uint32_t st = ST_COMMITTED;
    if (takeover && old - 1u >= cbase) st = sh.part[old - 1].status;
    const bool failB = takeover && (st == ST_PENDING || st == ST_DONE_NEW || st == ST_SUSPENDED);
    const bool dep = takeover && st == ST_DONE_OLD;
    const unsigned long long fm = __ballot(failA || failCl || failB);
    uint32_t old2 = old;
    if (!fm && takeover) old2 = SMX_A_CAS(&sh.stamp[c], old, me);
    const bool got = (fresh && !failCl) || (takeover && !fm && old2 == old);
    const bool failC = takeover && !fm && old2 != old && old2 != me;          // somebody took it meanwhile
    const uint32_t prev = fresh ? 0u : old;
    const unsigned long long gm = __ballot(got);
    const uint32_t cnt = (uint32_t)__popcll(gm);
    if (cnt) {                                               // (cells acquired are logged even if the block fails)
      // lane 0 makes sure the log chunks for entries [base, base+cnt) exist
      uint32_t base = 0, cidA = 0, cidB = 0, ovf = 0;
      if (lane == 0) {
        base = pol.nlog;
        const uint32_t first = base / LOG_CHUNK, last = (base + cnt - 1) / LOG_CHUNK;
        for (uint32_t ch = first; ch <= last; ch++) {
          if (ch >= pol.nchunks) {
            if (ch >= (uint32_t)MAX_CHUNKS) { ovf = 1; break; }
            const uint32_t cid = SMX_A_ADD(&sh.ctrl[SC_CHUNK_CURSOR], 1u);
            if (cid >= sh.arena_chunks) { ovf = 1; break; }
            sh.chunk_tab[(size_t)pid * MAX_CHUNKS + ch] = cid;
            pol.nchunks = ch + 1;
          }
        }
        if (!ovf) { cidA = sh.chunk_tab[(size_t)pid * MAX_CHUNKS + first]; cidB = sh.chunk_tab[(size_t)pid * MAX_CHUNKS + last]; }
        else pol.overflow = 1;
      }
      base = (uint32_t)rfl((int)base); cidA = (uint32_t)rfl((int)cidA); cidB = (uint32_t)rfl((int)cidB); ovf = (uint32_t)rfl((int)ovf);
      if (ovf) {                                             // log full: give the cells back, the attempt is void
        if (got) SMX_A_STORE(&sh.stamp[c], prev);
        if (lane == 0) pol.fail_cell = (uint32_t)c0;
        return false;
      }
      if (got) {
        const uint32_t e = base + (uint32_t)__popcll(gm & ((1ull << lane) - 1ull));
        const uint32_t cid = (e / LOG_CHUNK == base / LOG_CHUNK) ? cidA : cidB;
        SpecLog* ent = sh.arena + (size_t)cid * LOG_CHUNK + (e % LOG_CHUNK);
        ent->cell = (uint32_t)c; ent->prev = prev; ent->rec = r; ent->wtrack = utr; ent->windfreq = uwf; ent->flags = fl;
      }
      if (lane == 0) {
        pol.nlog = base + cnt;
        if (pol.nlog % LOG_CHUNK) pol.chunk = sh.arena + (size_t)cidB * LOG_CHUNK;
      }
      unsigned long long dm = __ballot(dep && got);
      while (dm) {                                           // cells taken over from tentative particles: dependencies
        const int dk = __builtin_ctzll(dm);
        dm &= dm - 1;
        const uint32_t owner = (uint32_t)__shfl((int)old, dk) - 1u;
        if (lane == 0) pol.add_dep(owner);
      }
    }
    const unsigned long long bad = fm | __ballot(failC);
    if (bad) {
      const uint32_t fc = (uint32_t)__shfl((int)(uint32_t)c, __builtin_ctzll(bad));
      if (lane == 0) pol.fail_cell = fc;
      return false;
    }
    if (lane == 0) pol.last = c0;
  }
  if constexpr (std::is_same<P, ScoutPolicy>::value) {       // read-only scout: the block's cells go to the footprint list
    ScoutPolicy& pol = sim.pol;
    const unsigned long long vm = __ballot(valid);
    const uint32_t base = (uint32_t)rfl((int)pol.len);
    const uint32_t e = base + (uint32_t)__popcll(vm & ((1ull << lane) - 1ull));
    if (valid && e < sh_cap(pol)) pol.out[e] = (uint32_t)c;
    if (lane == 0) {
      const uint32_t nl = base + (uint32_t)__popcll(vm);
      pol.len = nl < sh_cap(pol) ? nl : sh_cap(pol);
      pol.last2 = pol.last; pol.last = c0;
    }
  }
  // hand the nine views to lane 0 (v_readlane with constant lane numbers; executed by every lane)
  // (the asm pins the broadcast values in VGPRs: left to itself the compiler keeps all 27 of them, for both
  // ping-pong blocks, in SGPRs -- they are wave-uniform -- and then spills hundreds of SGPRs inside the step loop)
#define X(i) b.v##i.size = rl64<i>(r.size); b.v##i.floor = rl64<i>(r.floor); b.v##i.type = rl32<i>(r.type); \
  asm volatile("" : "+v"(b.v##i.size), "+v"(b.v##i.floor), "+v"(b.v##i.type));
  SMX_FOR9(X)
#undef X
  b.c0 = c0;
  b.inb = (uint32_t)(__ballot(in && lane < 8) & 0xffull);
  b.fr = fr; b.tr = tr;
  return ok;
}

#endif  // SMX_HOSTSIM

// while(move && interact) for the wave: soil_core.h water_loop with the block fetches done cooperatively
template <class P>
SMX_D void water_loop_coop(Sim<P>& sim, Water& p, bool toplevel, bool resume) {
  const bool l0 = coop_lane() == 0;
  Blk A, B;
  int ucx = 0, ucy = 0;                                      // centre of the block the particle holds (wave-uniform)
  if (resume) { if (l0) { sim.pol.load_blk(A); ucx = p.ix; ucy = p.iy; } ucx = rfl(ucx); ucy = rfl(ucy); }
  else {
    if (l0) { p.ix = (int)roundf(p.pos.x); p.iy = (int)roundf(p.pos.y); ucx = p.ix; ucy = p.iy; }
    ucx = rfl(ucx); ucy = rfl(ucy);
    if (!coop_load_block<P, 0>(sim, ucx, ucy, A)) { if (l0) sim.ab = true; return; }
  }
  while (true) {
#define SMX_COOP_WATER_STEP(CUR, NXT)                                                                         \
    {                                                                                                         \
      int go = 1;                                                                                             \
      if (l0 && !resume) go = sim.water_move(p, toplevel, CUR) ? 1 : 0;                                       \
      resume = false;                                                                                         \
      if (!rfl(go)) return;                                                                                   \
      int nx = 0, ny = 0;                                                                                     \
      if (l0) { nx = (int)roundf(p.pos.x); ny = (int)roundf(p.pos.y); }                                       \
      nx = rfl(nx); ny = rfl(ny);                                                                             \
      const bool ok = coop_load_block<P, 0>(sim, nx, ny, NXT, true, ucx, ucy);                                \
      int r = -1;                                                                                             \
      if (l0) { if (ok) r = sim.water_interact(p, CUR, NXT, true); else sim.ab = true; }                      \
      r = rfl(r);                                                                                             \
      ucx = nx; ucy = ny;                                                                                     \
      if (r <= 0) { if (l0 && r < 0 && sim.susp_ok) { sim.susp = 2; sim.pol.save_blk(CUR); } return; }        \
    }
    SMX_COOP_WATER_STEP(A, B)
    A = B;                                                   // (one body, as in wind_loop_coop)
#undef SMX_COOP_WATER_STEP
  }
}

template <class P>
SMX_D void wind_loop_coop(Sim<P>& sim, Wind& p, bool resume) {
  const bool l0 = coop_lane() == 0;
  Blk A, B;
  int ucx = 0, ucy = 0;
  if (resume) { if (l0) { sim.pol.load_blk(A); ucx = p.ix; ucy = p.iy; } ucx = rfl(ucx); ucy = rfl(ucy); }
  else {
    int live = 0;
    if (l0) {
      live = !(sim.soils[p.contains].suspension == 0.0);     // wind.h:56-57
      p.ix = (int)roundf(p.pos.x); p.iy = (int)roundf(p.pos.y); ucx = p.ix; ucy = p.iy;
    }
    if (!rfl(live)) return;
    ucx = rfl(ucx); ucy = rfl(ucy);
    if (!coop_load_block<P, 1>(sim, ucx, ucy, A)) { if (l0) sim.ab = true; return; }
  }
  while (true) {
#define SMX_COOP_WIND_STEP(CUR, NXT)                                                                          \
    {                                                                                                         \
      int go = 1;                                                                                             \
      if (l0 && !resume) go = sim.wind_move(p, CUR) ? 1 : 0;                                                  \
      resume = false;                                                                                         \
      if (!rfl(go)) return;                                                                                   \
      int nx = 0, ny = 0;                                                                                     \
      if (l0) { nx = (int)roundf(p.pos.x); ny = (int)roundf(p.pos.y); }                                       \
      nx = rfl(nx); ny = rfl(ny);                                                                             \
      const bool ok = coop_load_block<P, 1>(sim, nx, ny, NXT, true, ucx, ucy);                                \
      int r = -1;                                                                                             \
      if (l0) { if (ok) r = sim.wind_interact(p, CUR, NXT, true); else sim.ab = true; }                       \
      r = rfl(r);                                                                                             \
      ucx = nx; ucy = ny;                                                                                     \
      if (r <= 0) { if (l0 && r < 0 && sim.susp_ok) { sim.susp = 2; sim.pol.save_blk(CUR); } return; }        \
    }
    SMX_COOP_WIND_STEP(A, B)
    A = B;                                                   // the new block becomes the carried one (B is dead until the next load):
                                                             // one body instead of an A/B ping-pong halves the kernel's code
#undef SMX_COOP_WIND_STEP
  }
}

// What follows a top-level particle's loop -- flood, water-table cascade, nested particles (water.h:123-283) -- is
// rare, long and scalar, and is kept OUT of line. Two reasons, both measured on gfx950:
//  * inlined and specialised into the wave kernel it pushed the step loop's register allocation over the edge, and
//    the -O3 build of exactly that variant produced wrong maps (DESIGN.md "A compiler note");
//  * it must not take the step loop's objects by reference: an object whose address escapes into a call lives in
//    scratch memory for its whole life, and the loop then pays a scratch access for every touch of the particle
//    and of the policy's cursors (PMC: 78 vector-memory loads per water step, 63 % of wave cycles waiting).
// So policy and particle travel BY VALUE, the tail builds its own Sim, and the results come back by value.
template <class P> struct TailRes { P pol; Water p; bool ab; uint32_t steps; };
template <class P>
__attribute__((noinline)) SMX_D TailRes<P> coop_water_tail(DevState s, const SoilP* soils, P pol, Water p) {
  TailRes<P> r{pol, p, false, 0u};
  Sim<P> sim(s, soils, r.pol);
  Frame st[MAX_FRAMES]; int depth = 0;
  sim.water_drive(r.p, true, true, st, depth, false, true);
  r.ab = sim.ab;
  r.steps = (uint32_t)sim.n_steps_w_all;
  sim.flush_counters();
  return r;
}

// round 0: the read-only trajectory of one particle, walked by a whole wave (cf. spec_scout_one in soil_spec.h)
template <bool WIND>
SMX_D void spec_scout_one_coop(const DevState& s, const SoilP* soils, const SpecShared& sh, uint32_t pid) {
  const bool l0 = coop_lane() == 0;
  ScoutPolicy pol(sh, pid);
  Sim<ScoutPolicy> sim(s, soils, pol);
  const uint32_t k = 2u * pid;                               // particle id = slot of the draw-pair sequence
  const int ry = (int)(sh.draws[k] >> 1) % s.dimy, rx = (int)(sh.draws[k + 1] >> 1) % s.dimx;
  if constexpr (WIND) {
    Wind p;
    p.pos = {0.f, 0.f}; p.speed = {0.f, 0.f, 0.f}; p.sediment = p.height = p.sheight = 0.0; p.ix = p.iy = 0; p.surface = p.contains = 0;
    if (l0) sim.wind_init(p, rx, ry);
    wind_loop_coop(sim, p, false);
  } else {
    Water p;
    p.pos = {0.f, 0.f}; p.speed = {0.f, 0.f}; p.volume = p.sediment = p.evaprate = 0.0; p.spill = 0; p.ix = p.iy = 0;
    p.friction = 0.f; p.surface = p.contains = 0;
    if (l0) sim.water_init(p, rx, ry);
    water_loop_coop(sim, p, true, false);
    if (l0) {                                                // (read-only: claims where a flood would act, water.h:123)
      TailRes<ScoutPolicy> r = coop_water_tail<ScoutPolicy>(s, soils, pol, p);
      r.pol.finish();
    }
    return;
  }
  if (l0) pol.finish();
}

// one particle of a speculative phase, executed by a whole wave (cf. spec_exec_one in soil_spec.h)
template <bool WIND>
SMX_D void spec_exec_one_coop(const DevState& s, const SoilP* soils, const SpecShared& sh, uint32_t pid) {
  const bool l0 = coop_lane() == 0;
  if ((sh.mode & 1u) && sh.part[pid].blocked) { if (l0) SMX_A_ADD(&sh.ctrl[SC_SKIPPED], 1u); return; }
  if (l0) SMX_A_ADD(&sh.ctrl[SC_EXECUTED], 1u);
  const bool resume = sh.part[pid].status == ST_SUSPENDED;
  SpecPolicy pol(sh, pid, resume);
  Sim<SpecPolicy> sim(s, soils, pol);
  sim.susp_ok = (sh.mode & 2u) != 0;
  if constexpr (WIND) {
    Wind p;
    p.pos = {0.f, 0.f}; p.speed = {0.f, 0.f, 0.f}; p.sediment = p.height = p.sheight = 0.0; p.ix = p.iy = 0; p.surface = p.contains = 0;
    if (l0) {
      if (resume) p = sh.ckpt[pid].d;
      else { int d1, d2; pol.spawn_draws(d1, d2); sim.wind_init(p, d2 % s.dimx, d1 % s.dimy); }
    }
    if (!rfl(l0 && sim.ab ? 1 : 0)) wind_loop_coop(sim, p, resume);
    if (!l0) return;
    SMX_A_MAX(&sh.ctrl[SC_ROUND_MAXSTEPS], (uint32_t)sim.n_steps_wind);
    if (sim.susp && !pol.overflow) { sim.flush_counters(); sh.ckpt[pid].d = p; pol.suspend(s); return; }
  } else {
    Water p;
    p.pos = {0.f, 0.f}; p.speed = {0.f, 0.f}; p.volume = p.sediment = p.evaprate = 0.0; p.spill = 0; p.ix = p.iy = 0;
    p.friction = 0.f; p.surface = p.contains = 0;
    if (l0) {
      if (resume) p = sh.ckpt[pid].w;
      else { int d1, d2; pol.spawn_draws(d1, d2); sim.water_init(p, d2 % s.dimx, d1 % s.dimy); }
    }
    if (!rfl(l0 && sim.ab ? 1 : 0)) water_loop_coop(sim, p, true, resume);
    if (!l0) return;
    if (!sim.ab && !sim.susp) {
      const uint32_t steps = (uint32_t)sim.n_steps_w_all;
      sim.flush_counters();                                  // the loop's counters travel inside the policy
      TailRes<SpecPolicy> r = coop_water_tail<SpecPolicy>(s, soils, pol, p);
      SMX_A_MAX(&sh.ctrl[SC_ROUND_MAXSTEPS], steps + r.steps);
      if (r.ab || r.pol.overflow) r.pol.self_rollback(s); else r.pol.finish_done(s);
      return;
    }
    SMX_A_MAX(&sh.ctrl[SC_ROUND_MAXSTEPS], (uint32_t)sim.n_steps_w_all);
    if (sim.susp && !pol.overflow) { sim.flush_counters(); sh.ckpt[pid].w = p; pol.suspend(s); return; }
  }
  if (sim.ab || pol.overflow) { pol.self_rollback(s); return; }
  sim.flush_counters();
  pol.finish_done(s);
}

}  // namespace smx
