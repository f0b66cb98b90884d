// hostsim -- TEST-ONLY host build of the DEVICE sources (soilmachine_amd/csrc/soil_core.h + policies).
//
// The GPU is only reachable through short, metered gpurun calls, so the kernel logic (state machine of the
// recursion-free water cascade, column primitives, speculative protocol) is unit-tested here by compiling the
// very same headers with g++ and running the "kernels" as plain loops. The product never loads this library:
// it lives under tests/, is built by tests/hostsim_lib.py, and soilmachine_amd/ has no reference to it.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <vector>

#define SMX_D inline
#define SMX_HOSTSIM 1
#include "../../soilmachine_amd/csrc/soil_core.h"
#include "../../soilmachine_amd/csrc/soil_serial.h"
#include "../../soilmachine_amd/csrc/soil_spec.h"
#include <type_traits>
#include "../../soilmachine_amd/csrc/soil_coop.h"
#include "../../soilmachine_amd/csrc/soil_batch.h"
#include "../../soilmachine_amd/csrc/soil_relax.h"
#include <pthread.h>
#include <atomic>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <sched.h>
#include <algorithm>
#include <random>
#include <map>

using namespace smx;

struct hs_ctx {
  DevState d;
  std::vector<Sec> cells, pool;
  std::vector<uint32_t> freelist;
  uint32_t free_count;
  std::vector<uint8_t> flags;
  std::vector<float> wfreq, wtrack, windfreq;
  std::vector<SoilP> soils;
  RandState rnd;
  unsigned long long ctr[C_COUNT];
  std::vector<unsigned long long> a0, a1, a2;
  size_t n;
  std::vector<uint8_t> dirty;                                // strips: 4x4 grid tiles changed since the last halo exchange
  int own_x0 = 0, own_x1 = -1;
};

static void rebind(hs_ctx* c) {
  c->d.cells = c->cells.data(); c->d.pool = c->pool.data(); c->d.freelist = c->freelist.data();
  c->d.free_count = &c->free_count; c->d.flags = c->flags.data();
  c->d.wfreq = c->wfreq.data(); c->d.wtrack = c->wtrack.data(); c->d.windfreq = c->windfreq.data();
  c->d.soils = c->soils.data(); c->d.nsoils = (uint32_t)c->soils.size(); c->d.rnd = &c->rnd; c->d.ctr = c->ctr;
  c->d.active = c->a0.data(); c->d.active1 = c->a1.data(); c->d.active2 = c->a2.data();
}

extern "C" {

hs_ctx* hs_create(int dimx, int dimy, int scale, uint64_t cap) {
  hs_ctx* c = new hs_ctx();
  memset(&c->d, 0, sizeof(c->d));
  c->d.dimx = dimx; c->d.dimy = dimy; c->d.scale = scale; c->d.pool_capacity = cap; c->d.x_lo = 0; c->d.x_hi = dimx;
  c->n = (size_t)dimx * dimy;
  c->cells.resize(c->n); c->pool.resize(cap); c->freelist.resize(cap); c->flags.assign(c->n + 8, 0);   // (whole 32-bit words: SerialPolicy::set_flag)
  for (size_t i = 0; i < c->n; i++) { Sec& s = c->cells[i]; s.size = s.floor = s.sat = 0; s.type = EMPTY; s.prev = NIL; }
  for (uint64_t i = 0; i < cap; i++) c->freelist[i] = (uint32_t)(cap - 1 - i);
  c->free_count = (uint32_t)cap;
  c->wfreq.assign(c->n, 0); c->wtrack.assign(c->n, 0); c->windfreq.assign(c->n, 0);
  memset(c->ctr, 0, sizeof(c->ctr));
  size_t nw0 = (c->n + 63) >> 6, nw1 = (nw0 + 63) >> 6, nw2 = (nw1 + 63) >> 6;
  c->a0.assign(nw0 + 64, 0); c->a1.assign(nw1 + 64, 0); c->a2.assign(nw2 + 64, 0);
  c->dirty.assign((size_t)((dimx + 3) >> 2) * ((dimy + 3) >> 2), 0);
  c->soils.resize(1);
  memset(&c->soils[0], 0, sizeof(SoilP)); c->soils[0].porosity = 1.0f;
  rebind(c);
  return c;
}
void hs_destroy(hs_ctx* c) { delete c; }
void hs_set_soils(hs_ctx* c, const SoilP* s, int n) { c->soils.assign(s, s + n); rebind(c); }
void hs_srand(hs_ctx* c, unsigned seed) {
  RandState& r = c->rnd;
  if (seed == 0) seed = 1;
  int32_t word = (int32_t)seed;
  r.ring[0] = (uint32_t)word;
  for (int i = 1; i < 31; i++) {
    long hi = word / 127773, lo = word % 127773;
    word = (int32_t)(16807 * lo - 2836 * hi);
    if (word < 0) word += 2147483647;
    r.ring[i] = (uint32_t)word;
  }
  uint32_t idx = 34;
  for (int i = 0; i < 310; i++) { uint32_t v = r.ring[idx % 31] + r.ring[(idx - 3) % 31]; r.ring[idx % 31] = v; idx++; }
  r.idx = idx; r.calls = 0;
}
void hs_rand_advance(hs_ctx* c, uint64_t nd) {
  RandState& r = c->rnd;
  for (uint64_t i = 0; i < nd; i++) { uint32_t v = r.ring[r.idx % 31] + r.ring[(r.idx - 3) % 31]; r.ring[r.idx % 31] = v; r.idx++; }
  r.calls += nd;
}
int hs_import(hs_ctx* c, const uint32_t* count, const uint32_t* type, const double* size, const double* floor,
              const double* sat, const float* wfreq, const float* wtrack, const float* windfreq) {
  size_t off = 0; uint32_t used = 0; unsigned long long live = 0;
  for (size_t i = 0; i < c->n; i++) {
    Sec cell; cell.size = cell.floor = cell.sat = 0; cell.type = EMPTY; cell.prev = NIL;
    uint32_t pv = NIL; c->flags[i] = 0;
    for (uint32_t j = 0; j < count[i]; j++, off++, live++) {
      Sec r; r.size = size[off]; r.floor = floor[off]; r.sat = sat[off]; r.type = type[off]; r.prev = pv;
      if (r.sat != 0.0) c->flags[i] |= F_SAT;
      if (j == count[i] - 1) cell = r; else { pv = used; c->pool[used++] = r; }
    }
    if (cell.type == AIR) c->flags[i] |= F_AIR;
    c->cells[i] = cell;
  }
  uint64_t cap = c->d.pool_capacity; uint32_t nf = 0;
  for (uint64_t i = cap; i-- > used;) c->freelist[nf++] = (uint32_t)i;
  c->free_count = nf; c->ctr[C_LIVE_SECTIONS] = live;
  if (wfreq) c->wfreq.assign(wfreq, wfreq + c->n);
  if (wtrack) c->wtrack.assign(wtrack, wtrack + c->n);
  if (windfreq) c->windfreq.assign(windfreq, windfreq + c->n);
  rebind(c);
  return 0;
}
uint64_t hs_num_sections(hs_ctx* c) { return c->ctr[C_LIVE_SECTIONS]; }
void hs_export(hs_ctx* c, uint32_t* count, uint32_t* type, double* size, double* floor, double* sat,
               float* wfreq, float* wtrack, float* windfreq) {
  size_t off = 0; std::vector<const Sec*> col;
  for (size_t i = 0; i < c->n; i++) {
    col.clear();
    if (c->cells[i].type != EMPTY) { const Sec* t = &c->cells[i]; while (true) { col.push_back(t); if (t->prev == NIL) break; t = &c->pool[t->prev]; } }
    count[i] = (uint32_t)col.size();
    for (size_t k = col.size(); k-- > 0; off++) { type[off] = col[k]->type; size[off] = col[k]->size; floor[off] = col[k]->floor; sat[off] = col[k]->sat; }
  }
  memcpy(wfreq, c->wfreq.data(), c->n * 4); memcpy(wtrack, c->wtrack.data(), c->n * 4); memcpy(windfreq, c->windfreq.data(), c->n * 4);
}
void hs_counters(hs_ctx* c, unsigned long long* out, uint64_t* rand_calls) { memcpy(out, c->ctr, sizeof(c->ctr)); *rand_calls = c->rnd.calls; }

// ---- the "kernels" (bodies mirror soilmx.hip; keep in sync) ----
void hs_water(hs_ctx* c, int n) {
  const DevState& s = c->d;
  SerialPolicy pol(s);
  Sim<SerialPolicy> sim(s, s.soils, pol);
  Frame st[MAX_FRAMES]; int depth = 0;
  for (int i = 0; i < n; i++) {
    int ry, rx; pol.rand2(s, ry, rx);
    Water p; sim.water_init(p, rx % s.dimx, ry % s.dimy);
    sim.water_drive(p, true, true, st, depth);
  }
  sim.flush_counters(); pol.finish(s);
}
void hs_wind(hs_ctx* c, int n) {
  const DevState& s = c->d;
  SerialPolicy pol(s);
  Sim<SerialPolicy> sim(s, s.soils, pol);
  for (int i = 0; i < n; i++) {
    int ry, rx; pol.rand2(s, ry, rx);
    Wind p; sim.wind_init(p, rx % s.dimx, ry % s.dimy);
    sim.wind_run(p);
  }
  sim.flush_counters(); pol.finish(s);
}
static void grid_classify(hs_ctx* c) {                      // k_grid_classify
  const DevState& s = c->d;
  const size_t n = c->n;
  std::fill(c->a0.begin(), c->a0.end(), 0); std::fill(c->a1.begin(), c->a1.end(), 0); std::fill(c->a2.begin(), c->a2.end(), 0);
  for (size_t cc = 0; cc < n; cc++) {
    const int x = (int)(cc / s.dimy), y = (int)(cc % s.dimy);
    bool act = (s.flags[cc] & F_SAT) != 0;
    for (int dx = -1; dx <= 1 && !act; dx++) for (int dy = -1; dy <= 1; dy++) {
      int nx = x + dx, ny = y + dy; if (nx < 0 || ny < 0 || nx >= s.dimx || ny >= s.dimy) continue;
      if (s.flags[(size_t)nx * s.dimy + ny] & F_AIR) { act = true; break; }
    }
    if (act) { c->a0[cc >> 6] |= 1ull << (cc & 63); c->a1[cc >> 12] |= 1ull << ((cc >> 6) & 63); c->a2[cc >> 18] |= 1ull << ((cc >> 12) & 63); }
  }
}
void hs_grid(hs_ctx* c) {
  const DevState& s = c->d;
  const size_t n = c->n;
  grid_classify(c);
  SerialPolicy pol(s);
  Sim<SerialPolicy> sim(s, s.soils, pol);
  sim.grid_mode = true;
  Frame st[MAX_FRAMES]; int depth = 0;
  size_t cc = sim.next_active(0);
  unsigned long long visited = 0;
  Water dummy; memset(&dummy, 0, sizeof(dummy));
  while (cc < n) {
    const int x = (int)(cc / s.dimy), y = (int)(cc % s.dimy);
    sim.seep(x, y);
    sim.push_frame(st, depth, x, y, 3);
    sim.water_drive(dummy, false, false, st, depth);
    visited++;
    cc = sim.next_active(cc + 1);
  }
  sim.n_wcasc += n - visited;
  sim.flush_counters();
  pol.add_counter(s, C_GRID_ACTIVE, visited);
  pol.finish(s);
}
void hs_freq(hs_ctx* c) {
  const float lrate = 0.01f, K = 50.0f;
  for (size_t i = 0; i < c->n; i++) c->wfreq[i] = (1.0f - lrate) * c->wfreq[i] + lrate * K * c->wtrack[i] / (1.0f + K * c->wtrack[i]);
  std::fill(c->wtrack.begin(), c->wtrack.end(), 0.0f);
}
void hs_tick(hs_ctx* c, int nwater, int nwind, int dowater, int dowind) {
  if (dowater) { hs_water(c, nwater); hs_grid(c); }
  if (dowind) hs_wind(c, nwind);
  if (dowater) hs_freq(c);
}
void hs_add(hs_ctx* c, int x, int y, double size, uint32_t type) {
  const DevState& s = c->d; SerialPolicy pol(s); Sim<SerialPolicy> sim(s, s.soils, pol);
  sim.add(x, y, size, type); sim.flush_counters(); pol.finish(s);
}
double hs_remove(hs_ctx* c, int x, int y, double h) {
  const DevState& s = c->d; SerialPolicy pol(s); Sim<SerialPolicy> sim(s, s.soils, pol);
  double r = sim.remove(x, y, h); sim.flush_counters(); pol.finish(s); return r;
}


}  // extern "C" (serial engine)

// ---------------- speculative engine, emulated: "kernels" = groups of std::threads over the same memory ----------
struct SpecBuffers {
  std::vector<uint32_t> stamp, claim, chunk_tab, alloc_log, free_log, pending, ctrl, reclaim, scout_cells, scout_len;
  std::vector<SpecPart> part;
  SpecLog* arena = nullptr;                                 // big and sparse: malloc'd, never zero-filled
  uint32_t* scout_cells_raw = nullptr;
  ~SpecBuffers() { free(arena); free(scout_cells_raw); }
  std::vector<SpecCkpt> ckpt;
  std::vector<uint32_t> draws;
  SpecShared sh;
};
static uint32_t g_spec_maxnest = SPEC_MAXNEST;
extern "C" void hs_set_spec_maxnest(int n) { g_spec_maxnest = n < 128 ? 128u : (uint32_t)n; }
static void spec_setup(hs_ctx* c, SpecBuffers& b, uint32_t nparticles) {
  b.sh.maxnest = g_spec_maxnest;
  const uint32_t n = nparticles + b.sh.maxnest;             // one record per slot
  b.stamp.assign(c->n, 0); b.claim.assign(c->n, NOCLAIM);
  b.part.resize(n); b.chunk_tab.assign((size_t)n * MAX_CHUNKS, 0);
  const uint32_t arena_chunks = std::max<uint32_t>(4096, nparticles * 64);
  b.arena = (SpecLog*)malloc((size_t)arena_chunks * LOG_CHUNK * sizeof(SpecLog));
  b.alloc_log.assign((size_t)n * MAX_NODELOG, 0); b.free_log.assign((size_t)n * MAX_NODELOG, 0);
  b.pending.assign(n, 0); b.ctrl.assign(SC_COUNT, 0); b.reclaim.assign(1 << 20, 0);
  b.draws.assign(2 * n, 0);
  b.sh.scout_cap = 65536; b.scout_cells_raw = (uint32_t*)malloc((size_t)(nparticles + SPEC_SCOUT_MARGIN) * b.sh.scout_cap * 4); b.scout_len.assign(n, 0);   // only scouted slots own a list
  SpecShared& sh = b.sh;
  sh.stamp = b.stamp.data(); sh.claim = b.claim.data(); sh.part = b.part.data(); sh.chunk_tab = b.chunk_tab.data();
  sh.arena = b.arena; sh.arena_chunks = arena_chunks; sh.alloc_log = b.alloc_log.data(); sh.free_log = b.free_log.data();
  sh.pending = b.pending.data(); sh.ctrl = b.ctrl.data(); sh.draws = b.draws.data(); sh.ndraws = (uint32_t)b.draws.size();
  sh.reclaim = b.reclaim.data(); sh.reclaim_cap = (uint32_t)b.reclaim.size();
  sh.scout_cells = b.scout_cells_raw; sh.scout_len = b.scout_len.data();
  b.ckpt.resize(n); sh.ckpt = b.ckpt.data(); sh.mode = 2u;
}
// The workers of one calling thread (a "kernel launch" = one job for every worker): kept alive between launches -- a tick is
// thousands of launches, and creating the threads anew each time was most of the suite's system time. thread_local: ranks that run
// as threads of one process (loop-back strips) each have their own workers.
struct WorkerPool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv_start, cv_done;
  const std::function<void(int)>* job = nullptr;
  uint64_t gen = 0;
  int active = 0, pending = 0;
  bool stop = false;
  void worker(int t) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)>* j;
      {
        std::unique_lock<std::mutex> l(m);
        cv_start.wait(l, [&] { return stop || (gen != seen && t < active); });
        if (stop) return;
        seen = gen; j = job;
      }
      (*j)(t);
      { std::lock_guard<std::mutex> l(m); if (--pending == 0) cv_done.notify_one(); }
    }
  }
  void run(int n, const std::function<void(int)>& f) {
    std::unique_lock<std::mutex> l(m);
    while ((int)th.size() < n) { const int t = (int)th.size(); th.emplace_back([this, t] { worker(t); }); }
    job = &f; active = n; pending = n; gen++;
    cv_start.notify_all();
    cv_done.wait(l, [&] { return pending == 0; });
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> l(m); stop = true; }
    cv_start.notify_all();
    for (auto& t : th) t.join();
  }
};
template <class F> static void parallel_for(uint32_t n, int nthreads, F f, unsigned shuffle_seed) {
  std::vector<uint32_t> order(n);
  for (uint32_t i = 0; i < n; i++) order[i] = i;
  if (shuffle_seed) { std::mt19937 g(shuffle_seed); std::shuffle(order.begin(), order.end(), g); }
  if (nthreads <= 1 || n < 4u * (uint32_t)nthreads) { for (uint32_t i : order) f(i); return; }   // (a handful of items: not worth the threads -- the shuffle still varies the order)
  static thread_local WorkerPool pool;
  const std::function<void(int)> job = [&](int t) { for (uint32_t i = (uint32_t)t; i < n; i += (uint32_t)nthreads) f(order[i]); };
  pool.run(nthreads, job);
}
static int g_boundary_threads = 1;
extern "C" void hs_set_boundary_threads(int n) { g_boundary_threads = n < 1 ? 1 : n; }
static uint32_t g_spec_sub = 4096;                          // top-level particles per sub-phase (soilmx.hip spec_phase: SPEC_SUB)
extern "C" void hs_set_spec_sub(int n) { g_spec_sub = n < 1 ? 1u : (uint32_t)n; }
// one sub-phase (mirrors soilmx.hip spec_subphase): returns 0 = all n committed, 1 = cut (nested slots used up), 2 = give the rest to the serial walker
template <bool WIND>
static int spec_subphase(hs_ctx* c, uint32_t n, int nthreads, int scout, unsigned shuffle, uint32_t* stats, uint32_t* committed) {
  const DevState& s = c->d;
  SpecBuffers b; spec_setup(c, b, n);
  const SpecShared& sh = b.sh;
  spec_begin(s, sh, n);
  uint32_t last_base = 0, stall = 0; int fell_back = 0;
  b.sh.mode = (uint32_t)(scout >> 4);                       // scout argument: bit0 = trajectory scout, bits 4.. = engine mode
  scout &= 1;
  if (scout) parallel_for(n + SPEC_SCOUT_MARGIN, nthreads, [&](uint32_t i) { if (b.sh.mode & 4u) spec_scout_one_coop<WIND>(s, s.soils, sh, i); else spec_scout_one<WIND>(s, s.soils, sh, i); }, shuffle ? shuffle + 1 : 0);
  while (true) {
    std::fill(b.claim.begin(), b.claim.end(), NOCLAIM);
    const uint32_t np = sh.ctrl[SC_NPENDING];
    parallel_for(np, nthreads, [&](uint32_t i) { spec_claim_one(sh, sh.pending[i], 0u, 1u); }, shuffle ? shuffle + 1 : 0);
    if (sh.mode & 1u) parallel_for(np, nthreads, [&](uint32_t i) { const uint32_t pid = sh.pending[i]; sh.part[pid].blocked = spec_blocked(sh, pid, 0u, 1u) ? 1u : 0u; }, 0);
    const bool coop = (sh.mode & 4u) != 0;                   // mode bit 2 (host only): the wave-loop structure of soil_coop.h
    parallel_for(np, nthreads, [&](uint32_t i) { if (coop) spec_exec_one_coop<WIND>(s, s.soils, sh, sh.pending[i]); else spec_exec_one<WIND>(s, s.soils, sh, sh.pending[i]); }, shuffle ? shuffle + 2 + sh.ctrl[SC_ROUNDS] : 0);
    if (getenv("HS_TRACE") && np == 1) {
      const SpecPart& q = sh.part[sh.pending[0]];
      if (q.fail_cell != NOCLAIM) fprintf(stderr, "[hostsim]   solo %u failed at cell %u stamp %u (status of that holder %d)\n", sh.pending[0], q.fail_cell, sh.stamp[q.fail_cell],
                                          sh.stamp[q.fail_cell] && sh.stamp[q.fail_cell] != POISON ? (int)sh.part[sh.stamp[q.fail_cell] - 1].status : -1);
    }
    {
      static SpecWork* w = new SpecWork();
      const int bt = g_boundary_threads;                     // >1: the boundary runs as a TEAM (threads + barrier), as on the device
      if (bt <= 1) spec_boundary_team(s, sh, *w, 0u, 1u, []() {});
      else {
        // a team of threads with a barrier between the stages of the boundary, as on the device. The stages are tiny: a spinning
        // barrier (yielding when the machine is oversubscribed) instead of pthread_barrier's futex sleep per stage
        struct SpinBarrier {
          std::atomic<int> count{0}, gen{0}; int n;
          void wait() {
            const int g = gen.load(std::memory_order_acquire);
            if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) { count.store(0, std::memory_order_relaxed); gen.fetch_add(1, std::memory_order_release); }
            else { int spins = 0; while (gen.load(std::memory_order_acquire) == g) if (++spins > 4000) sched_yield(); }
          }
        } bar; bar.n = bt;
        std::vector<std::thread> th;
        for (int t = 0; t < bt; t++)
          th.emplace_back([&, t]() { spec_boundary_team(s, sh, *w, (uint32_t)t, (uint32_t)bt, [&]() { bar.wait(); }); });
        for (auto& t : th) t.join();
      }
    }
    if (sh.ctrl[SC_TOP] == n) break;
    if (sh.ctrl[SC_BASE] == last_base) stall++; else { stall = 0; last_base = sh.ctrl[SC_BASE]; }
    if (sh.ctrl[SC_NESTED_COMMITTED] > sh.maxnest - 64) { fell_back = 1; break; }   // nested slots used up: commit, the caller re-arms
    if (sh.ctrl[SC_NPENDING] == 0 || stall >= SPEC_STALL_LIMIT) {
      if (getenv("HS_TRACE")) {
        const SpecPart& q = sh.part[sh.ctrl[SC_BASE]];
        if (q.fail_cell != NOCLAIM) {
          const uint32_t fc = q.fail_cell, v = sh.stamp[fc];
          fprintf(stderr, "[hostsim]   fail_cell %u stamp %u claim %u owner-status %d owner-kill %d\n", fc, v, sh.claim[fc],
                  (v && v != POISON) ? (int)sh.part[v - 1].status : -1, (v && v != POISON) ? (int)sh.part[v - 1].kill : -1);
        }
        fprintf(stderr, "[hostsim] fallback: npending %u stall %u nested %u base %u top %u | base particle: status %u overflow %u attempts %u nlog %u nalloc %u nfree %u nested %u\n",
                sh.ctrl[SC_NPENDING], stall, sh.ctrl[SC_NESTED_COMMITTED], sh.ctrl[SC_BASE], sh.ctrl[SC_TOP], q.status, q.overflow, q.attempts, q.nlog, q.nalloc, q.nfree, q.nested);
      }
      fell_back = 2; break;
    }
  }
  if (stats) { stats[0] += sh.ctrl[SC_ROUNDS]; stats[1] += sh.ctrl[SC_ABORTS]; stats[2] += sh.ctrl[SC_KILLS]; stats[4] += sh.ctrl[SC_EXECUTED]; stats[5] += sh.ctrl[SC_SUM_MAXSTEPS]; }
  if (sh.ctrl[SC_N] == 0) { *committed = 0; return 2; }     // no pool headroom: nothing was executed
  if (fell_back) {
    spec_abandon(s, sh);
    const uint32_t base = sh.ctrl[SC_TOP];                   // top-level particles committed so far
    spec_end(s, sh, base, sh.ctrl[SC_NESTED_COMMITTED]);
    *committed = base;
    return fell_back;
  }
  spec_end(s, sh, n, sh.ctrl[SC_NESTED_COMMITTED]);
  *committed = n;
  return 0;
}
// a particle phase in sub-phases of at most g_spec_sub top-level particles (mirrors soilmx.hip spec_phase); returns 1 if any part of the
// phase went to the serial walker. stats[3] = (sub-phases cut << 16) | phases handed to the serial walker | single serial particles << 8
template <bool WIND>
static int spec_phase(hs_ctx* c, uint32_t n, int nthreads, int scout, unsigned shuffle, uint32_t* stats) {
  if (stats) for (int i = 0; i < 6; i++) stats[i] = 0;
  uint32_t remaining = n; int any_serial = 0;
  while (remaining > 0) {
    const uint32_t m = remaining < g_spec_sub ? remaining : g_spec_sub;
    uint32_t committed = 0;
    const int outcome = spec_subphase<WIND>(c, m, nthreads, scout, shuffle, stats, &committed);
    remaining -= committed;
    if (outcome == 0) continue;
    if (stats) stats[3] += 1u << 16;
    if (outcome == 2) { if (WIND) hs_wind(c, (int)remaining); else hs_water(c, (int)remaining); remaining = 0; any_serial = 1; if (stats) stats[3] += 1u; }
    else if (committed == 0) { if (WIND) hs_wind(c, 1); else hs_water(c, 1); remaining -= 1u; any_serial = 1; if (stats) stats[3] += 1u << 8; }
  }
  return any_serial;
}
extern "C" {
int hs_spec_water(hs_ctx* c, int n, int nthreads, int scout, unsigned shuffle, uint32_t* stats) { return spec_phase<false>(c, (uint32_t)n, nthreads, scout, shuffle, stats); }
int hs_spec_wind(hs_ctx* c, int n, int nthreads, int scout, unsigned shuffle, uint32_t* stats) { return spec_phase<true>(c, (uint32_t)n, nthreads, scout, shuffle, stats); }
void hs_spec_tick(hs_ctx* c, int nwater, int nwind, int dowater, int dowind, int nthreads, int scout, unsigned shuffle, uint32_t* stats) {
  uint32_t st[8] = {0};
  if (dowater) { hs_spec_water(c, nwater, nthreads, scout & 0xff, shuffle, st); if (stats) for (int i = 0; i < 6; i++) stats[i] += st[i]; hs_grid(c); }
  // scout: bit0 trajectory scout, bits 4-5 water engine mode, bits 8-9 wind engine mode (0 = same as water)
  const int wmode = (scout >> 8) & 3;
  const int wscout = wmode ? ((scout & 1) | (wmode << 4) | (scout & 0x40)) : (scout & 0xff);
  if (dowind) { hs_spec_wind(c, nwind, nthreads, wscout, shuffle, st); if (stats) for (int i = 0; i < 6; i++) stats[6 + i] += st[i]; }
  if (dowater) hs_freq(c);
}
}  // extern "C"

// ---------------- batched engine (soil_batch.h), emulated: the host driver below mirrors soilmx.hip batch_* ----------
static int g_batch_dilate = 0;
static uint32_t g_batch_cell_above = 0xFFFFFFFFu;
static int g_strips_n = 1, g_strips_a = 16, g_strips_b = 48;
struct BatchBuffers {
  std::vector<unsigned long long> claim0, claim1;
  std::vector<BWater> water; std::vector<BWind> wind;
  std::vector<uint32_t> live, freed, ctrl, draws;
  std::vector<BChild> children;
  std::vector<RReq> req; std::vector<uint32_t> rnext, rcount, rhead, cflag, flist0, flist1, clist0, clist1, flagl0, flagl1;   // relaxed schedule (soil_relax.h)
  BatchShared bs;
};
static int g_relaxed = 0;                                   // 1: particle phases run the relaxed schedule (SMX_ENGINE_RELAXED)
static uint32_t g_relax_wind_min = 0xFFFFFFFFu;                   // wind: relaxed while more particles than this run ...
static int g_relax_wind_steps = 4;                          // ... taking up to this many steps per epoch (smx_set_relax_wind)
static std::atomic<uint32_t> g_relax_tag{0};   // (atomic: several HostSim instances may tick on different threads of one process, tools/p2_ensemble.py)
extern "C" void hs_set_relaxed(int on) { g_relaxed = on; }
static uint32_t g_stagger_k = 0, g_stagger_gap = SMX_WATER_STAGGER;   // smx_set_water_stagger: one water phase = k batches born `gap` epochs apart (gap 0: off; k 0: smx_set_water_generations' k)
extern "C" void hs_set_water_stagger(uint32_t k, uint32_t gap) { g_stagger_k = k; g_stagger_gap = gap; }
static int g_water_generations = SMX_WATER_GENERATIONS;      // smx_set_water_generations
static uint32_t stagger_k() { return g_stagger_k ? g_stagger_k : (uint32_t)g_water_generations; }
static int g_relax_water_steps = 1;                         // relaxed water: steps per epoch (smx_set_relax_water)
extern "C" void hs_set_relax_water(int steps) { g_relax_water_steps = steps < 1 ? 1 : (steps > R_MAXSTEPS ? R_MAXSTEPS : steps); }
extern "C" void hs_set_relax_wind(uint32_t n, int steps) { g_relax_wind_min = n; g_relax_wind_steps = steps < 1 ? 1 : (steps > R_MAXSTEPS ? R_MAXSTEPS : steps); }
static void batch_setup(hs_ctx* c, BatchBuffers& b, uint32_t nslots, bool wind) {
  const DevState& s = c->d;
  b.bs.tshift = -1; b.bs.ntx = s.dimx; b.bs.nty = s.dimy; b.bs.txb = 0; b.bs.txe = s.dimx;
  const size_t nt = (size_t)s.dimx * s.dimy;                // one claim word per cell: enough for every granularity
  b.claim0.assign(nt, 0); b.claim1.assign(nt, 0);
  if (wind) b.wind.resize(nslots); else b.water.resize(nslots);
  b.live.resize(nslots ? nslots : 1); b.ctrl.assign(BC_COUNT, 0); b.draws.assign(2 * (size_t)nslots + 2, 0);
  if (b.children.size() < (1u << 16)) b.children.resize(1u << 16);
  if (b.freed.size() < (1u << 20)) b.freed.resize(1u << 20);
  b.bs.claim[0] = b.claim0.data(); b.bs.claim[1] = b.claim1.data();
  b.bs.water = b.water.data(); b.bs.wind = b.wind.data(); b.bs.live = b.live.data();
  b.bs.children = b.children.data(); b.bs.child_cap = (uint32_t)b.children.size();
  b.bs.dilate = g_batch_dilate; b.bs.maxsteps = 1;
  b.bs.strips = batch_make_strips(s.dimx, g_strips_n, g_strips_a, g_strips_b); b.bs.sphase = STRIP_INTERIOR;
  b.bs.dirty = c->dirty.data(); b.bs.own_x0 = c->own_x0; b.bs.own_x1 = c->own_x1 < 0 ? s.dimx : c->own_x1;
  b.bs.freed = b.freed.data(); b.bs.freed_cap = (uint32_t)b.freed.size(); b.bs.ctrl = b.ctrl.data(); b.bs.draws = b.draws.data();
  b.bs.relaxed = g_relaxed ? 1 : 0;
  b.bs.birth_n = 0u; b.bs.birth_gap = 0u;
  if (!wind && g_relaxed && stagger_k() > 1u && g_stagger_gap > 0u && nslots) { b.bs.birth_n = (nslots + stagger_k() - 1u) / stagger_k(); b.bs.birth_gap = g_stagger_gap; }   // (children: toplevel == 0, never delayed)
  b.bs.req = nullptr; b.bs.rnext = nullptr; b.bs.rcount = nullptr; b.bs.rhead = nullptr; b.bs.cflag = nullptr; b.bs.cstate = nullptr;
  b.bs.flist[0] = b.bs.flist[1] = b.bs.clist[0] = b.bs.clist[1] = b.bs.flagl[0] = b.bs.flagl[1] = nullptr; b.bs.rstride = 2; b.bs.rsteps = 1; b.bs.rtag = 0;
  if (b.bs.relaxed) {
    const size_t ent = (size_t)(nslots ? nslots : 1) * 2 * R_MAXSTEPS;
    b.req.resize(ent); b.rnext.assign(ent, NIL); b.rcount.assign(nslots ? nslots : 1, 0u); b.rhead.assign(nt, NIL); b.cflag.assign(nt, 0u);
    b.bs.list_cap = nslots ? nslots : 1; b.bs.flag_cap = (uint32_t)ent; b.bs.clist_cap = (uint32_t)std::max<size_t>(4096, ent / 2);
    b.flagl0.assign(ent, 0); b.flagl1.assign(ent, 0); b.bs.flagl[0] = b.flagl0.data(); b.bs.flagl[1] = b.flagl1.data();
    b.flist0.assign(b.bs.list_cap, 0); b.flist1.assign(b.bs.list_cap, 0); b.clist0.assign((size_t)9 * b.bs.clist_cap, 0); b.clist1.assign((size_t)9 * b.bs.clist_cap, 0);
    b.bs.req = b.req.data(); b.bs.rnext = b.rnext.data(); b.bs.rcount = b.rcount.data(); b.bs.rhead = b.rhead.data(); b.bs.cflag = b.cflag.data();
    b.bs.flist[0] = b.flist0.data(); b.bs.flist[1] = b.flist1.data(); b.bs.clist[0] = b.clist0.data(); b.bs.clist[1] = b.clist1.data();
  }
}
static void batch_draw(hs_ctx* c, BatchBuffers& b, uint32_t nslots) {   // the next 2*nslots rand() values (host-side generator)
  RandState& r = c->rnd;
  for (uint32_t i = 0; i < 2 * nslots; i++) {
    const uint32_t v = r.ring[r.idx % 31] + r.ring[(r.idx - 3) % 31];
    r.ring[r.idx % 31] = v; r.idx++; r.calls++;
    b.draws[i] = v >> 1;
  }
}
static void batch_merge_freed(hs_ctx* c, BatchBuffers& b) {
  uint32_t nf = b.ctrl[BC_NFREED]; if (nf > b.bs.freed_cap) nf = b.bs.freed_cap;
  for (uint32_t i = 0; i < nf; i++) c->freelist[c->free_count++] = b.freed[i];
  b.ctrl[BC_NFREED] = 0;
}
static uint32_t g_batch_stats[8];
extern "C" void hs_set_batch_dilate(int d) { g_batch_dilate = d; }
extern "C" void hs_set_batch_cell_above(uint32_t n) { g_batch_cell_above = n; }
extern "C" void hs_set_batch_strips(int n, int a, int b) { g_strips_n = n; g_strips_a = a; g_strips_b = b; }
// runs generations until no particle and no child is left; `kids` = children to start from (grid pass), else top-level
static void batch_generations(hs_ctx* c, BatchBuffers& b, bool wind, uint32_t n, std::vector<BChild> kids, int nthreads, unsigned shuffle) {
  const DevState& s = c->d;
  bool first = kids.empty();
  uint32_t nslots = first ? n : (uint32_t)kids.size();
  while (nslots > 0) {
    batch_setup(c, b, nslots, wind);
    if (first) batch_draw(c, b, nslots);                     // (suspended nested particles carry their state: no draws)
    const BatchShared& bs = b.bs;
    parallel_for(nslots, nthreads, [&](uint32_t i) {
      BatchPolicy pol(s, bs); Sim<BatchPolicy> sim(s, s.soils, pol);
      if (wind) batch_wind_spawn(sim, bs, i); else batch_water_spawn(sim, bs, i, first ? nullptr : &kids[i]);
      sim.flush_counters(); pol.finish(s);
    }, shuffle ? shuffle + 7 : 0);
    if (!wind && !first && b.bs.relaxed) for (uint32_t i = 0; i < nslots; i++) relax_flood_later(bs, i, 0u, b.water[i].state == B_ENDED);   // k_batch_spawn
    for (uint32_t i = 0; i < nslots; i++) b.live[i] = i;
    uint32_t nlive = nslots, epoch = 0, chunk = 0;
    const bool relax_gen = b.bs.relaxed != 0;
    while (nlive > 0) {
      BSched sc = batch_schedule(wind, nlive, g_batch_dilate, g_batch_cell_above);
      if (const char* e = getenv("HS_SCHED")) { int a, d, k; if (sscanf(e, "%d,%d,%d", &a, &d, &k) == 3) { sc.tshift = a; sc.dilate = d; sc.maxsteps = k; } }   // exploration only
      if (const char* e = wind ? nullptr : getenv("HS_SCHED_WATER")) { int a, d, k; if (sscanf(e, "%d,%d,%d", &a, &d, &k) == 3) { sc.tshift = a; sc.dilate = d; sc.maxsteps = k; } }   // ... water (floods) only
      const bool strips = b.bs.strips.n > 1;
      const bool relaxed_now = relax_gen && (!wind || nlive > g_relax_wind_min);
      bool regrain = sc.tshift != b.bs.tshift;               // first chunk, or the claim granularity changes
      if (relax_gen && !relaxed_now && b.bs.relaxed) { b.bs.relaxed = 0; regrain = true; }   // survivors go on under the exclusive schedule
      batch_apply_schedule(b.bs, s, sc);
      b.bs.sphase = strips ? (int)(chunk & 1u) : STRIP_INTERIOR;   // column strips: INTERIOR and SEAM chunks alternate
      if (strips && chunk > 0) epoch++;                      // (reservations of the previous chunk must not look current)
      chunk++;
      if (regrain || strips) {                               // under strips every chunk starts with fresh reservations: eligibility changed
        if (regrain) { std::fill(b.claim0.begin(), b.claim0.end(), 0ull); std::fill(b.claim1.begin(), b.claim1.end(), 0ull); }
        parallel_for(nlive, nthreads, [&](uint32_t i) {
          BatchPolicy pol(s, bs); Sim<BatchPolicy> sim(s, s.soils, pol);
          if (wind) batch_wind_reclaim(sim, bs, bs.live[i], epoch); else batch_water_reclaim(sim, bs, bs.live[i], epoch);
        }, shuffle ? shuffle + 5 : 0);
      }
      static FILE* estats = getenv("HS_EPOCH_STATS") ? fopen(getenv("HS_EPOCH_STATS"), "w") : nullptr;   // analysis only
      std::vector<uint32_t> pc, ps; std::vector<uint32_t> pst;
      int chunk_epochs = B_CHUNK;
      if (relaxed_now) {                                     // mirrors soilmx.hip: step, apply, nine cascade colours, floods
        chunk_epochs = strips ? B_CHUNK : (first ? B_CHUNK : 8);   // (with strips the chunk is the unit of the exchange: soilmx.hip run_chunk)
        b.bs.rstride = wind ? 2 * g_relax_wind_steps : (g_relax_water_steps > 2 ? g_relax_water_steps : 2); b.bs.rsteps = wind ? g_relax_wind_steps : g_relax_water_steps;
        if (strips) for (uint32_t q = BC_NFLOOD; q < (uint32_t)BC_LISTS_END; q++) b.ctrl[q] = 0u;   // (the epoch skipped between two chunks breaks the lists' parity hand-over)
        if (strips && !wind) {                                // k_relax_rebuild_floods
          for (uint32_t i = 0; i < nslots; i++) relax_flood_later(bs, i, epoch & 1u, b.water[i].state == B_ENDED);
        }
        for (int k = 0; k < chunk_epochs; k++, epoch++) {
          b.bs.rtag = g_relax_tag.fetch_add(1u) + 1u; if (b.bs.rtag == 0u) b.bs.rtag = g_relax_tag.fetch_add(1u) + 1u;
          parallel_for(nlive, nthreads, [&](uint32_t i) {                                       // k_relax_step
            BatchPolicy pol(s, bs); Sim<BatchPolicy> sim(s, s.soils, pol);
            if (wind) relax_wind_step(sim, bs, bs.live[i], epoch); else relax_water_step(sim, bs, bs.live[i], epoch);
          }, shuffle ? shuffle + 11 + epoch : 0);
          { const uint32_t par = (epoch + 1u) & 1u; b.ctrl[BC_NFLOOD + par] = 0u; b.ctrl[BC_NFLAG + par] = 0u; b.ctrl[BC_NRUN + par] = 0u; for (uint32_t q = 0; q < 9u; q++) b.ctrl[BC_NCASC + 9u * par + q] = 0u; }
          parallel_for(nlive, nthreads, [&](uint32_t i) {                                       // k_relax_apply
            BatchPolicy pol(s, bs); Sim<BatchPolicy> sim(s, s.soils, pol);
            relax_apply_slot(sim, bs, bs.live[i]);
            sim.flush_counters(); pol.finish(s);
          }, shuffle ? shuffle + 13 + epoch : 0);
          {                                                                                     // k_relax_filter
            const uint32_t par = epoch & 1u;
            uint32_t nf = b.ctrl[BC_NFLAG + par]; if (nf > bs.flag_cap) nf = bs.flag_cap;
            parallel_for(nf, nthreads, [&](uint32_t i) {
              BatchPolicy pol(s, bs); Sim<BatchPolicy> sim(s, s.soils, pol);
              relax_filter(sim, bs, bs.flagl[par][i], true, epoch);
              sim.flush_counters(); pol.finish(s);
              if (pol.a_casc) SMX_A_ADD64(&s.ctr[C_CASCADE_CALLS], (unsigned long long)pol.a_casc);
              if (pol.a_tr) SMX_A_ADD64(&s.ctr[C_CASCADE_TRANSFERS], (unsigned long long)pol.a_tr);
            }, shuffle ? shuffle + 15 + epoch : 0);
          }
          static FILE* rstats = getenv("HS_RESIDUE_STATS") ? fopen(getenv("HS_RESIDUE_STATS"), "w") : nullptr;   // analysis only: what is left for the colour passes?
          if (rstats) {
            const uint32_t par = epoch & 1u;
            uint32_t nf = b.ctrl[BC_NFLAG + par]; if (nf > bs.flag_cap) nf = bs.flag_cap;
            std::vector<uint32_t> fl(bs.flagl[par], bs.flagl[par] + nf);
            std::map<uint32_t, uint32_t> id; for (uint32_t i = 0; i < nf; i++) id[fl[i]] = i;
            std::vector<uint32_t> parent(nf); for (uint32_t i = 0; i < nf; i++) parent[i] = i;
            std::function<uint32_t(uint32_t)> find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
            for (uint32_t i = 0; i < nf; i++) {
              const int x = (int)(fl[i] / (uint32_t)s.dimy), y = (int)(fl[i] % (uint32_t)s.dimy);
              for (int dx = -2; dx <= 2; dx++) for (int dy = -2; dy <= 2; dy++) {
                const int xx = x + dx, yy = y + dy; if ((!dx && !dy) || xx < 0 || yy < 0 || xx >= s.dimx || yy >= s.dimy) continue;
                auto it = id.find((uint32_t)xx * (uint32_t)s.dimy + (uint32_t)yy); if (it != id.end()) { const uint32_t a = find(i), c = find(it->second); if (a != c) parent[a] = c; }
              }
            }
            std::vector<uint32_t> csize(nf, 0), cneedy(nf, 0); std::vector<uint32_t> ccol(nf, 0);
            for (uint32_t i = 0; i < nf; i++) csize[find(i)]++;
            uint32_t residue = 0, ncol_used = 0;
            for (uint32_t col = 0; col < 9u; col++) {
              uint32_t nc = b.ctrl[BC_NCASC + 9u * par + col]; if (nc > bs.clist_cap) nc = bs.clist_cap;
              residue += nc; if (nc) ncol_used++;
              for (uint32_t i = 0; i < nc; i++) { const uint32_t r = find(id[bs.clist[par][(size_t)col * bs.clist_cap + i]]); cneedy[r]++; ccol[r] |= 1u << col; }
            }
            uint32_t hist[6] = {0, 0, 0, 0, 0, 0}, maxdepth = 0;   // residue cells by the size of their cluster: 2, 3, 4, 5-8, 9-16, >16
            for (uint32_t i = 0; i < nf; i++) if (cneedy[i]) {
              const uint32_t z = csize[i]; const int bkt = z <= 2 ? 0 : z == 3 ? 1 : z == 4 ? 2 : z <= 8 ? 3 : z <= 16 ? 4 : 5; hist[bkt] += cneedy[i];
              const uint32_t d = (uint32_t)__builtin_popcount(ccol[i]); if (d > maxdepth) maxdepth = d;
            }
            fprintf(rstats, "%s %u %u live %u flagged %u residue %u colours %u maxdepth %u  bysize2/3/4/8/16/more %u %u %u %u %u %u\n", wind ? "wind" : "water", nslots, epoch, nlive, nf, residue, ncol_used, maxdepth,
                    hist[0], hist[1], hist[2], hist[3], hist[4], hist[5]);
          }
          for (uint32_t col = 0; col < 9u; col++) {                                            // k_relax_cascade(_small)
            const uint32_t par = epoch & 1u;
            uint32_t nc = b.ctrl[BC_NCASC + 9u * par + col]; if (nc > bs.clist_cap) nc = bs.clist_cap;
            parallel_for(nc, nthreads, [&](uint32_t i) {
              BatchPolicy pol(s, bs); Sim<BatchPolicy> sim(s, s.soils, pol);
              relax_cascade(sim, bs.clist[par][(size_t)col * bs.clist_cap + i]);
              sim.flush_counters(); pol.finish(s);
              if (pol.a_casc) SMX_A_ADD64(&s.ctr[C_CASCADE_CALLS], (unsigned long long)pol.a_casc);
              if (pol.a_tr) SMX_A_ADD64(&s.ctr[C_CASCADE_TRANSFERS], (unsigned long long)pol.a_tr);
            }, shuffle ? shuffle + 17 + epoch + col : 0);
          }
          if (!wind) {                                                                          // k_relax_floods
            const uint32_t par = epoch & 1u;
            uint32_t nf = b.ctrl[BC_NFLOOD + par]; if (nf > bs.list_cap) nf = bs.list_cap;
            const bool due = relax_floods_due(bs, epoch, nf, b.ctrl[BC_NRUN + par]);
            parallel_for(nf, nthreads, [&](uint32_t i) {
              const uint32_t slot = bs.flist[par][i];
              if (!due) { relax_flood_later(bs, slot, par ^ 1u, true); return; }
              BatchPolicy pol(s, bs); Sim<BatchPolicy> sim(s, s.soils, pol);
              Frame st[MAX_FRAMES];
              batch_water_epoch(sim, bs, slot, epoch, st);
              sim.flush_counters(); pol.finish(s);
              relax_flood_later(bs, slot, par ^ 1u, b.water[slot].state == B_ENDED);
            }, shuffle ? shuffle + 19 + epoch : 0);
          }
        }
        if (b.ctrl[BC_LIST_LOST]) { fprintf(stderr, "[hostsim] relaxed schedule: list overflow\n"); abort(); }
      } else
      for (int k = 0; k < B_CHUNK; k++, epoch++) {
        if (estats && !wind) { pc.resize(nlive); ps.resize(nlive); pst.resize(nlive); for (uint32_t i = 0; i < nlive; i++) { const BWater& q = b.water[bs.live[i]]; pc[i] = q.c_casc; ps[i] = q.c_steps; pst[i] = q.state; } }
        parallel_for(nlive, nthreads, [&](uint32_t i) {
          BatchPolicy pol(s, bs); Sim<BatchPolicy> sim(s, s.soils, pol);
          Frame st[MAX_FRAMES];
          if (wind) batch_wind_epoch(sim, bs, bs.live[i], epoch); else batch_water_epoch(sim, bs, bs.live[i], epoch, st);
          sim.flush_counters(); pol.finish(s);
        }, shuffle ? shuffle + 11 + epoch : 0);
        if (estats && !wind) {
          uint32_t actors = 0, done = 0, maxc = 0, maxs = 0; unsigned long long sumc = 0, sums = 0;
          for (uint32_t i = 0; i < nlive; i++) { const BWater& q = b.water[bs.live[i]]; const uint32_t dc = q.c_casc - pc[i], ds = q.c_steps - ps[i];
            if (ds || dc || q.state != pst[i]) actors++; if (q.state == B_DONE && pst[i] != B_DONE) done++; if (dc > maxc) maxc = dc; if (ds > maxs) maxs = ds; sumc += dc; sums += ds; }
          fprintf(estats, "%u %u %u %u %u %u %llu %llu %u\n", nslots, epoch, nlive, actors, done, maxc, sumc, sums, maxs);
        }
      }
      batch_merge_freed(c, b);
      uint32_t m = 0;
      for (uint32_t i = 0; i < nslots; i++) { const uint32_t st_ = wind ? b.wind[i].state : b.water[i].state; if (st_ != B_DONE && st_ != B_AWAY) b.live[m++] = i; }
      if (getenv("HS_TRACE_BATCH") && atoi(getenv("HS_TRACE_BATCH")) > 1) {
        unsigned long long acc[4] = {0, 0, 0, 0}; uint32_t mx = 0;
        for (uint32_t i = 0; i < nslots; i++) { batch_slot_counters(bs, wind, i, acc); const uint32_t cs = wind ? b.wind[i].c_steps : b.water[i].c_steps; if (cs > mx) mx = cs; }
        fprintf(stderr, "[hostsim] %s slots %u chunk %u live %u -> %u tshift %d maxsteps %d steps %llu maxsteps_slot %u\n", wind ? "wind" : "water", nslots, chunk, nlive, m, sc.tshift, sc.maxsteps, acc[0], mx);
      }
      nlive = m;
      g_batch_stats[0] += (uint32_t)chunk_epochs;
    }
    g_batch_stats[1]++;
    if (getenv("HS_TRACE_BATCH")) fprintf(stderr, "[hostsim] batched %s generation: %u slots, %u epochs\n", wind ? "wind" : "water", nslots, epoch);
    { unsigned long long acc[4] = {0, 0, 0, 0}; for (uint32_t i = 0; i < nslots; i++) batch_slot_counters(bs, wind, i, acc); batch_commit_counters(s, wind, acc); }
    // children -> next generation, in key order
    uint32_t nc = b.ctrl[BC_NCHILD]; if (nc > b.bs.child_cap) nc = b.bs.child_cap;
    g_batch_stats[2] += b.ctrl[BC_CHILD_LOST];
    kids.assign(b.children.begin(), b.children.begin() + nc);
    std::sort(kids.begin(), kids.end(), [](const BChild& a, const BChild& d) { return a.key < d.key; });
    b.ctrl[BC_NCHILD] = 0;
    first = false;
    nslots = wind ? 0 : nc;
  }
}
extern "C" {
void hs_batched_water(hs_ctx* c, int n, int nthreads, unsigned shuffle) { BatchBuffers b; batch_generations(c, b, false, (uint32_t)n, {}, nthreads, shuffle); }
void hs_batched_wind(hs_ctx* c, int n, int nthreads, unsigned shuffle) { BatchBuffers b; batch_generations(c, b, true, (uint32_t)n, {}, nthreads, shuffle); }
// One sweep of the grid pass over the tile columns [tx_lo, tx_hi): the tiles with work colour by colour (tx mod 3, ty mod 3) -- the order the
// device's dataflow kernel enforces between tiles whose regions overlap (soilmx.hip k_grid_tiles_flow); the tiles of one colour own disjoint
// regions and run on the host threads in any order.
static unsigned long long grid_sweep(hs_ctx* c, BatchBuffers& b, int phase, int tx_lo, int tx_hi, int nthreads, unsigned shuffle) {
  const DevState& s = c->d;
  const int gtx = (s.dimx + (1 << GRID_SHIFT) - 1) >> GRID_SHIFT, gty = (s.dimy + (1 << GRID_SHIFT) - 1) >> GRID_SHIFT;
  if (tx_lo < 0) tx_lo = 0;
  if (tx_hi > gtx) tx_hi = gtx;
  b.bs.tshift = GRID_SHIFT; b.bs.sphase = phase;
  const BatchShared& bs = b.bs;
  const unsigned long long keybase = (unsigned long long)phase * 9ull * (unsigned long long)gtx * (unsigned long long)gty;   // key prefix = (phase * 9 + colour) * tiles + tile
  unsigned long long visited = 0;
  for (uint32_t col = 0; col < 9u; col++) {
    std::vector<uint32_t> tiles;
    for (int tx = tx_lo; tx < tx_hi; tx++) {
      if (tx % 3 != (int)(col / 3u) || !batch_grid_tile_phase(s, bs, tx)) continue;
      for (int ty = (int)(col % 3u); ty < gty; ty += 3) tiles.push_back((uint32_t)tx * (uint32_t)gty + (uint32_t)ty);
    }
    std::vector<unsigned long long> vis(tiles.size(), 0);
    parallel_for((uint32_t)tiles.size(), nthreads, [&](uint32_t i) {
      const int tx = (int)(tiles[i] / (uint32_t)gty), ty = (int)(tiles[i] % (uint32_t)gty);
      BatchPolicy pol(s, bs); Sim<BatchPolicy> sim(s, s.soils, pol);
      Frame st[MAX_FRAMES];
      vis[i] = batch_grid_tile(sim, tx, ty, keybase + (unsigned long long)col * (unsigned long long)gtx * gty + (unsigned long long)tiles[i], st);
      sim.flush_counters(); pol.finish(s);
    }, shuffle ? shuffle + 3 + col : 0);
    for (unsigned long long v : vis) visited += v;
    batch_merge_freed(c, b);
  }
  return visited;
}
void hs_batched_grid(hs_ctx* c, int nthreads, unsigned shuffle) {
  const DevState& s = c->d;
  const size_t n = c->n;
  std::fill(c->a0.begin(), c->a0.end(), 0); std::fill(c->a1.begin(), c->a1.end(), 0); std::fill(c->a2.begin(), c->a2.end(), 0);
  for (size_t cc = 0; cc < n; cc++) {                       // k_grid_classify
    const int x = (int)(cc / s.dimy), y = (int)(cc % s.dimy);
    bool act = (s.flags[cc] & F_SAT) != 0;
    for (int dx = -1; dx <= 1 && !act; dx++) for (int dy = -1; dy <= 1; dy++) {
      int nx = x + dx, ny = y + dy; if (nx < 0 || ny < 0 || nx >= s.dimx || ny >= s.dimy) continue;
      if (s.flags[(size_t)nx * s.dimy + ny] & F_AIR) { act = true; break; }
    }
    if (act) c->a0[cc >> 6] |= 1ull << (cc & 63);
  }
  BatchBuffers b; batch_setup(c, b, 0, false);
  const int gtx = (s.dimx + (1 << GRID_SHIFT) - 1) >> GRID_SHIFT;
  unsigned long long visited = 0;
  for (int ph = 0; ph < (b.bs.strips.n > 1 ? 2 : 1); ph++) visited += grid_sweep(c, b, ph, 0, gtx, nthreads, shuffle);
  c->ctr[C_WCASCADE_CALLS] += n - visited;                   // every other cell still "calls" WaterParticle::cascade once
  c->ctr[C_GRID_ACTIVE] += visited;
  uint32_t nc = b.ctrl[BC_NCHILD]; if (nc > b.bs.child_cap) nc = b.bs.child_cap;
  std::vector<BChild> kids(b.children.begin(), b.children.begin() + nc);
  std::sort(kids.begin(), kids.end(), [](const BChild& a, const BChild& d) { return a.key < d.key; });
  if (!kids.empty()) { BatchBuffers b2; batch_generations(c, b2, false, 0, kids, nthreads, shuffle); }   // the nested particles that left their tile's region
}
void hs_set_water_generations(int k) { g_water_generations = k < 1 ? 1 : k; }
void hs_reset_water_schedule() { g_water_generations = SMX_WATER_GENERATIONS; g_stagger_k = 0; g_stagger_gap = SMX_WATER_STAGGER; }   // the library's defaults (tests restore them with this)
void hs_batched_tick(hs_ctx* c, int nwater, int nwind, int dowater, int dowind, int nthreads, unsigned shuffle) {
  if (dowater) {                                              // smx_tick_water: k consecutive top-level generations of n/k -- or ONE staggered phase (smx_set_water_stagger)
    uint32_t left = nwater > 0 ? (uint32_t)nwater : 0u;
    if (g_relaxed && g_stagger_gap > 0u && stagger_k() > 1u && left > 0u) { hs_batched_water(c, (int)left, nthreads, shuffle); left = 0u; }
    for (uint32_t q = 0, k = (uint32_t)g_water_generations; q < k && left > 0; q++) { const uint32_t nq = left / (k - q); left -= nq; if (nq) hs_batched_water(c, (int)nq, nthreads, shuffle); }
    hs_batched_grid(c, nthreads, shuffle);
  }
  if (dowind) hs_batched_wind(c, nwind, nthreads, shuffle);
  if (dowater) hs_freq(c);
}
void hs_batch_stats(uint32_t* out) { for (int i = 0; i < 8; i++) { out[i] = g_batch_stats[i]; g_batch_stats[i] = 0; } }
}  // extern "C"

// ---------------- column strips on several contexts (soilmachine_amd/strips.py drives this step by step) ----------------
// One hs_ctx per strip, each holding the FULL-size map but keeping only its strip + right halo current. The same
// step-wise entry points exist in libsoilmx (smx_d_*): generation begin / spawn / chunk / generation end, halo columns
// and migrating particles packed into byte buffers that the driver moves between contexts (in-process, gloo or RCCL).
struct DPhase { BatchBuffers b; bool wind = false, first = true; uint32_t nslots = 0, epoch = 0, chunk = 0; std::vector<BChild> kids; };
static std::map<hs_ctx*, DPhase*> g_dp;
static DPhase& dp_of(hs_ctx* c) { DPhase*& p = g_dp[c]; if (!p) p = new DPhase(); return *p; }
extern "C" {
void hs_d_set_own(hs_ctx* c, int x0, int x1) { c->own_x0 = x0; c->own_x1 = x1; }
// generation begin: the generation's draws (identical on every context) + the children list; contains_out[i] = what this
// context can say about child i's `contains` (0xFFFFFFFF if the random cell is not in its strip)
void hs_d_gen_begin(hs_ctx* c, int wind, uint32_t nslots, const BChild* kids, uint32_t* contains_out) {
  DPhase& d = dp_of(c);
  d.wind = wind != 0; d.first = kids == nullptr; d.nslots = nslots; d.epoch = 0; d.chunk = 0;
  d.kids.assign(kids ? kids : nullptr, kids ? kids + nslots : nullptr);
  batch_setup(c, d.b, nslots, d.wind);
  if (!kids) batch_draw(c, d.b, nslots);                      // (suspended nested particles carry their state: no draws)
  d.b.ctrl[BC_NCHILD] = 0; d.b.ctrl[BC_CHILD_LOST] = 0;
  if (kids && contains_out) for (uint32_t i = 0; i < nslots; i++) contains_out[i] = batch_child_contains(kids, i);
}
void hs_d_gen_spawn(hs_ctx* c, const uint32_t* contains, uint32_t* nlive_local) {
  DPhase& d = dp_of(c); const DevState& s = c->d; const BatchShared& bs = d.b.bs;
  uint32_t m = 0;
  for (uint32_t i = 0; i < d.nslots; i++) {
    BatchPolicy pol(s, bs); Sim<BatchPolicy> sim(s, s.soils, pol);
    if (d.wind) batch_wind_spawn(sim, bs, i); else batch_water_spawn(sim, bs, i, d.first ? nullptr : &d.kids[i], d.first ? nullptr : contains);
    const uint32_t st = d.wind ? d.b.wind[i].state : d.b.water[i].state;
    if (st != B_DONE && st != B_AWAY) d.b.live[m++] = i;
  }
  *nlive_local = m;
}
int hs_d_next_phase(hs_ctx* c) { DPhase& d = dp_of(c); return d.b.bs.strips.n > 1 ? (int)(d.chunk & 1u) : STRIP_INTERIOR; }
// one chunk of B_CHUNK epochs; the schedule follows the GLOBAL number of running particles
void hs_d_chunk(hs_ctx* c, uint32_t nlive_global, uint32_t* nlive_local) {
  DPhase& d = dp_of(c); const DevState& s = c->d; BatchBuffers& b = d.b; const BatchShared& bs = b.bs;
  uint32_t nlive = 0;                                        // residents that run: rebuild the list (particles may have arrived / left)
  for (uint32_t i = 0; i < d.nslots; i++) { const uint32_t st = d.wind ? b.wind[i].state : b.water[i].state; if (st != B_DONE && st != B_AWAY) b.live[nlive++] = i; }
  const BSched sc = batch_schedule(d.wind, nlive_global, g_batch_dilate, g_batch_cell_above);
  const bool strips = b.bs.strips.n > 1, regrain = sc.tshift != b.bs.tshift;
  batch_apply_schedule(b.bs, s, sc);
  b.bs.sphase = strips ? (int)(d.chunk & 1u) : STRIP_INTERIOR;
  if (strips && d.chunk > 0) d.epoch++;
  d.chunk++;
  if (regrain) { std::fill(b.claim0.begin(), b.claim0.end(), 0ull); std::fill(b.claim1.begin(), b.claim1.end(), 0ull); }
  if (regrain || strips)
    for (uint32_t i = 0; i < nlive; i++) {
      BatchPolicy pol(s, bs); Sim<BatchPolicy> sim(s, s.soils, pol);
      if (d.wind) batch_wind_reclaim(sim, bs, bs.live[i], d.epoch); else batch_water_reclaim(sim, bs, bs.live[i], d.epoch);
    }
  for (int k = 0; k < B_CHUNK; k++, d.epoch++)
    for (uint32_t i = 0; i < nlive; i++) {
      BatchPolicy pol(s, bs); Sim<BatchPolicy> sim(s, s.soils, pol);
      Frame st[MAX_FRAMES];
      if (d.wind) batch_wind_epoch(sim, bs, bs.live[i], d.epoch); else batch_water_epoch(sim, bs, bs.live[i], d.epoch, st);
      sim.flush_counters(); pol.finish(s);
    }
  batch_merge_freed(c, b);
  uint32_t m = 0;
  for (uint32_t i = 0; i < d.nslots; i++) { const uint32_t st = d.wind ? b.wind[i].state : b.water[i].state; if (st != B_DONE && st != B_AWAY) m++; }
  *nlive_local = m;
}
void hs_d_gen_end(hs_ctx* c, BChild* out, uint32_t cap, uint32_t* n) {
  DPhase& d = dp_of(c); const DevState& s = c->d;
  { unsigned long long acc[4] = {0, 0, 0, 0}; for (uint32_t i = 0; i < d.nslots; i++) batch_slot_counters(d.b.bs, d.wind, i, acc); batch_commit_counters(s, d.wind, acc); }
  uint32_t nc = d.b.ctrl[BC_NCHILD]; if (nc > d.b.bs.child_cap) nc = d.b.bs.child_cap; if (nc > cap) nc = cap;
  for (uint32_t i = 0; i < nc; i++) out[i] = d.b.children[i];
  *n = nc;
  d.b.ctrl[BC_NCHILD] = 0;
}
// halo columns: every column of the DIRTY 4x4 grid tiles inside x in [x0, x1) -> [u32 nrec][u32 data bytes][u32 offs[nrec]][records],
// record = {u32 cell, u32 count, f32 wtrack, f32 windfreq, u32 flags, u32 pad, count x {f64 size, floor, sat; u32 type, pad}} bottom -> top
uint64_t hs_d_pack_columns(hs_ctx* c, int x0, int x1, uint8_t* buf, uint64_t cap) {
  const DevState& s = c->d;
  const int gty = (s.dimy + 3) >> 2;
  std::vector<uint32_t> offs; std::vector<uint8_t> data;
  for (int gx = x0 >> 2; gx < ((x1 + 3) >> 2); gx++)
    for (int gy = 0; gy < gty; gy++) {
      uint8_t& dflag = c->dirty[(size_t)gx * gty + gy];
      if (!dflag) continue;
      dflag = 0;
      for (int x = gx << 2; x < (gx << 2) + 4 && x < s.dimx; x++)
        for (int y = gy << 2; y < (gy << 2) + 4 && y < s.dimy; y++) {
          const size_t cc = (size_t)x * s.dimy + y, fi = (size_t)y * s.dimx + x;
          std::vector<const Sec*> col;
          if (c->cells[cc].type != EMPTY) { const Sec* t = &c->cells[cc]; while (true) { col.push_back(t); if (t->prev == NIL) break; t = &c->pool[t->prev]; } }
          offs.push_back((uint32_t)data.size());
          const uint32_t hdr[6] = {(uint32_t)cc, (uint32_t)col.size(), 0, 0, c->flags[cc], 0};
          size_t o = data.size(); data.resize(o + 24 + 32 * col.size());
          memcpy(&data[o], hdr, 24); memcpy(&data[o + 8], &c->wtrack[fi], 4); memcpy(&data[o + 12], &c->windfreq[fi], 4);
          for (size_t k = col.size(); k-- > 0;) { uint8_t* r = &data[o + 24 + 32 * (col.size() - 1 - k)]; memcpy(r, &col[k]->size, 8); memcpy(r + 8, &col[k]->floor, 8); memcpy(r + 16, &col[k]->sat, 8); memcpy(r + 24, &col[k]->type, 4); memset(r + 28, 0, 4); }
        }
    }
  const uint64_t total = 8 + 4 * offs.size() + data.size();
  if (total > cap) return ~0ull;
  const uint32_t nrec = (uint32_t)offs.size(), nb = (uint32_t)data.size();
  memcpy(buf, &nrec, 4); memcpy(buf + 4, &nb, 4);
  if (nrec) memcpy(buf + 8, offs.data(), 4 * offs.size());
  if (nb) memcpy(buf + 8 + 4 * offs.size(), data.data(), data.size());
  return total;
}
void hs_d_unpack_columns(hs_ctx* c, const uint8_t* buf, uint64_t bytes) {
  if (bytes < 8) return;
  uint32_t nrec, nb; memcpy(&nrec, buf, 4); memcpy(&nb, buf + 4, 4);
  const uint8_t* data = buf + 8 + 4 * (size_t)nrec;
  const DevState& s = c->d;
  for (uint32_t i = 0; i < nrec; i++) {
    uint32_t off; memcpy(&off, buf + 8 + 4 * (size_t)i, 4);
    const uint8_t* r = data + off;
    uint32_t cc, count, flags; memcpy(&cc, r, 4); memcpy(&count, r + 4, 4); memcpy(&flags, r + 16, 4);
    const size_t x = cc / (size_t)s.dimy, y = cc % (size_t)s.dimy, fi = y * (size_t)s.dimx + x;
    memcpy(&c->wtrack[fi], r + 8, 4); memcpy(&c->windfreq[fi], r + 12, 4);
    long long live = 0;                                       // give the old chain back, build the new one
    if (c->cells[cc].type != EMPTY) { live--; uint32_t pv = c->cells[cc].prev; while (pv != NIL) { const uint32_t nx = c->pool[pv].prev; c->freelist[c->free_count++] = pv; pv = nx; live--; } }
    Sec cell; cell.size = cell.floor = cell.sat = 0; cell.type = EMPTY; cell.prev = NIL;
    uint32_t pv = NIL;
    for (uint32_t k = 0; k < count; k++) {
      Sec t; const uint8_t* q = r + 24 + 32 * (size_t)k;
      memcpy(&t.size, q, 8); memcpy(&t.floor, q + 8, 8); memcpy(&t.sat, q + 16, 8); memcpy(&t.type, q + 24, 4); t.prev = pv;
      if (k + 1 == count) cell = t; else { const uint32_t nd = c->freelist[--c->free_count]; c->pool[nd] = t; pv = nd; }
      live++;
    }
    c->cells[cc] = cell; c->flags[cc] = (uint8_t)flags;
    c->ctr[C_LIVE_SECTIONS] += (unsigned long long)live;
  }
}
// migrating particles: residents (running) with ipos.x in [x0, x1) -> {u32 slot, record}; they become B_AWAY here
uint64_t hs_d_pack_particles(hs_ctx* c, int x0, int x1, uint8_t* buf, uint64_t cap) {
  DPhase& d = dp_of(c);
  const size_t rs = d.wind ? sizeof(BWind) : sizeof(BWater);
  uint32_t n = 0; uint64_t o = 4;
  for (uint32_t i = 0; i < d.nslots; i++) {
    uint32_t& st = d.wind ? d.b.wind[i].state : d.b.water[i].state;
    if (st == B_DONE || st == B_AWAY) continue;
    const int ix = d.wind ? d.b.wind[i].d.ix : d.b.water[i].w.ix;
    if (ix < x0 || ix >= x1) continue;
    if (o + 4 + rs > cap) return ~0ull;
    memcpy(buf + o, &i, 4); memcpy(buf + o + 4, d.wind ? (const void*)&d.b.wind[i] : (const void*)&d.b.water[i], rs);
    o += 4 + rs; n++;
    st = B_AWAY;
  }
  memcpy(buf, &n, 4);
  return o;
}
void hs_d_unpack_particles(hs_ctx* c, const uint8_t* buf, uint64_t bytes) {
  DPhase& d = dp_of(c);
  if (bytes < 4) return;
  const size_t rs = d.wind ? sizeof(BWind) : sizeof(BWater);
  uint32_t n; memcpy(&n, buf, 4);
  for (uint32_t k = 0; k < n; k++) {
    uint32_t i; memcpy(&i, buf + 4 + k * (4 + rs), 4);
    memcpy(d.wind ? (void*)&d.b.wind[i] : (void*)&d.b.water[i], buf + 8 + k * (4 + rs), rs);
  }
}
// grid pass, strip by strip: classification, then one sweep (9 colours) over the 8x8 tiles of the columns [x_lo, x_hi) that belong to `phase`
void hs_d_grid_begin(hs_ctx* c) {
  const DevState& s = c->d; const size_t n = c->n;
  std::fill(c->a0.begin(), c->a0.end(), 0);
  for (size_t cc = 0; cc < n; cc++) {
    const int x = (int)(cc / s.dimy), y = (int)(cc % s.dimy);
    bool act = (s.flags[cc] & F_SAT) != 0;
    for (int dx = -1; dx <= 1 && !act; dx++) for (int dy = -1; dy <= 1; dy++) {
      int nx = x + dx, ny = y + dy; if (nx < 0 || ny < 0 || nx >= s.dimx || ny >= s.dimy) continue;
      if (s.flags[(size_t)nx * s.dimy + ny] & F_AIR) { act = true; break; }
    }
    if (act) c->a0[cc >> 6] |= 1ull << (cc & 63);
  }
  DPhase& d = dp_of(c);
  d.wind = false; d.nslots = 0; d.first = true;
  batch_setup(c, d.b, 0, false);
  d.b.ctrl[BC_NCHILD] = 0;
}
void hs_d_grid_sweep(hs_ctx* c, int phase, int x_lo, int x_hi) {   // (cell columns, as smx_d_grid_sweep_cols)
  DPhase& d = dp_of(c);
  c->ctr[C_GRID_ACTIVE] += grid_sweep(c, d.b, phase, x_lo >> GRID_SHIFT, (x_hi + (1 << GRID_SHIFT) - 1) >> GRID_SHIFT, 1, 0);
}
}  // extern "C"
