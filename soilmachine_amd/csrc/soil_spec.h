// soil_spec.h -- the SPECULATIVE exact engine: particles of one phase run concurrently IN PLACE, yet the
// result is bit-identical to the reference's sequential order (SoilMachine.cpp:287-307).
//
// Idea (DESIGN.md "Speculative engine"): the reference is sequential because particle i+1 sees what particle i
// wrote. Two particles commute exactly when their cell footprints are disjoint, and footprints are thin curves,
// so at 4096^2 the conflict DAG of a 1000-particle phase is ~20 levels deep (oracle/soil_oracle.cpp so_dep_*).
// We execute level by level without knowing the levels in advance:
//   * stamp[cell]  : id+1 of the last particle that touched the cell in this phase (0 = nobody). Touching a
//                    cell = winning a CAS on its stamp; the first touch snapshots everything the particle may
//                    change there (32-byte top-section record, track / frequency floats, flag byte) into the
//                    particle's undo log. Within one kernel a cell's data is only ever accessed by the ONE
//                    particle that holds its stamp -> plain loads/stores, visibility by kernel boundaries only.
//   * forwarding   : a particle may take over a cell from a LOWER particle that finished in an EARLIER round
//                    (it then sees that particle's writes, as the sequential order demands) and records the
//                    dependency; it must never read a cell last touched by a HIGHER particle -> it flags that
//                    particle (kill) and gives up for this round.
//   * claim[cell]  : lowest still-pending particle that is expected to need the cell (from a read-only scout
//                    run and from the footprint of its previous attempts). A particle yields at cells claimed
//                    by a lower pending particle, so the lower id always wins a conflict (deterministic
//                    reservations) and every round completes one full level of the conflict DAG.
//   * boundary     : between rounds one workgroup computes the kill closure (flags, dependencies, the slot
//                    chain below), rolls killed particles back in DESCENDING id order, commits the finished
//                    prefix and rebuilds the pending list.
//   * rand() SLOTS : the phase's rand() stream is a sequence of draw pairs ("slots"). The reference hands slot m to
//                    the next top-level particle, or -- while a particle floods -- to the nested particles it spawns
//                    (water.h:251), which take the slots right after their parent's. Particle ids ARE slot numbers:
//                    id m spawns from draws 2m, 2m+1 and its nested particles use slots m+1, m+2, ...  Every slot
//                    is run speculatively as a top-level particle; when a lower particle turns out to spawn k
//                    nested ones, only the k slots they consume are voided (their tentative runs are undone) -- the
//                    trajectories of all later slots stay valid, and the phase simply ends k slots later. The
//                    phase is over when n top-level particles are committed.
//   * suspend      : a particle that cannot acquire the block of its next step (claimed by a lower pending
//                    particle, or held by one that is still running) does not throw its work away: it keeps its
//                    cells, checkpoints {particle, carried block, log cursor} and resumes at that very step in the
//                    next round. Work is done once; a round only costs the distance the particles advance in it.
// Progress: the lowest pending particle can only be stopped by higher particles, which yield to its claims;
// the host falls back to the serial walker if the committed prefix stalls.
#pragma once
#include "soil_core.h"

#ifdef SMX_HOSTSIM
#define SMX_A_LOAD(p) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define SMX_A_STORE(p, v) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)
#define SMX_A_ADD(p, v) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
#define SMX_A_MIN(p, v) smx::host_atomic_min((p), (v))
#define SMX_A_MAX(p, v) smx::host_atomic_max((p), (v))
namespace smx {
inline uint32_t host_atomic_min(uint32_t* p, uint32_t v) {
  uint32_t o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (v < o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return o;
}
inline uint32_t host_atomic_max(uint32_t* p, uint32_t v) {
  uint32_t o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return o;
}
inline uint32_t host_atomic_cas(uint32_t* p, uint32_t expected, uint32_t desired) {
  __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return expected;
}
}  // namespace smx
#define SMX_A_CAS(p, e, d) smx::host_atomic_cas((p), (e), (d))
#define SMX_A_SUB(p, v) __atomic_fetch_sub((p), (v), __ATOMIC_SEQ_CST)
#define SMX_A_ADD64(p, v) __atomic_fetch_add((p), (unsigned long long)(v), __ATOMIC_SEQ_CST)
#define SMX_A_MAX64(p, v) smx::host_atomic_max64((p), (v))
#define SMX_A_XCHG(p, v) __atomic_exchange_n((p), (v), __ATOMIC_SEQ_CST)
namespace smx {
inline unsigned long long host_atomic_max64(unsigned long long* p, unsigned long long v) {
  unsigned long long o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return o;
}
}  // namespace smx
#else
#define SMX_A_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SMX_A_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SMX_A_ADD(p, v) atomicAdd((p), (v))
#define SMX_A_MIN(p, v) atomicMin((p), (v))
#define SMX_A_MAX(p, v) atomicMax((p), (v))
#define SMX_A_CAS(p, e, d) atomicCAS((p), (e), (d))
#define SMX_A_SUB(p, v) atomicSub((p), (v))
#define SMX_A_ADD64(p, v) atomicAdd((p), (unsigned long long)(v))
#define SMX_A_MAX64(p, v) atomicMax((p), (unsigned long long)(v))
#define SMX_A_XCHG(p, v) atomicExch((p), (v))
#endif

namespace smx {

constexpr uint32_t ST_PENDING = 0, ST_DONE_NEW = 1, ST_DONE_OLD = 2, ST_COMMITTED = 3, ST_SUSPENDED = 4;
constexpr uint32_t POISON = 0xFFFFFFFEu;      // stamp of a cell released by a particle that aborted in this round
constexpr uint32_t NOCLAIM = 0xFFFFFFFFu;
constexpr uint32_t NODE_ENTRY = 0x80000000u;  // log entry describes pool node (cell & 0x7fffffff), not a cell
constexpr int LOG_CHUNK = 512;                // log entries per arena chunk
constexpr int MAX_CHUNKS = 1024;              // chunks per particle (524288 touched cells)
constexpr int MAX_DEPS = 28;
constexpr int MAX_NODELOG = 2048;             // node allocations / frees per particle attempt
constexpr int SPEC_NCTR = 9;
constexpr uint32_t SPEC_SOLO_AFTER = 3;       // rounds without a commit after which the lowest particle runs alone
constexpr uint32_t SPEC_STALL_LIMIT = 24;     // ... and after which the host gives the phase to the serial walker
constexpr uint32_t SPEC_SCOUT_MARGIN = 64;    // slots past the n-th that are scouted too (they run when nested particles void lower slots)
constexpr uint32_t SPEC_MAXNEST = 1024;        // default of SpecShared::maxnest: nested particles one speculative sub-phase can absorb before it
                                              // is cut and re-armed; a sub-phase of n particles owns n + maxnest slots (<= 65535: SpecWork::nested)

struct alignas(64) SpecLog {                  // undo record of one touched cell (or one modified pool node)
  uint32_t cell, prev;                        // cell index, stamp value it replaced
  Sec rec;                                    // the 32-byte record at first touch
  float wtrack, windfreq;
  uint32_t flags, pad;
};

struct SpecPart {                             // one particle of the phase
  uint32_t status, kill, aborted, fail_cell;
  uint32_t noff, nested, nlog, nchunks;            // (noff: unused since particle ids are rand() slots; keeps the layout)
  uint32_t ndeps, dep_overflow, nalloc, nfree;
  uint32_t overflow, attempts, nnode, blocked;     // nnode: NODE_ENTRY records in the log
  long long live_delta;
  unsigned long long ctr[SPEC_NCTR];
  uint32_t deps[MAX_DEPS];
};

struct SpecCkpt {                             // checkpoint of a suspended particle (resumes at interact())
  Water w; Wind d; Blk blk;
  uint32_t nlog, nchunks, ndeps, dep_overflow, nalloc, nfree, nested_drawn, nnode;
  long long live_delta;
  unsigned long long ctr[SPEC_NCTR];
};

enum SpecCtrl { SC_NPENDING = 0, SC_BASE, SC_N, SC_NESTED_COMMITTED, SC_NODE_CURSOR, SC_NRECLAIM, SC_NFREE_START,
                SC_ROUNDS, SC_ABORTS, SC_KILLS, SC_RECLAIM_LOST, SC_CHUNK_CURSOR, SC_SKIPPED, SC_EXECUTED, SC_SUSPENDS, SC_ROUND_MAXSTEPS, SC_SUM_MAXSTEPS,
                SC_NSLOTS, SC_TOP, SC_STALL, SC_SOLO, SC_HIGH, SC_POOL_CROSSED, SC_COUNT = 24 };

struct SpecShared {
  uint32_t maxnest = SPEC_MAXNEST;   // nested-particle slots of a sub-phase (host: smx SMX_SPEC_MAXNEST / hs_set_spec_maxnest)
  uint32_t* stamp;
  uint32_t* claim;
  SpecPart* part;
  uint32_t* chunk_tab;      // [n][MAX_CHUNKS]
  SpecLog* arena;
  uint32_t arena_chunks;
  uint32_t* alloc_log;      // [n][MAX_NODELOG]
  uint32_t* free_log;       // [n][MAX_NODELOG]
  uint32_t* pending;        // ids of the particles to run this round
  uint32_t* ctrl;           // SpecCtrl
  uint32_t* draws;          // the phase's rand() stream, peeked ahead: RAW 32-bit generator outputs (rand() = raw >> 1)
  uint32_t ndraws;
  uint32_t* reclaim;        // pool nodes that return to the free stack at phase end
  uint32_t reclaim_cap;
  uint32_t* scout_cells;    // [n][scout_cap] cells of the read-only scout run (predicted footprint)
  uint32_t scout_cap;       // cells one scouted footprint can hold (longer ones are truncated)
  uint32_t* scout_len;      // [n]
  SpecCkpt* ckpt;           // [n]
  uint32_t mode;            // bit0: blocked particles sit the round out (select pass); bit1: suspend/resume
};

SMX_D SpecLog* spec_entry(const SpecShared& sh, uint32_t pid, uint32_t k) {
  const uint32_t cid = sh.chunk_tab[(size_t)pid * MAX_CHUNKS + k / LOG_CHUNK];
  return sh.arena + (size_t)cid * LOG_CHUNK + (k % LOG_CHUNK);
}
SMX_D size_t spec_fidx(const DevState& s, size_t c) {        // cell order -> frequency order
  const size_t x = c / (size_t)s.dimy, y = c - x * (size_t)s.dimy;
  return y * (size_t)s.dimx + x;
}

// ---------------- read-only scout: records the predicted footprint of a particle ----------------
struct ScoutPolicy {
  static constexpr bool READ_ONLY = true;
  static constexpr bool DEFER_NESTED = false;
  const SpecShared& sh;
  uint32_t pid, len = 0;
  uint32_t* out;
  size_t last = ~(size_t)0, last2 = ~(size_t)0;
  SMX_D ScoutPolicy(const SpecShared& sh_, uint32_t p) : sh(sh_), pid(p), out(sh_.scout_cells + (size_t)p * sh_.scout_cap) {}
  template <int N> struct Pre {};
  SMX_D void save_blk(const Blk&) {}
  SMX_D void load_blk(Blk&) {}
  SMX_D bool touch(const DevState&, size_t c, size_t = 0) {
    if (c != last && c != last2) { last2 = last; last = c; if (len < sh.scout_cap) out[len++] = (uint32_t)c; }
    return true;
  }
  template <int N> SMX_D void prefetch(const DevState&, const size_t (&)[N], const size_t (&)[N], Pre<N>&) {}
  template <int N> SMX_D bool acquire_log(const DevState& s, const size_t (&c)[N], const Pre<N>&, const Sec (&)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) touch(s, c[i]);
    return true;
  }
  SMX_D void finish() { sh.scout_len[pid] = len; }
  SMX_D void pre_write_node(const DevState&, uint32_t) {}
  SMX_D void set_flag(const DevState&, size_t, uint8_t) {}
  SMX_D void clear_flag(const DevState&, size_t, uint8_t) {}
  SMX_D bool can_get(const DevState&) const { return true; }
  SMX_D void live_add(const DevState&, int) {}
  SMX_D uint32_t node_alloc(const DevState&) { return NIL; }
  SMX_D void node_free(const DevState&, uint32_t) {}
  SMX_D void rand2(const DevState&, int& a, int& b) { a = 0; b = 0; }
  SMX_D void add_counter(const DevState&, int, unsigned long long) {}
};

// ---------------- in-place speculative execution ----------------
struct SpecPolicy {
  static constexpr bool READ_ONLY = false;
  static constexpr bool DEFER_NESTED = false;
  static constexpr bool OUTLINE_RECURSION = true;           // soil_core.h wind_recurse
  // take over the cursors of a copy that ran a piece of this particle out of line (same particle, same log)
  SMX_D void adopt(const SpecPolicy& o) {
    nlog = o.nlog; nchunks = o.nchunks; ndeps = o.ndeps; dep_overflow = o.dep_overflow; nalloc = o.nalloc; nfree = o.nfree;
    nested_drawn = o.nested_drawn; overflow = o.overflow; nnode = o.nnode; fail_cell = o.fail_cell; live_delta = o.live_delta;
    for (int i = 0; i < SPEC_NCTR; i++) ctr[i] = o.ctr[i];
    last = o.last; chunk = o.chunk;
  }
  const SpecShared& sh;
  uint32_t pid, me;
  SpecPart* part;
  uint32_t nlog = 0, nchunks, ndeps = 0, dep_overflow = 0, nalloc = 0, nfree = 0, nested_drawn = 0, overflow = 0, nnode = 0;
  uint32_t fail_cell = NOCLAIM;
  long long live_delta = 0;
  unsigned long long ctr[SPEC_NCTR];
  size_t last = ~(size_t)0;

  SMX_D SpecPolicy(const SpecShared& sh_, uint32_t p, bool resume = false) : sh(sh_), pid(p), me(p + 1), part(sh_.part + p) {
    nchunks = part->nchunks;
    for (int i = 0; i < SPEC_NCTR; i++) ctr[i] = 0;
    if (resume) {                                           // continue the undo log etc. of the suspended attempt
      const SpecCkpt& k = sh.ckpt[pid];
      nlog = k.nlog; nchunks = k.nchunks; ndeps = k.ndeps; dep_overflow = k.dep_overflow; nalloc = k.nalloc; nfree = k.nfree;
      nested_drawn = k.nested_drawn; nnode = k.nnode; live_delta = k.live_delta;
      for (int i = 0; i < SPEC_NCTR; i++) ctr[i] = k.ctr[i];
      if (nlog % LOG_CHUNK) chunk = sh.arena + (size_t)sh.chunk_tab[(size_t)pid * MAX_CHUNKS + nlog / LOG_CHUNK] * LOG_CHUNK;
    }
  }
  SMX_D void save_blk(const Blk& b) { sh.ckpt[pid].blk = b; }
  SMX_D void load_blk(Blk& b) { b = sh.ckpt[pid].blk; }
  // the particle could not acquire the block of its next step: keep everything, continue next round
  SMX_D void suspend(const DevState&) {
    SpecCkpt& k = sh.ckpt[pid];
    k.nlog = nlog; k.nchunks = nchunks; k.ndeps = ndeps; k.dep_overflow = dep_overflow; k.nalloc = nalloc; k.nfree = nfree;
    k.nested_drawn = nested_drawn; k.nnode = nnode; k.live_delta = live_delta;
    for (int i = 0; i < SPEC_NCTR; i++) k.ctr[i] = ctr[i];
    part->nlog = nlog; part->nchunks = nchunks; part->ndeps = ndeps; part->dep_overflow = dep_overflow;
    part->nalloc = nalloc; part->nfree = nfree; part->nnode = nnode; part->fail_cell = fail_cell; part->nested = 0;
    part->aborted = 0; part->attempts++;
    SMX_A_ADD(&sh.ctrl[SC_SUSPENDS], 1u);
    SMX_A_STORE(&part->status, ST_SUSPENDED);
  }

  SpecLog* chunk = nullptr;                                 // arena chunk that receives the next log entries
  SMX_D bool next_chunk() {                                 // nlog is a multiple of LOG_CHUNK: open chunk nlog / LOG_CHUNK
    const uint32_t ch = nlog / LOG_CHUNK;
    uint32_t cid;
    if (ch < nchunks) cid = sh.chunk_tab[(size_t)pid * MAX_CHUNKS + ch];      // chunk kept from an earlier attempt
    else {
      if (ch >= (uint32_t)MAX_CHUNKS) { overflow = 1; return false; }
      cid = SMX_A_ADD(&sh.ctrl[SC_CHUNK_CURSOR], 1u);
      if (cid >= sh.arena_chunks) { overflow = 1; return false; }
      sh.chunk_tab[(size_t)pid * MAX_CHUNKS + ch] = cid;
      nchunks = ch + 1;
    }
    chunk = sh.arena + (size_t)cid * LOG_CHUNK;
    return true;
  }
  SMX_D SpecLog* append(uint32_t cellkey, uint32_t prev) {
    const uint32_t off = nlog % LOG_CHUNK;
    if (off == 0 && !next_chunk()) return nullptr;
    nlog++;
    SpecLog* e = chunk + off;
    e->cell = cellkey; e->prev = prev;
    return e;
  }

  SMX_D bool touch(const DevState& s, size_t c, size_t fi) {
    if (c == last) return true;
    const uint32_t v = SMX_A_LOAD(&sh.stamp[c]);
    if (v == me) { last = c; return true; }
    return touch_slow(s, c, fi, v);
  }
  SMX_D void add_dep(uint32_t owner) {
    bool have = false;
    for (uint32_t i = 0; i < ndeps; i++) have |= (part->deps[i] == owner);
    if (!have) { if (ndeps < (uint32_t)MAX_DEPS) part->deps[ndeps++] = owner; else dep_overflow = 1; }
  }
  // ---- batched acquisition. prefetch() issues EVERY load a batch can need in one go -- stamps, claims and the
  // words an undo record stores besides the 32-byte section record -- for all N cells, owned or not. Reading a
  // cell's data before owning it is safe: data only changes under an owner, and if anybody owned the cell between
  // our stamp load and our CAS the CAS fails (a stamp never returns to an old value inside a round) and the attempt
  // is thrown away. acquire_log() then needs at most one more round trip (owner status, only for cells taken over)
  // before the CAS. Cells that are already ours cost nothing but the prefetch.
  template <int N> struct Pre { uint32_t v[N], cl[N], fl[N]; float tr[N], wf[N]; };
  template <int N> SMX_D void prefetch(const DevState& s, const size_t (&c)[N], const size_t (&f)[N], Pre<N>& p) {
#pragma unroll
    for (int i = 0; i < N; i++) {
      // PLAIN loads (L1-cacheable): claims do not change during the kernel; a stale stamp is harmless -- a cell
      // somebody else took meanwhile makes our CAS fail, and a cell we own but whose cached stamp is older than
      // our own CAS is recognised by the CAS returning `me`
      p.v[i] = sh.stamp[c[i]]; p.cl[i] = sh.claim[c[i]];
      p.tr[i] = s.wtrack[f[i]]; p.wf[i] = s.windfreq[f[i]]; p.fl[i] = s.flags[c[i]];
    }
  }
  template <int N> SMX_D bool acquire_log(const DevState&, const size_t (&c)[N], const Pre<N>& p, const Sec (&r)[N]) {
    uint32_t need = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      bool nd = (p.v[i] != me);
#pragma unroll
      for (int j = 0; j < i; j++) if (c[j] == c[i]) nd = false;   // a batch may name a cell twice (map border)
      need |= nd ? (1u << i) : 0u;
    }
    if (!need) return true;
    bool fail = false;
    uint32_t takeover = 0;                                  // cells last touched by a lower particle: its status decides
#pragma unroll
    for (int i = 0; i < N; i++) {
      if ((need >> i) & 1) {
        if (p.v[i] == POISON || p.v[i] > me) {
          if (p.v[i] != POISON) SMX_A_STORE(&sh.part[p.v[i] - 1].kill, 1u);   // a later particle got here first: undo it
          fail_cell = (uint32_t)c[i]; fail = true;
        } else if (p.cl[i] < pid) { fail_cell = (uint32_t)c[i]; fail = true; }   // yield to a lower pending particle
        else if (p.v[i] != 0) takeover |= 1u << i;
      }
    }
    if (fail) return false;
    uint32_t depmask = 0;
    if (takeover) {
      uint32_t stv[N];
#pragma unroll
      for (int i = 0; i < N; i++) stv[i] = ((takeover >> i) & 1) ? sh.part[p.v[i] - 1].status : ST_COMMITTED;   // plain: finished states are final
#pragma unroll
      for (int i = 0; i < N; i++) {
        if (stv[i] == ST_PENDING || stv[i] == ST_DONE_NEW || stv[i] == ST_SUSPENDED) { fail_cell = (uint32_t)c[i]; fail = true; }   // not finished before this round
        if (stv[i] == ST_DONE_OLD) depmask |= 1u << i;
      }
      if (fail) return false;
    }
    uint32_t got = 0;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; i++) {
      if ((need >> i) & 1) {
        const uint32_t old = SMX_A_CAS(&sh.stamp[c[i]], p.v[i], me);
        if (old == p.v[i]) got |= 1u << i;
        else if (old != me) { fail_cell = (uint32_t)c[i]; ok = false; }   // somebody took it meanwhile: give up
        // (old == me: the cached stamp was older than our own earlier acquisition -- already ours and logged)
      }
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
      if ((got >> i) & 1) {
        SpecLog* e = append((uint32_t)c[i], p.v[i]);
        if (!e) { SMX_A_STORE(&sh.stamp[c[i]], p.v[i]); ok = false; continue; }   // log full: overflow is set, the attempt is void
        e->rec = r[i]; e->wtrack = p.tr[i]; e->windfreq = p.wf[i]; e->flags = p.fl[i];
        if ((depmask >> i) & 1) add_dep(p.v[i] - 1);
      }
    }
    if (ok) last = c[N - 1];
    return ok;
  }
  SMX_D bool touch_slow(const DevState& s, size_t c, size_t fi, uint32_t v) {
    while (true) {
      if (v == me) return true;                             // (a batch may name the same cell twice)
      if (v == POISON) { fail_cell = (uint32_t)c; return false; }
      if (v > me) {                                         // a later particle got here first: it must be undone
        SMX_A_STORE(&sh.part[v - 1].kill, 1u);
        fail_cell = (uint32_t)c;
        return false;
      }
      bool dep = false;
      if (v != 0) {                                         // last touched by a lower particle
        const uint32_t st = SMX_A_LOAD(&sh.part[v - 1].status);
        if (st == ST_PENDING || st == ST_DONE_NEW || st == ST_SUSPENDED) { fail_cell = (uint32_t)c; return false; }   // not finished before this round
        dep = (st == ST_DONE_OLD);
      }
      if (SMX_A_LOAD(&sh.claim[c]) < pid) { fail_cell = (uint32_t)c; return false; }   // yield to a lower pending particle
      const uint32_t old = SMX_A_CAS(&sh.stamp[c], v, me);
      if (old != v) { v = old; continue; }
      SpecLog* e = append((uint32_t)c, v);
      if (!e) { SMX_A_STORE(&sh.stamp[c], v); fail_cell = (uint32_t)c; return false; }
      e->rec = s.cells[c]; e->wtrack = s.wtrack[fi]; e->windfreq = s.windfreq[fi]; e->flags = s.flags[c];
      if (dep) add_dep(v - 1);
      last = c;
      return true;
    }
  }
  SMX_D void pre_write_node(const DevState& s, uint32_t node) {
    SpecLog* e = append(NODE_ENTRY | node, 0);
    if (e) { e->rec = s.pool[node]; nnode++; }
  }
  SMX_D void set_flag(const DevState& s, size_t c, uint8_t f) { const uint8_t v = s.flags[c]; if (!(v & f)) s.flags[c] = v | f; }
  SMX_D void clear_flag(const DevState& s, size_t c, uint8_t f) { const uint8_t v = s.flags[c]; if (v & f) s.flags[c] = v & (uint8_t)~f; }
  SMX_D bool can_get(const DevState&) const { return true; }   // the host only starts a speculative phase with pool headroom
  SMX_D void live_add(const DevState&, int d) { live_delta += d; }
  SMX_D uint32_t node_alloc(const DevState& s) {
    if (nalloc >= (uint32_t)MAX_NODELOG) { overflow = 1; return NIL; }
    const uint32_t cur = SMX_A_ADD(&sh.ctrl[SC_NODE_CURSOR], 1u);
    const uint32_t nstart = sh.ctrl[SC_NFREE_START];
    if (cur >= nstart) { overflow = 1; return NIL; }
    const uint32_t nd = s.freelist[nstart - 1 - cur];
    sh.alloc_log[(size_t)pid * MAX_NODELOG + nalloc++] = nd;
    return nd;
  }
  SMX_D void node_free(const DevState&, uint32_t e) {       // deferred: the node keeps its content until the phase ends
    if (nfree >= (uint32_t)MAX_NODELOG) { overflow = 1; return; }
    sh.free_log[(size_t)pid * MAX_NODELOG + nfree++] = e;
  }
  SMX_D void rand2(const DevState&, int& first, int& second) {   // draws of a NESTED particle (water.h:251)
    const uint32_t k = 2u * (pid + 1u + nested_drawn);     // the slots right after this particle's own
    nested_drawn++;
    if (k + 1 >= sh.ndraws) { overflow = 1; first = second = 0; return; }
    first = (int)(sh.draws[k] >> 1); second = (int)(sh.draws[k + 1] >> 1);
  }
  SMX_D void spawn_draws(int& first, int& second) const {
    const uint32_t k = 2u * pid;                           // particle id = slot of the phase's draw-pair sequence
    first = (int)(sh.draws[k] >> 1); second = (int)(sh.draws[k + 1] >> 1);
  }
  SMX_D void add_counter(const DevState&, int which, unsigned long long v) { if (which < SPEC_NCTR) ctr[which] += v; }

  SMX_D void reclaim_nodes(const uint32_t* list, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t k = SMX_A_ADD(&sh.ctrl[SC_NRECLAIM], 1u);
      if (k < sh.reclaim_cap) sh.reclaim[k] = list[i]; else SMX_A_ADD(&sh.ctrl[SC_RECLAIM_LOST], 1u);
    }
  }
  // the particle aborted: undo its writes now (it still holds every cell it touched), then poison those cells
  SMX_D void self_rollback(const DevState& s) {
    for (uint32_t k = nlog; k-- > 0;) {
      const SpecLog* e = spec_entry(sh, pid, k);
      if (e->cell & NODE_ENTRY) { s.pool[e->cell & ~NODE_ENTRY] = e->rec; continue; }
      const size_t c = e->cell, fi = spec_fidx(s, c);
      s.cells[c] = e->rec; s.wtrack[fi] = e->wtrack; s.windfreq[fi] = e->windfreq; s.flags[c] = (uint8_t)e->flags;
      SMX_A_STORE(&sh.stamp[c], POISON);
    }
    reclaim_nodes(sh.alloc_log + (size_t)pid * MAX_NODELOG, nalloc);
    part->nlog = nlog; part->nchunks = nchunks; part->fail_cell = fail_cell; part->overflow = overflow; part->nnode = nnode;
    part->ndeps = 0; part->dep_overflow = 0; part->nalloc = 0; part->nfree = 0; part->nested = 0;
    part->aborted = 1; part->attempts++;
    SMX_A_ADD(&sh.ctrl[SC_ABORTS], 1u);
    SMX_A_STORE(&part->status, ST_PENDING);
  }
  SMX_D void finish_done(const DevState&) {
    part->nlog = nlog; part->nchunks = nchunks; part->ndeps = ndeps; part->dep_overflow = dep_overflow;
    part->nalloc = nalloc; part->nfree = nfree; part->nested = nested_drawn; part->live_delta = live_delta;
    part->overflow = overflow; part->fail_cell = NOCLAIM; part->aborted = 0; part->attempts++; part->nnode = nnode;
    for (int i = 0; i < SPEC_NCTR; i++) part->ctr[i] = ctr[i];
    SMX_A_STORE(&part->status, ST_DONE_NEW);
  }
};

// ---------------- round boundary (ONE workgroup; everything it reads was written by earlier kernels) --------
// 1. kill closure over the window [base, end): kill flags set by lower particles, log/node overflows, dependencies
//    on killed particles, tentative runs on slots that a lower particle's nested particles consumed (slot chain).
// 2. roll killed particles back (data + stamps) and release the stamps of particles that aborted by themselves,
//    in DESCENDING id order (a cell taken over from a lower tentative particle must be unwound top-down);
//    2b. clear the whole take-over chain on the cell the lowest particle gave up at.
// 3. DONE_NEW -> DONE_OLD, commit the finished prefix (and void the slots its nested particles used), rebuild the
//    pending list (or, after SPEC_SOLO_AFTER rounds without a commit, just the lowest particle).
// Written for a TEAM of `nt` threads (thread `tid`); `sync` is the team barrier. The scalar passes run on
// thread 0 over the staged arrays in `w` (LDS on the device).
constexpr uint32_t SPEC_WINDOW = 4096;        // particles one boundary pass looks at (ids base .. base+window)
// (thread 0's scans read the byte arrays FOUR SLOTS PER LOAD: one scan is a chain of LDS round trips, and the five
//  of them were most of the boundary's time -- profiles/r01_kernel_stats.csv, k_spec_boundary)
SMX_D uint32_t spec_w4(const uint8_t* a, uint32_t i0) { uint32_t v; __builtin_memcpy(&v, a + i0, 4); return v; }
SMX_D void spec_w4_put(uint8_t* a, uint32_t i0, uint32_t v) { __builtin_memcpy(a + i0, &v, 4); }
struct alignas(8) SpecWork {                  // staged per-particle state, index = pid - base
  alignas(8) uint8_t st[SPEC_WINDOW], kl[SPEC_WINDOW], ab[SPEC_WINDOW], dov[SPEC_WINDOW];
  uint16_t nested[SPEC_WINDOW];               // <= SPEC_MAXNEST
  uint8_t mask[SPEC_WINDOW];                  // slot is not a top-level particle (now): consumed by a lower particle's
                                              // nested draws, or beyond the phase's n-th particle
  uint16_t list[SPEC_WINDOW];                 // particles that need a rollback / stamp release (descending);
                                              // reused by the commit step: nfree of the committed prefix
  uint32_t nlist, changed, base, end;
  uint32_t ncommit, rbase;                    // slots leaving the window now; running (uncapped) reclaim cursor
  uint32_t next_end, rank_end, top;           // slot chain / particle rank at the window's end; committed particles
  unsigned long long acc[SPEC_NCTR + 3];      // commit sums: ctr[0..NCTR), live_delta, nested, freed nodes
};

template <class SyncF>
SMX_D void spec_boundary_team(const DevState& s, const SpecShared& sh, SpecWork& w, uint32_t tid, uint32_t nt, SyncF sync) {
  uint32_t* ctrl = sh.ctrl;
  const uint32_t n = ctrl[SC_N], nslots = ctrl[SC_NSLOTS];
  if (tid == 0) {                                          // the window stops at the highest slot that ever became pending:
    w.base = ctrl[SC_BASE];                                // the slots above never ran and hold nothing
    uint32_t e = (nslots - w.base > SPEC_WINDOW) ? w.base + SPEC_WINDOW : nslots;
    const uint32_t high = ctrl[SC_HIGH] > w.base ? ctrl[SC_HIGH] : w.base + 1u;
    w.end = e < high ? e : high;
  }
  sync();
  const uint32_t base = w.base, end = w.end, cnt = end - base;
  for (uint32_t i = tid; i < cnt; i += nt) {
    const SpecPart& q = sh.part[base + i];
    w.st[i] = (uint8_t)q.status; w.kl[i] = (uint8_t)((q.kill != 0) || (q.overflow != 0 && q.status != ST_PENDING));
    w.ab[i] = (uint8_t)q.aborted; w.dov[i] = (uint8_t)q.dep_overflow; w.nested[i] = (uint16_t)q.nested;
  }
  sync();
  while (true) {                                           // (repeated while the lowest particle's way is being cleared, see 2b)
  // ---- 1. kill closure
  while (true) {
    if (tid == 0) w.changed = 0;
    sync();
    for (uint32_t i = tid; i < cnt; i += nt) {             // dependencies (always lower ids)
      if ((w.st[i] == ST_DONE_NEW || w.st[i] == ST_DONE_OLD || w.st[i] == ST_SUSPENDED) && !w.kl[i]) {
        const SpecPart& q = sh.part[base + i];
        bool bad = false;
        for (uint32_t k = 0; k < q.ndeps; k++) { const uint32_t d = q.deps[k]; if (d >= base && w.kl[d - base]) { bad = true; break; } }
        if (bad) { w.kl[i] = 1; w.changed = 1; }
      }
    }
    sync();
    if (tid == 0) {                                        // slot chain + dependency-list overflows, in id order
      // `next` = the slot the reference hands to its next top-level particle: a finished particle's nested
      // particles consumed the slots right after its own (an unfinished one is speculated to spawn none).
      // A tentative run on a slot in between -- or past the phase's n-th particle -- is void.
      uint32_t next = base, rank = ctrl[SC_TOP];
      bool any = false, chg = false;
      for (uint32_t i0 = 0; i0 < cnt; i0 += 4u) {
        const uint32_t st4 = spec_w4(w.st, i0), kl4 = spec_w4(w.kl, i0), dov4 = spec_w4(w.dov, i0);
        const uint32_t ne01 = spec_w4((const uint8_t*)w.nested, 2u * i0), ne23 = spec_w4((const uint8_t*)w.nested, 2u * i0 + 4u);
        uint32_t kln = kl4, mk4 = 0;
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
          if (i0 + u >= cnt) break;
          const uint32_t sh8 = 8u * u, st = (st4 >> sh8) & 0xffu;
          const bool holds = (st == ST_DONE_NEW || st == ST_DONE_OLD || st == ST_SUSPENDED);
          bool kl = ((kln >> sh8) & 0xffu) != 0;
          if (base + i0 + u < next || rank >= n) {
            mk4 |= 1u << sh8;
            if (holds) { if (!kl) { kln |= 1u << sh8; chg = true; } any = true; }
            continue;
          }
          if (holds) {
            if (!kl && ((dov4 >> sh8) & 0xffu) && any) { kln |= 1u << sh8; chg = true; kl = true; }
            if (kl) any = true;
          }
          const uint32_t ne = ((u < 2u ? ne01 : ne23) >> (16u * (u & 1u))) & 0xffffu;
          next = base + i0 + u + 1u + ((holds && !kl) ? ne : 0u);   // (a suspended particle has spawned nothing yet)
          rank++;
        }
        if (kln != kl4) spec_w4_put(w.kl, i0, kln);
        spec_w4_put(w.mask, i0, mk4);
      }
      if (chg) w.changed = 1;
      w.next_end = next; w.rank_end = rank;
    }
    sync();
    const bool again = w.changed != 0;                     // everybody reads the verdict before thread 0 resets it
    sync();
    if (!again) break;
  }
  // ---- 2. rollback list, highest id first
  if (tid == 0) {
    uint32_t m = 0;
    for (uint32_t i0 = (cnt + 3u) & ~3u; i0 >= 4u;) {
      i0 -= 4u;
      const uint32_t st4 = spec_w4(w.st, i0), kl4 = spec_w4(w.kl, i0), ab4 = spec_w4(w.ab, i0);
#pragma unroll
      for (uint32_t u = 4u; u-- > 0u;) {
        if (i0 + u >= cnt) continue;
        const uint32_t sh8 = 8u * u, st = (st4 >> sh8) & 0xffu;
        const bool done = (st == ST_DONE_NEW || st == ST_DONE_OLD || st == ST_SUSPENDED);   // holds cells + log
        if ((done && ((kl4 >> sh8) & 0xffu)) || (!done && ((ab4 >> sh8) & 0xffu))) w.list[m++] = (uint16_t)(i0 + u);
      }
    }
    w.nlist = m;
    w.rbase = ctrl[SC_NRECLAIM];                           // reclaim cursor of this pass (written back once, at the end)
  }
  sync();
  // The whole team works on one particle at a time (descending ids): cells and reclaimed nodes are spread over
  // the threads; thread 0 only unwinds the (rare) buried-section records, whose duplicates must go newest-first.
  for (uint32_t j = 0; j < w.nlist; j++) {
    const uint32_t i = w.list[j], p = base + i;
    SpecPart& q = sh.part[p];
    const bool done = (w.st[i] == ST_DONE_NEW || w.st[i] == ST_DONE_OLD || w.st[i] == ST_SUSPENDED);
    const uint32_t nl = q.nlog;
    if (done) {                                            // killed: restore data and stamps
      const uint32_t na = q.nalloc, rb = w.rbase;
      for (uint32_t k = tid; k < nl; k += nt) {
        const SpecLog* e = spec_entry(sh, p, k);
        if (e->cell & NODE_ENTRY) continue;
        const size_t c = e->cell, fi = spec_fidx(s, c);
        s.cells[c] = e->rec; s.wtrack[fi] = e->wtrack; s.windfreq[fi] = e->windfreq; s.flags[c] = (uint8_t)e->flags;
        sh.stamp[c] = e->prev;
      }
      for (uint32_t k = tid; k < na; k += nt)              // nodes it allocated go back to the pool at phase end
        if (rb + k < sh.reclaim_cap) sh.reclaim[rb + k] = sh.alloc_log[(size_t)p * MAX_NODELOG + k];
      if (tid == 0 && q.nnode)                              // buried sections changed by seep(): oldest snapshot must win
        for (uint32_t k = nl; k-- > 0;) {
          const SpecLog* e = spec_entry(sh, p, k);
          if (e->cell & NODE_ENTRY) s.pool[e->cell & ~NODE_ENTRY] = e->rec;
        }
      sync();                                              // everybody has read q.nalloc / w.rbase
      if (tid == 0) {
        const uint32_t cap = sh.reclaim_cap, hi = rb + na;
        if (hi > cap) ctrl[SC_RECLAIM_LOST] += hi - (rb > cap ? rb : cap);
        w.rbase = hi;
        q.nalloc = 0; q.nfree = 0; q.ndeps = 0; q.dep_overflow = 0; q.nested = 0; q.overflow = 0;
        ctrl[SC_KILLS]++;
        w.st[i] = ST_PENDING; w.nested[i] = 0;
      }
    } else {                                               // aborted by itself: data already restored, release stamps
      for (uint32_t k = tid; k < nl; k += nt) {
        const SpecLog* e = spec_entry(sh, p, k);
        if (e->cell & NODE_ENTRY) continue;
        sh.stamp[e->cell] = e->prev;
      }
      if (tid == 0) w.ab[i] = 0;                           // (released: a repeated pass must not do it again)
    }
    sync();
  }
  // ---- 2b. The lowest particle decides the phase's progress. If it gave up at a cell that tentative holders had
  // handed on to each other (h1 took it over from h2, h2 from h3, ...), undoing the one it met only exposes the
  // next: unwind the whole chain now instead of one holder per round (a lake cell can have dozens).
  if (tid == 0) {
    w.changed = 0;
    const uint32_t st0 = w.st[0], fc = sh.part[base].fail_cell;
    if ((st0 == ST_PENDING || st0 == ST_SUSPENDED) && !w.mask[0] && fc != NOCLAIM) {
      const uint32_t v = sh.stamp[fc];
      if (v != POISON && v > base + 1u && v - 1u - base < cnt) {
        const uint32_t h = v - 1u - base, sth = w.st[h];
        if ((sth == ST_DONE_NEW || sth == ST_DONE_OLD || sth == ST_SUSPENDED) && !w.kl[h]) { w.kl[h] = 1; w.changed = 1; }
      }
    }
  }
  sync();
  const bool more = w.changed != 0;
  sync();
  if (!more) break;
  }
  // ---- 3. promote, commit the finished prefix (team-parallel: counters by LDS sums, freed nodes by ranges)
  for (uint32_t i = tid; i < cnt; i += nt) if (w.st[i] == ST_DONE_NEW) w.st[i] = ST_DONE_OLD;
  sync();
  if (tid == 0) {
    uint32_t b = 0, top = ctrl[SC_TOP];                    // slots leaving the window; committed top-level particles
    while (b < cnt && top < n && !w.mask[b] && w.st[b] == ST_DONE_OLD) {
      const uint32_t k = w.nested[b];                      // its nested particles consumed slots b+1 .. b+k: void for good
      w.st[b] = ST_COMMITTED; top++;
      for (uint32_t j = 1; j <= k && b + j < cnt; j++) w.st[b + j] = ST_COMMITTED;   // (masked; rolled back above if they ran)
      b += 1u + k;
    }
    w.ncommit = b; w.top = top;
    for (int k = 0; k < SPEC_NCTR + 3; k++) w.acc[k] = 0;
  }
  sync();
  const uint32_t b = w.ncommit, bc = b < cnt ? b : cnt;    // (the last particle's void slots may reach past the window)
  for (uint32_t i = tid; i < bc; i += nt) w.list[i] = w.mask[i] ? (uint16_t)0 : (uint16_t)sh.part[base + i].nfree;   // <= MAX_NODELOG
  sync();
  {
    unsigned long long a[SPEC_NCTR + 2];
    for (int k = 0; k < SPEC_NCTR + 2; k++) a[k] = 0;
    bool any = false;
    const uint32_t rb = w.rbase;
    for (uint32_t i = tid; i < bc; i += nt) {
      if (w.mask[i]) continue;                             // a void slot: nothing to commit
      const SpecPart& q = sh.part[base + i];
      for (int k = 0; k < SPEC_NCTR; k++) a[k] += q.ctr[k];
      a[SPEC_NCTR] += (unsigned long long)q.live_delta; a[SPEC_NCTR + 1] += q.nested;
      any = true;
      uint32_t off = rb;                                   // deterministic: particle i's nodes follow those of 0..i-1
      for (uint32_t k = 0; k < i; k++) off += w.list[k];
      const uint32_t nf = w.list[i];
      for (uint32_t k = 0; k < nf; k++)
        if (off + k < sh.reclaim_cap) sh.reclaim[off + k] = sh.free_log[(size_t)(base + i) * MAX_NODELOG + k];
    }
    if (any) for (int k = 0; k < SPEC_NCTR + 2; k++) if (a[k]) SMX_A_ADD(&w.acc[k], a[k]);
  }
  sync();
  if (tid == 0) {
    for (int k = 0; k < SPEC_NCTR; k++) if (w.acc[k]) s.ctr[k] += w.acc[k];
    if (w.acc[SPEC_NCTR]) s.ctr[C_LIVE_SECTIONS] += w.acc[SPEC_NCTR];
    ctrl[SC_NESTED_COMMITTED] += (uint32_t)w.acc[SPEC_NCTR + 1];
    uint32_t rb = w.rbase, nfreed = 0;
    for (uint32_t i = 0; i < bc; i++) nfreed += w.list[i];
    const uint32_t cap = sh.reclaim_cap, hi = rb + nfreed;
    if (hi > cap) ctrl[SC_RECLAIM_LOST] += hi - (rb > cap ? rb : cap);
    ctrl[SC_NRECLAIM] = hi;
    // pending list: the top-level slots of the window that still have to run (ranks below n only)
    uint32_t np = 0;
    const uint32_t stall = b ? 0u : ctrl[SC_STALL] + 1u;   // consecutive rounds without a commit
    ctrl[SC_STALL] = stall;
    if (w.top < n && stall >= SPEC_SOLO_AFTER && bc < cnt && !w.mask[bc] && (w.st[bc] == ST_PENDING || w.st[bc] == ST_SUSPENDED)) {
      // The lowest particle keeps failing (typically a flooding particle whose nested particles run into cells that
      // higher particles re-take every round): it runs ALONE until it commits. Nobody else moves, so each failure
      // flags -- and the next boundary unwinds -- one more tentative holder in its way; the others keep their work.
      sh.pending[np++] = base + bc;
      ctrl[SC_SOLO]++;
    } else if (w.top < n) {
      for (uint32_t i0 = bc & ~3u; i0 < cnt; i0 += 4u) {
        const uint32_t st4 = spec_w4(w.st, i0), mk4 = spec_w4(w.mask, i0);
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
          const uint32_t i = i0 + u, st = (st4 >> (8u * u)) & 0xffu;
          if (i < bc || i >= cnt || ((mk4 >> (8u * u)) & 0xffu)) continue;
          if (st == ST_PENDING || st == ST_SUSPENDED) sh.pending[np++] = base + i;
        }
      }
      // slots beyond the window become eligible as it slides; the slot chain continues from the window's end
      const uint32_t newbase = base + b;
      const uint32_t newend = (nslots - newbase > SPEC_WINDOW) ? newbase + SPEC_WINDOW : nslots;
      uint32_t next = w.next_end > newbase ? w.next_end : newbase, rank = w.rank_end;
      for (uint32_t p = end > newbase ? end : newbase; p < newend && rank < n; p++) {
        if (p < next) continue;
        sh.pending[np++] = p; rank++;
      }
      if (np && sh.pending[np - 1] + 1u > ctrl[SC_HIGH]) ctrl[SC_HIGH] = sh.pending[np - 1] + 1u;   // (ascending list)
    }
    ctrl[SC_BASE] = base + b;
    ctrl[SC_TOP] = w.top;
    ctrl[SC_NPENDING] = np;
    ctrl[SC_ROUNDS]++;
    ctrl[SC_SUM_MAXSTEPS] += ctrl[SC_ROUND_MAXSTEPS]; ctrl[SC_ROUND_MAXSTEPS] = 0;
  }
  sync();
  for (uint32_t i = tid; i < cnt; i += nt) {
    SpecPart& q = sh.part[base + i];
    q.status = w.st[i]; q.kill = 0; q.aborted = 0;
  }
  sync();
}

// give up speculation: undo every tentative particle so that the state equals "particles [0, base) executed"
SMX_D void spec_abandon(const DevState& s, const SpecShared& sh) {
  uint32_t* ctrl = sh.ctrl;
  const uint32_t n = ctrl[SC_NSLOTS], base = ctrl[SC_BASE];
  for (uint32_t p = base; p < n; p++) if (sh.part[p].status == ST_DONE_OLD || sh.part[p].status == ST_DONE_NEW || sh.part[p].status == ST_SUSPENDED) sh.part[p].kill = 1;
  for (uint32_t p = n; p-- > base;) {
    SpecPart& q = sh.part[p];
    if (q.kill) {
      for (uint32_t k = q.nlog; k-- > 0;) {
        const SpecLog* e = spec_entry(sh, p, k);
        if (e->cell & NODE_ENTRY) { s.pool[e->cell & ~NODE_ENTRY] = e->rec; continue; }
        const size_t c = e->cell, fi = spec_fidx(s, c);
        s.cells[c] = e->rec; s.wtrack[fi] = e->wtrack; s.windfreq[fi] = e->windfreq; s.flags[c] = (uint8_t)e->flags;
      }
      for (uint32_t i = 0; i < q.nalloc; i++) {
        const uint32_t k = ctrl[SC_NRECLAIM]++;
        if (k < sh.reclaim_cap) sh.reclaim[k] = sh.alloc_log[(size_t)p * MAX_NODELOG + i]; else ctrl[SC_RECLAIM_LOST]++;
      }
      q.status = ST_PENDING; q.kill = 0;
    }
  }
}

// phase end: the free-node stack = what the bump allocator did not consume + the reclaimed nodes
SMX_D void spec_finish_pool(const DevState& s, const SpecShared& sh) {
  uint32_t* ctrl = sh.ctrl;
  const uint32_t nstart = ctrl[SC_NFREE_START];
  uint32_t used = ctrl[SC_NODE_CURSOR];
  if (used > nstart) used = nstart;
  uint32_t nf = nstart - used;
  uint32_t nr = ctrl[SC_NRECLAIM];
  if (nr > sh.reclaim_cap) nr = sh.reclaim_cap;
  for (uint32_t i = 0; i < nr; i++) s.freelist[nf++] = sh.reclaim[i];
  *s.free_count = nf;
}

}  // namespace smx

// ------------------------------------------------------------------------------------------------
// kernel bodies shared by the HIP kernels (soilmx.hip) and the host-side unit-test build (tests/hostsim)
// ------------------------------------------------------------------------------------------------
namespace smx {


// phase start (team of nt threads): thread 0 peeks the rand() stream ahead and resets the control block,
// everybody resets the per-particle records. The speculative engine needs pool headroom (SpecPolicy::can_get
// never refuses): without it the phase is flagged (SC_N = 0) and the host runs the serial walker instead.
SMX_D void spec_begin_team(const DevState& s, const SpecShared& sh, uint32_t n, uint32_t tid, uint32_t nt) {
  if (tid == 0) {
    // Peek the stream ahead without advancing it (that happens when the phase ends). The ring is rotated once so
    // that a[j] is the word draw j overwrites; a revolution of 31 draws then has static indices only (registers,
    // not scratch): a[j] += a[j-3 mod 31].
    const RandState& r = *s.rnd;
    const uint32_t i0 = r.idx;
    uint32_t a[31];
    for (int j = 0; j < 31; j++) a[j] = r.ring[(i0 + (uint32_t)j) % 31];
    for (uint32_t k = 0; k < sh.ndraws; k += 31) {
#pragma unroll
      for (int j = 0; j < 31; j++) {
        a[j] += a[(j + 28) % 31];
        if (k + (uint32_t)j < sh.ndraws) sh.draws[k + j] = a[j];
      }
    }
    for (int i = 0; i < SC_COUNT; i++) sh.ctrl[i] = 0;
    const unsigned long long live = s.ctr[C_LIVE_SECTIONS];
    // headroom: SpecPolicy::can_get() never refuses, so a phase only runs speculatively when the pool cannot run out
    // inside it (measured growth: ~80 sections per particle on rgps 64^2; 512 per particle + 64 Ki is the margin). Without
    // room the phase runs on the serial walker, whose get() fails exactly where the reference's does (layermap.h:92-95);
    // spec_end_team re-checks the live count afterwards (SC_POOL_CROSSED -> the host reports an error).
    const bool room = live + 65536ull + 512ull * n < s.pool_capacity && (unsigned long long)*s.free_count > 65536ull + 512ull * n;
    sh.ctrl[SC_N] = room ? n : 0; sh.ctrl[SC_NPENDING] = room ? ((n > SPEC_WINDOW) ? SPEC_WINDOW : n) : 0;
    sh.ctrl[SC_NSLOTS] = n + sh.maxnest;
    sh.ctrl[SC_HIGH] = (n > SPEC_WINDOW) ? SPEC_WINDOW : n;
    sh.ctrl[SC_NFREE_START] = *s.free_count;
  }
  for (uint32_t p = tid; p < n + sh.maxnest; p += nt) {  // one record per SLOT (n particles + the nested ones' slots)
    SpecPart& q = sh.part[p];
    q.status = ST_PENDING; q.kill = 0; q.aborted = 0; q.fail_cell = NOCLAIM; q.noff = 0; q.nested = 0; q.nlog = 0;
    q.nchunks = 0; q.ndeps = 0; q.dep_overflow = 0; q.nalloc = 0; q.nfree = 0; q.overflow = 0; q.attempts = 0; q.nnode = 0;
    q.live_delta = 0; q.blocked = 0;
    sh.pending[p] = p;
    sh.scout_len[p] = 0;
  }
}
SMX_D void spec_begin(const DevState& s, const SpecShared& sh, uint32_t n) { spec_begin_team(s, sh, n, 0u, 1u); }
// phase end (team): consume the draws the phase used, rebuild the free-node stack. The generator state after
// nd draws is just its last 31 raw outputs, which the peeked stream already holds -- no replay loop.
SMX_D void spec_end_team(const DevState& s, const SpecShared& sh, uint32_t nparticles_done, uint32_t nested_done,
                         uint32_t tid, uint32_t nt) {
  if (tid == 0) {
    RandState& r = *s.rnd;
    const uint64_t nd = 2ull * nparticles_done + 2ull * nested_done;
    const uint32_t idx0 = r.idx;
    for (uint64_t t = nd > 31 ? nd - 31 : 0; t < nd; t++) r.ring[(idx0 + (uint32_t)t) % 31] = sh.draws[t];
    r.idx = idx0 + (uint32_t)nd;
    r.calls += nd;
  }
  uint32_t* ctrl = sh.ctrl;
  const uint32_t nstart = ctrl[SC_NFREE_START];
  uint32_t used = ctrl[SC_NODE_CURSOR];
  if (used > nstart) used = nstart;
  const uint32_t nf = nstart - used;
  uint32_t nr = ctrl[SC_NRECLAIM];
  if (nr > sh.reclaim_cap) nr = sh.reclaim_cap;
  for (uint32_t i = tid; i < nr; i += nt) s.freelist[nf + i] = sh.reclaim[i];
  if (tid == 0) {
    *s.free_count = nf + nr;
    if (s.ctr[C_LIVE_SECTIONS] > s.pool_capacity) ctrl[SC_POOL_CROSSED] = 1;   // can_get() would have refused somewhere inside
  }
}
SMX_D void spec_end(const DevState& s, const SpecShared& sh, uint32_t nparticles_done, uint32_t nested_done) {
  spec_end_team(s, sh, nparticles_done, nested_done, 0u, 1u);
}

// A pending particle's predicted footprint = cells of its scout run + cells its previous attempt really touched +
// the cell that attempt failed at. spec_claim_one publishes it (claim[c] = min pending id); spec_blocked asks
// whether a LOWER pending particle wants any of those cells -- then this particle sits the round out.
// The walk is batched four cells deep: `f` gets four cells at a time (NOCLAIM = none) so that the index loads of
// a batch -- and whatever f loads per cell -- are in flight together (the lists were written by earlier kernels
// and come from HBM); f returning false ends the walk.
template <class F>
SMX_D void spec_for_footprint(const SpecShared& sh, uint32_t pid, uint32_t lane, uint32_t nl, F f) {
  const SpecPart& q = sh.part[pid];
  const uint32_t sl = sh.scout_len[pid];
  const uint32_t* sc = sh.scout_cells + (size_t)pid * sh.scout_cap;
  for (uint32_t k = lane; k < sl; k += 4u * nl) {
    const uint32_t k1 = k + nl, k2 = k + 2u * nl, k3 = k + 3u * nl;
    const uint32_t c[4] = {sc[k], k1 < sl ? sc[k1] : NOCLAIM, k2 < sl ? sc[k2] : NOCLAIM, k3 < sl ? sc[k3] : NOCLAIM};
    if (!f(c)) return;
  }
  if (q.attempts != 0) {
    const uint32_t nlog = q.nlog;
    for (uint32_t k = lane; k < nlog; k += 4u * nl) {
      uint32_t c[4];
#pragma unroll
      for (uint32_t u = 0; u < 4u; u++) {
        const uint32_t ku = k + u * nl;
        c[u] = ku < nlog ? spec_entry(sh, pid, ku)->cell : NOCLAIM;
        if (c[u] & NODE_ENTRY) c[u] = NOCLAIM;             // a pool-node record, not a cell
      }
      if (!f(c)) return;
    }
    if (lane == 0 && q.fail_cell != NOCLAIM) { const uint32_t c[4] = {q.fail_cell, NOCLAIM, NOCLAIM, NOCLAIM}; f(c); }
  }
}
// (lane, nl): the footprint is walked by a team of nl lanes, lane `lane` takes every nl-th cell.
// The minimum is an unconditional atomic whose result nobody waits for (most cells have a single claimant, so a
// load-and-compare in front of it only adds a round trip).
SMX_D void spec_claim_one(const SpecShared& sh, uint32_t pid, uint32_t lane, uint32_t nl) {
  spec_for_footprint(sh, pid, lane, nl, [&](const uint32_t (&c)[4]) {
#pragma unroll
    for (int u = 0; u < 4; u++) if (c[u] != NOCLAIM) (void)SMX_A_MIN(&sh.claim[c[u]], pid);
    return true;
  });
}
SMX_D bool spec_blocked(const SpecShared& sh, uint32_t pid, uint32_t lane, uint32_t nl) {
  bool blocked = false;
  spec_for_footprint(sh, pid, lane, nl, [&](const uint32_t (&c)[4]) {
    uint32_t v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = c[u] != NOCLAIM ? SMX_A_LOAD(&sh.claim[c[u]]) : NOCLAIM;
    if (v[0] < pid || v[1] < pid || v[2] < pid || v[3] < pid) { blocked = true; return false; }
    return true;
  });
  return blocked;
}

// round 0 only: the read-only trajectory of every particle on the phase's initial state
template <bool WIND>
SMX_D void spec_scout_one(const DevState& s, const SoilP* soils, const SpecShared& sh, uint32_t pid) {
  ScoutPolicy pol(sh, pid);
  Sim<ScoutPolicy> sim(s, soils, pol);
  const uint32_t k = 2u * pid;
  const int ry = (int)(sh.draws[k] >> 1) % s.dimy, rx = (int)(sh.draws[k + 1] >> 1) % s.dimx;
  if constexpr (WIND) {
    Wind p; sim.wind_init(p, rx, ry); sim.wind_run(p);
  } else {
    Frame st[1]; int depth = 0;               // the scout never pushes frames (it stops before the flood)
    Water p; sim.water_init(p, rx, ry); sim.water_drive(p, true, true, st, depth);
  }
  pol.finish();
}

template <bool WIND>
SMX_D void spec_exec_one(const DevState& s, const SoilP* soils, const SpecShared& sh, uint32_t pid) {
  if ((sh.mode & 1u) && sh.part[pid].blocked) { SMX_A_ADD(&sh.ctrl[SC_SKIPPED], 1u); return; }
  SMX_A_ADD(&sh.ctrl[SC_EXECUTED], 1u);
  const bool resume = sh.part[pid].status == ST_SUSPENDED;
  SpecPolicy pol(sh, pid, resume);
  Sim<SpecPolicy> sim(s, soils, pol);
  sim.susp_ok = (sh.mode & 2u) != 0;
  if constexpr (WIND) {
    Wind p;
    if (resume) p = sh.ckpt[pid].d;
    else {
      int d1, d2; pol.spawn_draws(d1, d2);                  // 1st draw -> y, 2nd -> x (wind.h:15 under g++)
      sim.wind_init(p, d2 % s.dimx, d1 % s.dimy);
    }
    if (!sim.ab) sim.wind_run(p, resume);
    SMX_A_MAX(&sh.ctrl[SC_ROUND_MAXSTEPS], (uint32_t)sim.n_steps_wind);
    if (sim.susp && !pol.overflow) { sim.flush_counters(); sh.ckpt[pid].d = p; pol.suspend(s); return; }
  } else {
    Frame st[MAX_FRAMES]; int depth = 0;
    Water p;
    if (resume) p = sh.ckpt[pid].w;
    else {
      int d1, d2; pol.spawn_draws(d1, d2);                  // water.h:13
      sim.water_init(p, d2 % s.dimx, d1 % s.dimy);
    }
    if (!sim.ab) sim.water_drive(p, true, true, st, depth, resume);
    SMX_A_MAX(&sh.ctrl[SC_ROUND_MAXSTEPS], (uint32_t)sim.n_steps_w_all);
    if (sim.susp && !pol.overflow) { sim.flush_counters(); sh.ckpt[pid].w = p; pol.suspend(s); return; }
  }
  if (sim.ab || pol.overflow) { pol.self_rollback(s); return; }
  sim.flush_counters();
  pol.finish_done(s);
}

}  // namespace smx
