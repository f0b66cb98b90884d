// soilmx.hip -- HIP kernels (gfx950) and the C-ABI of include/soilmx.h.
//
// Kernels (DESIGN.md "Kernels"):
//   k_water_serial / k_wind_serial   exact engine: ONE lane walks the particles in reference order
//                                    (SoilMachine.cpp:287-298,303-307) -- the parity anchor
//   k_grid_classify                  O(cells), HBM-streaming: which cells can the grid pass change at all?
//                                    (water.h:285-343 is a provable no-op unless a water top is within the
//                                    3x3 neighbourhood or the column holds a non-zero saturation)
//   k_grid_serial                    exact walk over the (dynamic) active set in the reference's x-major order
//   k_map_frequency / k_reset_frequency   water.h:353-365, float4-vectorised streaming kernels
//   k_init_terrain                   Layermap::initialize (layermap.h:163-216): FBm OpenSimplex2 per cell + column build
//   k_heights / k_surface / k_normals / k_bilinear   whole-map read-side primitives (layermap.h:341-439)
//   k_fill_vertices                  Layermap::update(Vertexpool&) (layermap.h:475-555): the 44-byte vertex stream
//
// Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see __graft_entry__.build()).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <dlfcn.h>
#include <string>
#include <vector>

#include "../../include/soilmx.h"
#ifdef SMX_PROF_FLOOD
__device__ unsigned long long g_sect[32];                    // (experiment build: soil_core.h SMX_SECT)
#endif
#include "soil_core.h"
#include "soil_serial.h"
#include "soil_spec.h"
#include <type_traits>
#include <chrono>
#include "soil_coop.h"
#include "soil_noise.h"
#include "soil_batch.h"
#include "soil_relax.h"
#include "soil_lbm.h"
#include <algorithm>
#include <rocprim/rocprim.hpp>   // device radix sort of the nested particles' keys (children -> next generation, batch_generations)

using namespace smx;

static_assert(sizeof(SoilP) == sizeof(smx_soil), "smx_soil layout");
static_assert(sizeof(Sec) == 32, "Sec must be 32 bytes");

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// soil table -> LDS (<= 14 KB); every lane helps, then lane 0 walks
#define SMX_LOAD_SOILS(sh)                                                                          \
  __shared__ SoilP sh[MAX_SOILS];                                                                   \
  for (uint32_t i_ = threadIdx.x; i_ < s.nsoils * (sizeof(SoilP) / 4); i_ += blockDim.x)            \
    reinterpret_cast<uint32_t*>(sh)[i_] = reinterpret_cast<const uint32_t*>(s.soils)[i_];           \
  __syncthreads();

__global__ void __launch_bounds__(64) k_water_serial(DevState s, int n) {
  SMX_LOAD_SOILS(sh)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  SerialPolicy pol(s);
  Sim<SerialPolicy> sim(s, sh, pol);
  Frame st[MAX_FRAMES];
  int depth = 0;
  for (int i = 0; i < n; i++) {                            // SoilMachine.cpp:288-298
    int ry, rx;
    pol.rand2(s, ry, rx);                                  // water.h:13, g++ order: 1st draw -> y, 2nd -> x
    Water p;
    sim.water_init(p, rx % s.dimx, ry % s.dimy);
    sim.water_drive(p, true, true, st, depth);
  }
  sim.flush_counters();
  pol.finish(s);
}

__global__ void __launch_bounds__(64) k_wind_serial(DevState s, int n) {
  SMX_LOAD_SOILS(sh)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  SerialPolicy pol(s);
  Sim<SerialPolicy> sim(s, sh, pol);
  for (int i = 0; i < n; i++) {                            // SoilMachine.cpp:304-307
    int ry, rx;
    pol.rand2(s, ry, rx);                                  // wind.h:15
    Wind p;
    sim.wind_init(p, rx % s.dimx, ry % s.dimy);
    sim.wind_run(p);
  }
  sim.flush_counters();
  pol.finish(s);
}

// One thread per cell, 256 consecutive cells (cell order) per block = 4 level-0 bitmap words.
__global__ void __launch_bounds__(256) k_grid_classify(DevState s) {
  const size_t n = (size_t)s.x_hi * s.dimy;
  const size_t c = (size_t)s.x_lo * s.dimy + (size_t)blockIdx.x * 256 + threadIdx.x;   // (x_lo*dimy is a multiple of 64: launch_classify)
  bool act = false;
  if (c < n) {
    const int x = (int)(c / s.dimy), y = (int)(c % s.dimy);
    uint8_t f = s.flags[c];
    act = (f & F_SAT) != 0;
    uint8_t any = f;
    const bool xm = x > s.x_lo, xp = x < s.x_hi - 1, ym = y > 0, yp = y < s.dimy - 1;
    if (ym) any |= s.flags[c - 1];
    if (yp) any |= s.flags[c + 1];
    if (xm) { any |= s.flags[c - s.dimy]; if (ym) any |= s.flags[c - s.dimy - 1]; if (yp) any |= s.flags[c - s.dimy + 1]; }
    if (xp) { any |= s.flags[c + s.dimy]; if (ym) any |= s.flags[c + s.dimy - 1]; if (yp) any |= s.flags[c + s.dimy + 1]; }
    act = act || (any & F_AIR);
  }
  const unsigned long long m = __ballot(act);
  if ((threadIdx.x & 63) == 0 && c < n) {
    const size_t w0 = c >> 6;
    s.active[w0] = m;
    if (m) {
      atomicOr(&s.active1[w0 >> 6], 1ull << (w0 & 63));
      atomicOr(&s.active2[w0 >> 12], 1ull << ((w0 >> 6) & 63));
    }
  }
}

// The same, eight cells per thread (dimy a multiple of 8: a thread's cells lie in one map row x): three aligned 8-byte
// loads of flag bytes (rows x-1, x, x+1) + the two flanking bytes of each row instead of nine byte loads per cell; a
// wave writes eight complete 64-bit words of the bitmap. 1 B/cell read, 1 bit/cell written: an HBM-streaming kernel.
__global__ void __launch_bounds__(256) k_grid_classify8(DevState s) {
  __shared__ unsigned long long sm[32];                      // 256 threads x 8 bits
  const size_t n = (size_t)s.x_hi * s.dimy, c_lo = (size_t)s.x_lo * s.dimy;   // (c_lo is a multiple of 64: launch_classify)
  const size_t c = c_lo + ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
  uint32_t act = 0;
  if (c < n) {
    const int x = (int)(c / s.dimy), y = (int)(c % s.dimy);
    const bool xm = x > s.x_lo, xp = x < s.x_hi - 1, ym = y > 0, yp = y + 8 < s.dimy;
    const unsigned long long zero = 0ull;
    const unsigned long long r0 = *reinterpret_cast<const unsigned long long*>(s.flags + c);
    const unsigned long long rm = xm ? *reinterpret_cast<const unsigned long long*>(s.flags + c - s.dimy) : zero;
    const unsigned long long rp = xp ? *reinterpret_cast<const unsigned long long*>(s.flags + c + s.dimy) : zero;
    uint32_t lo = 0, hi = 0;                                 // flag bytes left of cell 0 / right of cell 7, the three rows or-ed
    if (ym) { lo |= s.flags[c - 1]; if (xm) lo |= s.flags[c - s.dimy - 1]; if (xp) lo |= s.flags[c + s.dimy - 1]; }
    if (yp) { hi |= s.flags[c + 8]; if (xm) hi |= s.flags[c - s.dimy + 8]; if (xp) hi |= s.flags[c + s.dimy + 8]; }
    const unsigned long long col = r0 | rm | rp;             // per cell: own column of three rows
    // F_AIR anywhere in the 3x3: the cell's own column or-ed with its left and right neighbours'
    const unsigned long long airc = col & 0x0101010101010101ull * F_AIR;
    const unsigned long long left = (airc << 8) | (unsigned long long)(lo & F_AIR);
    const unsigned long long right = (airc >> 8) | ((unsigned long long)(hi & F_AIR) << 56);
    const unsigned long long any = airc | left | right | ((r0 & (0x0101010101010101ull * F_SAT)) >> 1);   // F_SAT = 2 -> bit 0
#pragma unroll
    for (int k = 0; k < 8; k++) act |= (uint32_t)((any >> (8 * k)) & 1ull) << k;
  }
  reinterpret_cast<uint8_t*>(sm)[threadIdx.x] = (uint8_t)act;
  __syncthreads();
  if (threadIdx.x < 32) {
    const size_t w0 = (c_lo >> 6) + (size_t)blockIdx.x * 32 + threadIdx.x;   // 2048 cells per block = 32 bitmap words
    if (w0 * 64 < n) {
      const unsigned long long m = sm[threadIdx.x];
      s.active[w0] = m;
      if (m) {
        atomicOr(&s.active1[w0 >> 6], 1ull << (w0 & 63));
        atomicOr(&s.active2[w0 >> 12], 1ull << ((w0 >> 6) & 63));
      }
    }
  }
}

__global__ void __launch_bounds__(64) k_grid_serial(DevState s) {
  SMX_LOAD_SOILS(sh)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  SerialPolicy pol(s);
  Sim<SerialPolicy> sim(s, sh, pol);
  sim.grid_mode = true;
  Frame st[MAX_FRAMES];
  int depth = 0;
  const size_t n = (size_t)s.dimx * s.dimy;
  size_t c = sim.next_active(0);
  unsigned long long visited = 0;
  Water dummy;
  dummy.pos = {0.f, 0.f}; dummy.speed = {0.f, 0.f}; dummy.volume = 0.0; dummy.sediment = 0.0; dummy.evaprate = 0.0;
  dummy.spill = 0; dummy.ix = dummy.iy = 0; dummy.friction = 0.f; dummy.surface = dummy.contains = 0;
  while (c < n) {
    const int x = (int)(c / s.dimy), y = (int)(c % s.dimy);
    sim.seep(x, y);                                        // water.h:339
    sim.push_frame(st, depth, x, y, 3);                    // water.h:340 WaterParticle::cascade(ivec2(x,y), .., 3)
    sim.water_drive(dummy, false, false, st, depth);
    visited++;
    c = sim.next_active(c + 1);
  }
  // every non-active cell still "calls" WaterParticle::cascade once in the reference (counter parity)
  sim.n_wcasc += n - visited;
  sim.flush_counters();
  pol.add_counter(s, C_GRID_ACTIVE, visited);
  pol.finish(s);
}

// ---------------- speculative engine kernels (protocol: soil_spec.h) ----------------
// Particles per wave (env SMX_SPEC_LANES, default 1): a phase has only ~10^3 particles while the chip has 1024
// SIMDs, and one particle step is issue-bound (a few thousand dependent instructions), so one particle per wave
// is fastest (measured at 1024^2: 1 lane 349 ms, 8 lanes 429 ms, 32 lanes 537 ms per water phase).
static int spec_lanes() {
  static int v = 0;
  if (!v) { const char* e = getenv("SMX_SPEC_LANES"); v = e ? atoi(e) : 1; if (v < 1) v = 1; if (v > 64) v = 64; }
  return v;
}

__global__ void __launch_bounds__(256) k_spec_begin(DevState s, SpecShared sh, uint32_t n) {
  spec_begin_team(s, sh, n, threadIdx.x, blockDim.x);
}
template <bool WIND>
__global__ void __launch_bounds__(64) k_spec_scout(DevState s, SpecShared sh, uint32_t lanes) {
  SMX_LOAD_SOILS(soils_sh)
  const uint32_t n = sh.ctrl[SC_N] ? sh.ctrl[SC_N] + SPEC_SCOUT_MARGIN : 0u;
  const uint32_t pid = blockIdx.x * lanes + threadIdx.x;
  if (threadIdx.x >= lanes || pid >= n) return;
  spec_scout_one<WIND>(s, soils_sh, sh, pid);
}
template <bool WIND>
__global__ void __launch_bounds__(64) k_spec_scout_coop(DevState s, SpecShared sh) {
  SMX_LOAD_SOILS(soils_sh)
  if (sh.ctrl[SC_N] == 0 || blockIdx.x >= sh.ctrl[SC_N] + SPEC_SCOUT_MARGIN) return;
  spec_scout_one_coop<WIND>(s, soils_sh, sh, blockIdx.x);
}
// one wave per pending particle walks its predicted footprint: publish claims, then (next kernel) test them
__global__ void __launch_bounds__(64) k_spec_claim(SpecShared sh) {
  if (blockIdx.x >= sh.ctrl[SC_NPENDING]) return;
  spec_claim_one(sh, sh.pending[blockIdx.x], threadIdx.x, 64u);
}
__global__ void __launch_bounds__(64) k_spec_select(SpecShared sh) {
  if (blockIdx.x >= sh.ctrl[SC_NPENDING]) return;
  const uint32_t pid = sh.pending[blockIdx.x];
  const bool b = spec_blocked(sh, pid, threadIdx.x, 64u);
  const bool any = __any(b);
  if (threadIdx.x == 0) sh.part[pid].blocked = any ? 1u : 0u;
}
template <bool WIND>
__global__ void __launch_bounds__(64) k_spec_exec(DevState s, SpecShared sh, uint32_t lanes) {
  SMX_LOAD_SOILS(soils_sh)
  const uint32_t i = blockIdx.x * lanes + threadIdx.x;
  if (threadIdx.x >= lanes || i >= sh.ctrl[SC_NPENDING]) return;
  spec_exec_one<WIND>(s, soils_sh, sh, sh.pending[i]);
}
// one particle per WAVE: lanes 0..8 share the block fetch / acquisition of every step (soil_coop.h)
template <bool WIND>
__global__ void __launch_bounds__(64) k_spec_exec_coop(DevState s, SpecShared sh) {
  SMX_LOAD_SOILS(soils_sh)
  if (blockIdx.x >= sh.ctrl[SC_NPENDING]) return;
  spec_exec_one_coop<WIND>(s, soils_sh, sh, sh.pending[blockIdx.x]);
}
__global__ void __launch_bounds__(1024) k_spec_boundary(DevState s, SpecShared sh) {   // ONE workgroup: the round barrier
  __shared__ SpecWork w;
  spec_boundary_team(s, sh, w, threadIdx.x, blockDim.x, []() { __syncthreads(); });
}
__global__ void __launch_bounds__(64) k_spec_abandon(DevState s, SpecShared sh) {
  if (threadIdx.x == 0) spec_abandon(s, sh);
}
__global__ void __launch_bounds__(256) k_spec_end(DevState s, SpecShared sh, uint32_t whole_phase) {
  const uint32_t done = whole_phase ? sh.ctrl[SC_N] : sh.ctrl[SC_TOP];   // top-level particles executed
  spec_end_team(s, sh, done, sh.ctrl[SC_NESTED_COMMITTED], threadIdx.x, blockDim.x);
}


// ---------------- batched engine kernels (protocol: soil_batch.h) ----------------
// One particle per LANE. 64-thread workgroups: a phase has 10^3..10^6 particles and the chip has 1024 SIMDs, so
// single-wave workgroups spread the waves evenly over CUs and XCDs (workgroups go round-robin over the XCDs).
template <bool WIND>
__global__ void __launch_bounds__(64) k_batch_spawn(DevState s, BatchShared bs, uint32_t nslots, const BChild* kids, const uint32_t* contains) {
  SMX_LOAD_SOILS(sh)
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  if (i >= nslots) return;
  BatchPolicy pol(s, bs);
  Sim<BatchPolicy> sim(s, sh, pol);
  if constexpr (WIND) batch_wind_spawn(sim, bs, i); else batch_water_spawn(sim, bs, i, kids ? kids + i : nullptr, kids ? contains : nullptr);
  sim.flush_counters(); pol.finish(s);
  // relaxed schedule: a record whose flood is what did not fit its parent's region waits on the flood list of epoch 0
  if constexpr (!WIND) { if (bs.relaxed && kids) relax_flood_later(bs, i, 0u, bs.water[i].state == B_ENDED); }
}
template <bool WIND>
__global__ void __launch_bounds__(64) k_batch_reclaim(DevState s, BatchShared bs, uint32_t nlive, uint32_t epoch) {
  SMX_LOAD_SOILS(sh)
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  if (i >= nlive) return;
  BatchPolicy pol(s, bs);
  Sim<BatchPolicy> sim(s, sh, pol);
  if constexpr (WIND) batch_wind_reclaim(sim, bs, bs.live[i], epoch); else batch_water_reclaim(sim, bs, bs.live[i], epoch);
}
template <bool WIND>
__global__ void __launch_bounds__(64) k_batch_epoch(DevState s, BatchShared bs, uint32_t nlive, uint32_t epoch, uint32_t lpw) {
  SMX_LOAD_SOILS(sh)
  // `lpw` particles per wavefront (batch_lanes_per_wave): a wavefront costs the UNION of its lanes' paths, and the chip has
  // 1024 SIMDs -- a phase that is down to a few thousand particles runs them one per wavefront instead of 64.
  const uint32_t i = blockIdx.x * lpw + threadIdx.x;
  if (threadIdx.x >= lpw || i >= nlive) return;
  BatchPolicy pol(s, bs);
  Sim<BatchPolicy> sim(s, sh, pol);
  if constexpr (WIND) batch_wind_epoch(sim, bs, bs.live[i], epoch);
  else { Frame st[MAX_FRAMES]; batch_water_epoch(sim, bs, bs.live[i], epoch, st); }
  sim.flush_counters(); pol.finish(s);
}
// the same kernel held to 256 registers (a few spilled) = two wavefronts per SIMD: for phases with more wavefronts than SIMDs
template <bool WIND>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_batch_epoch2(DevState s, BatchShared bs, uint32_t nlive, uint32_t epoch, uint32_t lpw) {
  SMX_LOAD_SOILS(sh)
  // `lpw` particles per wavefront (batch_lanes_per_wave): a wavefront costs the UNION of its lanes' paths, and the chip has
  // 1024 SIMDs -- a phase that is down to a few thousand particles runs them one per wavefront instead of 64.
  const uint32_t i = blockIdx.x * lpw + threadIdx.x;
  if (threadIdx.x >= lpw || i >= nlive) return;
  BatchPolicy pol(s, bs);
  Sim<BatchPolicy> sim(s, sh, pol);
  if constexpr (WIND) batch_wind_epoch(sim, bs, bs.live[i], epoch);
  else { Frame st[MAX_FRAMES]; batch_water_epoch(sim, bs, bs.live[i], epoch, st); }
  sim.flush_counters(); pol.finish(s);
}
// ---------------- relaxed schedule (soil_relax.h): step -> apply -> nine cascade colours (-> floods), per epoch ----------------
template <bool WIND>
__global__ void __launch_bounds__(64) k_relax_step(DevState s, BatchShared bs, uint32_t nlive, uint32_t epoch, uint32_t lpw) {
  SMX_LOAD_SOILS(sh)
  const uint32_t i = blockIdx.x * lpw + threadIdx.x;
  if (threadIdx.x >= lpw || i >= nlive) return;
  BatchPolicy pol(s, bs);
  Sim<BatchPolicy> sim(s, sh, pol);
  if constexpr (WIND) relax_wind_step(sim, bs, bs.live[i], epoch); else relax_water_step(sim, bs, bs.live[i], epoch);
}
__global__ void __launch_bounds__(64) k_relax_apply(DevState s, BatchShared bs, uint32_t nlive, uint32_t epoch) {
  SMX_LOAD_SOILS(sh)
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  if (i == 0u) {                                             // the lists of the NEXT epoch start empty (their last readers finished an epoch ago)
    const uint32_t par = (epoch + 1u) & 1u;
    bs.ctrl[BC_NFLOOD + par] = 0u; bs.ctrl[BC_NFLAG + par] = 0u; bs.ctrl[BC_NRUN + par] = 0u;
    for (uint32_t k = 0; k < 9u; k++) bs.ctrl[BC_NCASC + 9u * par + k] = 0u;
    bs.ctrl[BC_CASC_CURSOR] = 0u;                            // (this epoch's cascade dataflow starts at its first listed cell)
  }
  if (i >= nlive) return;
  BatchPolicy pol(s, bs);
  Sim<BatchPolicy> sim(s, sh, pol);
  relax_apply_slot(sim, bs, bs.live[i]);
  sim.flush_counters(); pol.finish(s);
}
// wave-reduced commit of the cascade counters of the relaxed kernels (one atomic per wavefront)
__device__ __forceinline__ void relax_commit_cascade_counters(const DevState& s, unsigned long long nc, unsigned long long nt) {
  for (int off = 32; off > 0; off >>= 1) { nc += (unsigned long long)__shfl_down((long long)nc, off); nt += (unsigned long long)__shfl_down((long long)nt, off); }
  if ((threadIdx.x & 63u) == 0) { if (nc) atomicAdd(&s.ctr[C_CASCADE_CALLS], nc); if (nt) atomicAdd(&s.ctr[C_CASCADE_TRANSFERS], nt); }
}
// steps of the nested particles that ran inside this launch's floods (C_FLOOD_NESTED_STEPS: what bench.py prices a flood launch with)
__device__ __forceinline__ void relax_commit_flood_steps(const DevState& s, unsigned long long n) {
  for (int off = 32; off > 0; off >>= 1) n += (unsigned long long)__shfl_down((long long)n, off);
  if ((threadIdx.x & 63u) == 0 && n) atomicAdd(&s.ctr[C_FLOOD_NESTED_STEPS], n);
}
__global__ void __launch_bounds__(64) k_relax_filter(DevState s, BatchShared bs, uint32_t epoch) {
  const uint32_t par = epoch & 1u;
  uint32_t n = bs.ctrl[BC_NFLAG + par];
  if (n > bs.flag_cap) n = bs.flag_cap;
  if (blockIdx.x * 64u >= n) return;                         // (the launch is sized for the worst case: most blocks leave here)
  SMX_LOAD_SOILS(sh)
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  BatchPolicy pol(s, bs);
  Sim<BatchPolicy> sim(s, sh, pol);
  relax_filter(sim, bs, i < n ? bs.flagl[par][i] : 0u, i < n, epoch);   // (isolated cells cascade in here)
  sim.flush_counters(); pol.finish(s);
  relax_commit_cascade_counters(s, pol.a_casc, pol.a_tr);
}
// (a fixed, small grid with a grid-stride loop: the colour lists hold the residue of crowded cells -- a few hundred entries, often none --
//  and a launch sized for the worst case spends its ~13 us dispatching a thousand workgroups that read the count and leave; round 5)
__global__ void __launch_bounds__(64) k_relax_cascade(DevState s, BatchShared bs, uint32_t epoch, uint32_t colour) {
  const uint32_t par = epoch & 1u;
  uint32_t n = bs.ctrl[BC_NCASC + 9u * par + colour];
  if (n > bs.clist_cap) n = bs.clist_cap;
  if (blockIdx.x * 64u >= n) return;
  SMX_LOAD_SOILS(sh)
  unsigned long long nc = 0, nt = 0;
  for (uint32_t i = blockIdx.x * 64u + threadIdx.x; i < n; i += gridDim.x * 64u) {
    BatchPolicy pol(s, bs);
    Sim<BatchPolicy> sim(s, sh, pol);
    relax_cascade(sim, bs.clist[par][(size_t)colour * bs.clist_cap + i]);
    sim.flush_counters(); pol.finish(s);
    nc += pol.a_casc; nt += pol.a_tr;
  }
  relax_commit_cascade_counters(s, nc, nt);
}
// The nine colour lists of an epoch as ONE launch (round 5): a DATAFLOW over the listed cells in colour order -- a wavefront takes the
// next cell, waits until every listed cell of an earlier colour within two cells of it (3x3 blocks that overlap its own) is done, runs
// Particle::cascade there (lane 0) and publishes it (bs.cstate). Same order between conflicting cells as nine launches with a barrier
// each, so the same result -- without eight kernel boundaries of ~8 us in an epoch that lasts ~230 (2 300 epochs per tick: 19.8 % of it).
// Spin budget of the two dataflow kernels of the default schedule: a wavefront that polls a dependency longer than `budget` polls -- tens of seconds; the
// dynamic cursor guarantees that every dependency belongs to a wavefront that is already running, so only a fault gets there -- raises bs.ctrl[BC_STALLED], every other
// waiter sees the flag within 1 024 polls and stops waiting too, the kernel drains, and the host fails the call (-9) instead of hanging the stream.
// SMX_SPIN_BUDGET overrides the number of polls (tests/test_gpu_relaxed.py sets 1 to see the error).
static uint32_t spin_budget(uint32_t dflt) { static const long v = [] { const char* e = getenv("SMX_SPIN_BUDGET"); return e ? atol(e) : -1L; }(); return v >= 0 ? (uint32_t)v : dflt; }
__device__ __forceinline__ bool spin_wait_while(const uint32_t* f, uint32_t busy, uint32_t* stalled, uint32_t budget, uint32_t sleep_a, uint32_t naps) {
  uint32_t spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == busy) {
    if (sleep_a == 2u) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(8);
    for (uint32_t r = 0; r < naps; r++) __builtin_amdgcn_s_sleep(127);
    if (++spins > budget) { __hip_atomic_store(stalled, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    if ((spins & 1023u) == 0u && __hip_atomic_load(stalled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
  }
  return true;
}
__global__ void __launch_bounds__(64) k_relax_cascade_flow(DevState s, BatchShared bs, uint32_t epoch, uint32_t budget) {
  const uint32_t par = epoch & 1u;
  uint32_t cnt[9], total = 0;
#pragma unroll
  for (uint32_t q = 0; q < 9u; q++) { uint32_t n = bs.ctrl[BC_NCASC + 9u * par + q]; if (n > bs.clist_cap) n = bs.clist_cap; cnt[q] = n; total += n; }
  if (total == 0u) return;
  SMX_LOAD_SOILS(sh)
  __shared__ uint32_t s_i;
  unsigned long long nc = 0, nt = 0;
  const uint32_t pending = 2u * bs.rtag;
  bool live = true;
  while (true) {
    if (threadIdx.x == 0) s_i = atomicAdd(&bs.ctrl[BC_CASC_CURSOR], 1u);
    __syncthreads();
    uint32_t i = s_i;
    __syncthreads();
    if (i >= total) break;
    if (threadIdx.x == 0) {
      uint32_t q = 0;
#pragma unroll
      for (uint32_t k = 0; k < 9u; k++) { if (q == k && i >= cnt[k]) { i -= cnt[k]; q = k + 1u; } }
      const uint32_t c = bs.clist[par][(size_t)q * bs.clist_cap + i];
      const int x = (int)(c / (uint32_t)s.dimy), y = (int)(c - (uint32_t)x * (uint32_t)s.dimy);
      const int xa = x - 2 < s.x_lo ? s.x_lo : x - 2, xb = x + 2 >= s.x_hi ? s.x_hi - 1 : x + 2;
      const int ya = y - 2 < 0 ? 0 : y - 2, yb = y + 2 >= s.dimy ? s.dimy - 1 : y + 2;
      for (int xx = xa; xx <= xb; xx++)
        for (int yy = ya; yy <= yb; yy++) {
          if ((uint32_t)((xx % 3) * 3 + (yy % 3)) >= q) continue;            // (only cells of an earlier colour; the cell itself has colour q)
          uint32_t* f = &bs.cstate[(size_t)xx * (size_t)s.dimy + (size_t)yy];
          if (live && !spin_wait_while(f, pending, &bs.ctrl[BC_STALLED], budget, 2u, 0u)) live = false;   // (stalled: the call fails; stop waiting, let the kernel drain)
        }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      BatchPolicy pol(s, bs);
      Sim<BatchPolicy> sim(s, sh, pol);
      relax_cascade(sim, c);
      sim.flush_counters(); pol.finish(s);
      nc += pol.a_casc; nt += pol.a_tr;
      __hip_atomic_store(&bs.cstate[c], pending + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (threadIdx.x == 0) { if (nc) atomicAdd(&s.ctr[C_CASCADE_CALLS], nc); if (nt) atomicAdd(&s.ctr[C_CASCADE_TRANSFERS], nt); }
}
// the floods of the relaxed schedule: batch_water_epoch over this epoch's list of ended particles; who is not done joins the next
// list. The launch has a fixed number of wavefronts; the floods are spread over them as thinly as the list allows (a wavefront costs
// the union of its lanes' paths, and a flood is a long, branchy path).
// (Round 6, measured and dropped -- profiles/r06_floods.md: the 63 idle lanes fetching a flood's whole rectangle -- cells, flag bytes,
//  frequency / track words, one load per 128-byte line -- before lane 0 starts, and the same for a grid tile's 24 x 24-cell region: no
//  change within 0.5 % although the kernel spends 80 % of its wave cycles waiting (rocprofv3 --pmc): the chain's ~100 dependent accesses
//  are L2 / Infinity-Cache hits already, 0.3-0.5 us each; what a flood costs is their NUMBER. 256 ... 16 384 wavefronts for the list: no
//  change either -- a launch lasts as long as its longest flood.)
// (experiments, SMX_FLOOD_PROF=1: how long does ONE flood take, and how many steps of nested particles run inside it? [0..31] floods by log2 of
//  their duration in 100 MHz ticks, [32..63] the nested steps of those floods, [64] floods that acted, [65] their ticks, [66] listed floods that
//  did not hold their tiles; printed by smx_destroy)
__device__ unsigned long long g_flood_prof[72];
__global__ void __launch_bounds__(64) k_relax_floods(DevState s, BatchShared bs, uint32_t epoch, uint32_t prof) {
  const uint32_t par = epoch & 1u;
  uint32_t n = bs.ctrl[BC_NFLOOD + par];
  if (n > bs.list_cap) n = bs.list_cap;
  uint32_t lpw = (n + gridDim.x - 1u) / gridDim.x;
  if (lpw > 64u) lpw = 64u;
  if (blockIdx.x * lpw >= n) return;
  SMX_LOAD_SOILS(sh)
  // (n > 64 * gridDim.x cannot happen: the host sizes the launch for the generation's slots)
  const uint32_t i = blockIdx.x * lpw + threadIdx.x;
  const bool act = threadIdx.x < lpw && i < n;
  bool again = false; uint32_t slot = 0u;
  if (!relax_floods_due(bs, epoch, n, bs.ctrl[BC_NRUN + par])) {  // not this epoch: the whole list moves on
    if (act) slot = bs.flist[par][i];
    relax_flood_later(bs, slot, par ^ 1u, act);
    return;
  }
  unsigned long long nst = 0;
  if (act) slot = bs.flist[par][i];
  if (act) {
    BatchPolicy pol(s, bs);
    Sim<BatchPolicy> sim(s, sh, pol);
    Frame st[MAX_FRAMES];
    const unsigned long long t0 = prof ? wall_clock64() : 0ull;
    batch_water_epoch(sim, bs, slot, epoch, st);
    sim.flush_counters(); nst = pol.a_nsteps; pol.finish(s);
    again = bs.water[slot].state == B_ENDED;
    if (prof) {
      const unsigned long long dt = wall_clock64() - t0;
      if (again) atomicAdd(&g_flood_prof[66], 1ull);
      else { const int b = 63 - __clzll((long long)(dt | 1ull)); atomicAdd(&g_flood_prof[b > 31 ? 31 : b], 1ull); atomicAdd(&g_flood_prof[32 + (b > 31 ? 31 : b)], nst); atomicAdd(&g_flood_prof[64], 1ull); atomicAdd(&g_flood_prof[65], dt); }
    }
  }
  relax_flood_later(bs, slot, par ^ 1u, again);
  relax_commit_flood_steps(s, nst);
}
// The tail of a relaxed water generation (a few hundred particles left, for hundreds of epochs; and every nested generation after
// its first steps): ONE workgroup runs whole epochs back to back -- step, apply, filter, nine cascade colours, floods -- with
// workgroup barriers where the dense schedule has kernel boundaries. Same phases, same order, same results; no launch per phase.
constexpr uint32_t RELAX_TAIL = 256;                          // running particles up to which the tail kernel takes over (one per thread of its
                                                              // workgroup; round 3: 1 024 = four per thread -- 2.6 % slower on the headline, r04 sweep)
__global__ void __launch_bounds__(256) k_relax_tail(DevState s, BatchShared bs, uint32_t nlive, uint32_t epoch0, uint32_t nepochs, uint32_t tag0) {
  SMX_LOAD_SOILS(sh)
  __shared__ uint32_t busy;
  for (uint32_t k = 0; k < nepochs; k++) {
    const uint32_t epoch = epoch0 + k, par = epoch & 1u;
    bs.rtag = tag0 + k;
    if (threadIdx.x == 0) busy = 0u;
    __syncthreads();
    uint32_t mine = 0u;
    for (uint32_t i = threadIdx.x; i < nlive; i += 256u) {                                       // step
      BatchPolicy pol(s, bs);
      Sim<BatchPolicy> sim(s, sh, pol);
      const uint32_t slot = bs.live[i];
      relax_water_step(sim, bs, slot, epoch);
      mine |= (bs.water[slot].state != B_DONE || bs.rcount[slot] != 0u) ? 1u : 0u;   // (a particle's last step still has edits to apply)
    }
    if (mine) busy = 1u;
    if (threadIdx.x == 0) {                                                                      // the lists of the next epoch start empty
      bs.ctrl[BC_NFLOOD + (par ^ 1u)] = 0u; bs.ctrl[BC_NFLAG + (par ^ 1u)] = 0u; bs.ctrl[BC_NRUN + (par ^ 1u)] = 0u;
      for (uint32_t q = 0; q < 9u; q++) bs.ctrl[BC_NCASC + 9u * (par ^ 1u) + q] = 0u;
    }
    __syncthreads();
    if (!busy) break;                                                                            // every particle of the generation is done
    for (uint32_t i = threadIdx.x; i < nlive; i += 256u) {                                       // apply
      BatchPolicy pol(s, bs);
      Sim<BatchPolicy> sim(s, sh, pol);
      relax_apply_slot(sim, bs, bs.live[i]);
      sim.flush_counters(); pol.finish(s);
    }
    __syncthreads();
    unsigned long long nc = 0, nt = 0;
    {                                                                                            // filter
      uint32_t n = SMX_A_LOAD(&bs.ctrl[BC_NFLAG + par]);
      if (n > bs.flag_cap) n = bs.flag_cap;
      for (uint32_t i0 = 0; i0 < n; i0 += 256u) {
        const uint32_t i = i0 + threadIdx.x;
        BatchPolicy pol(s, bs);
        Sim<BatchPolicy> sim(s, sh, pol);
        relax_filter(sim, bs, i < n ? bs.flagl[par][i] : 0u, i < n, epoch);
        sim.flush_counters(); pol.finish(s);
        nc += pol.a_casc; nt += pol.a_tr;
      }
    }
    __syncthreads();
    for (uint32_t q = 0; q < 9u; q++) {                                                          // the nine cascade colours
      uint32_t n = SMX_A_LOAD(&bs.ctrl[BC_NCASC + 9u * par + q]);
      if (n > bs.clist_cap) n = bs.clist_cap;
      if (n == 0u) continue;                                                                     // (uniform: every thread reads the same count)
      for (uint32_t i = threadIdx.x; i < n; i += 256u) {
        BatchPolicy pol(s, bs);
        Sim<BatchPolicy> sim(s, sh, pol);
        relax_cascade(sim, bs.clist[par][(size_t)q * bs.clist_cap + i]);
        sim.flush_counters(); pol.finish(s);
        nc += pol.a_casc; nt += pol.a_tr;
      }
      __syncthreads();
    }
    relax_commit_cascade_counters(s, nc, nt);
    {                                                                                            // floods
      uint32_t n = SMX_A_LOAD(&bs.ctrl[BC_NFLOOD + par]);
      if (n > bs.list_cap) n = bs.list_cap;
      const bool due = relax_floods_due(bs, epoch, n, SMX_A_LOAD(&bs.ctrl[BC_NRUN + par]));
      unsigned long long nst = 0;
      for (uint32_t i0 = 0; i0 < n; i0 += 256u) {
        const uint32_t i = i0 + threadIdx.x;
        bool again = false; uint32_t slot = 0u;
        if (i < n && !due) { slot = bs.flist[par][i]; again = true; }                            // not this epoch: the list moves on
        else if (i < n) {
          slot = bs.flist[par][i];
          BatchPolicy pol(s, bs);
          Sim<BatchPolicy> sim(s, sh, pol);
          Frame st[MAX_FRAMES];
          batch_water_epoch(sim, bs, slot, epoch, st);
          sim.flush_counters(); nst += pol.a_nsteps; pol.finish(s);
          again = bs.water[slot].state == B_ENDED;
        }
        relax_flood_later(bs, slot, par ^ 1u, again);
      }
      relax_commit_flood_steps(s, nst);
    }
    __syncthreads();
  }
}
// The DENSE epochs of a relaxed water generation as ONE persistent launch (round 6; OFF by default -- smx_set_relax_launch(1, ..) /
// SMX_RELAX_MEGA=1 turn it on): the same five phases in the same order -- step, apply, filter (isolated cells cascade there), the colour
// lists as a dataflow, floods -- run by a grid of co-resident workgroups (hipLaunchCooperativeKernel) with a device-wide barrier where the
// per-phase launches have kernel boundaries; phases whose list is empty are skipped together with their barrier (the count is read by
// everybody after a barrier, so the decision is uniform). Which lane runs a slot / a cell has no influence on any result (soil_relax.h): the
// state after the launch is the one the per-phase launches leave, bit for bit (tests/test_gpu_relaxed.py runs every launch shape against the
// same host-thread states and against the restatement).
// MEASURED, and why it is not the default (profiles/r06_persistent_epochs.md): the idea was that 5 launches per epoch x 4 352 dense epochs per
// tick are launch-bound. They are not. The barrier itself is cheap -- 6-7 us for 256 workgroups with thread 0 fencing, 2.4 us for 64
// (tools/barrier_bench.hip; no fence: stale reads, counted) -- but an epoch inside this kernel takes 183 us against 143 us as five launches
// (headline workload, k = 8, ticks 4-7: water phase 904 ms against 623; 64 workgroups: 714): workgroup 0's clock puts the step phase at
// 23 + 15 us (its own work + waiting for the slowest workgroup) where k_relax_step takes 9.7 us as a launch of its own, apply at 9 + 18
// (13), and the flood phase at 20 + 48. One kernel holds all five phases' code (240 KB against a 64 KB instruction cache that every
// wavefront walks through once per epoch), is register-allocated for its worst phase (256 VGPR + 96 AGPR, 1.1 KB of scratch per lane: one
// wavefront per SIMD, where the step / apply / filter kernels run at 64-144 registers), and every barrier's acquire invalidates the L2 the
// next phase's first loads -- instructions included -- then miss. Kernel boundaries do the same flushes in hardware, overlapped with the
// dispatch of a kernel compiled for that phase alone. What bounds an epoch is the slowest flood and the slowest cascade, not the launches.
struct GridBar { uint32_t count, gen, busy[2], abort, pad[3]; unsigned long long prof[16]; };   // prof: (flags & 2) ticks of the 100 MHz clock per phase, workgroup 0's view
constexpr uint32_t GB_SPIN_LIMIT = 1u << 22;                  // ~ seconds: a barrier nobody releases sets `abort`, everybody leaves, the host reports it
// The fences are thread 0's alone: a workgroup's stores are acknowledged by its XCD's L2 when __syncthreads() lets it through (the per-CU
// cache writes through), thread 0's release then writes that L2 back, and its acquire after the wait invalidates the CU's cache and the
// L2's stale lines for everybody behind the second __syncthreads(). (Every thread fencing -- 1 024 wavefronts x write-back + invalidate per
// barrier -- cost 65 us per barrier: the first build of this kernel ran the water phase three times slower than five launches per epoch.)
__device__ __forceinline__ bool grid_sync(GridBar* b, uint32_t nblocks, uint32_t& gen, bool every_thread_fences) {
  if (every_thread_fences) __threadfence();
  __syncthreads();
  gen++;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (__hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1u) {
      __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&b->gen, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      uint32_t spins = 0;
      while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > GB_SPIN_LIMIT) { __hip_atomic_store(&b->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        if ((spins & 0x3FFu) == 0u && __hip_atomic_load(&b->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (every_thread_fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return __hip_atomic_load(&b->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;   // (the host reads it through bs.ctrl[BC_STALLED], k_relax_epochs)
}
__global__ void __launch_bounds__(256) k_relax_epochs(DevState s, BatchShared bs, uint32_t nlive, uint32_t epoch0, uint32_t nepochs, uint32_t tag0, GridBar* bar, uint32_t dense_waves, uint32_t flags) {
  SMX_LOAD_SOILS(sh)
  // entry i of a list -> lane (i / W) % 64 of wavefront i % W, W wavefronts dealt round robin over the workgroups (wavefront w of the deal =
  // wave w / nblocks of workgroup w % nblocks: the first `nblocks` wavefronts sit on different CUs). Steps, applies and the filter use
  // `dense_waves` wavefronts (one per CU by default, as batch_lanes_per_wave: their paths are uniform, and wavefronts of one CU compete for the
  // instruction cache); floods -- long, branchy, one lane each -- are dealt over ALL wavefronts.
  const uint32_t NW = gridDim.x * 4u, lane = threadIdx.x & 63u, gw = (threadIdx.x >> 6) * gridDim.x + blockIdx.x;
  const bool fence_all = (flags & 1u) != 0u, prof = (flags & 2u) != 0u && blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long t_prev = prof ? wall_clock64() : 0ull;
#define SMX_PROF(slot) if (prof) { const unsigned long long t_now = wall_clock64(); bar->prof[slot] += t_now - t_prev; t_prev = t_now; }
  uint32_t DW = dense_waves < 1u ? 1u : (dense_waves > NW ? NW : dense_waves);
  { const uint32_t need = (nlive + 63u) / 64u; if (need > DW) DW = need > NW ? NW : need; }
  const uint32_t first = lane * DW + gw, stride = 64u * DW;    // (gw >= DW: this wavefront sits the dense phases out)
  uint32_t gen = __hip_atomic_load(&bar->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (uint32_t k = 0; k < nepochs; k++) {
    const uint32_t epoch = epoch0 + k, par = epoch & 1u;
    bs.rtag = tag0 + k;
    unsigned long long nc = 0, nt = 0;
    {                                                                                            // ---- step
      uint32_t mine = 0u;
      for (uint32_t i = first; gw < DW && i < nlive; i += stride) {
        BatchPolicy pol(s, bs);
        Sim<BatchPolicy> sim(s, sh, pol);
        const uint32_t slot = bs.live[i];
        relax_water_step(sim, bs, slot, epoch);
        mine |= (bs.water[slot].state != B_DONE || bs.rcount[slot] != 0u) ? 1u : 0u;            // (a particle's last step still has edits to apply)
      }
      if (mine) __hip_atomic_store(&bar->busy[k & 1u], bs.rtag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (tags are unique and never 0: nothing to reset)
      if (blockIdx.x == 0 && threadIdx.x == 0) {                                                 // the lists of the next epoch start empty
        bs.ctrl[BC_NFLOOD + (par ^ 1u)] = 0u; bs.ctrl[BC_NFLAG + (par ^ 1u)] = 0u; bs.ctrl[BC_NRUN + (par ^ 1u)] = 0u;
        for (uint32_t q = 0; q < 9u; q++) bs.ctrl[BC_NCASC + 9u * (par ^ 1u) + q] = 0u;
        bs.ctrl[BC_CASC_CURSOR] = 0u;
      }
    }
    SMX_PROF(0)
    if (!grid_sync(bar, gridDim.x, gen, fence_all)) { bs.ctrl[BC_STALLED] = 1u; return; }
    SMX_PROF(1)
    if (__hip_atomic_load(&bar->busy[k & 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != bs.rtag) break;   // every particle of the generation is done
    for (uint32_t i = first; gw < DW && i < nlive; i += stride) {                                // ---- apply
      BatchPolicy pol(s, bs);
      Sim<BatchPolicy> sim(s, sh, pol);
      relax_apply_slot(sim, bs, bs.live[i]);
      sim.flush_counters(); pol.finish(s);
    }
    SMX_PROF(2)
    if (!grid_sync(bar, gridDim.x, gen, fence_all)) { bs.ctrl[BC_STALLED] = 1u; return; }
    SMX_PROF(3)
    {                                                                                            // ---- filter (isolated cells cascade in here)
      uint32_t n = __hip_atomic_load(&bs.ctrl[BC_NFLAG + par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (n > bs.flag_cap) n = bs.flag_cap;
      if (n) {
        uint32_t FW = DW; { const uint32_t need = (n + 63u) / 64u; if (need > FW) FW = need > NW ? NW : need; }
        for (uint32_t i0 = gw; gw < FW && i0 < n; i0 += 64u * FW) {                              // (wavefront-uniform trip count: relax_filter appends with ballots)
          const uint32_t i = i0 + lane * FW;
          BatchPolicy pol(s, bs);
          Sim<BatchPolicy> sim(s, sh, pol);
          relax_filter(sim, bs, i < n ? bs.flagl[par][i] : 0u, i < n, epoch);
          sim.flush_counters(); pol.finish(s);
          nc += pol.a_casc; nt += pol.a_tr;
        }
        SMX_PROF(4)
        if (!grid_sync(bar, gridDim.x, gen, fence_all)) { bs.ctrl[BC_STALLED] = 1u; return; }
        SMX_PROF(5)
      }
    }
    {                                                                                            // ---- the colour lists: the dataflow of k_relax_cascade_flow, a cell per wavefront
      uint32_t cnt[9], total = 0;
#pragma unroll
      for (uint32_t q = 0; q < 9u; q++) {
        uint32_t n = __hip_atomic_load(&bs.ctrl[BC_NCASC + 9u * par + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n > bs.clist_cap) n = bs.clist_cap;
        cnt[q] = n; total += n;
      }
      if (total) {
        const uint32_t pending = 2u * bs.rtag;
        while (true) {
          uint32_t i = 0u;
          if (lane == 0u) i = atomicAdd(&bs.ctrl[BC_CASC_CURSOR], 1u);
          i = (uint32_t)__shfl((int)i, 0);
          if (i >= total) break;
          if (lane == 0u) {
            uint32_t q = 0;
#pragma unroll
            for (uint32_t c9 = 0; c9 < 9u; c9++) { if (q == c9 && i >= cnt[c9]) { i -= cnt[c9]; q = c9 + 1u; } }
            const uint32_t c = bs.clist[par][(size_t)q * bs.clist_cap + i];
            const int x = (int)(c / (uint32_t)s.dimy), y = (int)(c - (uint32_t)x * (uint32_t)s.dimy);
            const int xa = x - 2 < s.x_lo ? s.x_lo : x - 2, xb = x + 2 >= s.x_hi ? s.x_hi - 1 : x + 2;
            const int ya = y - 2 < 0 ? 0 : y - 2, yb = y + 2 >= s.dimy ? s.dimy - 1 : y + 2;
            uint32_t spins = 0;
            for (int xx = xa; xx <= xb; xx++)
              for (int yy = ya; yy <= yb; yy++) {
                if ((uint32_t)((xx % 3) * 3 + (yy % 3)) >= q) continue;                          // (only cells of an earlier colour; the cell itself has colour q)
                uint32_t* f = &bs.cstate[(size_t)xx * (size_t)s.dimy + (size_t)yy];
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pending) {
                  __builtin_amdgcn_s_sleep(2);
                  if (++spins > GB_SPIN_LIMIT) { __hip_atomic_store(&bar->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
              }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            BatchPolicy pol(s, bs);
            Sim<BatchPolicy> sim(s, sh, pol);
            relax_cascade(sim, c);
            sim.flush_counters(); pol.finish(s);
            nc += pol.a_casc; nt += pol.a_tr;
            __hip_atomic_store(&bs.cstate[c], pending + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        SMX_PROF(6)
        if (!grid_sync(bar, gridDim.x, gen, fence_all)) { bs.ctrl[BC_STALLED] = 1u; return; }
        SMX_PROF(7)
      }
    }
    relax_commit_cascade_counters(s, nc, nt);
    {                                                                                            // ---- floods (and their nested particles), one per wavefront while the list allows
      uint32_t n = __hip_atomic_load(&bs.ctrl[BC_NFLOOD + par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (n > bs.list_cap) n = bs.list_cap;
      if (n) {
        const bool due = relax_floods_due(bs, epoch, n, __hip_atomic_load(&bs.ctrl[BC_NRUN + par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        unsigned long long nst = 0;
        for (uint32_t i0 = gw; i0 < n; i0 += 64u * NW) {
          const uint32_t i = i0 + lane * NW;
          bool again = false; uint32_t slot = 0u;
          if (i < n && !due) { slot = bs.flist[par][i]; again = true; }                          // not this epoch: the list moves on
          else if (i < n) {
            slot = bs.flist[par][i];
            BatchPolicy pol(s, bs);
            Sim<BatchPolicy> sim(s, sh, pol);
            Frame st[MAX_FRAMES];
            batch_water_epoch(sim, bs, slot, epoch, st);
            sim.flush_counters(); nst += pol.a_nsteps; pol.finish(s);
            again = bs.water[slot].state == B_ENDED;
          }
          relax_flood_later(bs, slot, par ^ 1u, again);
        }
        relax_commit_flood_steps(s, nst);
        SMX_PROF(8)
        if (!grid_sync(bar, gridDim.x, gen, fence_all)) { bs.ctrl[BC_STALLED] = 1u; return; }
        SMX_PROF(9)
      }
    }
    if (prof) bar->prof[10] += 1ull;
  }
#undef SMX_PROF
}
// how many particles share a wavefront: as few as fill `SMX_BATCH_WAVES` wavefronts (default 256 = one per CU; 0 = always 64;
// measured in profiles/r02_batched_schedules.md, sweep 5). Which lane runs a slot has no influence on results.
static uint32_t batch_lanes_per_wave(uint32_t nlive) {
  static const uint32_t waves = [] { const char* e = getenv("SMX_BATCH_WAVES"); return e ? (uint32_t)atoi(e) : 256u; }();
  if (!waves) return 64u;
  const uint32_t l = (nlive + waves - 1u) / waves;
  return l < 1u ? 1u : (l > 64u ? 64u : l);
}
// rebuild the list of slots that still run (order is irrelevant: a slot's identity is its index, not its lane)
template <bool WIND>
__global__ void __launch_bounds__(256) k_batch_compact(BatchShared bs, uint32_t nslots) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t st_ = i < nslots ? (WIND ? bs.wind[i].state : bs.water[i].state) : B_DONE;
  const bool alive = st_ != B_DONE && st_ != B_AWAY;
  const unsigned long long m = __ballot(alive);
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t base = 0;
  if (lane == 0 && m) base = atomicAdd(&bs.ctrl[BC_NLIVE], (uint32_t)__popcll(m));
  base = (uint32_t)__shfl((int)base, 0);
  if (alive) bs.live[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
}
__global__ void __launch_bounds__(256) k_batch_iota(uint32_t* live, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) live[i] = i;
}
// between kernels: nodes released during the last kernel go back on the free stack
__global__ void __launch_bounds__(256) k_batch_merge_freed(DevState s, BatchShared bs) {
  uint32_t nf = bs.ctrl[BC_NFREED];
  if (nf > bs.freed_cap) nf = bs.freed_cap;
  const uint32_t base = *s.free_count;
  for (uint32_t i = threadIdx.x; i < nf; i += 256u) s.freelist[base + i] = bs.freed[i];
  __syncthreads();
  if (threadIdx.x == 0) { *s.free_count = base + nf; bs.ctrl[BC_NFREED] = 0; }
}
template <bool WIND>
__global__ void __launch_bounds__(256) k_batch_counters(DevState s, BatchShared bs, uint32_t nslots) {
  unsigned long long acc[4] = {0, 0, 0, 0};
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nslots; i += gridDim.x * 256u) batch_slot_counters(bs, WIND, i, acc);
#pragma unroll
  for (int k = 0; k < 4; k++)
    for (int off = 32; off > 0; off >>= 1) acc[k] += (unsigned long long)__shfl_down((long long)acc[k], off);
  if ((threadIdx.x & 63u) == 0) batch_commit_counters(s, WIND, acc);
}
// ---------------- grid pass (water.h:335-343) over the 8x8-cell tiles (soil_batch.h GRID_SHIFT) ----------------
// A tile's seep + cascade(.., 3) -- and the nested particles that run inside it -- stay in the tile and the ring of tiles around it (its
// REGION). Regions of tiles two or less apart overlap; such tiles have different colours (tx mod 3, ty mod 3) and are swept in colour
// order. The sweep is a DATAFLOW over the tiles that have work, sorted by (colour, tile): a wavefront takes the next entry, waits until
// every tile of a smaller colour within two tiles of its own that has work is done, sweeps its tile (lane 0) and publishes it. A tile only
// waits for entries before its own, taken by wavefronts that are running: no deadlock; chains of waiting tiles are at most nine long, and
// a tile in a quiet corner never waits for the slowest tile of a colour somewhere else (nine launches with a barrier each: 9 x 5 ms in
// round 4, with 16-cell tiles and no nested particles inside). One tile per wavefront: a tile is a long, branchy path, and lanes of one
// wavefront cannot wait for each other. The order of conflicting tiles is fixed, so the result is the colour-by-colour sweep's.
constexpr uint32_t GT_PENDING = 0xFFFFFFFFu;
struct GridTiles {
  uint32_t* tdone;                            // [held tile columns][gty]: 0 = nothing to do here, GT_PENDING, 1 = swept
  uint32_t* pend[2];                          // entries colour * (gtx * gty) + tile (tile = tx * gty + ty): as appended / sorted
  uint32_t* count;                            // [0] entries, [1] the dataflow kernel's cursor
  int gtx, gty, txb, txe;                     // tiles of the map; tile columns [txb, txe) held by tdone
  int tx_lo, tx_hi;                           // tile columns of this sweep
  unsigned long long keybase;                 // key prefix of a tile's children: keybase + entry
};
__global__ void __launch_bounds__(256) k_grid_tiles_init(DevState s, BatchShared bs, GridTiles g) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t ncol = (uint32_t)(g.tx_hi - g.tx_lo);
  bool act = false; uint32_t entry = 0;
  if (i < ncol * (uint32_t)g.gty) {
    const int tx = g.tx_lo + (int)(i / (uint32_t)g.gty), ty = (int)(i % (uint32_t)g.gty);
    entry = (uint32_t)((tx % 3) * 3 + (ty % 3)) * ((uint32_t)g.gtx * (uint32_t)g.gty) + (uint32_t)tx * (uint32_t)g.gty + (uint32_t)ty;
    if (batch_grid_tile_phase(s, bs, tx)) {
      const int x0 = tx << GRID_SHIFT, y0 = ty << GRID_SHIFT;
      for (int x = x0; x < x0 + (1 << GRID_SHIFT) && x < s.dimx; x++) {   // (a tile row = 8 consecutive bits of the bitmap)
        const size_t c = (size_t)x * s.dimy + y0;
        int len = s.dimy - y0; if (len > (1 << GRID_SHIFT)) len = 1 << GRID_SHIFT;
        for (int y = 0; y < len; y++) act |= ((s.active[(c + y) >> 6] >> ((c + y) & 63)) & 1ull) != 0ull;
      }
    }
    g.tdone[(size_t)(tx - g.txb) * g.gty + ty] = act ? GT_PENDING : 0u;
  }
  const uint32_t k = relax_append(&g.count[0], act);
  if (act) g.pend[0][k] = entry;
}
// (experiments, SMX_FLOOD_PROF=1: where does a sweep go? [0..31] tiles by log2 of their sweep time in 100 MHz ticks, [32..63] their active cells, [72..103] the steps of
//  their nested particles, [64] tiles, [65] ticks sweeping, [66] ticks waiting for tiles of earlier colours, [67] the longest sweep; printed by smx_destroy)
__device__ unsigned long long g_tile_prof[104];
static bool flood_prof_on() { static const bool on = [] { const char* e = getenv("SMX_FLOOD_PROF"); return e && atoi(e) != 0; }(); return on; }
__global__ void __launch_bounds__(64) k_grid_tiles_flow(DevState s, BatchShared bs, GridTiles g, const uint32_t* __restrict__ order, uint32_t n, uint32_t poll_naps_prof, uint32_t budget) {
  const uint32_t poll_naps = poll_naps_prof & 0x7FFFFFFFu; const bool prof = (poll_naps_prof >> 31) != 0u;
  SMX_LOAD_SOILS(sh)
  __shared__ uint32_t s_i;
  unsigned long long vis = 0, gns = 0;
  const uint32_t nt = (uint32_t)g.gtx * (uint32_t)g.gty;
  bool live = true;
  while (true) {
    if (threadIdx.x == 0) s_i = atomicAdd(&g.count[1], 1u);
    __syncthreads();
    const uint32_t i = s_i;
    __syncthreads();
    if (i >= n) break;
    if (threadIdx.x == 0) {
      const uint32_t entry = order[i], col = entry / nt, tile = entry - col * nt;
      const int tx = (int)(tile / (uint32_t)g.gty), ty = (int)(tile % (uint32_t)g.gty);
      const unsigned long long tw0 = prof ? wall_clock64() : 0ull;
      for (int dx = -2; dx <= 2; dx++)                         // the tiles whose region overlaps this one's and whose colour comes first
        for (int dy = -2; dy <= 2; dy++) {
          const int ax = tx + dx, ay = ty + dy;
          if ((!dx && !dy) || ax < g.tx_lo || ax >= g.tx_hi || ax < g.txb || ax >= g.txe || ay < 0 || ay >= g.gty) continue;
          if ((uint32_t)((ax % 3) * 3 + (ay % 3)) >= col) continue;
          uint32_t* f = &g.tdone[(size_t)(ax - g.txb) * g.gty + ay];
          // (a tile takes milliseconds: its dependants poll every few microseconds -- rocprofv3 counted 122 M polling loads per launch at one poll per 0.2 us, profiles/r06_pmc.json)
          if (live && !spin_wait_while(f, GT_PENDING, &bs.ctrl[BC_STALLED], budget, 8u, poll_naps)) live = false;   // (stalled: the call fails; stop waiting, let the kernel drain)
        }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);                 // what the tiles before wrote to the map is visible from here on
      BatchPolicy pol(s, bs);
      Sim<BatchPolicy> sim(s, sh, pol);
      Frame st[MAX_FRAMES];
      const unsigned long long ts0 = prof ? wall_clock64() : 0ull;
      const unsigned long long vt = batch_grid_tile(sim, tx, ty, g.keybase + (unsigned long long)entry, st);
      vis += vt;
      sim.flush_counters(); gns += pol.a_nsteps;
      if (prof) {
        const unsigned long long dt = wall_clock64() - ts0;
        int b = 63 - __clzll((long long)(dt | 1ull)); if (b > 31) b = 31;
        atomicAdd(&g_tile_prof[b], 1ull); atomicAdd(&g_tile_prof[32 + b], vt); atomicAdd(&g_tile_prof[72 + b], (unsigned long long)pol.a_nsteps);
        atomicAdd(&g_tile_prof[64], 1ull); atomicAdd(&g_tile_prof[65], dt); atomicAdd(&g_tile_prof[66], ts0 - tw0); atomicMax(&g_tile_prof[67], dt);
      }
      pol.finish(s);
      __hip_atomic_store(&g.tdone[(size_t)(tx - g.txb) * g.gty + ty], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (threadIdx.x == 0 && vis) atomicAdd(&s.ctr[C_GRID_ACTIVE], vis);
  if (threadIdx.x == 0 && gns) atomicAdd(&s.ctr[C_GRID_NESTED_STEPS], gns);
}
__global__ void k_batch_grid_finish(DevState s, unsigned long long visited0) {   // counter parity: every other cell "calls" cascade once
  const unsigned long long n = (unsigned long long)s.dimx * s.dimy, v = s.ctr[C_GRID_ACTIVE] - visited0;
  s.ctr[C_WCASCADE_CALLS] += n - v;
}

// ---------------- column strips on several devices: halo columns, migrating particles (library driver: soil_strips_host.h; Python restatement: tests/strips_ref.py) ----------------
__global__ void __launch_bounds__(64) k_batch_child_contains(const BChild* kids, uint32_t nslots, uint32_t* out) {
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  if (i < nslots) out[i] = batch_child_contains(kids, i);
}
struct XCursor { unsigned long long bytes; uint32_t nrec, lost; };
// every column of the DIRTY 4x4 grid tiles whose x lies in [gx0, gx1) tiles: one thread per tile walks its 16 columns twice
// (size, then content); space comes from two atomic cursors, so records land in any order (they are independent)
__global__ void __launch_bounds__(64) k_strip_pack_columns(DevState s, BatchShared bs, int gx0, int gx1, uint8_t* data, unsigned long long cap,
                                                            uint32_t* offs, uint32_t offs_cap, XCursor* cur) {
  const int gty = (s.dimy + 3) >> 2;
  const long long t = (long long)blockIdx.x * 64 + threadIdx.x;
  if (t >= (long long)(gx1 - gx0) * gty) return;
  const int gx = gx0 + (int)(t / gty), gy = (int)(t % gty);
  uint8_t& dflag = bs.dirty[(size_t)gx * gty + gy];
  if (!dflag) return;
  dflag = 0;
  uint32_t ncol = 0; unsigned long long need = 0;
  for (int x = gx << 2; x < (gx << 2) + 4 && x < s.dimx; x++)
    for (int y = gy << 2; y < (gy << 2) + 4 && y < s.dimy; y++) {
      const Sec& c = s.cells[(size_t)x * s.dimy + y];
      uint32_t k = 0;
      if (c.type != EMPTY) { k = 1; for (uint32_t pv = c.prev; pv != NIL; pv = s.pool[pv].prev) k++; }
      need += 24ull + 32ull * k; ncol++;
    }
  const unsigned long long base = atomicAdd(&cur->bytes, need);
  const uint32_t rbase = atomicAdd(&cur->nrec, ncol);
  if (base + need > cap || rbase + ncol > offs_cap) { atomicAdd(&cur->lost, 1u); return; }
  unsigned long long o = base; uint32_t r = rbase;
  for (int x = gx << 2; x < (gx << 2) + 4 && x < s.dimx; x++)
    for (int y = gy << 2; y < (gy << 2) + 4 && y < s.dimy; y++) {
      const size_t cc = (size_t)x * s.dimy + y, fi = (size_t)y * s.dimx + x;
      const Sec& c = s.cells[cc];
      uint32_t k = 0;
      if (c.type != EMPTY) { k = 1; for (uint32_t pv = c.prev; pv != NIL; pv = s.pool[pv].prev) k++; }
      offs[r++] = (uint32_t)(o - base) + (uint32_t)base;       // (offsets are relative to the start of the data area; < 4 GiB)
      uint32_t* h = reinterpret_cast<uint32_t*>(data + o);
      h[0] = (uint32_t)cc; h[1] = k; h[2] = __float_as_uint(s.wtrack[fi]); h[3] = __float_as_uint(s.windfreq[fi]); h[4] = s.flags[cc]; h[5] = 0;
      // sections bottom -> top: the chain is top -> bottom, so fill from the back
      const Sec* t2 = &c;
      for (uint32_t j = k; j-- > 0;) {
        double* q = reinterpret_cast<double*>(data + o + 24 + 32ull * j);
        q[0] = t2->size; q[1] = t2->floor; q[2] = t2->sat;
        reinterpret_cast<uint32_t*>(q + 3)[0] = t2->type; reinterpret_cast<uint32_t*>(q + 3)[1] = 0;
        if (t2->prev != NIL) t2 = &s.pool[t2->prev];
      }
      o += 24ull + 32ull * k;
    }
}
// one thread per received column: give the old chain back (deferred, like every free of a batched kernel), build the new one
__global__ void __launch_bounds__(64) k_strip_unpack_columns(DevState s, BatchShared bs, const uint8_t* data, const uint32_t* offs, uint32_t nrec, uint32_t nbytes) {
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  if (i >= nrec) return;
  if ((unsigned long long)offs[i] + 24ull > nbytes) return;  // (records must lie inside the message)
  const uint8_t* r = data + offs[i];
  const uint32_t* h = reinterpret_cast<const uint32_t*>(r);
  const uint32_t cc = h[0], count = h[1];
  if (cc >= (unsigned long long)s.dimx * s.dimy || (unsigned long long)offs[i] + 24ull + 32ull * count > nbytes) return;
  if (cc < (unsigned long long)s.x_lo * s.dimy || cc >= (unsigned long long)s.x_hi * s.dimy) return;   // (a strip context holds [x_lo, x_hi) only: cells / flags are biased pointers)
  const size_t x = cc / (size_t)s.dimy, y = cc % (size_t)s.dimy, fi = y * (size_t)s.dimx + x;
  s.wtrack[fi] = __uint_as_float(h[2]); s.windfreq[fi] = __uint_as_float(h[3]);
  BatchPolicy pol(s, bs);
  long long live = 0;
  Sec& c = s.cells[cc];
  if (c.type != EMPTY) { live--; uint32_t pv = c.prev; while (pv != NIL) { const uint32_t nx = s.pool[pv].prev; pol.node_free(s, pv); pv = nx; live--; } }
  Sec cell; cell.size = 0.0; cell.floor = 0.0; cell.sat = 0.0; cell.type = EMPTY; cell.prev = NIL;
  uint32_t pv = NIL;
  for (uint32_t k = 0; k < count; k++) {
    const double* q = reinterpret_cast<const double*>(r + 24 + 32ull * k);
    Sec t; t.size = q[0]; t.floor = q[1]; t.sat = q[2]; t.type = reinterpret_cast<const uint32_t*>(q + 3)[0]; t.prev = pv;
    live++;
    if (k + 1 == count) cell = t;
    else {
      const uint32_t nd = pol.node_alloc(s);
      if (nd == NIL) { atomicAdd(&s.ctr[C_POOL_OVERFLOW], 1ull); cell = t; break; }   // out of nodes: the column keeps what fits (reported, not silent)
      s.pool[nd] = t; pv = nd;
    }
  }
  c = cell; s.flags[cc] = (uint8_t)h[4];
  if (live) atomicAdd(&s.ctr[C_LIVE_SECTIONS], (unsigned long long)live);
}
template <bool WIND>
__global__ void __launch_bounds__(256) k_strip_pack_particles(BatchShared bs, uint32_t nslots, int x0, int x1, uint8_t* data, unsigned long long cap, XCursor* cur) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= nslots) return;
  uint32_t& st = WIND ? bs.wind[i].state : bs.water[i].state;
  if (st == B_DONE || st == B_AWAY) return;
  const int ix = WIND ? bs.wind[i].d.ix : bs.water[i].w.ix;
  if (ix < x0 || ix >= x1) return;
  constexpr unsigned long long rs = 4 + (WIND ? sizeof(BWind) : sizeof(BWater));
  const uint32_t k = atomicAdd(&cur->nrec, 1u);
  if ((k + 1ull) * rs > cap) { atomicAdd(&cur->lost, 1u); return; }
  uint8_t* o = data + k * rs;
  memcpy(o, &i, 4);
  if constexpr (WIND) memcpy(o + 4, &bs.wind[i], sizeof(BWind)); else memcpy(o + 4, &bs.water[i], sizeof(BWater));
  st = B_AWAY;
}
template <bool WIND>
__global__ void __launch_bounds__(256) k_strip_unpack_particles(BatchShared bs, const uint8_t* data, uint32_t n) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n) return;
  constexpr unsigned long long rs = 4 + (WIND ? sizeof(BWind) : sizeof(BWater));
  const uint8_t* o = data + k * rs;
  uint32_t i; memcpy(&i, o, 4);
  if constexpr (WIND) memcpy(&bs.wind[i], o + 4, sizeof(BWind)); else memcpy(&bs.water[i], o + 4, sizeof(BWater));
}
__global__ void __launch_bounds__(256) k_map_frequency(float* __restrict__ freq, const float* __restrict__ track, size_t n4, size_t n) {
  const float lrate = 0.01f, K = 50.0f;      // water.h:359-360
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = i; k < n4; k += stride) {
    float4 f = reinterpret_cast<float4*>(freq)[k];
    const float4 t = reinterpret_cast<const float4*>(track)[k];
    f.x = (1.0f - lrate) * f.x + lrate * K * t.x / (1.0f + K * t.x);
    f.y = (1.0f - lrate) * f.y + lrate * K * t.y / (1.0f + K * t.y);
    f.z = (1.0f - lrate) * f.z + lrate * K * t.z / (1.0f + K * t.z);
    f.w = (1.0f - lrate) * f.w + lrate * K * t.w / (1.0f + K * t.w);
    reinterpret_cast<float4*>(freq)[k] = f;
  }
  for (size_t k = n4 * 4 + i; k < n; k += stride)
    freq[k] = (1.0f - lrate) * freq[k] + lrate * K * track[k] / (1.0f + K * track[k]);
}

// Layermap::initialize (layermap.h:163-216). One thread per cell builds its whole column: the layers are
// added in file order with the rules of Layermap::add (merge same type, water swap, push). Pool nodes for
// buried sections come from an atomic bump on the free stack (node identity is unobservable).
struct LayerDev { uint32_t type; float min, bias, scale; NoiseCfg noise; float z; };
constexpr int MAX_LAYERS = 16;
struct LayersDev { int n; LayerDev l[MAX_LAYERS]; };

__global__ void __launch_bounds__(256) k_init_terrain(DevState s, LayersDev L) {
  const size_t n = (size_t)s.x_hi * s.dimy;
  const size_t c = (size_t)s.x_lo * s.dimy + (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  const int i = (int)(c / s.dimy), j = (int)(c % s.dimy);
  double csize[MAX_LAYERS]; uint32_t ctype[MAX_LAYERS];
  int ns = 0;
  const float px = (float)i / (float)s.dimx, py = (float)j / (float)s.dimy;
  for (int l = 0; l < L.n; l++) {
    float val = L.l[l].bias + L.l[l].scale * noise_fbm(L.l[l].noise, px, py, L.l[l].z);   // SurfLayer::get surface.h:95-99
    if (val < L.l[l].min) val = L.l[l].min;
    const double h = val;
    const uint32_t t = L.l[l].type;
    if (h <= 0) continue;                                        // layermap.h:237-240
    if (ns == 0) { csize[0] = h; ctype[0] = t; ns = 1; continue; }
    if (ctype[ns - 1] == t) { csize[ns - 1] += h; continue; }    // :249-253
    if (ctype[ns - 1] == AIR) {                                  // :258-275 water swap
      const double wsize = csize[ns - 1];
      ns--;
      if (ns > 0 && ctype[ns - 1] == t) csize[ns - 1] += h;
      else { csize[ns] = h; ctype[ns] = t; ns++; }
      csize[ns] = wsize; ctype[ns] = AIR; ns++;
      continue;
    }
    csize[ns] = h; ctype[ns] = t; ns++;
  }
  Sec cell;
  cell.size = 0.0; cell.floor = 0.0; cell.sat = 0.0; cell.type = EMPTY; cell.prev = NIL;
  uint8_t flag = 0;
  if (ns > 0) {
    uint32_t base = 0;
    if (ns > 1) {
      base = atomicSub(s.free_count, (uint32_t)(ns - 1));             // old count: our nodes are freelist[base-ns+1 .. base-1]
      if (base < (uint32_t)(ns - 1) || base > (uint32_t)s.pool_capacity) {   // out of nodes (smx_initialize rules this out: guard only)
        atomicAdd(&s.ctr[C_POOL_OVERFLOW], 1ull);
        ns = 1;                                                       // keep the bottom section only, touch nothing out of bounds
      }
    }
    double fl = 0.0;
    uint32_t pv = NIL;
    for (int k = 0; k < ns; k++) {
      Sec r; r.size = csize[k]; r.floor = fl; r.sat = 0.0; r.type = ctype[k]; r.prev = pv;
      if (k == ns - 1) cell = r;
      else { const uint32_t nd = s.freelist[base - 1 - k]; s.pool[nd] = r; pv = nd; }
      fl = r.floor + r.size;                                     // E->floor = height(pos) (:304)
    }
    if (cell.type == AIR) flag |= F_AIR;
    atomicAdd(&s.ctr[C_LIVE_SECTIONS], (unsigned long long)ns);
  }
  s.cells[c] = cell;
  s.flags[c] = flag;
}

__global__ void __launch_bounds__(256) k_heights(DevState s, double* __restrict__ out) {   // (out: the held range, first cell at out[0])
  const size_t n = (size_t)s.x_hi * s.dimy, c_lo = (size_t)s.x_lo * s.dimy;
  const size_t c = c_lo + (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c < n) out[c - c_lo] = s.cells[c].floor + s.cells[c].size;
}
__global__ void __launch_bounds__(256) k_surface(DevState s, uint32_t* __restrict__ out) {
  const size_t n = (size_t)s.x_hi * s.dimy, c_lo = (size_t)s.x_lo * s.dimy;
  const size_t c = c_lo + (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c < n) { uint32_t t = s.cells[c].type; out[c - c_lo] = (t == EMPTY) ? 0u : t; }
}
__global__ void __launch_bounds__(256) k_normals(DevState s, float* __restrict__ out) {
  const size_t n = (size_t)s.dimx * s.dimy;
  const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  SerialPolicy pol(s);
  Sim<SerialPolicy> sim(s, s.soils, pol);
  f3 v = sim.normal((int)(c / s.dimy), (int)(c % s.dimy));
  out[3 * c] = v.x; out[3 * c + 1] = v.y; out[3 * c + 2] = v.z;
}
// Layermap::update(Vertexpool&) for the whole map (layermap.h:475-555): one 44-byte vertex per column. A thread
// builds its vertex in LDS, the block then streams its 256 x 11 floats out as one contiguous run (a 44-byte
// stride per thread would scatter every store over two lines).
// mode -1: no cut (the top section). mode 0: Layermap::update's SLICE cut (layermap.h:477-510). mode 1: Layermap::slice(s)
// (layermap.h:557-613). thr = the cut height in map units, ycut = the y written for a cut column.
struct VCut { int mode; double thr; float ycut; unsigned block0 = 0, rel = 0; };   // block0: first 256-cell block of the launch; rel: `out` holds that block only (one column: smx_fill_vertex_cut)
__global__ void __launch_bounds__(256) k_fill_vertices(DevState s, const float* __restrict__ colors4, uint32_t ncolors,
                                                       float* __restrict__ out, VCut cut) {
  __shared__ float stage[256 * 11];
  const size_t n = (size_t)s.dimx * s.dimy;
  const size_t c0 = ((size_t)blockIdx.x + cut.block0) * 256, c = c0 + threadIdx.x;
  if (c < n) {
    const int x = (int)(c / s.dimy), y = (int)(c % s.dimy);
    Sec r = s.cells[c];
    bool have = r.type != EMPTY;
    if (cut.mode >= 0)                                        // the first section that starts at or below the cut
      while (have && r.floor > cut.thr) { if (r.prev == NIL) have = false; else r = s.pool[r.prev]; }
    const uint32_t ty = have ? r.type : 0u;
    const double h = have ? r.floor + r.size : 0.0;
    float* v = stage + threadIdx.x * 11;
    f3 nrm = {0.f, 1.f, 0.f};
    float py = 0.f;
    uint32_t ci = 0;
    float col[4];
    bool plain = true;                                        // colour = colors4[ci]
    if (cut.mode >= 0 && have && h > cut.thr) {               // the section is cut: flat top at the cut height
      py = cut.ycut;
      const bool wet = r.floor + r.size * r.sat > cut.thr;
      ci = (cut.mode == 0 && wet) ? 0u : ty;
      if (wet) {                                              // mix(a, colour, 0.6) in double (glm), a = Air's colour / red
        plain = false;
        const uint32_t ct = ty < ncolors ? ty : 0u;
        const float a[4] = {cut.mode == 0 ? colors4[0] : 1.f, cut.mode == 0 ? colors4[1] : 0.f, cut.mode == 0 ? colors4[2] : 0.f, cut.mode == 0 ? colors4[3] : 1.f};
        for (int k = 0; k < 4; k++) col[k] = (float)((double)a[k] * (1.0 - 0.6) + (double)colors4[4 * ct + k] * 0.6);
      }
    } else if (cut.mode >= 0 ? have : !(h == 0.0 && ty == 0u)) {
      SerialPolicy pol(s);
      Sim<SerialPolicy> sim(s, s.soils, pol);
      nrm = sim.normal(x, y);
      py = (float)(s.scale * h);
      ci = ty;
      if (cut.mode == 1 && r.sat == 0.0) { plain = false; col[0] = 1.f; col[1] = 0.f; col[2] = 0.f; col[3] = 1.f; }   // :598-604
    }
    const uint32_t cc = ci < ncolors ? ci : 0u;
    if (plain) { col[0] = colors4[4 * cc]; col[1] = colors4[4 * cc + 1]; col[2] = colors4[4 * cc + 2]; col[3] = colors4[4 * cc + 3]; }
    v[0] = (float)x; v[1] = py; v[2] = (float)y;
    v[3] = nrm.x; v[4] = nrm.y; v[5] = nrm.z;
    v[6] = col[0]; v[7] = col[1]; v[8] = col[2]; v[9] = col[3];
    v[10] = (float)(int)ci;
  }
  __syncthreads();
  const size_t left = n - c0, cells = left < 256 ? left : 256, nf = cells * 11;
  for (size_t i = threadIdx.x; i < nf; i += 256) out[(cut.rel ? (size_t)blockIdx.x * 256 : c0) * 11 + i] = stage[i];
}
__global__ void __launch_bounds__(256) k_bilinear(DevState s, const float* __restrict__ pos, int n, double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  SerialPolicy pol(s);
  Sim<SerialPolicy> sim(s, s.soils, pol);
  out[i] = sim.height_bilinear(f2{pos[2 * i], pos[2 * i + 1]});
}

// single-operation kernels behind the point API (host-driven Layermap::add/remove etc.)
struct PointOp { int op; int x, y; double a; uint32_t type; float px, py; int k; };
__global__ void __launch_bounds__(64) k_point_op(DevState s, PointOp o, double* out) {
  SMX_LOAD_SOILS(sh)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  SerialPolicy pol(s);
  Sim<SerialPolicy> sim(s, sh, pol);
  Frame st[MAX_FRAMES];
  int depth = 0;
  Water dummy;
  dummy.pos = {0.f, 0.f}; dummy.speed = {0.f, 0.f}; dummy.volume = 0.0; dummy.sediment = 0.0; dummy.evaprate = 0.0;
  dummy.spill = 0; dummy.ix = dummy.iy = 0; dummy.friction = 0.f; dummy.surface = dummy.contains = 0;
  switch (o.op) {
    case 0: sim.add(o.x, o.y, o.a, o.type); break;
    case 1: *out = sim.remove(o.x, o.y, o.a); break;
    case 2: if (o.k > 0) sim.template particle_cascade<1>(f2{o.px, o.py}); else sim.template particle_cascade<0>(f2{o.px, o.py}); break;
    case 3: sim.push_frame(st, depth, o.x, o.y, o.k); sim.water_drive(dummy, false, false, st, depth); break;
    case 4: sim.seep(o.x, o.y); break;
    case 5: *out = (double)pol.rand1(s); break;
  }
  sim.flush_counters();
  pol.finish(s);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// roctx ranges per phase: resolved at run time, so the library does not depend on the profiler's marker library
struct RoctxApi {
  int (*push)(const char*) = nullptr; int (*pop)() = nullptr; bool tried = false;
  void load() {
    if (tried) return;
    tried = true;
    // only a marker library that is ALREADY in the process (a profiler preloads it; PyTorch links libroctx64): nothing is loaded for ranges
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD)) {
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA")); pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push && pop) return;
      }
    }
    push = nullptr; pop = nullptr;
  }
};
static RoctxApi g_roctx;
struct roctx_range {
  bool on;
  explicit roctx_range(const char* name) { g_roctx.load(); on = g_roctx.push != nullptr; if (on) g_roctx.push(name); }
  ~roctx_range() { if (on) g_roctx.pop(); }
};
struct EvPair { hipEvent_t a, b; int phase; uint32_t launches; };

struct smx_ctx {
  smx_config cfg;
  DevState d;
  hipStream_t stream = nullptr;
  std::string err;
  size_t ncells = 0, nw0 = 0, nw1 = 0, nw2 = 0;
  size_t c_lo = 0, c_hi = 0, lcells = 0;   // cells HELD by this context: [c_lo, c_hi) = columns [x_lo, x_hi) (the whole map unless smx_create_strip)
  Sec* cells_alloc = nullptr; uint8_t* flags_alloc = nullptr; uint32_t* rhead_alloc = nullptr; uint32_t* cflag_alloc = nullptr; uint32_t* cstate_alloc = nullptr;   // unbiased allocations
  bool partial() const { return lcells != ncells; }
  SoilP* d_soils = nullptr;
  int nsoils = 0;
  double* d_scratch = nullptr;        // 1 double for point ops
  float* d_vtx = nullptr; float* d_vcol = nullptr; float* d_vtx1 = nullptr;   // smx_fill_vertices scratch (kept); d_vtx1: one 256-cell block (smx_fill_vertex_cut)
  float vcol_host[256 * 4]; int vcol_n = -1;          // the colour table on the device
  void* d_plane = nullptr; size_t plane_cap = 0;     // smx_read_heights / _surface / smx_normals scratch (kept)
  std::vector<EvPair> pending;
  std::vector<hipEvent_t> evpool;
  smx_timing timing{};
  uint32_t spec_sub = 0, spec_maxnest = 0; bool spec_realloc = false;   // smx_set_spec_limits (0 = defaults)
  int water_generations = SMX_WATER_GENERATIONS;             // smx_set_water_generations (throughput engines: 8)
  int water_stagger = SMX_WATER_STAGGER;                     // smx_set_water_stagger: > 0 = the k generations of a water phase are BORN this many epochs apart inside ONE phase (relaxed engine)
  uint32_t stagger_phase_n = 0;                              // ... the top-level particle count of the phase that is running staggered (0: none)
  uint64_t spec_rounds = 0, spec_aborts = 0, spec_kills = 0, spec_fallbacks = 0, spec_executed = 0, spec_subphases_cut = 0, spec_serial_particles = 0;
  SpecShared sh{};                    // speculative engine buffers (allocated on first use)
  uint32_t spec_cap = 0;              // particles the buffers are sized for
  uint32_t* h_ctrl = nullptr;         // pinned host copy of the control block
  // batched engine (allocated on first use)
  BatchShared bs{};
  uint32_t batch_cap = 0;             // slots the particle buffers are sized for
  bool batch_wind_buf = false;
  uint32_t* d_draws = nullptr; uint32_t* h_draws = nullptr; uint32_t draws_cap = 0;
  BChild* d_kids = nullptr; uint32_t kids_cap = 0;
  unsigned long long* d_skeys[2] = {nullptr, nullptr}; uint32_t* d_sidx[2] = {nullptr, nullptr}; void* d_stemp = nullptr; size_t stemp_bytes = 0; uint32_t sort_cap = 0;   // children sort (device)
  uint32_t* h_bctrl = nullptr;
  int batch_dilate = 0;
  uint32_t relax_wind_min = 0xFFFFFFFFu;   // relaxed schedule: wind runs relaxed while more particles than this are running (default: never --
                                      // measured, profiles/r03_relaxed.md: the exclusive schedule's 8-32 steps per launch serve wind better)
  int relax_wind_steps = 4;           // ... taking up to this many steps per epoch (<= R_MAXSTEPS)
  int relax_water_steps = 1;          // relaxed water: steps per epoch (1 = an edit is visible to everybody's next step)
  uint32_t relax_tag = 0;             // epoch tags of the cascade flags (unique over the context's life)
  struct GridBar* d_gbar = nullptr;   // k_relax_epochs: the device-wide barrier of the persistent launch
  int mega_blocks = 0;                // ... its grid (0: not sized yet): co-resident workgroups, hipLaunchCooperativeKernel
  bool mega_off = false;              // ... the device refused a cooperative launch: the per-phase launches run instead (said once on stderr)
  int relax_persistent = -1, relax_tail_at = -1;   // smx_set_relax_launch (-1: the defaults / the environment's)
  uint32_t flood_launch_no = 0;       // k_relax_floods launches so far (every 7th is bracketed by HIP events)
  uint32_t relax_cap = 0;             // slots the relaxed schedule's buffers are sized for
  int strips_n = 1, strips_a = 16, strips_b = 48;
  int own_x0 = 0, own_x1 = -1;        // strips on several devices: the columns this context owns (-1: to the map's edge)
  // step-wise generation state (smx_d_*)
  bool d_wind = false, d_first = true; uint32_t d_nslots = 0, d_epoch = 0, d_chunk = 0;
  uint32_t* d_contains = nullptr; uint32_t contains_cap = 0;
  uint8_t* d_xdata = nullptr; uint32_t* d_xoffs = nullptr; XCursor* d_xcur = nullptr;
  unsigned long long xdata_cap = 0; uint32_t xoffs_cap = 0;
  uint8_t* d_rdata = nullptr; uint32_t* d_roffs = nullptr; uint8_t* d_xpart = nullptr; uint8_t* d_rpart = nullptr; unsigned long long xpart_cap = 0;   // strips in the library
  struct StripState* strip = nullptr;
  uint32_t* d_tdone = nullptr; uint32_t* d_tpend[2] = {nullptr, nullptr}; uint32_t* d_tcount = nullptr; uint32_t* h_tcount = nullptr; void* d_tsort = nullptr; size_t tsort_bytes = 0; uint32_t tsort_cap = 0;   // grid pass: tile states, pending lists
  uint64_t batch_epochs = 0, batch_generations = 0, batch_children_lost = 0, grid_passes = 0;
};

#define HIPCHK(call)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (call);                                                                           \
    if (e_ != hipSuccess) {                                                                           \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                                   \
      return -1;                                                                                      \
    }                                                                                                 \
  } while (0)

static hipEvent_t ev_get(smx_ctx* ctx) {
  if (!ctx->evpool.empty()) { hipEvent_t e = ctx->evpool.back(); ctx->evpool.pop_back(); return e; }
  hipEvent_t e; hipEventCreate(&e); return e;
}
static void drain_events(smx_ctx* ctx) {
  for (EvPair& p : ctx->pending) {
    hipEventSynchronize(p.b);
    float ms = 0; hipEventElapsedTime(&ms, p.a, p.b);
    switch (p.phase) {
      case 0: ctx->timing.ms_water += ms; ctx->timing.launches_water++; break;
      case 1: ctx->timing.ms_grid += ms; ctx->timing.launches_grid++; break;
      case 2: ctx->timing.ms_wind += ms; ctx->timing.launches_wind++; break;
      case 4: ctx->timing.ms_kernel_water += ms; ctx->timing.launches_kernel_water += p.launches; break;
      case 5: ctx->timing.ms_kernel_wind += ms; ctx->timing.launches_kernel_wind += p.launches; break;
      case 6: ctx->timing.ms_kernel_classify += ms; ctx->timing.launches_kernel_classify += p.launches; break;
      case 7: ctx->timing.ms_kernel_gridtiles += ms; ctx->timing.launches_kernel_gridtiles += p.launches; break;
      case 8: ctx->timing.ms_kernel_mapfreq += ms; ctx->timing.launches_kernel_mapfreq += p.launches; break;
      case 9: ctx->timing.ms_kernel_tail += ms; ctx->timing.launches_kernel_tail++; ctx->timing.epochs_kernel_tail += p.launches; break;          // k_relax_tail (p.launches = its epochs)
      case 10: ctx->timing.ms_kernel_epochs += ms; ctx->timing.launches_kernel_epochs++; ctx->timing.epochs_kernel_epochs += p.launches; break;   // k_relax_epochs
      case 11: ctx->timing.ms_kernel_grid_children += ms; ctx->timing.launches_kernel_grid_children += p.launches; break;
      case 12: ctx->timing.ms_kernel_floods += ms; ctx->timing.launches_kernel_floods += p.launches; break;       // k_relax_floods, the bracketed launches
      default: ctx->timing.ms_freq += ms; ctx->timing.launches_freq++; break;
    }
    ctx->evpool.push_back(p.a); ctx->evpool.push_back(p.b);
  }
  ctx->pending.clear();
}
struct PhaseTimer {
  smx_ctx* ctx; EvPair p;
  PhaseTimer(smx_ctx* c, int phase, uint32_t launches = 1) : ctx(c) { p.a = ev_get(c); p.b = ev_get(c); p.phase = phase; p.launches = launches; hipEventRecord(p.a, c->stream); }
  ~PhaseTimer() { hipEventRecord(p.b, ctx->stream); ctx->pending.push_back(p); if (ctx->pending.size() > 512) drain_events(ctx); }
};

static int reset_pool(smx_ctx* ctx) {
  // free stack holds every node index; cells empty; counters zero
  const uint64_t cap = ctx->cfg.pool_capacity;
  std::vector<uint32_t> fl(cap);
  for (uint64_t i = 0; i < cap; i++) fl[i] = (uint32_t)(cap - 1 - i);
  HIPCHK(hipMemcpyAsync(ctx->d.freelist, fl.data(), cap * 4, hipMemcpyHostToDevice, ctx->stream));
  uint32_t fc = (uint32_t)cap;
  HIPCHK(hipMemcpyAsync(ctx->d.free_count, &fc, 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" {

const char* smx_last_error(smx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

static int create_range(const smx_config* cfg, int x_lo, int x_hi, smx_ctx** out) {
  if (!cfg || !out || cfg->dimx <= 0 || cfg->dimy <= 0 || cfg->pool_capacity == 0 || cfg->pool_capacity >= 0x7FFFFFFFull) return -2;
  if (x_lo < 0 || x_hi > cfg->dimx || x_lo >= x_hi) return -2;
  smx_ctx* ctx = new smx_ctx();
  ctx->cfg = *cfg;
  *out = ctx;   // handed out even on failure so the caller can read smx_last_error(); smx_destroy() is safe
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { ctx->err = "no HIP device visible (the soilmx product path has no CPU fallback)"; return -3; }
  HIPCHK(hipSetDevice(cfg->device));
  HIPCHK(hipStreamCreate(&ctx->stream));
  hipDeviceSetLimit(hipLimitStackSize, 16384);   // recursion depth of cascade<->nested particles is bounded by spill (water.h:34)
  const size_t n = (size_t)cfg->dimx * cfg->dimy;
  ctx->ncells = n;
  ctx->nw0 = (n + 63) >> 6; ctx->nw1 = (ctx->nw0 + 63) >> 6; ctx->nw2 = (ctx->nw1 + 63) >> 6;
  DevState& d = ctx->d;
  memset(&d, 0, sizeof(d));
  d.dimx = cfg->dimx; d.dimy = cfg->dimy; d.scale = cfg->scale; d.pool_capacity = cfg->pool_capacity;
  d.x_lo = x_lo; d.x_hi = x_hi;
  ctx->c_lo = (size_t)x_lo * cfg->dimy; ctx->c_hi = (size_t)x_hi * cfg->dimy; ctx->lcells = ctx->c_hi - ctx->c_lo;
  const size_t ln = ctx->lcells;
  HIPCHK(hipMalloc(&ctx->cells_alloc, ln * sizeof(Sec)));
  d.cells = ctx->cells_alloc - ctx->c_lo;                     // global cell index, held range only (DevState::x_lo)
  HIPCHK(hipMalloc(&d.pool, cfg->pool_capacity * sizeof(Sec)));
  HIPCHK(hipMalloc(&d.freelist, cfg->pool_capacity * 4));
  HIPCHK(hipMalloc(&d.free_count, 64));
  HIPCHK(hipMalloc(&ctx->flags_alloc, ln + 8));   // (whole 32-bit words: SerialPolicy::set_flag; c_lo is a multiple of 4 or the context holds everything)
  d.flags = ctx->flags_alloc - ctx->c_lo;
  HIPCHK(hipMalloc(&d.wfreq, n * 4)); HIPCHK(hipMalloc(&d.wtrack, n * 4)); HIPCHK(hipMalloc(&d.windfreq, n * 4));
  HIPCHK(hipMalloc(&d.rnd, sizeof(RandState)));
  HIPCHK(hipMalloc(&d.ctr, C_COUNT * 8));
  HIPCHK(hipMalloc(&d.active, (ctx->nw0 + 64) * 8)); HIPCHK(hipMalloc(&d.active1, (ctx->nw1 + 64) * 8)); HIPCHK(hipMalloc(&d.active2, (ctx->nw2 + 64) * 8));
  HIPCHK(hipMalloc(&ctx->d_scratch, 64));
  HIPCHK(hipMalloc(&ctx->d_soils, 256 * sizeof(SoilP)));
  d.soils = ctx->d_soils;
  HIPCHK(hipMemsetAsync(d.wfreq, 0, n * 4, ctx->stream)); HIPCHK(hipMemsetAsync(d.wtrack, 0, n * 4, ctx->stream));
  HIPCHK(hipMemsetAsync(d.windfreq, 0, n * 4, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->flags_alloc, 0, ln + 8, ctx->stream));
  HIPCHK(hipMemsetAsync(d.ctr, 0, C_COUNT * 8, ctx->stream));
  {   // all columns empty
    std::vector<Sec> empty(ln);
    for (size_t i = 0; i < ln; i++) { empty[i].size = 0; empty[i].floor = 0; empty[i].sat = 0; empty[i].type = EMPTY; empty[i].prev = NIL; }
    HIPCHK(hipMemcpyAsync(ctx->cells_alloc, empty.data(), ln * sizeof(Sec), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  if (reset_pool(ctx)) return -1;
  const smx_soil air = {0, 0, 0, 0, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};   // surface.h:43-49
  if (smx_set_soils(ctx, &air, 1)) return -1;
  if (smx_srand(ctx, 1)) return -1;
  if (const char* e = getenv("SMX_RELAX_WIND_MIN")) ctx->relax_wind_min = (uint32_t)atoll(e);   // (experiments)
  if (const char* e = getenv("SMX_RELAX_WATER_STEPS")) { const int v = atoi(e); if (v >= 1 && v <= R_MAXSTEPS) ctx->relax_water_steps = v; }   // (experiment)
  if (const char* e = getenv("SMX_RELAX_WIND_STEPS")) { const int v = atoi(e); if (v >= 1 && v <= R_MAXSTEPS) ctx->relax_wind_steps = v; }
  return 0;
}
int smx_create(const smx_config* cfg, smx_ctx** out) { return create_range(cfg, 0, cfg ? cfg->dimx : 0, out); }
// A context that HOLDS only the columns [x_lo, x_hi) of a dimx x dimy map -- a column strip and its halo (smx_strips_*): cells, flags
// and the engines' per-cell planes are allocated for that range, per-rank memory follows the strip. x_lo*dimy must be a multiple of 64.
int smx_create_strip(const smx_config* cfg, int32_t x_lo, int32_t x_hi, smx_ctx** out) {
  if (cfg && ((size_t)x_lo * (size_t)cfg->dimy) % 64 != 0) return -2;
  return create_range(cfg, x_lo, x_hi, out);
}
#define FULLMAP(what)                                                                                                 \
  if (ctx->partial()) { ctx->err = what ": not available on a strip context (smx_create_strip): it holds only a part of the map"; return -2; }

static void strips_free(smx_ctx* ctx);
void smx_destroy(smx_ctx* ctx) {
  if (!ctx) return;
  strips_free(ctx);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  if (getenv("SMX_FLOOD_PROF") && atoi(getenv("SMX_FLOOD_PROF"))) {
    unsigned long long h[72];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_flood_prof), sizeof(h)) == hipSuccess && h[64]) {
      fprintf(stderr, "[soilmx] k_relax_floods: %llu floods acted, mean %.1f us; %llu listed floods waited (did not hold their tiles). duration bucket (us): floods, nested steps per flood\n", h[64], 0.01 * (double)h[65] / (double)h[64], h[66]);
      for (int b = 0; b < 32; b++) if (h[b]) fprintf(stderr, "   [%8.1f, %8.1f) %9llu  %7.1f\n", 0.01 * (double)(1ull << b), 0.01 * (double)(2ull << b), h[b], (double)h[32 + b] / (double)h[b]);
    }
  }
  if (getenv("SMX_FLOOD_PROF") && atoi(getenv("SMX_FLOOD_PROF"))) {
    unsigned long long h[104];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tile_prof), sizeof(h)) == hipSuccess && h[64]) {
      fprintf(stderr, "[soilmx] k_grid_tiles_flow: %llu tiles swept, mean %.1f us, longest %.1f us; waiting for earlier colours: mean %.1f us per tile. sweep time bucket (us): tiles, active cells per tile, nested steps per tile, us per active cell\n",
              h[64], 0.01 * (double)h[65] / (double)h[64], 0.01 * (double)h[67], 0.01 * (double)h[66] / (double)h[64]);
      for (int b = 0; b < 32; b++) if (h[b]) fprintf(stderr, "   [%8.1f, %8.1f) %9llu  %6.1f  %7.1f  %7.1f\n", 0.01 * (double)(1ull << b), 0.01 * (double)(2ull << b), h[b], (double)h[32 + b] / (double)h[b],
                                                     (double)h[72 + b] / (double)h[b], h[32 + b] ? 0.01 * 1.5 * (double)(1ull << b) * (double)h[b] / (double)h[32 + b] : 0.0);
    }
  }
#ifdef SMX_PROF_FLOOD
  {
    unsigned long long h[32];
    static const char* nm[10] = {"prelude: add sediment", "prelude: Particle::cascade", "prelude: add water", "prelude: seep", "push_frame (9 views, sort, mask)", "neighbour: nothing to move",
                                 "neighbour: partial transfer", "neighbour: nested particle spawned", "run_nested (a nested particle's steps)", "grid tile: seep of an active cell"};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sect), sizeof(h)) == hipSuccess)
      for (int k = 0; k < 10; k++) if (h[2 * k + 1]) fprintf(stderr, "[soilmx] section %-40s %10llu x %7.2f us = %9.1f ms\n", nm[k], h[2 * k + 1], 0.01 * (double)h[2 * k] / (double)h[2 * k + 1], 1e-5 * (double)h[2 * k]);
  }
#endif
  if (ctx->d_gbar && getenv("SMX_RELAX_MEGA_FLAGS") && (atoi(getenv("SMX_RELAX_MEGA_FLAGS")) & 2)) {   // (experiments: where an epoch of k_relax_epochs goes, workgroup 0's clock)
    GridBar hb; if (hipMemcpy(&hb, ctx->d_gbar, sizeof(hb), hipMemcpyDeviceToHost) == hipSuccess && hb.prof[10]) {
      static const char* nm[10] = {"step", "sync", "apply", "sync", "filter", "sync", "colours", "sync", "floods", "sync"};
      fprintf(stderr, "[soilmx] k_relax_epochs, %llu epochs, microseconds per epoch (100 MHz clock):", hb.prof[10]);
      for (int k = 0; k < 10; k++) fprintf(stderr, " %s %.1f", nm[k], 0.01 * (double)hb.prof[k] / (double)hb.prof[10]);
      fprintf(stderr, "\n");
    }
  }
  drain_events(ctx);
  for (hipEvent_t e : ctx->evpool) hipEventDestroy(e);
  DevState& d = ctx->d;
  hipFree(ctx->cells_alloc); hipFree(d.pool); hipFree(d.freelist); hipFree(d.free_count); hipFree(ctx->flags_alloc);
  hipFree(d.wfreq); hipFree(d.wtrack); hipFree(d.windfreq); hipFree(d.rnd); hipFree(d.ctr);
  hipFree(d.active); hipFree(d.active1); hipFree(d.active2); hipFree(ctx->d_scratch); hipFree(ctx->d_soils);
  hipFree(ctx->d_vtx); hipFree(ctx->d_vcol); hipFree(ctx->d_plane);
  if (ctx->sh.stamp) {
    SpecShared& sh = ctx->sh;
    hipFree(sh.stamp); hipFree(sh.claim); hipFree(sh.ctrl); hipFree(sh.reclaim); hipHostFree(ctx->h_ctrl);
    // (hipFree(nullptr) is a no-op: after a failed spec_alloc some of these are null)
    hipFree(sh.part); hipFree(sh.chunk_tab); hipFree(sh.arena); hipFree(sh.alloc_log); hipFree(sh.free_log);
    hipFree(sh.pending); hipFree(sh.draws); hipFree(sh.scout_cells); hipFree(sh.scout_len); hipFree(sh.ckpt);
  }
  {
    BatchShared& bs = ctx->bs;
    hipFree(bs.claim[0]); hipFree(bs.claim[1]); hipFree(bs.water); hipFree(bs.wind); hipFree(bs.live); hipFree(bs.children);
    hipFree(bs.freed); hipFree(bs.ctrl); hipFree(bs.dirty); hipFree(bs.req); hipFree(bs.rnext); hipFree(bs.rcount); hipFree(ctx->rhead_alloc); hipFree(ctx->cflag_alloc); hipFree(ctx->cstate_alloc); hipFree(bs.flist[0]); hipFree(bs.flist[1]); hipFree(bs.clist[0]); hipFree(bs.clist[1]); hipFree(bs.flagl[0]); hipFree(bs.flagl[1]); hipFree(ctx->d_rdata); hipFree(ctx->d_roffs); hipFree(ctx->d_xpart); hipFree(ctx->d_rpart); hipFree(ctx->d_contains); hipFree(ctx->d_xdata); hipFree(ctx->d_xoffs); hipFree(ctx->d_xcur); hipFree(ctx->d_draws); hipFree(ctx->d_kids); hipFree(ctx->d_skeys[0]); hipFree(ctx->d_skeys[1]); hipFree(ctx->d_sidx[0]); hipFree(ctx->d_sidx[1]); hipFree(ctx->d_stemp); hipFree(ctx->d_vtx1); hipFree(ctx->d_tsort); hipFree(ctx->d_tdone); hipFree(ctx->d_tpend[0]); hipFree(ctx->d_tpend[1]); hipFree(ctx->d_tcount); hipFree(ctx->d_gbar); if (ctx->h_tcount) hipHostFree(ctx->h_tcount);
    if (ctx->h_draws) hipHostFree(ctx->h_draws);
    if (ctx->h_bctrl) hipHostFree(ctx->h_bctrl);
  }
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}

int smx_set_soils(smx_ctx* ctx, const smx_soil* soils, int32_t n) {
  if (n <= 0 || n > 256) { ctx->err = "smx_set_soils: 1..256 soils"; return -2; }
  for (int i = 0; i < n; i++)
    if (soils[i].transports >= (uint32_t)n || soils[i].erodes >= (uint32_t)n || soils[i].cascades >= (uint32_t)n || soils[i].abrades >= (uint32_t)n) {
      ctx->err = "smx_set_soils: soil index out of range"; return -2;
    }
  HIPCHK(hipMemcpyAsync(ctx->d_soils, soils, n * sizeof(SoilP), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->nsoils = n; ctx->d.nsoils = n;
  return 0;
}
int smx_set_scale(smx_ctx* ctx, int32_t scale) { ctx->d.scale = scale; ctx->cfg.scale = scale; return 0; }
int smx_set_engine(smx_ctx* ctx, int32_t engine) { ctx->cfg.engine = engine; return 0; }
// speculative engine: top-level particles per sub-phase / nested-particle slots of a sub-phase (0 = default: 4096 / 8192). Tests use
// small values to drive the sub-phase hand-over; results never depend on them.
int smx_set_spec_limits(smx_ctx* ctx, uint32_t particles_per_subphase, uint32_t nested_slots) {
  if (nested_slots && (nested_slots < 128u || nested_slots > 60000u)) { ctx->err = "smx_set_spec_limits: nested_slots must lie in [128, 60000]"; return -2; }
  ctx->spec_sub = particles_per_subphase ? std::max(64u, particles_per_subphase) : 0u; ctx->spec_maxnest = nested_slots; ctx->spec_realloc = true;
  return 0;
}
// Column strips need every reservation to fit between the inset and the seam half-width (soil_batch.h BStrips), or a particle
// near a boundary is eligible in neither kind of chunk and the phase never ends. Widest reservations of batch_schedule:
// water 3 tiles of 4 cells, wind 3 tiles of 4 / 2 tiles of 8, each plus the margin on both sides.
static bool batch_strips_fit(int dilate, int inset, int seam_halfwidth) {
  const int dw = dilate > 0 ? dilate : 0, dd = dilate > 1 ? dilate : 1;
  const int w_water = (3 + 2 * dw) * 4, w_wind = (3 + 2 * dd) * 4, w_tail = (2 + 2 * dd) * 8;
  return std::max(w_water, std::max(w_wind, w_tail)) <= seam_halfwidth - inset;
}
int smx_set_batch_dilate(smx_ctx* ctx, int32_t tiles) {
  if (tiles < 0) tiles = 0;
  if (ctx->strips_n > 1 && !batch_strips_fit(tiles, ctx->strips_a, ctx->strips_b)) {
    ctx->err = "smx_set_batch_dilate: reservations with this margin do not fit the seam zones of the column strips"; return -2;
  }
  ctx->batch_dilate = tiles;
  return 0;
}
int smx_get_water_generations(smx_ctx* ctx, int32_t* k) { if (!k) return -2; *k = ctx->water_generations; return 0; }
int smx_set_water_stagger(smx_ctx* ctx, int32_t gap_epochs) {
  if (gap_epochs < 0 || gap_epochs > 65536) { ctx->err = "smx_set_water_stagger: 0 (consecutive generations) .. 65536 epochs"; return -2; }
  ctx->water_stagger = gap_epochs;
  return 0;
}
int smx_get_water_stagger(smx_ctx* ctx, int32_t* gap_epochs) { if (!gap_epochs) return -2; *gap_epochs = ctx->water_stagger; return 0; }
int smx_set_water_generations(smx_ctx* ctx, int32_t k) {
  if (k < 1 || k > 64) { ctx->err = "smx_set_water_generations: 1..64"; return -2; }
  ctx->water_generations = k;
  return 0;
}
int smx_set_grid_interleave(smx_ctx* ctx, int32_t) {
  ctx->err = "smx_set_grid_interleave: removed in round 5 (nested particles run inside their parent's region; the grid pass has no interleave any more)";
  return -2;
}
int smx_set_relax_launch(smx_ctx* ctx, int32_t persistent, int32_t tail_at) {
  if (tail_at > 256) { ctx->err = "smx_set_relax_launch: the tail kernel holds at most 256 particles (one per thread of its workgroup)"; return -2; }
  ctx->relax_persistent = persistent < 0 ? -1 : (persistent ? 1 : 0); ctx->relax_tail_at = tail_at < 0 ? -1 : tail_at;
  return 0;
}
int smx_set_relax_wind(smx_ctx* ctx, uint32_t min_running, int32_t steps_per_epoch) {
  if (steps_per_epoch < 1 || steps_per_epoch > R_MAXSTEPS) { ctx->err = "smx_set_relax_wind: 1..8 steps per epoch"; return -2; }
  ctx->relax_wind_min = min_running; ctx->relax_wind_steps = steps_per_epoch;
  return 0;
}
int smx_set_batch_strips(smx_ctx* ctx, int32_t nstrips, int32_t inset, int32_t seam_halfwidth) {
  if (nstrips < 1 || nstrips > 8 || inset < 0 || seam_halfwidth - inset < 32) { ctx->err = "smx_set_batch_strips: 1..8 strips, seam_halfwidth - inset >= 32"; return -2; }
  if (nstrips > 1 && ctx->cfg.dimx / nstrips < 2 * seam_halfwidth + 16) { ctx->err = "smx_set_batch_strips: strips narrower than two seam zones"; return -2; }
  if (nstrips > 1 && !batch_strips_fit(ctx->batch_dilate, inset, seam_halfwidth)) {
    ctx->err = "smx_set_batch_strips: the reservations of the current margin (smx_set_batch_dilate) do not fit seam_halfwidth - inset"; return -2;
  }
  ctx->strips_n = nstrips; ctx->strips_a = inset; ctx->strips_b = seam_halfwidth;
  return 0;
}
int smx_get_batch_stats(smx_ctx* ctx, uint64_t* epochs, uint64_t* generations, uint64_t* children_lost) {
  if (epochs) *epochs = ctx->batch_epochs;
  if (generations) *generations = ctx->batch_generations;
  if (children_lost) *children_lost = ctx->batch_children_lost;
  return 0;
}
void* smx_stream(smx_ctx* ctx) { return (void*)ctx->stream; }

int smx_srand(smx_ctx* ctx, uint32_t seed) {   // glibc srandom_r, TYPE_3 (stdlib/random_r.c)
  RandState r;
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
  HIPCHK(hipMemcpyAsync(ctx->d.rnd, &r, sizeof(r), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}
int smx_rand(smx_ctx* ctx, int32_t* out) {
  PointOp o{}; o.op = 5;
  hipLaunchKernelGGL(k_point_op, dim3(1), dim3(64), 0, ctx->stream, ctx->d, o, ctx->d_scratch);
  double v = 0;
  HIPCHK(hipMemcpyAsync(&v, ctx->d_scratch, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *out = (int32_t)v;
  return 0;
}

int smx_get_rand_state(smx_ctx* ctx, uint32_t ring31[31], uint32_t* idx, uint64_t* calls) {
  RandState r;
  HIPCHK(hipMemcpyAsync(&r, ctx->d.rnd, sizeof(r), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  memcpy(ring31, r.ring, sizeof(r.ring)); if (idx) *idx = r.idx; if (calls) *calls = r.calls;
  return 0;
}
int smx_set_rand_state(smx_ctx* ctx, const uint32_t ring31[31], uint32_t idx, uint64_t calls) {
  RandState r; memcpy(r.ring, ring31, sizeof(r.ring)); r.idx = idx; r.calls = calls;
  HIPCHK(hipMemcpyAsync(ctx->d.rnd, &r, sizeof(r), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}
int smx_rand_advance(smx_ctx* ctx, uint64_t ndraws) {
  RandState r;
  HIPCHK(hipMemcpyAsync(&r, ctx->d.rnd, sizeof(r), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (uint64_t i = 0; i < ndraws; i++) { uint32_t v = r.ring[r.idx % 31] + r.ring[(r.idx - 3) % 31]; r.ring[r.idx % 31] = v; r.idx++; }
  r.calls += ndraws;
  HIPCHK(hipMemcpyAsync(ctx->d.rnd, &r, sizeof(r), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int smx_initialize(smx_ctx* ctx, int32_t seed, const smx_layer* layers, int32_t nlayers) {
  if (nlayers < 0 || nlayers > MAX_LAYERS) { ctx->err = "smx_initialize: too many layers"; return -2; }
  // Every (cell, layer) pair can become a section. The reference prints "Memory Pool Out-Of-Elements" and silently
  // drops the rest of the terrain (layermap.h:92-95,233); a device-parallel build cannot reproduce WHICH sections it
  // drops, so a pool that cannot hold the worst case is refused up front instead of producing a different map.
  if ((uint64_t)ctx->lcells * (uint64_t)(nlayers > 0 ? nlayers : 1) > ctx->cfg.pool_capacity) {
    ctx->err = "smx_initialize: pool_capacity < cells * layers (the reference's POOLSIZE would overflow during Layermap::initialize)";
    return -4;
  }
  if (reset_pool(ctx)) return -1;
  HIPCHK(hipMemsetAsync(ctx->d.ctr + C_LIVE_SECTIONS, 0, 8, ctx->stream));
  LayersDev L; L.n = nlayers;
  const int MAXSEED = 10000;
  for (int l = 0; l < nlayers; l++) {
    const float f = (float)l / (float)nlayers;                    // layermap.h:183-184
    const int Z = (int)((float)seed + f * (float)MAXSEED);
    LayerDev& d = L.l[l];
    d.type = layers[l].type; d.min = layers[l].min; d.bias = layers[l].bias; d.scale = layers[l].scale;
    d.noise.seed = 1337; d.noise.octaves = (int)layers[l].octaves; d.noise.frequency = layers[l].frequency;
    d.noise.lacunarity = layers[l].lacunarity; d.noise.gain = layers[l].gain;
    float g = d.noise.gain < 0 ? -d.noise.gain : d.noise.gain;    // CalculateFractalBounding (FastNoiseLite.h:473-484)
    float amp = g, ampFractal = 1.0f;
    for (int i = 1; i < d.noise.octaves; i++) { ampFractal += amp; amp *= g; }
    d.noise.bounding = 1 / ampFractal;
    d.z = (float)(Z % MAXSEED) / 1.0f;                            // vec3(i,j,Z%MAXSEED)/vec3(dim.x,dim.y,1) (:191)
  }
  const unsigned nb = (unsigned)((ctx->lcells + 255) / 256);
  hipLaunchKernelGGL(k_init_terrain, dim3(nb), dim3(256), 0, ctx->stream, ctx->d, L);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int smx_import_columns(smx_ctx* ctx, const uint32_t* count, const uint32_t* type, const double* size,
                       const double* floor, const double* sat) {
  const size_t n = ctx->ncells, lo = ctx->c_lo, hi = ctx->c_hi, ln = ctx->lcells;   // (the arrays describe the WHOLE map; a strip context keeps its range)
  const uint64_t cap = ctx->cfg.pool_capacity;
  std::vector<Sec> cells(ln);
  std::vector<uint8_t> flags(ln + 8, 0);
  std::vector<Sec> pool;
  size_t off = 0, total = 0;
  for (size_t c = lo; c < hi; c++) total += count[c];
  if (total > cap) { ctx->err = "smx_import_columns: more sections than pool_capacity"; return -4; }
  pool.reserve(total > ln ? total - ln + 1 : 1);
  for (size_t c = 0; c < n; c++) {
    const uint32_t k = count[c];
    if (c < lo || c >= hi) { off += k; continue; }
    Sec cell; cell.size = 0; cell.floor = 0; cell.sat = 0; cell.type = EMPTY; cell.prev = NIL;
    uint32_t pv = NIL;
    for (uint32_t j = 0; j < k; j++, off++) {
      Sec r; r.size = size[off]; r.floor = floor[off]; r.sat = sat[off]; r.type = type[off]; r.prev = pv;
      if (r.sat != 0.0) flags[c - lo] |= F_SAT;
      if (j == k - 1) cell = r;
      else { pv = (uint32_t)pool.size(); pool.push_back(r); }
    }
    if (cell.type == AIR) flags[c - lo] |= F_AIR;
    cells[c - lo] = cell;
  }
  // node i of `pool` lives at pool index i; the free stack holds the remaining indices
  const uint64_t used = pool.size();
  std::vector<uint32_t> fl(cap);
  uint64_t nf = 0;
  for (uint64_t i = cap; i-- > used;) fl[nf++] = (uint32_t)i;
  uint32_t fc = (uint32_t)nf;
  unsigned long long live = total;
  HIPCHK(hipMemcpyAsync(ctx->cells_alloc, cells.data(), ln * sizeof(Sec), hipMemcpyHostToDevice, ctx->stream));
  if (used) HIPCHK(hipMemcpyAsync(ctx->d.pool, pool.data(), used * sizeof(Sec), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d.freelist, fl.data(), cap * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d.free_count, &fc, 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->flags_alloc, flags.data(), ln, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d.ctr + C_LIVE_SECTIONS, &live, 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int smx_import_frequency(smx_ctx* ctx, const float* wfreq, const float* wtrack, const float* windfreq) {
  const size_t n = ctx->ncells;
  if (wfreq) HIPCHK(hipMemcpyAsync(ctx->d.wfreq, wfreq, n * 4, hipMemcpyHostToDevice, ctx->stream));
  if (wtrack) HIPCHK(hipMemcpyAsync(ctx->d.wtrack, wtrack, n * 4, hipMemcpyHostToDevice, ctx->stream));
  if (windfreq) HIPCHK(hipMemcpyAsync(ctx->d.windfreq, windfreq, n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int smx_num_sections(smx_ctx* ctx, uint64_t* out) {
  unsigned long long live = 0;
  HIPCHK(hipMemcpyAsync(&live, ctx->d.ctr + C_LIVE_SECTIONS, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *out = live;
  return 0;
}

int smx_export_columns(smx_ctx* ctx, uint32_t* count, uint32_t* type, double* size, double* floor, double* sat) {
  const size_t n = ctx->ncells, lo = ctx->c_lo, hi = ctx->c_hi, ln = ctx->lcells;   // (a strip context reports empty columns outside its range)
  const uint64_t cap = ctx->cfg.pool_capacity;
  std::vector<Sec> cells(ln), pool(cap);
  HIPCHK(hipMemcpyAsync(cells.data(), ctx->cells_alloc, ln * sizeof(Sec), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(pool.data(), ctx->d.pool, cap * sizeof(Sec), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  size_t off = 0;
  std::vector<const Sec*> col;
  for (size_t c = 0; c < n; c++) {
    col.clear();
    if (c < lo || c >= hi) { count[c] = 0; continue; }
    if (cells[c - lo].type != EMPTY) {
      const Sec* t = &cells[c - lo];
      while (true) {
        col.push_back(t);
        if (t->prev == NIL) break;
        if (t->prev >= cap || col.size() > cap) { ctx->err = "smx_export_columns: corrupt section chain"; return -5; }
        t = &pool[t->prev];
      }
    }
    count[c] = (uint32_t)col.size();
    for (size_t k = col.size(); k-- > 0; off++) {
      type[off] = col[k]->type; size[off] = col[k]->size; floor[off] = col[k]->floor; sat[off] = col[k]->sat;
    }
  }
  return 0;
}

struct SnapHdr { char magic[8]; int32_t dimx, dimy, scale, nsoils; uint64_t ncells, nsec, rand_calls, pool_free, reserved; };
static_assert(sizeof(SnapHdr) == 64, "snapshot header");
struct RandTrailer { char magic[8]; uint32_t ring[31]; uint32_t idx; uint64_t calls; };
int smx_save(smx_ctx* ctx, const char* path) {
  FULLMAP("smx_save")
  const size_t n = ctx->ncells;
  uint64_t ns = 0;
  if (smx_num_sections(ctx, &ns)) return -1;
  std::vector<uint32_t> count(n), type(ns);
  std::vector<double> size(ns), floor(ns), sat(ns);
  std::vector<float> wf(n), wt(n), wi(n);
  if (smx_export_columns(ctx, count.data(), type.data(), size.data(), floor.data(), sat.data())) return -1;
  if (smx_read_frequency(ctx, wf.data(), wt.data(), wi.data())) return -1;
  // the per-column flag plane (bit1 = "this column has held a saturation", sticky): context state the column records do not
  // carry; the throughput engines fix the active set of their grid pass from it, so a resume without it can differ
  std::vector<uint8_t> fl(n);
  HIPCHK(hipMemcpyAsync(fl.data(), ctx->d.flags, n, hipMemcpyDeviceToHost, ctx->stream));
  RandTrailer tr; memcpy(tr.magic, "SMXRAND1", 8);
  RandState r;
  HIPCHK(hipMemcpyAsync(&r, ctx->d.rnd, sizeof(r), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  memcpy(tr.ring, r.ring, sizeof(tr.ring)); tr.idx = r.idx; tr.calls = r.calls;
  SnapHdr h; memset(&h, 0, sizeof(h)); memcpy(h.magic, "SMXSNAP1", 8);
  h.dimx = ctx->cfg.dimx; h.dimy = ctx->cfg.dimy; h.scale = ctx->d.scale; h.nsoils = ctx->nsoils; h.ncells = n; h.nsec = ns;
  h.rand_calls = r.calls; h.pool_free = ctx->cfg.pool_capacity - ns;
  FILE* f = fopen(path, "wb");
  if (!f) { ctx->err = std::string("smx_save: cannot open ") + path; return -7; }
  bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
  ok = ok && fwrite(count.data(), 4, n, f) == n && fwrite(type.data(), 4, ns, f) == ns && fwrite(size.data(), 8, ns, f) == ns &&
       fwrite(floor.data(), 8, ns, f) == ns && fwrite(sat.data(), 8, ns, f) == ns && fwrite(wf.data(), 4, n, f) == n &&
       fwrite(wt.data(), 4, n, f) == n && fwrite(wi.data(), 4, n, f) == n && fwrite(&tr, sizeof(tr), 1, f) == 1 &&
       fwrite("SMXFLAG1", 1, 8, f) == 8 && fwrite(fl.data(), 1, n, f) == n;
  ok = (fclose(f) == 0) && ok;
  if (!ok) { ctx->err = std::string("smx_save: short write to ") + path; return -7; }
  return 0;
}
int smx_load(smx_ctx* ctx, const char* path) {
  FULLMAP("smx_load")
  FILE* f = fopen(path, "rb");
  if (!f) { ctx->err = std::string("smx_load: cannot open ") + path; return -7; }
  SnapHdr h;
  if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "SMXSNAP1", 8) != 0) { fclose(f); ctx->err = "smx_load: not a soilmx snapshot"; return -7; }
  if (h.dimx != ctx->cfg.dimx || h.dimy != ctx->cfg.dimy || h.ncells != ctx->ncells) { fclose(f); ctx->err = "smx_load: snapshot dimensions differ from the context's"; return -2; }
  const size_t n = ctx->ncells; const uint64_t ns = h.nsec;
  // the header is untrusted input: the section count sizes five arrays below and every later index
  {
    struct stat sb;
    const uint64_t need = sizeof(SnapHdr) + 4ull * n + 28ull * ns + 12ull * n;
    if (ns > ctx->cfg.pool_capacity) { fclose(f); ctx->err = "smx_load: the snapshot holds more sections than pool_capacity"; return -4; }
    if (fstat(fileno(f), &sb) != 0 || (uint64_t)sb.st_size < need) { fclose(f); ctx->err = "smx_load: truncated snapshot (shorter than its header promises)"; return -7; }
  }
  std::vector<uint32_t> count(n), type(ns);
  std::vector<double> size(ns), floor(ns), sat(ns);
  std::vector<float> wf(n), wt(n), wi(n);
  bool ok = fread(count.data(), 4, n, f) == n && fread(type.data(), 4, ns, f) == ns && fread(size.data(), 8, ns, f) == ns &&
            fread(floor.data(), 8, ns, f) == ns && fread(sat.data(), 8, ns, f) == ns && fread(wf.data(), 4, n, f) == n &&
            fread(wt.data(), 4, n, f) == n && fread(wi.data(), 4, n, f) == n;
  RandTrailer tr; const bool have_rand = ok && fread(&tr, sizeof(tr), 1, f) == 1 && memcmp(tr.magic, "SMXRAND1", 8) == 0;
  std::vector<uint8_t> fl; char fmagic[8];
  if (have_rand && fread(fmagic, 1, 8, f) == 8 && memcmp(fmagic, "SMXFLAG1", 8) == 0) { fl.resize(n); if (fread(fl.data(), 1, n, f) != n) fl.clear(); }
  fclose(f);
  if (!ok) { ctx->err = "smx_load: truncated snapshot"; return -7; }
  {
    uint64_t total = 0;
    for (size_t c = 0; c < n; c++) total += count[c];
    if (total != ns) { ctx->err = "smx_load: per-cell section counts do not add up to the header's section count"; return -2; }
    for (uint64_t i = 0; i < ns; i++)
      if (type[i] >= (uint32_t)ctx->nsoils) { ctx->err = "smx_load: a section's soil type is outside the context's soil table (smx_set_soils first)"; return -2; }
  }
  if (smx_import_columns(ctx, count.data(), type.data(), size.data(), floor.data(), sat.data())) return -1;
  if (smx_import_frequency(ctx, wf.data(), wt.data(), wi.data())) return -1;
  if (!fl.empty()) {                                         // the sticky bit on top of what the import derived from the columns
    std::vector<uint8_t> cur(n);
    HIPCHK(hipMemcpyAsync(cur.data(), ctx->d.flags, n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (size_t c = 0; c < n; c++) cur[c] |= (uint8_t)(fl[c] & F_SAT);
    HIPCHK(hipMemcpyAsync(ctx->d.flags, cur.data(), n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  if (have_rand) {
    RandState r; memcpy(r.ring, tr.ring, sizeof(r.ring)); r.idx = tr.idx; r.calls = tr.calls;
    HIPCHK(hipMemcpyAsync(ctx->d.rnd, &r, sizeof(r), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  ctx->d.scale = h.scale; ctx->cfg.scale = h.scale;
  return have_rand ? 0 : 1;   /* 1: state restored, but the file carries no generator state (a bare snapshot): re-seed it yourself */
}

int smx_read_frequency(smx_ctx* ctx, float* wfreq, float* wtrack, float* windfreq) {
  const size_t n = ctx->ncells;
  if (wfreq) HIPCHK(hipMemcpyAsync(wfreq, ctx->d.wfreq, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (wtrack) HIPCHK(hipMemcpyAsync(wtrack, ctx->d.wtrack, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (windfreq) HIPCHK(hipMemcpyAsync(windfreq, ctx->d.windfreq, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

static int read_plane(smx_ctx* ctx, void* host, size_t bytes, int which) {   // (a strip context fills its range; the rest of `host` reads 0)
  if (which == 2) { FULLMAP("smx_normals") }
  if (ctx->plane_cap < bytes) { hipFree(ctx->d_plane); ctx->d_plane = nullptr; ctx->plane_cap = 0; HIPCHK(hipMalloc(&ctx->d_plane, bytes)); ctx->plane_cap = bytes; }
  void* tmp = ctx->d_plane;
  const unsigned nb = (unsigned)((ctx->lcells + 255) / 256);
  const size_t el = which == 0 ? 8 : (which == 1 ? 4 : 12);
  if (which == 0) hipLaunchKernelGGL(k_heights, dim3(nb), dim3(256), 0, ctx->stream, ctx->d, (double*)tmp);
  else if (which == 1) hipLaunchKernelGGL(k_surface, dim3(nb), dim3(256), 0, ctx->stream, ctx->d, (uint32_t*)tmp);
  else hipLaunchKernelGGL(k_normals, dim3(nb), dim3(256), 0, ctx->stream, ctx->d, (float*)tmp);
  if (ctx->partial()) memset(host, 0, bytes);
  hipError_t e = hipMemcpyAsync((char*)host + ctx->c_lo * el, tmp, ctx->lcells * el, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { ctx->err = hipGetErrorString(e); return -1; }
  return 0;
}
int smx_read_heights(smx_ctx* ctx, double* out) { return read_plane(ctx, out, ctx->ncells * 8, 0); }
int smx_read_surface(smx_ctx* ctx, uint32_t* out) { return read_plane(ctx, out, ctx->ncells * 4, 1); }
int smx_normals(smx_ctx* ctx, float* out3) { return read_plane(ctx, out3, ctx->ncells * 12, 2); }

static int fill_vertices(smx_ctx* ctx, const float* colors4, int32_t ncolors, void* out_vertices44, VCut cut, long long one_cell = -1) {
  FULLMAP("smx_fill_vertices")
  if (!colors4 || ncolors <= 0 || !out_vertices44) { ctx->err = "smx_fill_vertices: colors4/out must be given"; return -2; }
  const size_t bytes = ctx->ncells * 44;
  if (!ctx->d_vcol) { HIPCHK(hipMalloc(&ctx->d_vcol, 256 * 16)); ctx->vcol_n = -1; }
  if (ncolors > 256) ncolors = 256;
  if (ctx->vcol_n != ncolors || memcmp(ctx->vcol_host, colors4, (size_t)ncolors * 16) != 0) {   // (the colour table rarely changes: one upload, not one per call)
    HIPCHK(hipMemcpyAsync(ctx->d_vcol, colors4, (size_t)ncolors * 16, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    memcpy(ctx->vcol_host, colors4, (size_t)ncolors * 16); ctx->vcol_n = ncolors;
  }
  if (one_cell >= 0) {                                        // one column (Layermap::update(ivec2, ...)): the 256-cell block that holds it, into a block-sized scratch
    if (!ctx->d_vtx1) HIPCHK(hipMalloc(&ctx->d_vtx1, 256 * 44));
    cut.block0 = (unsigned)((size_t)one_cell / 256); cut.rel = 1u;
    hipLaunchKernelGGL(k_fill_vertices, dim3(1), dim3(256), 0, ctx->stream, ctx->d, ctx->d_vcol, (uint32_t)ncolors, ctx->d_vtx1, cut);
    HIPCHK(hipMemcpyAsync(out_vertices44, reinterpret_cast<const char*>(ctx->d_vtx1) + ((size_t)one_cell % 256) * 44, 44, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  if (!ctx->d_vtx) {                                        // whole-map scratch kept for the context's lifetime (738 MB at 4096^2)
    if (hipMalloc(&ctx->d_vtx, bytes) != hipSuccess) { ctx->d_vtx = nullptr; ctx->err = "smx_fill_vertices: out of device memory"; return -1; }
  }
  hipLaunchKernelGGL(k_fill_vertices, dim3((unsigned)((ctx->ncells + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d, ctx->d_vcol,
                     (uint32_t)ncolors, ctx->d_vtx, cut);
  HIPCHK(hipMemcpyAsync(out_vertices44, ctx->d_vtx, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}
int smx_fill_vertices(smx_ctx* ctx, const float* colors4, int32_t ncolors, void* out_vertices44) {
  return fill_vertices(ctx, colors4, ncolors, out_vertices44, VCut{-1, 0.0, 0.f});
}
int smx_fill_vertices_cut(smx_ctx* ctx, const float* colors4, int32_t ncolors, int32_t mode, double cut, void* out_vertices44) {
  VCut c;
  c.mode = mode ? 1 : 0;
  if (mode == 0) { const int sl = (int)cut; c.thr = (double)((float)sl / (float)ctx->d.scale); c.ycut = (float)sl; }   // (float)SLICE/(float)SCALE, vec3(p.x, SLICE, p.y)
  else { c.thr = cut / (double)ctx->d.scale; c.ycut = (float)cut; }                                                    // s/SCALE, vec3(p.x, s, p.y)
  return fill_vertices(ctx, colors4, ncolors, out_vertices44, c);
}
int smx_fill_vertex_cut(smx_ctx* ctx, const float* colors4, int32_t ncolors, int32_t mode, double cut, int32_t x, int32_t y, void* out_vertex44) {
  if (x < 0 || y < 0 || x >= ctx->cfg.dimx || y >= ctx->cfg.dimy) { ctx->err = "smx_fill_vertex_cut: position outside the map"; return -2; }
  VCut c;
  c.mode = mode < 0 ? -1 : (mode ? 1 : 0); c.thr = 0.0; c.ycut = 0.f;
  if (mode == 0) { const int sl = (int)cut; c.thr = (double)((float)sl / (float)ctx->d.scale); c.ycut = (float)sl; }
  else if (mode > 0) { c.thr = cut / (double)ctx->d.scale; c.ycut = (float)cut; }
  return fill_vertices(ctx, colors4, ncolors, out_vertex44, c, (long long)x * ctx->cfg.dimy + y);
}

int smx_heights_bilinear(smx_ctx* ctx, const float* pos2, int32_t n, double* out) {
  FULLMAP("smx_heights_bilinear")
  float* dpos = nullptr; double* dout = nullptr;
  HIPCHK(hipMalloc(&dpos, (size_t)n * 8)); HIPCHK(hipMalloc(&dout, (size_t)n * 8));
  HIPCHK(hipMemcpyAsync(dpos, pos2, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_bilinear, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d, dpos, n, dout);
  hipError_t e = hipMemcpyAsync(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(dpos); hipFree(dout);
  if (e != hipSuccess) { ctx->err = hipGetErrorString(e); return -1; }
  return 0;
}

// ---------------- speculative engine: host driver ----------------
static int spec_alloc(smx_ctx* ctx, uint32_t n) {          // n = particles of a sub-phase; it owns n + SPEC_MAXNEST slots
  FULLMAP("the speculative engine")
  if (ctx->spec_cap >= n && !ctx->spec_realloc) return 0;
  ctx->spec_realloc = false;
  SpecShared& sh = ctx->sh;
  if (ctx->spec_cap) {
    hipFree(sh.part); hipFree(sh.chunk_tab); hipFree(sh.arena); hipFree(sh.alloc_log); hipFree(sh.free_log);
    hipFree(sh.pending); hipFree(sh.draws); hipFree(sh.scout_cells); hipFree(sh.scout_len); hipFree(sh.ckpt);
    sh.part = nullptr; sh.chunk_tab = nullptr; sh.arena = nullptr; sh.alloc_log = nullptr; sh.free_log = nullptr;
    sh.pending = nullptr; sh.draws = nullptr; sh.scout_cells = nullptr; sh.scout_len = nullptr; sh.ckpt = nullptr;
    ctx->spec_cap = 0;                                        // a failed hipMalloc below must not leave dangling pointers
  } else {
    HIPCHK(hipMalloc(&sh.stamp, ctx->ncells * 4)); HIPCHK(hipMalloc(&sh.claim, ctx->ncells * 4));
    HIPCHK(hipMalloc(&sh.ctrl, SC_COUNT * 4));
    sh.reclaim_cap = 1u << 22; HIPCHK(hipMalloc(&sh.reclaim, (size_t)sh.reclaim_cap * 4));
    HIPCHK(hipHostMalloc(&ctx->h_ctrl, SC_COUNT * 4));
  }
  const uint32_t cap = n < 1024 ? 1024 : n;                 // particles
  // nested-particle slots of a sub-phase: 8192 on the device (21 KB of per-slot records each: 170 MB). With the 1 024 of round 3 a
  // sub-phase of 4 096 particles at the headline's density was cut -- everything in flight rolled back on one lane and run again --
  // after ~650 committed particles (profiles/r04_exact_subphases.md)
  static const uint32_t maxnest = [] { const char* e = getenv("SMX_SPEC_MAXNEST"); const int v = e ? atoi(e) : 8192; return (uint32_t)(v < 128 ? 128 : (v > 60000 ? 60000 : v)); }();
  sh.maxnest = ctx->spec_maxnest ? ctx->spec_maxnest : maxnest;
  const uint32_t slots = cap + sh.maxnest;                  // per-slot records (soil_spec.h "rand() SLOTS")
  HIPCHK(hipMalloc(&sh.part, (size_t)slots * sizeof(SpecPart)));
  HIPCHK(hipMalloc(&sh.chunk_tab, (size_t)slots * MAX_CHUNKS * 4));
  sh.arena_chunks = cap * 32u < 16384u ? 16384u : cap * 32u;
  HIPCHK(hipMalloc(&sh.arena, (size_t)sh.arena_chunks * LOG_CHUNK * sizeof(SpecLog)));
  HIPCHK(hipMalloc(&sh.alloc_log, (size_t)slots * MAX_NODELOG * 4)); HIPCHK(hipMalloc(&sh.free_log, (size_t)slots * MAX_NODELOG * 4));
  HIPCHK(hipMalloc(&sh.pending, (size_t)slots * 4));
  sh.ndraws = 2 * slots; HIPCHK(hipMalloc(&sh.draws, (size_t)sh.ndraws * 4));
  // scouted footprints (the first n + SPEC_SCOUT_MARGIN slots only): water paths are a few thousand cells, wind paths up to ~10^5
  sh.scout_cap = cap <= 2048 ? 131072u : 16384u;
  HIPCHK(hipMalloc(&sh.scout_cells, (size_t)(cap + SPEC_SCOUT_MARGIN) * sh.scout_cap * 4)); HIPCHK(hipMalloc(&sh.scout_len, (size_t)slots * 4));
  HIPCHK(hipMalloc(&sh.ckpt, (size_t)slots * sizeof(SpecCkpt)));
  sh.mode = 3u;
  ctx->spec_cap = cap;
  return 0;
}

// One SUB-PHASE of a particle phase with the speculative engine: the next `n` top-level particles of the phase, from the generator
// state and the map as they stand. It ends when all n are committed (outcome 0), when the nested particles it has committed use up
// the slots a sub-phase owns (outcome 1: everything committed so far stays, the caller goes on with a fresh sub-phase from the
// continued rand() stream -- water.h:246-264 draws nested particles from the one stream, which is why the slots are finite), or
// when the speculation cannot go on (outcome 2: no pool headroom, stalled prefix -- the caller hands the rest to the serial walker).
// Results are bit-identical whichever way a phase is cut (tests/test_gpu_parity.py, tests/test_hostsim.py).
static int spec_subphase(smx_ctx* ctx, int32_t n, bool wind, uint32_t* committed, int* outcome) {
  // engine mode per phase: water = blocked-filter + suspend/resume, wind = suspend/resume only (long wind particles
  // are nearly always "blocked somewhere": better to let them run up to the contested cell) -- host-sim measurements
  static const uint32_t mode_water = [] { const char* e = getenv("SMX_SPEC_MODE"); return e ? (uint32_t)atoi(e) : 3u; }();
  static const uint32_t mode_wind = [] { const char* e = getenv("SMX_SPEC_MODE_WIND"); return e ? (uint32_t)atoi(e) : 2u; }();
  ctx->sh.mode = wind ? mode_wind : mode_water;
  const SpecShared& sh = ctx->sh;
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemsetAsync(sh.stamp, 0, ctx->ncells * 4, st));
  hipLaunchKernelGGL(k_spec_begin, dim3(1), dim3(256), 0, st, ctx->d, sh, (uint32_t)n);
  const uint32_t L = (uint32_t)spec_lanes();
  static const bool wind_scout = [] { const char* e = getenv("SMX_WIND_SCOUT"); return e ? atoi(e) != 0 : true; }();
  static const bool coop_scout = [] { const char* e = getenv("SMX_COOP"); return e ? atoi(e) != 0 : true; }();
  const uint32_t ns = (uint32_t)n + SPEC_SCOUT_MARGIN;       // scouted slots
  if (coop_scout) {
    if (!wind) hipLaunchKernelGGL(k_spec_scout_coop<false>, dim3(ns), dim3(64), 0, st, ctx->d, sh);
    else if (wind_scout) hipLaunchKernelGGL(k_spec_scout_coop<true>, dim3(ns), dim3(64), 0, st, ctx->d, sh);
  } else {
    if (!wind) hipLaunchKernelGGL(k_spec_scout<false>, dim3((ns + L - 1) / L), dim3(64), 0, st, ctx->d, sh, L);
    else if (wind_scout) hipLaunchKernelGGL(k_spec_scout<true>, dim3((ns + L - 1) / L), dim3(64), 0, st, ctx->d, sh, L);
  }
  uint32_t np = (uint32_t)n > SPEC_WINDOW ? SPEC_WINDOW : (uint32_t)n, last_base = 0, stall = 0;
  *outcome = 0;
  static const bool trace_rounds = [] { const char* e = getenv("SMX_TRACE_ROUNDS"); return e && atoi(e) != 0; }();
  uint32_t prev_exec = 0, prev_sum = 0;
  if (trace_rounds) HIPCHK(hipStreamSynchronize(st));
  auto t_round = std::chrono::steady_clock::now();
  for (int round = 0;; round++) {
    HIPCHK(hipMemsetAsync(sh.claim, 0xFF, ctx->ncells * 4, st));
    hipLaunchKernelGGL(k_spec_claim, dim3(np), dim3(64), 0, st, sh);
    if (sh.mode & 1u) hipLaunchKernelGGL(k_spec_select, dim3(np), dim3(64), 0, st, sh);
    {
      PhaseTimer tk(ctx, wind ? 5 : 4);
      static const bool coop = [] { const char* e = getenv("SMX_COOP"); return e ? atoi(e) != 0 : true; }();
      if (coop) {
        if (wind) hipLaunchKernelGGL(k_spec_exec_coop<true>, dim3(np), dim3(64), 0, st, ctx->d, sh);
        else hipLaunchKernelGGL(k_spec_exec_coop<false>, dim3(np), dim3(64), 0, st, ctx->d, sh);
      } else {
        if (wind) hipLaunchKernelGGL(k_spec_exec<true>, dim3((np + L - 1) / L), dim3(64), 0, st, ctx->d, sh, L);
        else hipLaunchKernelGGL(k_spec_exec<false>, dim3((np + L - 1) / L), dim3(64), 0, st, ctx->d, sh, L);
      }
    }
    hipLaunchKernelGGL(k_spec_boundary, dim3(1), dim3(1024), 0, st, ctx->d, sh);
    HIPCHK(hipMemcpyAsync(ctx->h_ctrl, sh.ctrl, SC_COUNT * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const uint32_t* c = ctx->h_ctrl;
    if (trace_rounds) {                                                   // SMX_TRACE_ROUNDS=1: one line per round on stderr
      const auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "[soilmx] %s round %3d: %7.3f ms  pending-in %4u  base %4u  executed %5u  longest %6u steps  aborts %u kills %u suspends %u\n",
              wind ? "wind " : "water", round, std::chrono::duration<double, std::milli>(now - t_round).count(), np, c[SC_BASE],
              c[SC_EXECUTED] - prev_exec, c[SC_SUM_MAXSTEPS] - prev_sum, c[SC_ABORTS], c[SC_KILLS], c[SC_SUSPENDS]);
      prev_exec = c[SC_EXECUTED]; prev_sum = c[SC_SUM_MAXSTEPS]; t_round = now;
    }
    if (c[SC_N] == 0) { *outcome = 2; break; }                          // no pool headroom: nothing was executed
    if (c[SC_TOP] == (uint32_t)n) break;                                // n top-level particles committed
    np = c[SC_NPENDING];
    if (c[SC_BASE] == last_base) stall++; else { stall = 0; last_base = c[SC_BASE]; }
    if (c[SC_NESTED_COMMITTED] + 64 > sh.maxnest) { *outcome = 1; break; }   // the sub-phase's nested slots are used up: commit, re-arm
    if (np == 0 || stall >= SPEC_STALL_LIMIT || round > 100000) { *outcome = 2; break; }
  }
  const uint32_t* c = ctx->h_ctrl;
  ctx->spec_rounds += c[SC_ROUNDS]; ctx->spec_aborts += c[SC_ABORTS]; ctx->spec_kills += c[SC_KILLS]; ctx->spec_executed += c[SC_EXECUTED];
  *committed = (c[SC_N] == 0) ? 0u : c[SC_TOP];                        // top-level particles the speculation committed
  if (*outcome != 0) {
    if (c[SC_N] != 0) {
      hipLaunchKernelGGL(k_spec_abandon, dim3(1), dim3(64), 0, st, ctx->d, sh);
      hipLaunchKernelGGL(k_spec_end, dim3(1), dim3(256), 0, st, ctx->d, sh, 0u);
    }
  } else {
    hipLaunchKernelGGL(k_spec_end, dim3(1), dim3(256), 0, st, ctx->d, sh, 1u);
  }
  HIPCHK(hipGetLastError());
  {
    uint32_t crossed = 0;
    HIPCHK(hipMemcpyAsync(&crossed, sh.ctrl + SC_POOL_CROSSED, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (crossed) { ctx->err = "speculative phase ran past pool_capacity (pool.get() would have failed inside it): raise pool_capacity"; return -6; }
  }
  if (c[SC_RECLAIM_LOST] != 0) { ctx->err = "speculative phase: reclaim list overflow (pool nodes leaked)"; return -6; }
  return 0;
}
// One particle phase (SoilMachine.cpp:287-298 or :303-307) with the speculative engine, in sub-phases of at most SPEC_SUB top-level
// particles (round 4; round 3 gave the WHOLE rest of a phase to the one-lane serial walker as soon as 1 024 nested particles had
// committed: 0.4 M steps/s on the headline workload). Only what the speculation cannot do at all goes to the serial walker.
static int spec_phase(smx_ctx* ctx, int32_t n, bool wind) {
  if (n <= 0) return 0;
  static const uint32_t SPEC_SUB_ENV = [] { const char* e = getenv("SMX_SPEC_SUB"); const int v = e ? atoi(e) : 4096; return (uint32_t)(v < 64 ? 64 : v); }();
  const uint32_t SPEC_SUB = ctx->spec_sub ? ctx->spec_sub : SPEC_SUB_ENV;
  if (spec_alloc(ctx, std::min<uint32_t>((uint32_t)n, SPEC_SUB))) return -1;
  hipStream_t st = ctx->stream;
  uint32_t remaining = (uint32_t)n;
  auto serial = [&](uint32_t k) {                            // the reference's order on one lane, from the state as it stands
    PhaseTimer tk(ctx, wind ? 5 : 4);
    if (wind) hipLaunchKernelGGL(k_wind_serial, dim3(1), dim3(64), 0, st, ctx->d, (int)k);
    else hipLaunchKernelGGL(k_water_serial, dim3(1), dim3(64), 0, st, ctx->d, (int)k);
  };
  while (remaining > 0) {
    const uint32_t m = std::min(remaining, SPEC_SUB);
    uint32_t committed = 0; int outcome = 0;
    const int rc = spec_subphase(ctx, (int32_t)m, wind, &committed, &outcome);
    if (rc) return rc;
    remaining -= committed;
    if (outcome == 0) continue;
    ctx->spec_subphases_cut++;
    if (outcome == 2) { ctx->spec_fallbacks++; serial(remaining); remaining = 0; }
    else if (committed == 0) { ctx->spec_serial_particles++; serial(1u); remaining -= 1u; }   // one particle with more nested particles than a sub-phase has slots
  }
  HIPCHK(hipGetLastError());
  return 0;
}


static void launch_classify(smx_ctx* ctx);
// ---------------- batched engine: host driver (mirrored for the CPU by tests/hostsim/hostsim.cpp batch_generations) ----------------
static int batch_alloc(smx_ctx* ctx, uint32_t nslots, bool wind) {
  BatchShared& bs = ctx->bs;
  if (!bs.ctrl) {
    bs.tshift = -1; bs.ntx = ctx->cfg.dimx; bs.nty = ctx->cfg.dimy; bs.txb = 0; bs.txe = ctx->cfg.dimx;
    const size_t nt = ctx->lcells + 2 * (size_t)ctx->cfg.dimy;   // one claim word per held cell (+ a row of slack for tiles cut by the range's edges): enough for every granularity
    HIPCHK(hipMalloc(&bs.claim[0], nt * 8)); HIPCHK(hipMalloc(&bs.claim[1], nt * 8));
    HIPCHK(hipMalloc(&bs.ctrl, BC_COUNT * 4)); HIPCHK(hipMemsetAsync(bs.ctrl, 0, BC_COUNT * 4, ctx->stream));
    // nested particles of one generation / pool nodes released in one chunk: sized by the map (a grid pass over a map full
    // of lakes emits children for a sizeable part of its cells); running out is counted (children_lost) / an error (freed)
    bs.child_cap = (uint32_t)std::max<size_t>((size_t)1 << 20, ctx->lcells / 2); HIPCHK(hipMalloc(&bs.children, (size_t)bs.child_cap * sizeof(BChild)));
    bs.freed_cap = (uint32_t)std::max<size_t>((size_t)1 << 22, ctx->lcells / 2); HIPCHK(hipMalloc(&bs.freed, (size_t)bs.freed_cap * 4));
    {
      const size_t ngt = (size_t)((ctx->cfg.dimx + 3) >> 2) * ((ctx->cfg.dimy + 3) >> 2);
      HIPCHK(hipMalloc(&bs.dirty, ngt)); HIPCHK(hipMemsetAsync(bs.dirty, 0, ngt, ctx->stream));
    }
    HIPCHK(hipHostMalloc(&ctx->h_bctrl, BC_COUNT * 4));
  }
  if (ctx->batch_cap < nslots) {
    hipFree(bs.water); hipFree(bs.wind); hipFree(bs.live); bs.water = nullptr; bs.wind = nullptr; bs.live = nullptr; ctx->batch_cap = 0;
    const uint32_t cap = nslots < 4096u ? 4096u : nslots;
    HIPCHK(hipMalloc(&bs.water, (size_t)cap * sizeof(BWater))); HIPCHK(hipMalloc(&bs.wind, (size_t)cap * sizeof(BWind)));
    HIPCHK(hipMalloc(&bs.live, (size_t)cap * 4));
    ctx->batch_cap = cap;
  }
  if (ctx->cfg.engine == SMX_ENGINE_RELAXED && ctx->relax_cap < ctx->batch_cap) {   // request entries (2 per step), per-epoch flood and cascade lists
    hipFree(bs.req); hipFree(bs.rnext); hipFree(bs.rcount); hipFree(bs.flist[0]); hipFree(bs.flist[1]); hipFree(bs.clist[0]); hipFree(bs.clist[1]);
    hipFree(bs.flagl[0]); hipFree(bs.flagl[1]); bs.flagl[0] = bs.flagl[1] = nullptr;
    bs.req = nullptr; bs.rnext = nullptr; bs.rcount = nullptr; bs.flist[0] = bs.flist[1] = bs.clist[0] = bs.clist[1] = nullptr; ctx->relax_cap = 0;
    const uint32_t cap = ctx->batch_cap;
    const size_t ent = (size_t)cap * 2 * R_MAXSTEPS;
    HIPCHK(hipMalloc(&bs.req, ent * sizeof(RReq))); HIPCHK(hipMalloc(&bs.rnext, ent * 4)); HIPCHK(hipMalloc(&bs.rcount, (size_t)cap * 4));
    HIPCHK(hipMemsetAsync(bs.rcount, 0, (size_t)cap * 4, ctx->stream));
    bs.list_cap = cap; bs.flag_cap = (uint32_t)ent; bs.clist_cap = (uint32_t)std::max<size_t>(4096, ent / 2);
    for (int k = 0; k < 2; k++) {
      HIPCHK(hipMalloc(&bs.flist[k], (size_t)bs.list_cap * 4)); HIPCHK(hipMalloc(&bs.flagl[k], (size_t)bs.flag_cap * 4));
      HIPCHK(hipMalloc(&bs.clist[k], (size_t)9 * bs.clist_cap * 4));
    }
    ctx->relax_cap = cap;
  }
  if (ctx->cfg.engine == SMX_ENGINE_RELAXED && !bs.rhead) {   // per-cell request lists and cascade flags (8 B / cell), empty between epochs
    HIPCHK(hipMalloc(&ctx->rhead_alloc, ctx->lcells * 4)); HIPCHK(hipMalloc(&ctx->cflag_alloc, ctx->lcells * 4));
    HIPCHK(hipMemsetAsync(ctx->rhead_alloc, 0xFF, ctx->lcells * 4, ctx->stream)); HIPCHK(hipMemsetAsync(ctx->cflag_alloc, 0, ctx->lcells * 4, ctx->stream));
    HIPCHK(hipMalloc(&ctx->cstate_alloc, ctx->lcells * 4)); HIPCHK(hipMemsetAsync(ctx->cstate_alloc, 0, ctx->lcells * 4, ctx->stream));
    bs.rhead = ctx->rhead_alloc - ctx->c_lo; bs.cflag = ctx->cflag_alloc - ctx->c_lo; bs.cstate = ctx->cstate_alloc - ctx->c_lo;   // (global cell index, held range only)
  }
  if (ctx->draws_cap < 2u * nslots) {
    hipFree(ctx->d_draws); if (ctx->h_draws) hipHostFree(ctx->h_draws); ctx->d_draws = nullptr; ctx->h_draws = nullptr; ctx->draws_cap = 0;
    const uint32_t cap = 2u * (nslots < 4096u ? 4096u : nslots);
    HIPCHK(hipMalloc(&ctx->d_draws, (size_t)cap * 4)); HIPCHK(hipHostMalloc(&ctx->h_draws, (size_t)cap * 4));
    ctx->draws_cap = cap;
  }
  bs.draws = ctx->d_draws; bs.dilate = ctx->batch_dilate; bs.maxsteps = 1;
  bs.strips = batch_make_strips(ctx->cfg.dimx, ctx->strips_n, ctx->strips_a, ctx->strips_b); bs.sphase = STRIP_INTERIOR;
  bs.own_x0 = ctx->own_x0; bs.own_x1 = ctx->own_x1 < 0 ? ctx->cfg.dimx : ctx->own_x1;
  bs.birth_n = 0u; bs.birth_gap = 0u;
  if (!wind && ctx->stagger_phase_n) {                       // staggered generations (smx_set_water_stagger): slot i of the phase's top-level particles is born in epoch (i / birth_n) * gap
    const uint32_t k = (uint32_t)(ctx->water_generations < 1 ? 1 : ctx->water_generations);
    bs.birth_n = (ctx->stagger_phase_n + k - 1u) / k; bs.birth_gap = (uint32_t)ctx->water_stagger;
  }
  return 0;
}
// generations of one phase: `kids` = children to start from (grid pass), else n top-level particles
// every slot whose particle waits for its flood joins the flood list of `parity` (column strips: particles arrive and leave between chunks)
__global__ void __launch_bounds__(256) k_relax_rebuild_floods(BatchShared bs, uint32_t nslots, uint32_t parity) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  relax_flood_later(bs, i, parity, i < nslots && bs.water[i].state == B_ENDED);
}
// Children of a generation -> the next generation's slots, in key order (the order decides slot numbers = priorities and rand() draws),
// without leaving the device: keys are unique ((parent << 8) | sequence), so any sort gives the one order. Round 3 copied the records to
// the host, std::sort-ed them there and copied them back: 5-6 ms of an idle device per generation once a tick spawns 10^6 nested
// particles (profiles/r04_relaxed.md).
__global__ void __launch_bounds__(256) k_child_keys(const BChild* __restrict__ ch, uint32_t n, unsigned long long* __restrict__ keys, uint32_t* __restrict__ idx) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) { keys[i] = ch[i].key; idx[i] = i; }
}
__global__ void __launch_bounds__(256) k_child_gather(const BChild* __restrict__ ch, const uint32_t* __restrict__ idx, uint32_t n, BChild* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) out[i] = ch[idx[i]];
}
static int sort_children(smx_ctx* ctx, uint32_t nc) {       // bs.children[0, nc) -> ctx->d_kids[0, nc), ascending key; asynchronous on the context's stream
  hipStream_t st = ctx->stream;
  if (nc == 0) return 0;
  if (ctx->kids_cap < nc) { hipFree(ctx->d_kids); ctx->d_kids = nullptr; ctx->kids_cap = 0; const uint32_t cap = nc + nc / 2; HIPCHK(hipMalloc(&ctx->d_kids, (size_t)cap * sizeof(BChild))); ctx->kids_cap = cap; }
  if (ctx->sort_cap < nc) {
    for (int k = 0; k < 2; k++) { hipFree(ctx->d_skeys[k]); hipFree(ctx->d_sidx[k]); ctx->d_skeys[k] = nullptr; ctx->d_sidx[k] = nullptr; }
    hipFree(ctx->d_stemp); ctx->d_stemp = nullptr; ctx->stemp_bytes = 0; ctx->sort_cap = 0;
    const uint32_t cap = nc + nc / 2;
    for (int k = 0; k < 2; k++) { HIPCHK(hipMalloc(&ctx->d_skeys[k], (size_t)cap * 8)); HIPCHK(hipMalloc(&ctx->d_sidx[k], (size_t)cap * 4)); }
    size_t tb = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, tb, ctx->d_skeys[0], ctx->d_skeys[1], ctx->d_sidx[0], ctx->d_sidx[1], (size_t)cap, 0u, 64u, st));
    HIPCHK(hipMalloc(&ctx->d_stemp, tb)); ctx->stemp_bytes = tb; ctx->sort_cap = cap;
  }
  const unsigned nb = (nc + 255u) / 256u;
  hipLaunchKernelGGL(k_child_keys, dim3(nb), dim3(256), 0, st, ctx->bs.children, nc, ctx->d_skeys[0], ctx->d_sidx[0]);
  size_t tb = ctx->stemp_bytes;
  HIPCHK(rocprim::radix_sort_pairs(ctx->d_stemp, tb, ctx->d_skeys[0], ctx->d_skeys[1], ctx->d_sidx[0], ctx->d_sidx[1], (size_t)nc, 0u, 64u, st));
  hipLaunchKernelGGL(k_child_gather, dim3(nb), dim3(256), 0, st, ctx->bs.children, ctx->d_sidx[1], nc, ctx->d_kids);
  return 0;
}
// `nepochs` dense epochs of the running relaxed water generation as one persistent cooperative launch (k_relax_epochs). 0 = launched.
static int relax_epochs_launch(smx_ctx* ctx, uint32_t nlive, uint32_t epoch, uint32_t nepochs) {
  hipStream_t st = ctx->stream;
  if (!ctx->d_gbar) {
    if (hipMalloc(&ctx->d_gbar, sizeof(GridBar)) != hipSuccess) { ctx->mega_off = true; return 1; }
    hipMemsetAsync(ctx->d_gbar, 0, sizeof(GridBar), st);
  }
  if (!ctx->mega_blocks) {
    int dev = 0, ncu = 0, coop = 0, per_cu = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_relax_epochs, 256, 0) != hipSuccess) per_cu = 0;
    static const int want = [] { const char* e = getenv("SMX_RELAX_MEGA_BLOCKS"); return e ? atoi(e) : 0; }();   // (0: one workgroup per CU)
    int nb = want > 0 ? want : ncu;
    if (nb > ncu * per_cu) nb = ncu * per_cu;
    if (!coop || nb < 1) {
      fprintf(stderr, "[soilmx] cooperative launch unavailable (attribute %d, %d CUs x %d workgroups): the relaxed epochs run as per-phase launches\n", coop, ncu, per_cu);
      ctx->mega_off = true; return 1;
    }
    ctx->mega_blocks = nb;
  }
  BatchShared bs = ctx->bs;
  uint32_t tag0 = ctx->relax_tag + 1u;
  GridBar* bar = ctx->d_gbar;
  static const uint32_t dense_waves_env = [] { const char* e = getenv("SMX_RELAX_MEGA_WAVES"); return e ? (uint32_t)atoi(e) : 0u; }();   // (0: one wavefront per workgroup)
  static const uint32_t flags_env = [] { const char* e = getenv("SMX_RELAX_MEGA_FLAGS"); return e ? (uint32_t)atoi(e) : 0u; }();          // (1: every thread fences at a barrier)
  uint32_t dense_waves = dense_waves_env ? dense_waves_env : (uint32_t)ctx->mega_blocks, flags = flags_env;
  void* args[] = {(void*)&ctx->d, (void*)&bs, (void*)&nlive, (void*)&epoch, (void*)&nepochs, (void*)&tag0, (void*)&bar, (void*)&dense_waves, (void*)&flags};
  PhaseTimer tk(ctx, 10, nepochs);
  const hipError_t e = hipLaunchCooperativeKernel((const void*)k_relax_epochs, dim3((unsigned)ctx->mega_blocks), dim3(256), args, 0, st);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    fprintf(stderr, "[soilmx] hipLaunchCooperativeKernel(k_relax_epochs, %d workgroups): %s -- the relaxed epochs run as per-phase launches\n", ctx->mega_blocks, hipGetErrorString(e));
    ctx->mega_off = true; return 1;
  }
  return 0;
}
// One CHUNK of a particle phase on the context's current generation: the schedule for `nlive_sched` running particles (all strips
// together), fresh reservations where the schedule asks for them, then the chunk's epochs over the `nlive` slots of bs.live.
// Shared by the single-context driver (batch_generations) and the step-wise one of the column strips (smx_d_chunk).
static int run_chunk(smx_ctx* ctx, bool wind, uint32_t nlive_sched, uint32_t nlive, uint32_t nslots, bool first, bool relax_gen, uint32_t& epoch, uint32_t& chunk,
                     int* epochs_run) {
  hipStream_t st = ctx->stream;
  const BatchShared& bs = ctx->bs;
  static const int CHUNK = [] { const char* e = getenv("SMX_BATCH_CHUNK"); int v = e ? atoi(e) : B_CHUNK; return v < 1 ? 1 : v; }();
  BSched sc = batch_schedule(wind, nlive_sched, ctx->batch_dilate);
  {   // experiments only: SMX_BATCH_SCHED="tshift,dilate,maxsteps" (water) / SMX_BATCH_SCHED_WIND pins the schedule
    static const char* ew = getenv("SMX_BATCH_SCHED"); static const char* ed = getenv("SMX_BATCH_SCHED_WIND");
    const char* e = wind ? ed : ew; int a, d, k;
    if (e && sscanf(e, "%d,%d,%d", &a, &d, &k) == 3) { sc.tshift = a; sc.dilate = d; sc.maxsteps = k; }
    static const char* etw = getenv("SMX_BATCH_SCHED_TAIL"); static const char* etd = getenv("SMX_BATCH_SCHED_TAIL_WIND");   // "thr:tshift,dilate,maxsteps[;thr2:...]" (ascending thr): first thr >= nlive wins
    const char* et = wind ? etd : etw;
    if (et) {
      const char* q = et; unsigned thr;
      while (q && sscanf(q, "%u:%d,%d,%d", &thr, &a, &d, &k) == 4) {
        if (nlive_sched <= thr) { sc.tshift = a; sc.dilate = d; sc.maxsteps = k; break; }
        q = strchr(q, ';'); if (q) q++;
      }
    }
  }
  const bool strips = bs.strips.n > 1;
  // relaxed schedule (SMX_ENGINE_RELAXED): water always; wind while many particles run -- its tail (a few thousand particles that
  // fly on for up to ~13 000 steps) is better served by the exclusive schedule's 32 steps per launch. One switch per generation.
  const bool relaxed_now = relax_gen && (!wind || nlive_sched > ctx->relax_wind_min);
  bool regrain = sc.tshift != bs.tshift;                   // first chunk, or the claim granularity changes
  if (relax_gen && !relaxed_now && ctx->bs.relaxed) { ctx->bs.relaxed = 0; regrain = true; }   // hand the survivors to the exclusive schedule: fresh reservations
  batch_apply_schedule(ctx->bs, ctx->d, sc);               // (bs aliases ctx->bs: passed by value per launch)
  ctx->bs.sphase = strips ? (int)(chunk & 1u) : STRIP_INTERIOR;   // column strips: INTERIOR and SEAM chunks alternate
  if (strips && chunk > 0) epoch++;                        // (reservations of the previous chunk must not look current)
  chunk++;
  if (regrain) {                                           // (the planes hold the tile columns of the context's range)
    const size_t nt = (size_t)(((ctx->d.x_hi - 1) >> bs.tshift) - (ctx->d.x_lo >> bs.tshift) + 1) * bs.nty;
    HIPCHK(hipMemsetAsync(bs.claim[0], 0, nt * 8, st)); HIPCHK(hipMemsetAsync(bs.claim[1], 0, nt * 8, st));
  }
  if ((regrain || strips) && nlive) {                      // under strips every chunk starts with fresh reservations
    if (wind) hipLaunchKernelGGL(k_batch_reclaim<true>, dim3((nlive + 63u) / 64u), dim3(64), 0, st, ctx->d, bs, nlive, epoch);
    else hipLaunchKernelGGL(k_batch_reclaim<false>, dim3((nlive + 63u) / 64u), dim3(64), 0, st, ctx->d, bs, nlive, epoch);
  }
  int chunk_epochs = CHUNK;
  if (relaxed_now) {
    // relaxed schedule: per epoch one step of every running particle, the queued edits per cell, the nine cascade colours,
    // then one flood epoch behind claim tiles. Nested generations live a few steps: shorter chunks there (without strips the
    // result does not depend on where a generation is cut into chunks; with strips the chunk is the unit of the exchange).
    static const int RCHUNK_KIDS = [] { const char* e = getenv("SMX_RELAX_CHUNK_KIDS"); int v = e ? atoi(e) : 8; return v < 1 ? 1 : v; }();
    static const bool use_tail = [] { const char* e = getenv("SMX_RELAX_TAIL"); return e ? atoi(e) != 0 : true; }();
    static const uint32_t tail_at_env = [] { const char* e = getenv("SMX_RELAX_TAIL_AT"); return e ? (uint32_t)atoi(e) : RELAX_TAIL; }();
    const uint32_t tail_at = ctx->relax_tail_at >= 0 ? (uint32_t)ctx->relax_tail_at : tail_at_env;
    const bool tail = !wind && use_tail && nlive <= tail_at;
    static const bool use_mega_env = [] { const char* e = getenv("SMX_RELAX_MEGA"); return e ? atoi(e) != 0 : false; }();   // (off by default: measured slower, see k_relax_epochs)
    const bool use_mega = ctx->relax_persistent >= 0 ? ctx->relax_persistent != 0 : use_mega_env;
    static const int mega_chunk = [] { const char* e = getenv("SMX_RELAX_MEGA_CHUNK"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : v; }();   // (0: the chunk lengths of the per-phase launches)
    const bool mega = !wind && !tail && use_mega && !ctx->mega_off && bs.cstate != nullptr;
    chunk_epochs = strips ? B_CHUNK : (tail ? 64 : (mega && mega_chunk ? mega_chunk : (first ? CHUNK : RCHUNK_KIDS)));
    ctx->bs.rstride = wind ? 2 * ctx->relax_wind_steps : std::max(2, ctx->relax_water_steps); ctx->bs.rsteps = wind ? ctx->relax_wind_steps : ctx->relax_water_steps;
    if (ctx->relax_tag > 0x7FFF0000u) { ctx->relax_tag = 0; HIPCHK(hipMemsetAsync(ctx->cflag_alloc, 0, ctx->lcells * 4, st)); HIPCHK(hipMemsetAsync(ctx->cstate_alloc, 0, ctx->lcells * 4, st)); }   // (tags never repeat; 2 * tag + 1 fits 32 bits)
    if (strips) HIPCHK(hipMemsetAsync(bs.ctrl + BC_NFLOOD, 0, (BC_LISTS_END - BC_NFLOOD) * 4, st));   // (the epoch skipped between two chunks breaks the lists' parity hand-over)
    if (strips && !wind) {                                   // who waits for a flood on this device right now?
      if (nslots) hipLaunchKernelGGL(k_relax_rebuild_floods, dim3((nslots + 255u) / 256u), dim3(256), 0, st, bs, nslots, epoch & 1u);
    }
    if (nlive == 0) { epoch += (uint32_t)chunk_epochs; ctx->relax_tag += (uint32_t)chunk_epochs; }   // (a strip with nobody home this chunk)
    else if (tail) {                                         // few particles left: whole epochs inside one workgroup (k_relax_tail)
      PhaseTimer tk(ctx, 9, (uint32_t)chunk_epochs);
      hipLaunchKernelGGL(k_relax_tail, dim3(1), dim3(256), 0, st, ctx->d, bs, nlive, epoch, (uint32_t)chunk_epochs, ctx->relax_tag + 1u);
      ctx->relax_tag += (uint32_t)chunk_epochs; epoch += (uint32_t)chunk_epochs;
    } else if (mega && relax_epochs_launch(ctx, nlive, epoch, (uint32_t)chunk_epochs) == 0) {   // the chunk's dense epochs as ONE persistent launch (k_relax_epochs)
      ctx->relax_tag += (uint32_t)chunk_epochs; epoch += (uint32_t)chunk_epochs;
    } else {
      const uint32_t lpw = batch_lanes_per_wave(nlive);
      const unsigned nbl = (nlive + lpw - 1u) / lpw, nb64 = (nlive + 63u) / 64u;
      const unsigned ncb = (unsigned)(((size_t)nlive * (size_t)(wind ? ctx->bs.rstride : ctx->bs.rsteps) + 63u) / 64u);   // worst case of the flagged / one cascade list
      static const unsigned fb_min = [] { const char* e = getenv("SMX_RELAX_FLOOD_BLOCKS"); return e ? (unsigned)atoi(e) : 1024u; }();
      const unsigned nfb = std::max(fb_min, (unsigned)((nslots + 63u) / 64u));   // floods: at least 1024 wavefronts to spread over
      for (int k = 0; k < chunk_epochs; k++, epoch++) {
        ctx->bs.rtag = ++ctx->relax_tag;
        {
          // the step kernel between its own pair of events -- every `sample`-th launch (an event pair costs a few microseconds of an epoch
          // that lasts a few hundred; the average launch duration bench.py reports is over the bracketed launches only)
          static const int sample = [] { const char* e = getenv("SMX_STEP_EVENT_SAMPLE"); const int v = e ? atoi(e) : 8; return v < 0 ? 0 : v; }();
          const bool timed = sample > 0 && (k % sample) == 0;
          if (wind) ctx->timing.launches_step_wind++; else ctx->timing.launches_step_water++;   // (every launch, bracketed or not)
          if (timed) { PhaseTimer tk(ctx, wind ? 5 : 4, 1u);
            if (wind) hipLaunchKernelGGL(k_relax_step<true>, dim3(nbl), dim3(64), 0, st, ctx->d, bs, nlive, epoch, lpw);
            else hipLaunchKernelGGL(k_relax_step<false>, dim3(nbl), dim3(64), 0, st, ctx->d, bs, nlive, epoch, lpw);
          } else if (wind) hipLaunchKernelGGL(k_relax_step<true>, dim3(nbl), dim3(64), 0, st, ctx->d, bs, nlive, epoch, lpw);
          else hipLaunchKernelGGL(k_relax_step<false>, dim3(nbl), dim3(64), 0, st, ctx->d, bs, nlive, epoch, lpw);
        }
        hipLaunchKernelGGL(k_relax_apply, dim3(nb64), dim3(64), 0, st, ctx->d, bs, nlive, epoch);
        hipLaunchKernelGGL(k_relax_filter, dim3(ncb), dim3(64), 0, st, ctx->d, bs, epoch);
        static const unsigned casc_blocks = [] { const char* e = getenv("SMX_RELAX_CASC_BLOCKS"); return e ? (unsigned)atoi(e) : 64u; }();
        static const unsigned casc_flow = [] { const char* e = getenv("SMX_RELAX_CASC_FLOW"); return e ? (unsigned)atoi(e) : 1024u; }();   // 0: nine launches (round 4)
        if (casc_flow && bs.cstate) hipLaunchKernelGGL(k_relax_cascade_flow, dim3(std::min(ncb, casc_flow)), dim3(64), 0, st, ctx->d, bs, epoch, spin_budget(1u << 24));
        else {
          const unsigned ncc = std::min(ncb, std::max(1u, casc_blocks));
          for (uint32_t col = 0; col < 9u; col++) hipLaunchKernelGGL(k_relax_cascade, dim3(ncc), dim3(64), 0, st, ctx->d, bs, epoch, col);
        }
        if (!wind) {                                         // (bracketed every 7th launch of the context: a stride coprime to the every-4th-epoch rule of relax_floods_due)
          ctx->timing.launches_floods_all++;
          static const uint32_t fprof = [] { const char* e = getenv("SMX_FLOOD_PROF"); return e ? (uint32_t)atoi(e) : 0u; }();
          if (ctx->flood_launch_no++ % 7u == 0u) { PhaseTimer tk(ctx, 12, 1u); hipLaunchKernelGGL(k_relax_floods, dim3(nfb), dim3(64), 0, st, ctx->d, bs, epoch, fprof); }
          else hipLaunchKernelGGL(k_relax_floods, dim3(nfb), dim3(64), 0, st, ctx->d, bs, epoch, fprof);
        }
      }
    }
  } else if (nlive) {
    PhaseTimer tk(ctx, wind ? 5 : 4, (uint32_t)CHUNK);   // CHUNK back-to-back launches of the epoch kernel, nothing else
    if (wind) ctx->timing.launches_step_wind += (uint64_t)CHUNK; else ctx->timing.launches_step_water += (uint64_t)CHUNK;
    const uint32_t lpw = batch_lanes_per_wave(nlive);
    const unsigned nbl = (nlive + lpw - 1u) / lpw;
    // more wavefronts than SIMDs (> 65 536 running particles): the 256-register build, two wavefronts per SIMD -- +5 % at 16384^2,
    // nothing below (profiles/r02_batched_schedules.md); results do not depend on it
    static const uint32_t occ2_above = [] { const char* e = getenv("SMX_BATCH_OCC2_ABOVE"); return e ? (uint32_t)atoll(e) : 65536u; }();
    const bool occ2 = nlive > occ2_above;
    for (int k = 0; k < CHUNK; k++, epoch++) {
      if (occ2) {
        if (wind) hipLaunchKernelGGL(k_batch_epoch2<true>, dim3(nbl), dim3(64), 0, st, ctx->d, bs, nlive, epoch, lpw);
        else hipLaunchKernelGGL(k_batch_epoch2<false>, dim3(nbl), dim3(64), 0, st, ctx->d, bs, nlive, epoch, lpw);
      } else if (wind) hipLaunchKernelGGL(k_batch_epoch<true>, dim3(nbl), dim3(64), 0, st, ctx->d, bs, nlive, epoch, lpw);
      else hipLaunchKernelGGL(k_batch_epoch<false>, dim3(nbl), dim3(64), 0, st, ctx->d, bs, nlive, epoch, lpw);
    }
  } else epoch += (uint32_t)CHUNK;
  hipLaunchKernelGGL(k_batch_merge_freed, dim3(1), dim3(256), 0, st, ctx->d, bs);   // frees of the chunk go back on the stack
  if (epochs_run) *epochs_run = chunk_epochs;
  return 0;
}
// generations of one phase: n top-level particles, or (nkids > 0) the children the grid pass left, already sorted into ctx->d_kids
static int batch_generations(smx_ctx* ctx, bool wind, uint32_t n, uint32_t nkids) {
  hipStream_t st = ctx->stream;
  bool first = nkids == 0;
  uint32_t nslots = first ? n : nkids;
  if (nslots == 0) return 0;
  RandState r;                                              // the host draws for the device: glibc rand() is a serial recurrence
  const bool draws = first;                                 // (suspended nested particles carry their state: no draws, no round trip)
  if (draws) {
    HIPCHK(hipMemcpyAsync(&r, ctx->d.rnd, sizeof(r), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  while (nslots > 0) {
    if (batch_alloc(ctx, nslots, wind)) return -1;
    const BatchShared& bs = ctx->bs;
    if (first) {                                              // (suspended nested particles carry their state: no draws)
      for (uint32_t i = 0; i < 2u * nslots; i++) {
        const uint32_t v = r.ring[r.idx % 31] + r.ring[(r.idx - 3) % 31];
        r.ring[r.idx % 31] = v; r.idx++; r.calls++;
        ctx->h_draws[i] = v >> 1;
      }
      HIPCHK(hipMemcpyAsync(ctx->d_draws, ctx->h_draws, (size_t)2u * nslots * 4, hipMemcpyHostToDevice, st));
    }
    ctx->bs.tshift = -1;                                      // (the first chunk clears the planes and issues the first reservations)
    const bool relax_gen = ctx->cfg.engine == SMX_ENGINE_RELAXED;
    ctx->bs.relaxed = relax_gen ? 1 : 0;
    if (relax_gen) HIPCHK(hipMemsetAsync(ctx->bs.ctrl + BC_NFLOOD, 0, (BC_LISTS_END - BC_NFLOOD) * 4, st));   // flood, flag and cascade lists start empty
    const BChild* dk = first ? nullptr : ctx->d_kids;        // (sorted there by sort_children)
    const unsigned nb = (nslots + 63u) / 64u;
    if (wind) hipLaunchKernelGGL(k_batch_spawn<true>, dim3(nb), dim3(64), 0, st, ctx->d, bs, nslots, dk, (const uint32_t*)nullptr);
    else hipLaunchKernelGGL(k_batch_spawn<false>, dim3(nb), dim3(64), 0, st, ctx->d, bs, nslots, dk, (const uint32_t*)nullptr);
    hipLaunchKernelGGL(k_batch_iota, dim3((nslots + 255u) / 256u), dim3(256), 0, st, bs.live, nslots);
    uint32_t nlive = nslots, epoch = 0, chunk = 0;
    static const bool trace = [] { const char* e = getenv("SMX_TRACE_BATCH"); return e && atoi(e) != 0; }();
    while (nlive > 0) {
      const auto t_chunk = std::chrono::steady_clock::now();
      const uint32_t live_in = nlive;
      int chunk_epochs = 0;
      if (run_chunk(ctx, wind, nlive, nlive, nslots, first, relax_gen, epoch, chunk, &chunk_epochs)) return -1;
      HIPCHK(hipMemsetAsync(bs.ctrl + BC_NLIVE, 0, 4, st));
      if (wind) hipLaunchKernelGGL(k_batch_compact<true>, dim3((nslots + 255u) / 256u), dim3(256), 0, st, bs, nslots);
      else hipLaunchKernelGGL(k_batch_compact<false>, dim3((nslots + 255u) / 256u), dim3(256), 0, st, bs, nslots);
      HIPCHK(hipMemcpyAsync(ctx->h_bctrl, bs.ctrl, BC_COUNT * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      nlive = ctx->h_bctrl[BC_NLIVE];
      ctx->batch_epochs += chunk_epochs;
      if (trace) fprintf(stderr, "[soilmx] batched %s gen-slots %u epochs %u..%u live %u -> %u grain %d dilate %d maxsteps %d  %.3f ms\n", wind ? "wind " : "water", nslots,
                         epoch - chunk_epochs, epoch, live_in, nlive, 1 << bs.tshift, bs.dilate, bs.maxsteps, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_chunk).count());
    }
    if (wind) hipLaunchKernelGGL(k_batch_counters<true>, dim3(64), dim3(256), 0, st, ctx->d, bs, nslots);
    else hipLaunchKernelGGL(k_batch_counters<false>, dim3(64), dim3(256), 0, st, ctx->d, bs, nslots);
    ctx->batch_generations++;
    // children -> the next generation, in key order (the order decides slot numbers = priorities and rand() draws)
    uint32_t nc = ctx->h_bctrl[BC_NCHILD];
    if (ctx->h_bctrl[BC_CHILD_LOST]) ctx->batch_children_lost += ctx->h_bctrl[BC_CHILD_LOST];
    if (ctx->h_bctrl[BC_FREED_LOST]) { ctx->err = "batched engine: freed-node list overflow (pool nodes leaked)"; return -6; }
    if (ctx->h_bctrl[BC_LIST_LOST]) { ctx->err = "relaxed schedule: flood / cascade list overflow (work was dropped)"; return -6; }
    if (ctx->h_bctrl[BC_STALLED]) { ctx->err = "throughput schedule: a device-side wait ran out of its spin budget (a dataflow dependency or a barrier was never released)"; return -9; }
    if (nc > bs.child_cap) nc = bs.child_cap;
    if (wind) nc = 0;
    if (nc && sort_children(ctx, nc)) return -1;             // -> ctx->d_kids, on the device
    HIPCHK(hipMemsetAsync(bs.ctrl + BC_NCHILD, 0, 8, st));   // NCHILD, CHILD_LOST
    first = false;
    nslots = nc;
  }
  if (draws) {
    HIPCHK(hipMemcpyAsync(ctx->d.rnd, &r, sizeof(r), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  HIPCHK(hipGetLastError());
  return 0;
}
// One sweep of the grid pass over the tile columns [tx_lo, tx_hi): the tiles with work, in (colour, tile) order, through the dataflow
// kernel (k_grid_tiles_flow); bs.sphase says which kind of tile the sweep takes under column strips.
static int grid_sweep(smx_ctx* ctx, int phase, int tx_lo, int tx_hi) {
  hipStream_t st = ctx->stream;
  const int gtx = (ctx->cfg.dimx + (1 << GRID_SHIFT) - 1) >> GRID_SHIFT, gty = (ctx->cfg.dimy + (1 << GRID_SHIFT) - 1) >> GRID_SHIFT;
  const int txb = ctx->d.x_lo >> GRID_SHIFT, txe = ((ctx->d.x_hi - 1) >> GRID_SHIFT) + 1;
  if (tx_lo < txb) tx_lo = txb;
  if (tx_hi > txe) tx_hi = txe;
  if (tx_hi <= tx_lo) return 0;
  if (!ctx->d_tdone) {
    const size_t nt = (size_t)(txe - txb) * gty;
    HIPCHK(hipMalloc(&ctx->d_tdone, nt * 4)); HIPCHK(hipMalloc(&ctx->d_tpend[0], nt * 4)); HIPCHK(hipMalloc(&ctx->d_tpend[1], nt * 4));
    HIPCHK(hipMalloc(&ctx->d_tcount, 8)); HIPCHK(hipHostMalloc(&ctx->h_tcount, 8));
  }
  ctx->bs.tshift = GRID_SHIFT; ctx->bs.sphase = phase;
  const BatchShared& bs = ctx->bs;
  GridTiles g;
  g.tdone = ctx->d_tdone; g.pend[0] = ctx->d_tpend[0]; g.pend[1] = ctx->d_tpend[1]; g.count = ctx->d_tcount;
  g.gtx = gtx; g.gty = gty; g.txb = txb; g.txe = txe; g.tx_lo = tx_lo; g.tx_hi = tx_hi;
  g.keybase = (unsigned long long)phase * 9ull * (unsigned long long)gtx * (unsigned long long)gty;   // key prefix = (phase * 9 + colour) * tiles + tile
  HIPCHK(hipMemsetAsync(ctx->d_tcount, 0, 8, st));
  const uint32_t ntl = (uint32_t)(tx_hi - tx_lo) * (uint32_t)gty;
  hipLaunchKernelGGL(k_grid_tiles_init, dim3((ntl + 255u) / 256u), dim3(256), 0, st, ctx->d, bs, g);
  HIPCHK(hipMemcpyAsync(ctx->h_tcount, ctx->d_tcount, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const uint32_t n = ctx->h_tcount[0];
  static const bool trace = [] { const char* e = getenv("SMX_TRACE_BATCH"); return e && atoi(e) != 0; }();
  const auto t_sweep = std::chrono::steady_clock::now();
  if (n > 0) {
    // the tiles with work in (colour, tile) order (the init kernel appended them as its wavefronts came); entries are unique
    if (ctx->tsort_cap < n) {
      hipFree(ctx->d_tsort); ctx->d_tsort = nullptr; ctx->tsort_cap = 0; ctx->tsort_bytes = 0;
      const uint32_t cap = (uint32_t)((size_t)(txe - txb) * gty);
      size_t tb = 0;
      HIPCHK(rocprim::radix_sort_keys(nullptr, tb, ctx->d_tpend[0], ctx->d_tpend[1], (size_t)cap, 0u, 32u, st));
      HIPCHK(hipMalloc(&ctx->d_tsort, tb)); ctx->tsort_bytes = tb; ctx->tsort_cap = cap;
    }
    size_t tb = ctx->tsort_bytes;
    HIPCHK(rocprim::radix_sort_keys(ctx->d_tsort, tb, ctx->d_tpend[0], ctx->d_tpend[1], (size_t)n, 0u, 32u, st));
    HIPCHK(hipMemsetAsync(ctx->d_tcount + 1, 0, 4, st));     // the cursor of the dataflow kernel
    static const unsigned flow_waves = [] { const char* e = getenv("SMX_GRID_FLOW_WAVES"); const int v = e ? atoi(e) : 2048; return (unsigned)(v < 1 ? 1 : v); }();
    static const uint32_t poll_naps = [] { const char* e = getenv("SMX_GRID_POLL_NAPS"); return e ? (uint32_t)atoi(e) : 2u; }();   // (x 3.4 us between two polls of a waiting tile)
    { PhaseTimer tk(ctx, 7); hipLaunchKernelGGL(k_grid_tiles_flow, dim3(std::min<unsigned>(n, flow_waves)), dim3(64), 0, st, ctx->d, bs, g, ctx->d_tpend[1], n, poll_naps | (flood_prof_on() ? 0x80000000u : 0u), spin_budget(1u << 22)); }
    hipLaunchKernelGGL(k_batch_merge_freed, dim3(1), dim3(256), 0, st, ctx->d, bs);
    HIPCHK(hipStreamSynchronize(st));
    ctx->grid_passes++;
  }
  if (trace) fprintf(stderr, "[soilmx] grid sweep (kind %d): %u tiles with work  %.3f ms\n", phase, n, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_sweep).count());
  HIPCHK(hipGetLastError());
  return 0;
}
static int batch_grid(smx_ctx* ctx) {
  hipStream_t st = ctx->stream;
  if (batch_alloc(ctx, 1, false)) return -1;
  const BatchShared& bs = ctx->bs;
  unsigned long long v0 = 0;
  HIPCHK(hipMemcpyAsync(&v0, ctx->d.ctr + C_GRID_ACTIVE, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemsetAsync(ctx->d.active1, 0, (ctx->nw1 + 64) * 8, st));
  HIPCHK(hipMemsetAsync(ctx->d.active2, 0, (ctx->nw2 + 64) * 8, st));
  launch_classify(ctx);
  HIPCHK(hipStreamSynchronize(st));
  const int gtx = (ctx->cfg.dimx + (1 << GRID_SHIFT) - 1) >> GRID_SHIFT;
  for (int ph = 0; ph < (bs.strips.n > 1 ? 2 : 1); ph++) { const int rc = grid_sweep(ctx, ph, 0, gtx); if (rc) return rc; }
  hipLaunchKernelGGL(k_batch_grid_finish, dim3(1), dim3(1), 0, st, ctx->d, v0);
  HIPCHK(hipMemcpyAsync(ctx->h_bctrl, bs.ctrl, BC_COUNT * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  uint32_t nc = ctx->h_bctrl[BC_NCHILD];
  if (ctx->h_bctrl[BC_CHILD_LOST]) ctx->batch_children_lost += ctx->h_bctrl[BC_CHILD_LOST];
  if (ctx->h_bctrl[BC_FREED_LOST]) { ctx->err = "batched engine: freed-node list overflow (pool nodes leaked)"; return -6; }
  if (ctx->h_bctrl[BC_LIST_LOST]) { ctx->err = "relaxed schedule: flood / cascade list overflow (work was dropped)"; return -6; }
  if (ctx->h_bctrl[BC_STALLED]) { ctx->err = "throughput schedule: a device-side wait ran out of its spin budget (a dataflow dependency or a barrier was never released)"; return -9; }
  if (nc > bs.child_cap) nc = bs.child_cap;
  if (nc && sort_children(ctx, nc)) return -1;
  HIPCHK(hipMemsetAsync(bs.ctrl + BC_NCHILD, 0, 8, st));
  if (nc) { PhaseTimer tk(ctx, 11); return batch_generations(ctx, false, 0, nc); }   // the nested particles that left their tile's region
  return 0;
}

static void launch_classify(smx_ctx* ctx) {                  // which cells can the grid pass change at all? (one flag byte per cell in, one bit out)
  PhaseTimer tk(ctx, 6);
  static const bool scalar = [] { const char* e = getenv("SMX_CLASSIFY_SCALAR"); return e && atoi(e) != 0; }();
  if (ctx->cfg.dimy % 8 == 0 && ctx->lcells % 64 == 0 && ctx->c_lo % 64 == 0 && !scalar)
    hipLaunchKernelGGL(k_grid_classify8, dim3((unsigned)((ctx->lcells / 8 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d);
  else
    hipLaunchKernelGGL(k_grid_classify, dim3((unsigned)((ctx->lcells + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d);
}

}  // extern "C"

// ---------------- column strips, step by step (the same entry points exist in tests/hostsim as hs_d_*) ----------------
static int xbuf_alloc(smx_ctx* ctx) {
  if (ctx->d_xdata) return 0;
  ctx->xdata_cap = 256ull << 20; ctx->xoffs_cap = 8u << 20;
  HIPCHK(hipMalloc(&ctx->d_xdata, ctx->xdata_cap)); HIPCHK(hipMalloc(&ctx->d_xoffs, (size_t)ctx->xoffs_cap * 4)); HIPCHK(hipMalloc(&ctx->d_xcur, 2 * sizeof(XCursor)));   // [0] columns, [1] particles (st_pack_band)
  return 0;
}
extern "C" {
int smx_d_set_own(smx_ctx* ctx, int32_t x0, int32_t x1) { ctx->own_x0 = x0; ctx->own_x1 = x1; return 0; }
int smx_d_gen_begin(smx_ctx* ctx, int32_t wind, uint32_t nslots, const void* kids_host, uint32_t* contains_out) {
  hipStream_t st = ctx->stream;
  ctx->d_wind = wind != 0; ctx->d_first = kids_host == nullptr; ctx->d_nslots = nslots; ctx->d_epoch = 0; ctx->d_chunk = 0;
  if (!ctx->strip)                                           // (the step-wise driver of tests/strips_ref.py: a top-level water generation of a staggered phase is the whole phase; smx_strips_tick sets it itself)
    ctx->stagger_phase_n = (!wind && !kids_host && ctx->cfg.engine == SMX_ENGINE_RELAXED && ctx->water_stagger > 0 && ctx->water_generations > 1) ? nslots : 0u;
  if (batch_alloc(ctx, nslots ? nslots : 1, ctx->d_wind)) return -1;
  RandState r;                                              // every rank holds the same generator state and draws the same values
  HIPCHK(hipMemcpyAsync(&r, ctx->d.rnd, sizeof(r), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const uint32_t ndraw = kids_host ? 0u : 2u * nslots;          // (suspended nested particles carry their state: no draws)
  for (uint32_t i = 0; i < ndraw; i++) {
    const uint32_t v = r.ring[r.idx % 31] + r.ring[(r.idx - 3) % 31];
    r.ring[r.idx % 31] = v; r.idx++; r.calls++;
    ctx->h_draws[i] = v >> 1;
  }
  HIPCHK(hipMemcpyAsync(ctx->d.rnd, &r, sizeof(r), hipMemcpyHostToDevice, st));
  if (ndraw) HIPCHK(hipMemcpyAsync(ctx->d_draws, ctx->h_draws, (size_t)ndraw * 4, hipMemcpyHostToDevice, st));
  ctx->bs.tshift = -1;
  HIPCHK(hipMemsetAsync(ctx->bs.ctrl + BC_NCHILD, 0, 8, st));
  ctx->bs.relaxed = ctx->cfg.engine == SMX_ENGINE_RELAXED ? 1 : 0;
  if (ctx->bs.relaxed) HIPCHK(hipMemsetAsync(ctx->bs.ctrl + BC_NFLOOD, 0, (BC_LISTS_END - BC_NFLOOD) * 4, st));   // flood, flag and cascade lists start empty
  if (kids_host) {
    if (ctx->kids_cap < nslots) { hipFree(ctx->d_kids); ctx->d_kids = nullptr; ctx->kids_cap = 0; HIPCHK(hipMalloc(&ctx->d_kids, (size_t)nslots * sizeof(BChild))); ctx->kids_cap = nslots; }
    if (ctx->contains_cap < nslots) { hipFree(ctx->d_contains); ctx->d_contains = nullptr; ctx->contains_cap = 0; HIPCHK(hipMalloc(&ctx->d_contains, (size_t)nslots * 4)); ctx->contains_cap = nslots; }
    HIPCHK(hipMemcpyAsync(ctx->d_kids, kids_host, (size_t)nslots * sizeof(BChild), hipMemcpyHostToDevice, st));
    if (contains_out) {
      hipLaunchKernelGGL(k_batch_child_contains, dim3((nslots + 63u) / 64u), dim3(64), 0, st, ctx->d_kids, nslots, ctx->d_contains);
      HIPCHK(hipMemcpyAsync(contains_out, ctx->d_contains, (size_t)nslots * 4, hipMemcpyDeviceToHost, st));
    }
  }
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}
static int d_compact(smx_ctx* ctx, uint32_t* nlive) {
  hipStream_t st = ctx->stream; const BatchShared& bs = ctx->bs; const uint32_t nslots = ctx->d_nslots;
  HIPCHK(hipMemsetAsync(bs.ctrl + BC_NLIVE, 0, 4, st));
  if (nslots) {
    if (ctx->d_wind) hipLaunchKernelGGL(k_batch_compact<true>, dim3((nslots + 255u) / 256u), dim3(256), 0, st, bs, nslots);
    else hipLaunchKernelGGL(k_batch_compact<false>, dim3((nslots + 255u) / 256u), dim3(256), 0, st, bs, nslots);
  }
  HIPCHK(hipMemcpyAsync(ctx->h_bctrl, bs.ctrl, BC_COUNT * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  *nlive = ctx->h_bctrl[BC_NLIVE];
  return 0;
}
int smx_d_gen_spawn(smx_ctx* ctx, const uint32_t* contains_host, uint32_t* nlive_local) {
  hipStream_t st = ctx->stream; const uint32_t nslots = ctx->d_nslots;
  if (nslots == 0) { *nlive_local = 0; return 0; }
  const BChild* dk = ctx->d_first ? nullptr : ctx->d_kids;
  if (dk && contains_host) HIPCHK(hipMemcpyAsync(ctx->d_contains, contains_host, (size_t)nslots * 4, hipMemcpyHostToDevice, st));
  const unsigned nb = (nslots + 63u) / 64u;
  if (ctx->d_wind) hipLaunchKernelGGL(k_batch_spawn<true>, dim3(nb), dim3(64), 0, st, ctx->d, ctx->bs, nslots, dk, (const uint32_t*)nullptr);
  else hipLaunchKernelGGL(k_batch_spawn<false>, dim3(nb), dim3(64), 0, st, ctx->d, ctx->bs, nslots, dk, (const uint32_t*)(dk && contains_host ? ctx->d_contains : nullptr));
  return d_compact(ctx, nlive_local);
}
int smx_d_next_phase(smx_ctx* ctx) { return ctx->strips_n > 1 ? (int)(ctx->d_chunk & 1u) : STRIP_INTERIOR; }
int smx_d_chunk(smx_ctx* ctx, uint32_t nlive_global, uint32_t* nlive_local) {
  uint32_t nlive = 0;
  if (d_compact(ctx, &nlive)) return -1;                      // residents that run (particles may have arrived or left)
  int epochs = 0;
  if (run_chunk(ctx, ctx->d_wind, nlive_global, nlive, ctx->d_nslots, ctx->d_first, ctx->cfg.engine == SMX_ENGINE_RELAXED, ctx->d_epoch, ctx->d_chunk, &epochs)) return -1;
  ctx->batch_epochs += (uint64_t)epochs;
  return d_compact(ctx, nlive_local);
}
static int d_gen_end_impl(smx_ctx* ctx, void* children_out, uint32_t cap, uint32_t* n, std::vector<BChild>* vec) {
  hipStream_t st = ctx->stream; const BatchShared& bs = ctx->bs; const uint32_t nslots = ctx->d_nslots;
  if (nslots) {
    if (ctx->d_wind) hipLaunchKernelGGL(k_batch_counters<true>, dim3(64), dim3(256), 0, st, ctx->d, bs, nslots);
    else hipLaunchKernelGGL(k_batch_counters<false>, dim3(64), dim3(256), 0, st, ctx->d, bs, nslots);
    ctx->d_nslots = 0;                                       // (the slots' counters are committed once, whatever happens below)
  }
  HIPCHK(hipMemcpyAsync(ctx->h_bctrl, bs.ctrl, BC_COUNT * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  uint32_t nc = ctx->h_bctrl[BC_NCHILD];
  if (ctx->h_bctrl[BC_CHILD_LOST]) ctx->batch_children_lost += ctx->h_bctrl[BC_CHILD_LOST];
  if (ctx->h_bctrl[BC_FREED_LOST]) { ctx->err = "batched engine: freed-node list overflow (pool nodes leaked)"; return -6; }
  if (ctx->h_bctrl[BC_LIST_LOST]) { ctx->err = "relaxed schedule: flood / cascade list overflow (work was dropped)"; return -6; }
  if (ctx->h_bctrl[BC_STALLED]) { ctx->err = "throughput schedule: a device-side wait ran out of its spin budget (a dataflow dependency or a barrier was never released)"; return -9; }
  if (nc > bs.child_cap) nc = bs.child_cap;
  if (vec) { vec->resize(nc); children_out = vec->data(); cap = nc; }
  if (nc > cap) { ctx->err = "smx_d_gen_end: children buffer too small (call again with room for *n records)"; *n = nc; return -2; }
  if (nc) HIPCHK(hipMemcpyAsync(children_out, bs.children, (size_t)nc * sizeof(BChild), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemsetAsync(bs.ctrl + BC_NCHILD, 0, 8, st));
  HIPCHK(hipStreamSynchronize(st));
  *n = nc;
  ctx->batch_generations++;
  return 0;
}
int smx_d_gen_end(smx_ctx* ctx, void* children_out, uint32_t cap, uint32_t* n) { return d_gen_end_impl(ctx, children_out, cap, n, nullptr); }
// -> [u32 nrec][u32 data bytes][u32 offs[nrec]][records]   (record layout: k_strip_pack_columns)
int smx_d_pack_columns(smx_ctx* ctx, int32_t x0, int32_t x1, void* host_buf, uint64_t cap, uint64_t* bytes) {
  hipStream_t st = ctx->stream;
  if (batch_alloc(ctx, 1, false) || xbuf_alloc(ctx)) return -1;
  if (x0 < 0) x0 = 0; if (x1 > ctx->cfg.dimx) x1 = ctx->cfg.dimx;
  const int gx0 = x0 >> 2, gx1 = (x1 + 3) >> 2, gty = (ctx->cfg.dimy + 3) >> 2;
  HIPCHK(hipMemsetAsync(ctx->d_xcur, 0, sizeof(XCursor), st));
  const long long nt = (long long)(gx1 - gx0) * gty;
  if (nt > 0) hipLaunchKernelGGL(k_strip_pack_columns, dim3((unsigned)((nt + 63) / 64)), dim3(64), 0, st, ctx->d, ctx->bs, gx0, gx1, ctx->d_xdata, ctx->xdata_cap, ctx->d_xoffs, ctx->xoffs_cap, ctx->d_xcur);
  XCursor c;
  HIPCHK(hipMemcpyAsync(&c, ctx->d_xcur, sizeof(c), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (c.lost) { ctx->err = "smx_d_pack_columns: exchange buffer too small"; return -2; }
  const uint64_t total = 8 + 4ull * c.nrec + c.bytes;
  if (total > cap) { ctx->err = "smx_d_pack_columns: host buffer too small"; return -2; }
  uint8_t* hb = (uint8_t*)host_buf;
  const uint32_t nb = (uint32_t)c.bytes;
  memcpy(hb, &c.nrec, 4); memcpy(hb + 4, &nb, 4);
  if (c.nrec) HIPCHK(hipMemcpyAsync(hb + 8, ctx->d_xoffs, 4ull * c.nrec, hipMemcpyDeviceToHost, st));
  if (c.bytes) HIPCHK(hipMemcpyAsync(hb + 8 + 4ull * c.nrec, ctx->d_xdata, c.bytes, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  *bytes = total;
  return 0;
}
int smx_d_unpack_columns(smx_ctx* ctx, const void* host_buf, uint64_t bytes) {
  hipStream_t st = ctx->stream;
  if (bytes < 8) return 0;
  if (batch_alloc(ctx, 1, false) || xbuf_alloc(ctx)) return -1;
  const uint8_t* hb = (const uint8_t*)host_buf;
  uint32_t nrec, nb; memcpy(&nrec, hb, 4); memcpy(&nb, hb + 4, 4);
  if (nrec == 0) return 0;
  if (nrec > ctx->xoffs_cap || nb > ctx->xdata_cap || 8 + 4ull * nrec + nb > bytes) { ctx->err = "smx_d_unpack_columns: bad buffer"; return -2; }
  for (uint32_t i = 0; i < nrec; i++) {                       // (the kernel follows these offsets and counts: they must stay inside the data)
    uint32_t o, hd[2]; memcpy(&o, hb + 8 + 4ull * i, 4);
    if ((uint64_t)o + 24 > nb) { ctx->err = "smx_d_unpack_columns: record offset outside the buffer"; return -2; }
    memcpy(hd, hb + 8 + 4ull * nrec + o, 8);                  // {cell, section count}
    if (hd[0] >= ctx->ncells || (uint64_t)o + 24 + 32ull * hd[1] > nb) { ctx->err = "smx_d_unpack_columns: bad record"; return -2; }
  }
  HIPCHK(hipMemcpyAsync(ctx->d_xoffs, hb + 8, 4ull * nrec, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ctx->d_xdata, hb + 8 + 4ull * nrec, nb, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_strip_unpack_columns, dim3((nrec + 63u) / 64u), dim3(64), 0, st, ctx->d, ctx->bs, ctx->d_xdata, ctx->d_xoffs, nrec, nb);
  hipLaunchKernelGGL(k_batch_merge_freed, dim3(1), dim3(256), 0, st, ctx->d, ctx->bs);
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  return 0;
}
int smx_d_pack_particles(smx_ctx* ctx, int32_t x0, int32_t x1, void* host_buf, uint64_t cap, uint64_t* bytes) {
  hipStream_t st = ctx->stream;
  if (xbuf_alloc(ctx)) return -1;
  const uint32_t nslots = ctx->d_nslots;
  const uint64_t rs = 4 + (ctx->d_wind ? sizeof(BWind) : sizeof(BWater));
  HIPCHK(hipMemsetAsync(ctx->d_xcur, 0, sizeof(XCursor), st));
  if (nslots) {
    if (ctx->d_wind) hipLaunchKernelGGL(k_strip_pack_particles<true>, dim3((nslots + 255u) / 256u), dim3(256), 0, st, ctx->bs, nslots, x0, x1, ctx->d_xdata, ctx->xdata_cap, ctx->d_xcur);
    else hipLaunchKernelGGL(k_strip_pack_particles<false>, dim3((nslots + 255u) / 256u), dim3(256), 0, st, ctx->bs, nslots, x0, x1, ctx->d_xdata, ctx->xdata_cap, ctx->d_xcur);
  }
  XCursor c;
  HIPCHK(hipMemcpyAsync(&c, ctx->d_xcur, sizeof(c), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (c.lost || 4 + c.nrec * rs > cap) { ctx->err = "smx_d_pack_particles: buffer too small"; return -2; }
  uint8_t* hb = (uint8_t*)host_buf;
  memcpy(hb, &c.nrec, 4);
  if (c.nrec) HIPCHK(hipMemcpyAsync(hb + 4, ctx->d_xdata, c.nrec * rs, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  *bytes = 4 + c.nrec * rs;
  return 0;
}
int smx_d_unpack_particles(smx_ctx* ctx, const void* host_buf, uint64_t bytes) {
  hipStream_t st = ctx->stream;
  if (bytes < 4) return 0;
  if (xbuf_alloc(ctx)) return -1;
  const uint8_t* hb = (const uint8_t*)host_buf;
  uint32_t n; memcpy(&n, hb, 4);
  if (n == 0) return 0;
  const uint64_t rs = 4 + (ctx->d_wind ? sizeof(BWind) : sizeof(BWater));
  if (4 + n * rs > bytes || n * rs > ctx->xdata_cap) { ctx->err = "smx_d_unpack_particles: bad buffer"; return -2; }
  HIPCHK(hipMemcpyAsync(ctx->d_xdata, hb + 4, n * rs, hipMemcpyHostToDevice, st));
  if (ctx->d_wind) hipLaunchKernelGGL(k_strip_unpack_particles<true>, dim3((n + 255u) / 256u), dim3(256), 0, st, ctx->bs, ctx->d_xdata, n);
  else hipLaunchKernelGGL(k_strip_unpack_particles<false>, dim3((n + 255u) / 256u), dim3(256), 0, st, ctx->bs, ctx->d_xdata, n);
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}
int smx_d_grid_begin(smx_ctx* ctx) {
  hipStream_t st = ctx->stream;
  if (batch_alloc(ctx, 1, false)) return -1;
  ctx->d_wind = false; ctx->d_nslots = 0; ctx->d_first = true;
  HIPCHK(hipMemsetAsync(ctx->d.active1, 0, (ctx->nw1 + 64) * 8, st));
  HIPCHK(hipMemsetAsync(ctx->d.active2, 0, (ctx->nw2 + 64) * 8, st));
  HIPCHK(hipMemsetAsync(ctx->bs.ctrl + BC_NCHILD, 0, 8, st));
  launch_classify(ctx);
  return 0;
}
int smx_d_grid_sweep_cols(smx_ctx* ctx, int32_t phase, int32_t x_lo, int32_t x_hi) {   // the tiles of the COLUMNS [x_lo, x_hi) (multiples of the tile size)
  const int rc = grid_sweep(ctx, phase, x_lo >> GRID_SHIFT, (x_hi + (1 << GRID_SHIFT) - 1) >> GRID_SHIFT);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}
}  // extern "C"

static bool is_batched(const smx_ctx* ctx) { return ctx->cfg.engine == SMX_ENGINE_BATCHED || ctx->cfg.engine == SMX_ENGINE_RELAXED; }
extern "C" {
// ---------------- the hot path ----------------
int smx_tick_water(smx_ctx* ctx, int32_t nwater) {
  FULLMAP("smx_tick_water (a strip ticks with smx_strips_tick)")
  roctx_range rr("soilmx:water");
  PhaseTimer t(ctx, 0);
  if (is_batched(ctx)) {
    // throughput engines: the phase's particles as `water_generations` consecutive top-level generations of n/k (smx_set_water_generations;
    // default 1). Particles of one generation advance together and do not see the water the same generation adds; k > 1 trades
    // throughput for the reference's feedback between particles of a tick (profiles/r04_p2_analysis.md). Exactly k phase calls of n/k.
    uint32_t left = nwater > 0 ? (uint32_t)nwater : 0u;
    const uint32_t k = ctx->water_generations < 1 ? 1u : (uint32_t)ctx->water_generations;
    if (ctx->cfg.engine == SMX_ENGINE_RELAXED && ctx->water_stagger > 0 && k > 1u && left > 0u) {
      // STAGGERED generations (smx_set_water_stagger): ONE phase; batch b of ceil(n / k) particles takes its first step in epoch b * gap. A particle of
      // batch b finds what the batches before it have done in the `gap` epochs they are ahead -- the lakes their stopped particles made on the way --, the
      // epoch chain of a tick is (k - 1) * gap + one generation long instead of k generations (relax_water_step)
      ctx->stagger_phase_n = left;
      const int rc = batch_generations(ctx, false, left, 0u);
      ctx->stagger_phase_n = 0u;
      return rc;
    }
    for (uint32_t q = 0; q < k && left > 0; q++) {
      const uint32_t nq = left / (k - q);
      left -= nq;
      if (nq) { const int rc = batch_generations(ctx, false, nq, 0u); if (rc) return rc; }
    }
    return 0;
  }
  if (ctx->cfg.engine == SMX_ENGINE_SPECULATIVE) return spec_phase(ctx, nwater, false);
  PhaseTimer tk(ctx, 4);
  hipLaunchKernelGGL(k_water_serial, dim3(1), dim3(64), 0, ctx->stream, ctx->d, nwater);
  HIPCHK(hipGetLastError());
  return 0;
}
int smx_grid_pass(smx_ctx* ctx) {
  FULLMAP("smx_grid_pass (a strip ticks with smx_strips_tick)")
  roctx_range rr("soilmx:grid");
  PhaseTimer t(ctx, 1);
  if (is_batched(ctx)) return batch_grid(ctx);
  HIPCHK(hipMemsetAsync(ctx->d.active1, 0, (ctx->nw1 + 64) * 8, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d.active2, 0, (ctx->nw2 + 64) * 8, ctx->stream));
  launch_classify(ctx);
  hipLaunchKernelGGL(k_grid_serial, dim3(1), dim3(64), 0, ctx->stream, ctx->d);
  HIPCHK(hipGetLastError());
  return 0;
}
int smx_tick_wind(smx_ctx* ctx, int32_t nwind) {
  FULLMAP("smx_tick_wind (a strip ticks with smx_strips_tick)")
  roctx_range rr("soilmx:wind");
  PhaseTimer t(ctx, 2);
  if (is_batched(ctx)) return batch_generations(ctx, true, nwind > 0 ? (uint32_t)nwind : 0u, 0u);
  if (ctx->cfg.engine == SMX_ENGINE_SPECULATIVE) return spec_phase(ctx, nwind, true);
  PhaseTimer tk(ctx, 5);
  hipLaunchKernelGGL(k_wind_serial, dim3(1), dim3(64), 0, ctx->stream, ctx->d, nwind);
  HIPCHK(hipGetLastError());
  return 0;
}
int smx_map_frequency(smx_ctx* ctx) {
  roctx_range rr("soilmx:frequency");
  PhaseTimer t(ctx, 3);
  const size_t n = ctx->ncells, n4 = n / 4;
  unsigned nb = (unsigned)((n4 + 255) / 256);
  if (nb > 2048) nb = 2048;
  if (nb == 0) nb = 1;
  { PhaseTimer tk(ctx, 8); hipLaunchKernelGGL(k_map_frequency, dim3(nb), dim3(256), 0, ctx->stream, ctx->d.wfreq, ctx->d.wtrack, n4, n); }
  HIPCHK(hipGetLastError());
  return 0;
}
int smx_reset_frequency(smx_ctx* ctx) {
  PhaseTimer t(ctx, 3);
  HIPCHK(hipMemsetAsync(ctx->d.wtrack, 0, ctx->ncells * 4, ctx->stream));
  return 0;
}
int smx_tick(smx_ctx* ctx, int32_t nwater, int32_t nwind, int32_t dowater, int32_t dowind) {   // SoilMachine.cpp:283-329
  int rc = 0;                                                 // (a phase's own code comes through: -6 work dropped, -9 a device-side wait gave up, ...)
  if (dowater) { if ((rc = smx_tick_water(ctx, nwater)) != 0) return rc; if ((rc = smx_grid_pass(ctx)) != 0) return rc; }
  if (dowind) { if ((rc = smx_tick_wind(ctx, nwind)) != 0) return rc; }
  if (dowater) { if ((rc = smx_map_frequency(ctx)) != 0) return rc; if ((rc = smx_reset_frequency(ctx)) != 0) return rc; }
  return 0;
}
int smx_sync(smx_ctx* ctx) {
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------- point operations ----------------
static int point_op(smx_ctx* ctx, const PointOp& o, double* out) {
  hipLaunchKernelGGL(k_point_op, dim3(1), dim3(64), 0, ctx->stream, ctx->d, o, ctx->d_scratch);
  if (out) HIPCHK(hipMemcpyAsync(out, ctx->d_scratch, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}
static bool inb(smx_ctx* ctx, int x, int y) {
  if (x < 0 || y < 0 || x >= ctx->cfg.dimx || y >= ctx->cfg.dimy) { ctx->err = "cell out of bounds"; return false; }
  if (ctx->partial()) { ctx->err = "point operations are not available on a strip context (smx_create_strip)"; return false; }
  return true;
}
int smx_add(smx_ctx* ctx, int32_t x, int32_t y, double size, uint32_t type) {
  if (!inb(ctx, x, y)) return -2;
  PointOp o{}; o.op = 0; o.x = x; o.y = y; o.a = size; o.type = type; return point_op(ctx, o, nullptr);
}
int smx_remove(smx_ctx* ctx, int32_t x, int32_t y, double h, double* remainder) {
  if (!inb(ctx, x, y)) return -2;
  PointOp o{}; o.op = 1; o.x = x; o.y = y; o.a = h; double r = 0; int rc = point_op(ctx, o, &r); if (remainder) *remainder = r; return rc;
}
int smx_particle_cascade(smx_ctx* ctx, float px, float py, int32_t transferloop) {
  FULLMAP("smx_particle_cascade")
  if (!(px == px) || !(py == py) || !inb(ctx, (int)roundf(px), (int)roundf(py))) { ctx->err = "cell out of bounds"; return -2; }
  PointOp o{}; o.op = 2; o.px = px; o.py = py; o.k = transferloop; return point_op(ctx, o, nullptr);
}
int smx_water_cascade(smx_ctx* ctx, int32_t x, int32_t y, int32_t spill) {
  if (!inb(ctx, x, y)) return -2;
  PointOp o{}; o.op = 3; o.x = x; o.y = y; o.k = spill; return point_op(ctx, o, nullptr);
}
int smx_seep(smx_ctx* ctx, int32_t x, int32_t y) {
  if (!inb(ctx, x, y)) return -2;
  PointOp o{}; o.op = 4; o.x = x; o.y = y; return point_op(ctx, o, nullptr);
}

int smx_top(smx_ctx* ctx, int32_t x, int32_t y, uint32_t* type, double* size, double* floor, double* sat, int32_t* empty) {
  if (!inb(ctx, x, y)) return -2;
  Sec r;
  HIPCHK(hipMemcpyAsync(&r, ctx->d.cells + ((size_t)x * ctx->cfg.dimy + y), sizeof(Sec), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const bool e = r.type == EMPTY;
  if (empty) *empty = e ? 1 : 0;
  if (type) *type = e ? 0u : r.type;
  if (size) *size = r.size;
  if (floor) *floor = r.floor;
  if (sat) *sat = r.sat;
  return 0;
}

// ---------------- observability ----------------
int smx_digest(smx_ctx* ctx, double* sumh, uint64_t* nsec, uint64_t* typehash) {   // (a strip context digests the range it holds)
  const size_t n = ctx->lcells;
  const uint64_t cap = ctx->cfg.pool_capacity;
  std::vector<Sec> cells(n);
  HIPCHK(hipMemcpyAsync(cells.data(), ctx->cells_alloc, n * sizeof(Sec), hipMemcpyDeviceToHost, ctx->stream));
  // {type, prev} = the last 8 bytes of every 32-byte pool record: one strided 2D copy instead of the whole pool
  std::vector<uint64_t> tp(cap);
  HIPCHK(hipMemcpy2DAsync(tp.data(), 8, reinterpret_cast<const char*>(ctx->d.pool) + 24, sizeof(Sec), 8, cap, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  double sh = 0.0; uint64_t ns = 0, h = 1469598103934665603ull;
  for (size_t c = 0; c < n; c++) {
    const Sec& t = cells[c];
    if (t.type == EMPTY) { sh += 0.0; continue; }
    sh += t.floor + t.size;
    h = (h ^ (uint64_t)t.type) * 1099511628211ull; ns++;
    uint32_t pv = t.prev; uint64_t guard = 0;
    while (pv != NIL) {
      if (pv >= cap || ++guard > cap) { ctx->err = "smx_digest: corrupt section chain"; return -5; }
      const uint64_t w = tp[pv];
      h = (h ^ (uint64_t)(uint32_t)w) * 1099511628211ull; ns++;
      pv = (uint32_t)(w >> 32);
    }
  }
  if (sumh) *sumh = sh;
  if (nsec) *nsec = ns;
  if (typehash) *typehash = h;
  return 0;
}
// The structs behind the two getters have grown (round 2: three streaming-kernel timers; round 4: two speculative-engine counters;
// round 5: step-kernel launch totals). A caller says how large ITS struct is and gets exactly that prefix; the unsized entry points
// keep writing the layout they had when they were introduced (16 / 18 words), so a binary built against an older header is never
// written past its struct.
int smx_get_counters_sized(smx_ctx* ctx, smx_counters* out, uint64_t struct_size) {
  unsigned long long c[C_COUNT];
  RandState r;
  HIPCHK(hipMemcpyAsync(c, ctx->d.ctr, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(&r, ctx->d.rnd, sizeof(r), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  smx_counters t;
  memset(&t, 0, sizeof(t));
  t.steps_water_top = c[C_STEPS_WATER_TOP]; t.steps_water_all = c[C_STEPS_WATER_ALL]; t.steps_wind = c[C_STEPS_WIND];
  t.nested_particles = c[C_NESTED]; t.floods = c[C_FLOODS]; t.cascade_calls = c[C_CASCADE_CALLS];
  t.cascade_transfers = c[C_CASCADE_TRANSFERS]; t.wcascade_calls = c[C_WCASCADE_CALLS];
  t.grid_active_cells = c[C_GRID_ACTIVE]; t.rand_calls = r.calls;
  t.pool_free = ctx->cfg.pool_capacity - c[C_LIVE_SECTIONS]; t.pool_overflow = c[C_POOL_OVERFLOW];
  t.spec_rounds = ctx->spec_rounds; t.spec_aborts = ctx->spec_aborts;
  t.reserved[0] = ctx->spec_fallbacks; t.reserved[1] = ctx->spec_executed;
  t.spec_subphases_cut = ctx->spec_subphases_cut; t.spec_serial_particles = ctx->spec_serial_particles;
  t.flood_nested_steps = c[C_FLOOD_NESTED_STEPS]; t.grid_nested_steps = c[C_GRID_NESTED_STEPS];
  memcpy(out, &t, struct_size < sizeof(t) ? (size_t)struct_size : sizeof(t));
  return 0;
}
int smx_get_counters(smx_ctx* ctx, smx_counters* out) { return smx_get_counters_sized(ctx, out, 16 * sizeof(uint64_t)); }   // the layout of rounds 1-3
int smx_get_timing_sized(smx_ctx* ctx, smx_timing* out, uint64_t struct_size) {
  HIPCHK(hipStreamSynchronize(ctx->stream));
  drain_events(ctx);
  memcpy(out, &ctx->timing, struct_size < sizeof(ctx->timing) ? (size_t)struct_size : sizeof(ctx->timing));
  return 0;
}
int smx_get_timing(smx_ctx* ctx, smx_timing* out) { return smx_get_timing_sized(ctx, out, 144); }   // the layout of rounds 2-4
int smx_timing_reset(smx_ctx* ctx) {
  HIPCHK(hipStreamSynchronize(ctx->stream));
  drain_events(ctx);
  memset(&ctx->timing, 0, sizeof(ctx->timing));
  return 0;
}

}  // extern "C"

#include "soil_strips_host.h"

// ---------------- LBM wind: lbmwind.h:75-197 + shader/LBM/*.cs, shader/move.cs -> soil_lbm.h ----------------
#define LBMCHK(call)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (call);                                                                           \
    if (e_ != hipSuccess) { l->err = std::string(#call) + ": " + hipGetErrorString(e_); return -1; }  \
  } while (0)
extern "C" {
void smx_lbm_destroy(smx_lbm* l) {
  if (!l) return;
  hipSetDevice(l->device);
  if (l->stream) hipStreamSynchronize(l->stream);
  hipFree(l->f[0]); hipFree(l->f[1]); hipFree(l->B); hipFree(l->rho); hipFree(l->v);
  if (l->ev0) hipEventDestroy(l->ev0); if (l->ev1) hipEventDestroy(l->ev1);
  if (l->stream) hipStreamDestroy(l->stream);
  delete l;
}
const char* smx_lbm_last_error(smx_lbm* l) { return l ? l->err.c_str() : "null handle"; }
int smx_lbm_create(int32_t nx, int32_t ny, int32_t nz, int32_t device, smx_lbm** out) {
  if (!out) return -2;
  smx_lbm* l = new smx_lbm;
  *out = l;
  if (nx < 2 || ny < 2 || nz < 2) { l->err = "smx_lbm_create: every dimension must be >= 2"; return -2; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { l->err = "no HIP device visible (soilmx has no CPU fallback)"; return -3; }
  l->device = device; l->d.nx = nx; l->d.ny = ny; l->d.nz = nz; l->d.n = (unsigned long long)nx * ny * nz;
  l->k = make_lbm_consts();
  LBMCHK(hipSetDevice(device));
  LBMCHK(hipStreamCreate(&l->stream));
  LBMCHK(hipEventCreate(&l->ev0)); LBMCHK(hipEventCreate(&l->ev1));
  const size_t n = (size_t)l->d.n;
  LBMCHK(hipMalloc(&l->f[0], n * LQ * 4)); LBMCHK(hipMalloc(&l->f[1], n * LQ * 4));
  LBMCHK(hipMalloc(&l->B, n * 4)); LBMCHK(hipMalloc(&l->rho, n * 4)); LBMCHK(hipMalloc(&l->v, n * 16));
  LBMCHK(hipMemsetAsync(l->f[0], 0, n * LQ * 4, l->stream)); LBMCHK(hipMemsetAsync(l->f[1], 0, n * LQ * 4, l->stream));
  LBMCHK(hipMemsetAsync(l->B, 0, n * 4, l->stream)); LBMCHK(hipMemsetAsync(l->rho, 0, n * 4, l->stream)); LBMCHK(hipMemsetAsync(l->v, 0, n * 16, l->stream));
  LBMCHK(hipStreamSynchronize(l->stream));
  return 0;
}
static unsigned lbm_grid(const smx_lbm* l) { const unsigned long long nb = (l->d.n + 255ull) / 256ull; return (unsigned)((nb + 7ull) / 8ull * 8ull); }
int smx_lbm_set_boundary(smx_lbm* l, const float* b) {
  LBMCHK(hipSetDevice(l->device));
  LBMCHK(hipMemcpyAsync(l->B, b, (size_t)l->d.n * 4, hipMemcpyHostToDevice, l->stream));
  LBMCHK(hipStreamSynchronize(l->stream));
  return 0;
}
int smx_lbm_boundary_from_map(smx_lbm* l, smx_ctx* ctx, float sx, float sy, float sz) {
  if (!ctx) { l->err = "smx_lbm_boundary_from_map: no map"; return -2; }
  if (ctx->partial()) { l->err = "smx_lbm_boundary_from_map: the map context holds only a strip"; return -2; }
  if (ctx->cfg.device != l->device) { l->err = "smx_lbm_boundary_from_map: map and lattice live on different devices"; return -2; }
  if ((int)(sx * (float)(l->d.nx - 1)) >= ctx->cfg.dimx || (int)(sz * (float)(l->d.nz - 1)) >= ctx->cfg.dimy || sx < 0.f || sz < 0.f) {
    l->err = "smx_lbm_boundary_from_map: the scaled lattice does not fit the map"; return -2;
  }
  if (ctx->plane_cap < ctx->ncells * 8) { hipFree(ctx->d_plane); ctx->d_plane = nullptr; ctx->plane_cap = 0; HIPCHK(hipMalloc(&ctx->d_plane, ctx->ncells * 8)); ctx->plane_cap = ctx->ncells * 8; }
  hipLaunchKernelGGL(k_heights, dim3((unsigned)((ctx->ncells + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d, (double*)ctx->d_plane);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  hipLaunchKernelGGL(k_lbm_boundary_from_heights, dim3((unsigned)((l->d.n + 255ull) / 256ull)), dim3(256), 0, l->stream, l->d, (const double*)ctx->d_plane,
                     ctx->cfg.dimy, sx, sy, sz, ctx->cfg.scale, l->B);
  LBMCHK(hipStreamSynchronize(l->stream));
  return 0;
}
int smx_lbm_initialize(smx_lbm* l) {
  LBMCHK(hipSetDevice(l->device));
  l->cur = 0;
  hipLaunchKernelGGL(k_lbm_init, dim3(lbm_grid(l)), dim3(256), 0, l->stream, l->d, l->k, l->B, l->f[0], l->rho, l->v);
  LBMCHK(hipGetLastError());
  return 0;
}
int smx_lbm_step(smx_lbm* l, int32_t n) {
  LBMCHK(hipSetDevice(l->device));
  if (n <= 0) return 0;
  LBMCHK(hipEventRecord(l->ev0, l->stream));
  static const int variant = [] {                             // measured switches (profiles/r02_lbm_bench.log); results do not depend on them
    const char* a = getenv("SMX_LBM_NT"); const char* b = getenv("SMX_LBM_XCD");
    return ((a ? atoi(a) != 0 : false) ? 1 : 0) | ((b ? atoi(b) != 0 : false) ? 2 : 0);
  }();
  for (int s = 0; s < n; s++) {
    const dim3 g(lbm_grid(l)), t(256);
    float *fa = l->f[l->cur], *fb = l->f[l->cur ^ 1];
    switch (variant) {
      case 0: hipLaunchKernelGGL((k_lbm_step<false, false>), g, t, 0, l->stream, l->d, l->k, l->B, fa, fb, l->rho, l->v); break;
      case 1: hipLaunchKernelGGL((k_lbm_step<true, false>), g, t, 0, l->stream, l->d, l->k, l->B, fa, fb, l->rho, l->v); break;
      case 2: hipLaunchKernelGGL((k_lbm_step<false, true>), g, t, 0, l->stream, l->d, l->k, l->B, fa, fb, l->rho, l->v); break;
      default: hipLaunchKernelGGL((k_lbm_step<true, true>), g, t, 0, l->stream, l->d, l->k, l->B, fa, fb, l->rho, l->v); break;
    }
    l->cur ^= 1;
  }
  LBMCHK(hipEventRecord(l->ev1, l->stream));
  LBMCHK(hipGetLastError());
  LBMCHK(hipEventSynchronize(l->ev1));
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, l->ev0, l->ev1) == hipSuccess) { l->ms_steps += ms; l->steps += (unsigned long long)n; }
  return 0;
}
int smx_lbm_get_timing(smx_lbm* l, double* ms_steps, uint64_t* steps, int32_t reset) {
  if (ms_steps) *ms_steps = l->ms_steps;
  if (steps) *steps = l->steps;
  if (reset) { l->ms_steps = 0.0; l->steps = 0; }
  return 0;
}
int smx_lbm_read(smx_lbm* l, float* rho, float* v4, float* f_aos) {
  LBMCHK(hipSetDevice(l->device));
  const size_t n = (size_t)l->d.n;
  if (rho) LBMCHK(hipMemcpyAsync(rho, l->rho, n * 4, hipMemcpyDeviceToHost, l->stream));
  if (v4) LBMCHK(hipMemcpyAsync(v4, l->v, n * 16, hipMemcpyDeviceToHost, l->stream));
  if (f_aos) {                                                // the spare lattice is scratch between steps (a step writes every slot of it)
    float* tmp = l->f[l->cur ^ 1];
    hipLaunchKernelGGL(k_lbm_to_aos, dim3((unsigned)((n * LQ + 255) / 256)), dim3(256), 0, l->stream, l->d, l->f[l->cur], tmp);
    LBMCHK(hipMemcpyAsync(f_aos, tmp, n * LQ * 4, hipMemcpyDeviceToHost, l->stream));
  }
  LBMCHK(hipStreamSynchronize(l->stream));
  return 0;
}
int smx_lbm_write_f(smx_lbm* l, const float* f_aos) {
  LBMCHK(hipSetDevice(l->device));
  const size_t n = (size_t)l->d.n;
  float* stage = nullptr;
  LBMCHK(hipMalloc(&stage, n * LQ * 4));
  hipError_t e = hipMemcpyAsync(stage, f_aos, n * LQ * 4, hipMemcpyHostToDevice, l->stream);
  hipLaunchKernelGGL(k_lbm_from_aos, dim3((unsigned)((n * LQ + 255) / 256)), dim3(256), 0, l->stream, l->d, stage, l->f[l->cur]);
  if (e == hipSuccess) e = hipStreamSynchronize(l->stream);
  hipFree(stage);
  if (e != hipSuccess) { l->err = hipGetErrorString(e); return -1; }
  return 0;
}
int smx_lbm_move(smx_lbm* l, float* pos4, int32_t n) {
  LBMCHK(hipSetDevice(l->device));
  if (n <= 0) return 0;
  float4* d = nullptr;
  LBMCHK(hipMalloc(&d, (size_t)n * 16));
  hipError_t e = hipMemcpyAsync(d, pos4, (size_t)n * 16, hipMemcpyHostToDevice, l->stream);
  hipLaunchKernelGGL(k_lbm_move, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, l->stream, l->d, l->v, d, n);
  if (e == hipSuccess) e = hipMemcpyAsync(pos4, d, (size_t)n * 16, hipMemcpyDeviceToHost, l->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(l->stream);
  hipFree(d);
  if (e != hipSuccess) { l->err = hipGetErrorString(e); return -1; }
  return 0;
}
}  // extern "C"
