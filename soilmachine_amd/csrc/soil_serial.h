// soil_serial.h -- access policy of the exact SERIAL engine: one walker, reference order, direct access.
#pragma once
#include "soil_core.h"

namespace smx {

struct SerialPolicy {
  static constexpr bool READ_ONLY = false;
  static constexpr bool DEFER_NESTED = false;
  static constexpr bool EXCLUSIVE = true;       // this thread owns every cell it touches while it acts (soil_core.h push_frame)
  unsigned long long live;       // live sections, cached in a register for the kernel's lifetime
  uint32_t nfree;                // entries on the free-node stack
  SMX_D explicit SerialPolicy(const DevState& s) : live(s.ctr[C_LIVE_SECTIONS]), nfree(*s.free_count) {}
  SMX_D void finish(const DevState& s) { s.ctr[C_LIVE_SECTIONS] = live; *s.free_count = nfree; }

  template <int N> struct Pre {};
  SMX_D void save_blk(const Blk&) {}
  SMX_D void load_blk(Blk&) {}
  SMX_D bool touch(const DevState&, size_t, size_t) { return true; }
  template <int N> SMX_D void prefetch(const DevState&, const size_t (&)[N], const size_t (&)[N], Pre<N>&) {}
  template <int N> SMX_D bool acquire_log(const DevState&, const size_t (&)[N], const Pre<N>&, const Sec (&)[N]) { return true; }
  SMX_D void pre_write_node(const DevState&, uint32_t) {}
  // fire-and-forget on the 32-bit word that holds the flag byte (the plane is allocated in whole words): a walker that first
  // LOADS the byte to see whether it must change pays a memory round trip per call
  SMX_D void set_flag(const DevState& s, size_t c, uint8_t f) {
    SMX_OR32_ASYNC(reinterpret_cast<uint32_t*>(s.flags + (c & ~(size_t)3)), (uint32_t)f << (8u * (uint32_t)(c & 3)));
  }
  SMX_D void set_flag_async(const DevState& s, size_t c, uint8_t f) { set_flag(s, c, f); }
  SMX_D void clear_flag(const DevState& s, size_t c, uint8_t f) {
    SMX_AND32_ASYNC(reinterpret_cast<uint32_t*>(s.flags + (c & ~(size_t)3)), ~((uint32_t)f << (8u * (uint32_t)(c & 3))));
  }
  // secpool::get / unget (layermap.h:89-111): get() fails exactly when live sections == capacity
  SMX_D bool can_get(const DevState& s) const { return live < s.pool_capacity; }
  SMX_D void live_add(const DevState&, int d) { live += (long long)d; }
  SMX_D uint32_t node_alloc(const DevState& s) {
    if (nfree == 0) return NIL;
    nfree--;
    return s.freelist[nfree];
  }
  SMX_D void node_free(const DevState& s, uint32_t e) { s.freelist[nfree] = e; nfree++; }
  SMX_D int rand1(const DevState& s) {          // glibc rand()
    RandState& r = *s.rnd;
    const uint32_t i = r.idx;
    const uint32_t v = r.ring[i % 31] + r.ring[(i - 3) % 31];
    r.ring[i % 31] = v;
    r.idx = i + 1;
    r.calls++;
    return (int)(v >> 1);
  }
  SMX_D void rand2(const DevState& s, int& first, int& second) { first = rand1(s); second = rand1(s); }
  SMX_D void add_counter(const DevState& s, int which, unsigned long long v) { if (v) s.ctr[which] += v; }
};

}  // namespace smx
