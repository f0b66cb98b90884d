// Binary state snapshot shared by the oracle binaries, the C-ABI export and the Python tests.
// TEST/INTERCHANGE FORMAT (the reference has no on-disk state format: SURVEY.md §5 "Checkpoint").
//
// Layout (little endian):
//   SnapHeader (64 B)
//   uint32 count[ncells]      sections per column, cells in Layermap order  x*dimy + y  (layermap.h:151)
//   uint32 type [nsec]        per section, columns in cell order, each column BOTTOM -> TOP
//   double size [nsec]
//   double floor[nsec]
//   double sat  [nsec]
//   float  wfreq [ncells]     WaterParticle::frequency, native index y*dimx + x (water.h:53,349)
//   float  wtrack[ncells]     WaterParticle::track
//   float  windfreq[ncells]   WindParticle::frequency (wind.h:50)
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

struct SnapHeader {
  char magic[8];        // "SMXSNAP1"
  int32_t dimx, dimy, scale, nsoils;
  uint64_t ncells, nsec;
  uint64_t rand_calls;  // rand() draws consumed since srand(SEED)
  uint64_t pool_free;   // free pool elements (map.pool.free.size())
  uint64_t reserved;
};
static_assert(sizeof(SnapHeader) == 64, "SnapHeader must be 64 bytes");

struct Snapshot {
  SnapHeader h;
  std::vector<uint32_t> count, type;
  std::vector<double> size, floor, sat;
  std::vector<float> wfreq, wtrack, windfreq;
};

static inline bool snapshot_write(const char* path, const Snapshot& s) {
  FILE* f = fopen(path, "wb");
  if (!f) return false;
  fwrite(&s.h, sizeof(SnapHeader), 1, f);
  fwrite(s.count.data(), 4, s.count.size(), f);
  fwrite(s.type.data(), 4, s.type.size(), f);
  fwrite(s.size.data(), 8, s.size.size(), f);
  fwrite(s.floor.data(), 8, s.floor.size(), f);
  fwrite(s.sat.data(), 8, s.sat.size(), f);
  fwrite(s.wfreq.data(), 4, s.wfreq.size(), f);
  fwrite(s.wtrack.data(), 4, s.wtrack.size(), f);
  fwrite(s.windfreq.data(), 4, s.windfreq.size(), f);
  fclose(f);
  return true;
}

// Digest of SURVEY.md Appendix E: sum of heights (x outer / y inner), section count,
// FNV-style hash over the types walking each column TOP -> BOTTOM.
struct Digest {
  double sumh;
  uint64_t nsec;
  uint64_t typehash;
};
static inline Digest snapshot_digest(const Snapshot& s) {
  Digest d{0.0, 0, 1469598103934665603ull};
  size_t off = 0;
  for (uint64_t c = 0; c < s.h.ncells; c++) {
    uint32_t n = s.count[c];
    if (n) d.sumh += s.floor[off + n - 1] + s.size[off + n - 1];
    for (uint32_t k = n; k-- > 0;) {
      d.nsec++;
      d.typehash = (d.typehash ^ (uint64_t)s.type[off + k]) * 1099511628211ull;
    }
    off += n;
  }
  return d;
}
