// Headless driver around the UNMODIFIED reference hot-path headers (TEST INFRASTRUCTURE ONLY).
//
// Builds oracle/_ref/soil_ref (see oracle/Makefile): the reference's own source/layermap.h,
// source/surface.h, source/particle/{particle,water,wind}.h and source/io.h are #included from
// where they lie under /root/reference (REF_ROOT on the include path) -- nothing is copied into
// this repository. Third-party pieces that are not vendored by the reference are replaced by
// oracle/shim/ (glm restatement, GL-free Vertexpool, image:: stub).
//
// The tick loop below restates SoilMachine.cpp:283-329 with rendering removed:
//   water loop (NWATER x {ctor; while(move && interact); flood}) -> WaterParticle::seep(map)
//   -> wind loop (NWIND x {ctor; while(move && interact)}) -> mapfrequency -> resetfrequency.
// Set-up restates SoilMachine.cpp:36-48,82-83: srand(SEED); loadsoil; init(); Layermap(SEED, dim).
//
// Must be compiled with g++ (argument evaluation order of vec2(rand()%x, rand()%y),
// water.h:13 / wind.h:15: first draw -> y) and without FMA contraction.
#include <algorithm>
#include <assert.h>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <math.h>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include <glm/glm.hpp>
#include "shim/ref_stubs.h"
#include "snapshot.h"

// ---- globals the reference headers read as free variables (SoilMachine.cpp:9-17) ----
int SIZEX = 256;
int SIZEY = 256;
int SCALE = 80;
int SLICE = 2 * SCALE;
int NWIND = 250;
int NWATER = 250;
int SEED = 0;
int g_poolsize = 10000000;
#define POOLSIZE g_poolsize

// ---- count libc rand() draws (all std headers are already included above) ----
static uint64_t g_rand_calls = 0;
static inline int oracle_rand() { g_rand_calls++; return (rand)(); }
static inline int oracle_rand_fn() { return oracle_rand(); }
#define rand() oracle_rand()

#define private public             // --load rebuilds the columns behind Layermap::dat (layermap.h:131) from a snapshot; nothing else uses it
#include "source/layermap.h"
#undef private
#include "source/particle/water.h"
#include "source/particle/wind.h"
#include "source/io.h"

#undef rand

static Snapshot take_snapshot(Layermap& map) {
  Snapshot s;
  memset(&s.h, 0, sizeof(s.h));
  memcpy(s.h.magic, "SMXSNAP1", 8);
  s.h.dimx = map.dim.x; s.h.dimy = map.dim.y; s.h.scale = SCALE; s.h.nsoils = (int)soils.size();
  s.h.ncells = (uint64_t)map.dim.x * map.dim.y;
  s.h.rand_calls = g_rand_calls;
  s.h.pool_free = map.pool.free.size();
  s.count.resize(s.h.ncells);
  std::vector<sec*> col;
  for (int x = 0; x < map.dim.x; x++)
    for (int y = 0; y < map.dim.y; y++) {
      col.clear();
      for (sec* t = map.top(ivec2(x, y)); t != NULL; t = t->prev) col.push_back(t);
      s.count[(size_t)x * map.dim.y + y] = (uint32_t)col.size();
      for (size_t k = col.size(); k-- > 0;) {
        s.type.push_back((uint32_t)col[k]->type);
        s.size.push_back(col[k]->size);
        s.floor.push_back(col[k]->floor);
        s.sat.push_back(col[k]->saturation);
      }
    }
  s.h.nsec = s.type.size();
  size_t n = s.h.ncells;
  s.wfreq.assign(WaterParticle::frequency, WaterParticle::frequency + n);
  s.wtrack.assign(WaterParticle::track, WaterParticle::track + n);
  s.windfreq.assign(WindParticle::frequency, WindParticle::frequency + n);
  return s;
}

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  std::string soil = "soil/default.soil", dump_prefix, dump_at, soils_out, load_path, heights_out;
  long long rand_seed = -1;          // --rand-seed S: srand(S) once the terrain stands (the same map under another rand() stream: the P2 control)
  int size = 0, sizex = 0, sizey = 0, ticks = 1, nwater = -1, nwind = -1, dowater = 1, dowind = 1, quiet = 1;
  for (int i = 1; i + 1 < argc; i += 2) {
    std::string k = argv[i], v = argv[i + 1];
    if (k == "--soil") soil = v;
    else if (k == "--seed") SEED = atoi(v.c_str());
    else if (k == "--size") size = atoi(v.c_str());
    else if (k == "--sizex") sizex = atoi(v.c_str());
    else if (k == "--sizey") sizey = atoi(v.c_str());
    else if (k == "--ticks") ticks = atoi(v.c_str());
    else if (k == "--nwater") nwater = atoi(v.c_str());
    else if (k == "--nwind") nwind = atoi(v.c_str());
    else if (k == "--water") dowater = atoi(v.c_str());
    else if (k == "--wind") dowind = atoi(v.c_str());
    else if (k == "--pool") g_poolsize = atoi(v.c_str());
    else if (k == "--dump-prefix") dump_prefix = v;
    else if (k == "--dump-at") dump_at = v;
    else if (k == "--soils-out") soils_out = v;
    else if (k == "--quiet") quiet = atoi(v.c_str());
    else if (k == "--load") load_path = v;
    else if (k == "--rand-seed") rand_seed = atoll(v.c_str());
    else if (k == "--heights-out") heights_out = v;
    else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
  }
  std::vector<int> dumps;
  { std::stringstream ss(dump_at); std::string t; while (getline(ss, t, ',')) if (!t.empty()) dumps.push_back(atoi(t.c_str())); }

  std::streambuf* old = nullptr;
  std::ostringstream sink;
  if (quiet) old = std::cout.rdbuf(sink.rdbuf());

  srand(SEED);                       // SoilMachine.cpp:41
  loadsoil(soil);                    // SoilMachine.cpp:43-45 (overwrites SIZEX/SIZEY/SCALE/NWIND/NWATER)
  if (size > 0) { SIZEX = size; SIZEY = size; }
  if (sizex > 0) SIZEX = sizex;
  if (sizey > 0) SIZEY = sizey;   // forced map size (SURVEY.md §5 "Config")
  if (nwater >= 0) NWATER = nwater;
  if (nwind >= 0) NWIND = nwind;
  SLICE = 2 * SCALE;
  WaterParticle::init();             // SoilMachine.cpp:47-48
  WindParticle::init();

  double t0 = now_s();
  Vertexpool<Vertex> vertexpool(SIZEX * SIZEY, 1);
  Layermap map(SEED, glm::ivec2(SIZEX, SIZEY), vertexpool);   // SoilMachine.cpp:82-83
  double t_init = now_s() - t0;

  if (!soils_out.empty()) {          // soil table as parsed by the reference's own loadsoil()
    FILE* f = fopen(soils_out.c_str(), "w");
    fprintf(f, "{\"scale\": %d, \"sizex\": %d, \"sizey\": %d, \"nwater\": %d, \"nwind\": %d, \"soils\": [", SCALE, SIZEX, SIZEY, NWATER, NWIND);
    for (size_t i = 0; i < soils.size(); i++) {
      const SurfParam& p = soils[i];
      fprintf(f, "%s{\"name\": \"%s\", \"transports\": %zu, \"erodes\": %zu, \"cascades\": %zu, \"abrades\": %zu, "
                 "\"density\": %.9g, \"porosity\": %.9g, \"solubility\": %.9g, \"equrate\": %.9g, \"friction\": %.9g, "
                 "\"erosionrate\": %.9g, \"maxdiff\": %.9g, \"settling\": %.9g, \"suspension\": %.9g, \"abrasion\": %.9g}",
              i ? ", " : "", p.name.c_str(), p.transports, p.erodes, p.cascades, p.abrades, p.density, p.porosity,
              p.solubility, p.equrate, p.friction, p.erosionrate, p.maxdiff, p.settling, p.suspension, p.abrasion);
    }
    fprintf(f, "], \"layers\": [");
    for (size_t i = 0; i < layers.size(); i++) {
      const SurfLayer& l = layers[i];
      fprintf(f, "%s{\"type\": %zu, \"min\": %.9g, \"bias\": %.9g, \"scale\": %.9g, \"octaves\": %.9g, \"lacunarity\": %.9g, \"gain\": %.9g, \"frequency\": %.9g}",
              i ? ", " : "", l.type, l.min, l.bias, l.scale, l.octaves, l.lacunarity, l.gain, l.frequency);
    }
    fprintf(f, "]}\n");
    fclose(f);
  }

  // --load <snapshot>: continue from a state written by smx_save / snapshot_write (the same SMXSNAP1 layout): every column is rebuilt
  // bottom -> top with the snapshot's type / size / floor / saturation, the frequency planes are restored and the rand() stream is
  // advanced to the snapshot's draw count (the stream is srand(SEED)'s either way). Used to time the CPU path on the very state the
  // GPU line's warm-up produced (bench.py cpu_baseline) -- test infrastructure.
  uint64_t rand_calls_at_start = 0;
  if (!load_path.empty()) {
    FILE* f = fopen(load_path.c_str(), "rb");
    SnapHeader h;
    if (!f || fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "SMXSNAP1", 8) != 0) { fprintf(stderr, "--load: cannot read %s\n", load_path.c_str()); return 3; }
    if (h.dimx != SIZEX || h.dimy != SIZEY) { fprintf(stderr, "--load: snapshot is %dx%d, the run is %dx%d\n", h.dimx, h.dimy, SIZEX, SIZEY); return 3; }
    const size_t n = (size_t)h.ncells, ns = (size_t)h.nsec;
    std::vector<uint32_t> count(n), type(ns);
    std::vector<double> sz(ns), fl(ns), sa(ns);
    std::vector<float> wf(n), wt(n), wi(n);
    bool ok = fread(count.data(), 4, n, f) == n && fread(type.data(), 4, ns, f) == ns && fread(sz.data(), 8, ns, f) == ns && fread(fl.data(), 8, ns, f) == ns &&
              fread(sa.data(), 8, ns, f) == ns && fread(wf.data(), 4, n, f) == n && fread(wt.data(), 4, n, f) == n && fread(wi.data(), 4, n, f) == n;
    fclose(f);
    if (!ok) { fprintf(stderr, "--load: truncated snapshot\n"); return 3; }
    size_t off = 0;
    for (int x = 0; x < map.dim.x; x++)
      for (int y = 0; y < map.dim.y; y++) {
        const size_t c = (size_t)x * map.dim.y + y;
        for (sec* t = map.dat[c]; t != NULL;) { sec* below = t->prev; map.pool.unget(t); t = below; }
        sec* below = NULL;
        for (uint32_t k = 0; k < count[c]; k++, off++) {
          sec* E = map.pool.get(sz[off], (SurfType)type[off]);
          if (E == NULL) { fprintf(stderr, "--load: pool exhausted\n"); return 3; }
          E->floor = fl[off]; E->saturation = sa[off]; E->prev = below; E->next = NULL;
          if (below) below->next = E;
          below = E;
        }
        map.dat[c] = below;
      }
    memcpy(WaterParticle::frequency, wf.data(), n * 4); memcpy(WaterParticle::track, wt.data(), n * 4); memcpy(WindParticle::frequency, wi.data(), n * 4);
    if (rand_seed < 0) { srand(SEED); g_rand_calls = 0; for (uint64_t i = 0; i < h.rand_calls; i++) (void)oracle_rand_fn(); }
  }
  if (rand_seed >= 0) { srand((unsigned)rand_seed); }
  rand_calls_at_start = g_rand_calls;

  auto maybe_dump = [&](int t) {
    if (dump_prefix.empty()) return;
    if (std::find(dumps.begin(), dumps.end(), t) == dumps.end()) return;
    Snapshot s = take_snapshot(map);
    char path[4096];
    snprintf(path, sizeof(path), "%s.t%d.snap", dump_prefix.c_str(), t);
    snapshot_write(path, s);
  };
  maybe_dump(0);

  double t_water = 0, t_grid = 0, t_wind = 0, t_freq = 0;
  uint64_t steps_water_top = 0, steps_wind = 0;
  for (int t = 1; t <= ticks; t++) {
    double a = now_s();
    if (dowater)
      for (int i = 0; i < NWATER; i++) {                                     // SoilMachine.cpp:287-298
        WaterParticle particle(map);
        while (true) {
          while (true) {
            if (!particle.move(map, vertexpool)) break;
            steps_water_top++;
            if (!particle.interact(map, vertexpool)) break;
          }
          if (!particle.flood(map, vertexpool)) break;
        }
      }
    double b = now_s();
    if (dowater) WaterParticle::seep(map, vertexpool);                       // SoilMachine.cpp:300-301
    double c = now_s();
    if (dowind)
      for (int i = 0; i < NWIND; i++) {                                      // SoilMachine.cpp:303-307
        WindParticle particle(map);
        while (true) {
          if (!particle.move(map, vertexpool)) break;
          steps_wind++;
          if (!particle.interact(map, vertexpool)) break;
        }
      }
    double d = now_s();
    if (dowater) {                                                           // SoilMachine.cpp:313-320
      WaterParticle::mapfrequency(map);
      WaterParticle::resetfrequency(map);
    }
    double e = now_s();
    t_water += b - a; t_grid += c - b; t_wind += d - c; t_freq += e - d;
    maybe_dump(t);
  }

  Snapshot s = take_snapshot(map);
  Digest dg = snapshot_digest(s);
  uint64_t rc = g_rand_calls;
  int next_rand = (rand)();
  // figures for the P2 envelope (tools/p2_envelope.py --reference): cells under standing water, its volume, nested particles
  // (every particle constructor draws twice, water.h:13 / wind.h:15: what the top-level constructors did not draw, nested ones did)
  uint64_t standing = 0; double water_volume = 0.0;
  {
    size_t off = 0;
    for (uint64_t c = 0; c < s.h.ncells; c++) {
      const uint32_t n = s.count[c];
      if (n && s.type[off + n - 1] == 0) { standing++; water_volume += s.size[off + n - 1]; }
      off += n;
    }
  }
  const uint64_t top_draws = 2ull * (uint64_t)ticks * ((dowater ? (uint64_t)NWATER : 0ull) + (dowind ? (uint64_t)NWIND : 0ull));
  const uint64_t nested = (rc - rand_calls_at_start - top_draws) / 2ull;
  if (!heights_out.empty()) {        // float64 heights, Layermap order x*dimy + y
    std::vector<double> hh(s.h.ncells);
    size_t off = 0;
    for (uint64_t c = 0; c < s.h.ncells; c++) { const uint32_t n = s.count[c]; hh[c] = n ? s.floor[off + n - 1] + s.size[off + n - 1] : 0.0; off += n; }
    FILE* f = fopen(heights_out.c_str(), "wb");
    if (f) { fwrite(hh.data(), 8, hh.size(), f); fclose(f); }
  }
  if (quiet) std::cout.rdbuf(old);
  printf("JSON {\"soil\": \"%s\", \"seed\": %d, \"dimx\": %d, \"dimy\": %d, \"scale\": %d, \"ticks\": %d, \"nwater\": %d, \"nwind\": %d, "
         "\"dowater\": %d, \"dowind\": %d, \"sumh\": %.17g, \"nsec\": %llu, \"typehash\": \"%016llx\", \"poolfree\": %llu, "
         "\"rand_calls\": %llu, \"next_rand\": %d, \"steps_water_top\": %llu, \"steps_wind\": %llu, "
         "\"t_init\": %.6f, \"t_water\": %.6f, \"t_grid\": %.6f, \"t_wind\": %.6f, \"t_freq\": %.6f, "
         "\"standing\": %llu, \"water_volume\": %.17g, \"nested\": %llu, \"rand_seed\": %lld, \"loaded\": %d}\n",
         soil.c_str(), SEED, SIZEX, SIZEY, SCALE, ticks, NWATER, NWIND, dowater, dowind, dg.sumh,
         (unsigned long long)dg.nsec, (unsigned long long)dg.typehash, (unsigned long long)map.pool.free.size(),
         (unsigned long long)rc, next_rand, (unsigned long long)steps_water_top, (unsigned long long)steps_wind,
         t_init, t_water, t_grid, t_wind, t_freq, (unsigned long long)standing, water_volume, (unsigned long long)nested, rand_seed, load_path.empty() ? 0 : 1);
  return 0;
}
