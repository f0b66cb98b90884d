// compat/source/layermap.h -- drop-in for the reference's source/layermap.h (layermap.h:1-613).
//
// Same public surface (`sec`, `secpool`, `class Layermap` with dim / pool / section, height x2, normal x4, surface,
// add, remove, top, meshpool, update x2, slice, initialize, both constructors), but the run-length sediment
// columns live in HBM behind the C-ABI of include/soilmx.h: `Layermap` owns an `smx_ctx`. Host-side point queries
// (SoilMachine.cpp:238 calls map.height() 163 840 times at start-up) are served from a host mirror of the surface
// plane that is refreshed after device work, never by a per-call device round trip.
//
// The unmodified SoilMachine.cpp includes this file through the quoted path "source/layermap.h"
// (INTEGRATION.md shows the include-path arrangement); it reads the host's globals SIZEX, SIZEY, SCALE, SLICE,
// NWIND, NWATER, SEED and the POOLSIZE macro exactly like the reference header does.
#pragma once
#include <soilmx.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <iostream>
#include <string>
#include <vector>

using namespace glm;
using namespace std;

#include "surface.h"

struct sec {                                  // layermap.h:37-62 (host-side value object; the device owns the real lists)
  sec* next = NULL;
  sec* prev = NULL;
  SurfType type = 0;
  double size = 0.0f;
  double floor = 0.0f;
  double saturation = 0.0f;
  sec() {}
  sec(double s, SurfType t) { size = s; type = t; }
  void reset() { next = NULL; prev = NULL; type = 0; size = 0.0f; floor = 0.0f; saturation = 0.0f; }
};

class Layermap;

class secpool {                               // layermap.h:64-119: capacity POOLSIZE, `free.size()` for the GUI (SoilMachine.cpp:116)
 public:
  int size = 0;
  sec* start = NULL;
  struct freelist_view {
    smx_ctx** ctx = nullptr;
    size_t size() const {
      smx_counters c;
      if (!ctx || !*ctx || smx_get_counters_sized(*ctx, &c, sizeof(c)) != 0) return 0;
      return (size_t)c.pool_free;
    }
    bool empty() const { return size() == 0; }
  } free;
  secpool() {}
  void reserve(const int N) { size = N; }
  template <typename... Args>
  sec* get(Args&&... args) { return new sec(std::forward<Args>(args)...); }   // a value handed to Layermap::add
  void unget(sec* E) { delete E; }
  void reset() {}
};

class Layermap {
 private:
  smx_ctx* ctx = nullptr;
  std::vector<double> hmirror;                // Layermap::height(ivec2) for every cell (cell order x*dim.y+y)
  std::vector<uint32_t> smirror;              // Layermap::surface(ivec2)
  std::vector<float> nmirror;                 // Layermap::normal(ivec2), xyz per cell (filled on first use after a change)
  bool dirty = true, nstale = true;
  sec topview;                                // what top() hands out

  void check(int rc, const char* what) {
    if (rc != 0) { cout << "soilmx: " << what << " failed: " << (ctx ? smx_last_error(ctx) : "no context") << endl; exit(1); }
  }
  void refresh() {
    if (!dirty) return;
    flush();
    hmirror.resize((size_t)dim.x * dim.y); smirror.resize((size_t)dim.x * dim.y);
    check(smx_read_heights(ctx, hmirror.data()), "smx_read_heights");
    check(smx_read_surface(ctx, smirror.data()), "smx_read_surface");
    dirty = false; nstale = true;
  }

 public:
  ivec2 dim;
  secpool pool;
  uint* section = NULL;

  // ---- deferred particle batches: the host's per-particle calls only count; the device runs the whole batch
  //      in reference order when the next phase starts (INTEGRATION.md "How the unmodified loop batches")
  int pending_water = 0, pending_wind = 0;
  smx_ctx* handle() { return ctx; }
  void touch() { dirty = true; vstale = true; }
  // The reference keeps the vertex pool current edit by edit (add/remove/cascade call update(ivec2, vertexpool)); here the
  // edits happen in device batches, so the pool is brought up to date once per batch, at the first host call after it that
  // carries the pool (WaterParticle::seep(map, pool), the particle methods): the renderer and the exporters of the host's
  // own io.h (io.h:234-252) read it without calling update(). SMX_LAZY_VERTICES=1 leaves it to explicit update() calls.
  bool vstale = false;
  void present(Vertexpool<Vertex>& vertexpool) {
    static const bool lazy = [] { const char* e = getenv("SMX_LAZY_VERTICES"); return e && atoi(e) != 0; }();
    if (section != NULL && vstale && !lazy) update(vertexpool);
  }
  void flush_water() {
    if (pending_water) { check(smx_tick_water(ctx, pending_water), "smx_tick_water"); pending_water = 0; dirty = true; vstale = true; }
  }
  void flush_wind() {
    if (pending_wind) { check(smx_tick_wind(ctx, pending_wind), "smx_tick_wind"); pending_wind = 0; dirty = true; vstale = true; }
  }
  void flush() { flush_water(); flush_wind(); }
  void upload_tables() {                      // soils[] and SCALE are live-editable from the GUI (SoilMachine.cpp:118,167-186)
    std::vector<smx_soil> t(soils.size());
    for (size_t i = 0; i < soils.size(); i++) {
      const SurfParam& p = soils[i];
      t[i] = smx_soil{(uint32_t)p.transports, (uint32_t)p.erodes, (uint32_t)p.cascades, (uint32_t)p.abrades, p.density, p.porosity,
                      p.solubility, p.equrate, p.friction, p.erosionrate, p.maxdiff, p.settling, p.suspension, p.abrasion};
    }
    check(smx_set_soils(ctx, t.data(), (int32_t)t.size()), "smx_set_soils");
    check(smx_set_scale(ctx, SCALE), "smx_set_scale");
  }

  // ---- queries
  double height(ivec2 pos) { refresh(); return hmirror[(size_t)pos.x * dim.y + pos.y]; }          // layermap.h:422-425
  double height(vec2 pos) {                                       // layermap.h:427-439: evaluated on the device (k_bilinear), bit-identical
    flush();
    const float xy[2] = {pos.x, pos.y};
    double h = 0.0;
    check(smx_heights_bilinear(ctx, xy, 1, &h), "smx_heights_bilinear");
    return h;
  }
  SurfType surface(ivec2 pos) { refresh(); return smirror[(size_t)pos.x * dim.y + pos.y]; }        // layermap.h:417-420
  // Layermap::normal(ivec2) (layermap.h:341-377) is a device kernel (k_normals behind smx_normals: the same code the particles
  // use); the host serves point queries from a mirror of that plane, refreshed after device work like the height mirror
  vec3 normal(ivec2 pos) {
    refresh();
    if (nstale) { nmirror.resize(3 * (size_t)dim.x * dim.y); check(smx_normals(ctx, nmirror.data()), "smx_normals"); nstale = false; }
    const float* q = &nmirror[3 * ((size_t)pos.x * dim.y + pos.y)];
    return vec3(q[0], q[1], q[2]);
  }
  // layermap.h:379-392: the four corner normals blended with the bilinear weights (float; corner order (0,0) (1,0) (0,1) (1,1)
  // with the weight pairing of the reference, i.e. crossed like height(vec2))
  vec3 normal(vec2 pos) {
    const ivec2 cell = floor(pos);
    const vec2 f = fract(pos);
    const float wx[2] = {1.0f - f.x, f.x}, wy[2] = {1.0f - f.y, f.y};
    const int corner[4][4] = {{0, 0, 0, 0}, {1, 0, 0, 1}, {0, 1, 1, 0}, {1, 1, 1, 1}};   // {dx, dy, index into wx, index into wy}
    vec3 acc = vec3(0);
    for (int k = 0; k < 4; k++) acc += wx[corner[k][2]] * wy[corner[k][3]] * normal(cell + ivec2(corner[k][0], corner[k][1]));
    return acc;
  }
  vec3 normal(ivec2 pos, Vertexpool<Vertex>&) { return normal(pos); }                               // layermap.h:394-398
  vec3 normal(vec2 pos, Vertexpool<Vertex>&) { return normal(pos); }                                // layermap.h:400-413

  // ---- modifiers (point operations on the device; host code outside the tick loop)
  void add(ivec2 pos, sec* E) {                                                                     // layermap.h:230-307
    if (E == NULL) return;
    flush();
    check(smx_add(ctx, pos.x, pos.y, E->size, (uint32_t)E->type), "smx_add");
    pool.unget(E);
    dirty = true;
  }
  double remove(ivec2 pos, double h) {                                                              // layermap.h:310-339
    flush();
    double rem = 0.0;
    check(smx_remove(ctx, pos.x, pos.y, h, &rem), "smx_remove");
    dirty = true;
    return rem;
  }
  sec* top(ivec2 pos) {                                                                             // layermap.h:150-152 (a copy, not a list node)
    flush();
    uint32_t ty = 0; int32_t empty = 0;
    check(smx_top(ctx, pos.x, pos.y, &ty, &topview.size, &topview.floor, &topview.saturation, &empty), "smx_top");
    if (empty) return NULL;
    topview.type = ty;
    return &topview;
  }

  // ---- meshing / visualisation (render side): one whole-map pass from the device planes
  void meshpool(Vertexpool<Vertex>& vertexpool) {                                                   // layermap.h:443-473
    if (section != NULL) { vertexpool.unsection(section); vertexpool.indices.clear(); }
    section = vertexpool.section(dim.x * dim.y, 0, glm::vec3(0));
    update(vertexpool);
    // two triangles per grid quad, vertices numbered in cell order (layermap.h:458-469): (q, q+1, q+dim.y) and (q+dim.y, q+1, q+dim.y+1)
    vertexpool.indices.reserve((size_t)6 * (dim.x - 1) * (dim.y - 1));
    const int tri[6] = {0, 1, dim.y, dim.y, 1, dim.y + 1};
    for (int q = 0; q < (dim.x - 1) * dim.y; q++) {
      if (q % dim.y == dim.y - 1) continue;                    // last cell of a row: no quad to its right
      for (int t = 0; t < 6; t++) vertexpool.indices.push_back(q + tri[t]);
    }
    vertexpool.resize(section, vertexpool.indices.size());
    vertexpool.index();
    vertexpool.update();
  }
  // one column (host-driven edits), with the SLICE cut of layermap.h:477-510 (it needs the buried sections, which live on the device:
  // smx_fill_vertex_cut runs the whole-map kernel on the one block that holds the column and brings the 44-byte record back)
  void update(ivec2 p, Vertexpool<Vertex>& vertexpool) {                                            // layermap.h:475-549
    if (section == NULL) return;
    flush();
    static_assert(sizeof(Vertex) == 44, "Vertex is the 44-byte record smx_fill_vertex_cut writes");
    std::vector<float> col(4 * soils.size());
    for (size_t k = 0; k < soils.size(); k++) { col[4 * k] = soils[k].color.x; col[4 * k + 1] = soils[k].color.y; col[4 * k + 2] = soils[k].color.z; col[4 * k + 3] = soils[k].color.w; }
    const size_t c = (size_t)p.x * dim.y + p.y;
    check(smx_fill_vertex_cut(ctx, col.data(), (int32_t)soils.size(), 0, (double)SLICE, p.x, p.y, vertexpool.get(section, (int)c)), "smx_fill_vertex_cut");
  }
  // The whole map in one device pass (smx_fill_vertices_cut writes the 44-byte records straight into the pool's memory). A
  // section is one bucket of contiguous vertices -- in the host tree's own pool (source/include/vertexpool.h:97-100: start of
  // the persistently mapped buffer + baseVert + k) as in any stand-in -- so get(section, 0) is the destination.
  void update(Vertexpool<Vertex>& vertexpool) {                                                     // layermap.h:551-555
    if (section == NULL) return;
    refresh();
    static_assert(sizeof(Vertex) == 44, "Vertex is the 44-byte record smx_fill_vertices writes");
    std::vector<float> col(4 * soils.size());
    for (size_t k = 0; k < soils.size(); k++) { col[4 * k] = soils[k].color.x; col[4 * k + 1] = soils[k].color.y; col[4 * k + 2] = soils[k].color.z; col[4 * k + 3] = soils[k].color.w; }
    check(smx_fill_vertices_cut(ctx, col.data(), (int32_t)soils.size(), 0, (double)SLICE, vertexpool.get(section, 0)), "smx_fill_vertices_cut");   // SLICE cut: layermap.h:477-510
    vstale = false;
  }
  void slice(Vertexpool<Vertex>& vertexpool, double s = SCALE) {                                    // layermap.h:557-613
    if (section == NULL) return;
    refresh();
    std::vector<float> col(4 * soils.size());
    for (size_t k = 0; k < soils.size(); k++) { col[4 * k] = soils[k].color.x; col[4 * k + 1] = soils[k].color.y; col[4 * k + 2] = soils[k].color.z; col[4 * k + 3] = soils[k].color.w; }
    check(smx_fill_vertices_cut(ctx, col.data(), (int32_t)soils.size(), 1, s, vertexpool.get(section, 0)), "smx_fill_vertices_cut");
  }

  // the device context for a map of `_dim` cells. POOLSIZE is the reference's capacity (SoilMachine.cpp:16); maps that
  // cannot even hold their initial terrain in it (>= 2048^2 with two layers -- the reference would silently drop
  // columns there, layermap.h:92-95) get 4 sections per cell instead, and say so.
  void create_context(ivec2 _dim) {
    smx_config cfg;
    uint64_t cap = (uint64_t)POOLSIZE;
    const uint64_t need = 4ull * (uint64_t)_dim.x * (uint64_t)_dim.y;
    if (cap < (uint64_t)_dim.x * _dim.y * (layers.size() ? layers.size() : 1)) {
      cout << "soilmx: POOLSIZE " << cap << " cannot hold a " << _dim.x << "x" << _dim.y << " map with " << layers.size() << " layers: using " << need << endl;
      cap = need;
    }
    cfg.dimx = _dim.x; cfg.dimy = _dim.y; cfg.scale = SCALE; cfg.device = 0; cfg.pool_capacity = cap;
    const char* e = getenv("SMX_ENGINE");
    cfg.engine = (e && std::string(e) == "serial") ? SMX_ENGINE_SERIAL : (e && std::string(e) == "batched") ? SMX_ENGINE_BATCHED
               : (e && std::string(e) == "relaxed") ? SMX_ENGINE_RELAXED : SMX_ENGINE_SPECULATIVE;
    cfg.reserved = 0;
    int rc = smx_create(&cfg, &ctx);
    if (rc != 0) { cout << "soilmx: smx_create failed: " << (ctx ? smx_last_error(ctx) : "") << endl; exit(1); }
    upload_tables();
  }

  void initialize(int SEED, ivec2 _dim) {                                                           // layermap.h:163-216
    if (ctx && (_dim.x != dim.x || _dim.y != dim.y)) {       // the GUI re-seeds after SIZEX/SIZEY edits (SoilMachine.cpp:111-114): the
      uint32_t ring[31], idx = 0; uint64_t calls = 0;          // reference reallocates `dat` (layermap.h:165-172); the libc stream goes on
      check(smx_get_rand_state(ctx, ring, &idx, &calls), "smx_get_rand_state");
      smx_destroy(ctx); ctx = nullptr;
      create_context(_dim);
      check(smx_set_rand_state(ctx, ring, idx, calls), "smx_set_rand_state");
      hmirror.clear(); smirror.clear(); nmirror.clear();
    }
    dim = _dim;
    std::vector<smx_layer> L(layers.size());
    for (size_t l = 0; l < layers.size(); l++)
      L[l] = smx_layer{(uint32_t)layers[l].type, layers[l].min, layers[l].bias, layers[l].scale, layers[l].octaves,
                       layers[l].lacunarity, layers[l].gain, layers[l].frequency};
    check(smx_initialize(ctx, SEED, L.data(), (int32_t)L.size()), "smx_initialize");
    pending_water = pending_wind = 0;
    dirty = true;
  }

  Layermap(int SEED, ivec2 _dim) {                                                                  // layermap.h:219-222
    pool.reserve(POOLSIZE);
    pool.free.ctx = &ctx;
    dim = _dim;
    create_context(_dim);
    check(smx_srand(ctx, (uint32_t)SEED), "smx_srand");        // the host has just called srand(SEED) (SoilMachine.cpp:41)
    initialize(SEED, _dim);
  }
  Layermap(int SEED, ivec2 _dim, Vertexpool<Vertex>& vertexpool) : Layermap(SEED, _dim) { meshpool(vertexpool); }   // :224-226
  ~Layermap() { if (ctx) smx_destroy(ctx); }
  Layermap(const Layermap&) = delete;
  Layermap& operator=(const Layermap&) = delete;
};
