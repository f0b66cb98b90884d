// compat/source/particle/water.h -- drop-in for water.h:9-373 (WaterParticle).
//
// How the unmodified host loop batches (SoilMachine.cpp:287-301): `WaterParticle particle(map)` only counts a
// spawn request -- the device draws the particle's two rand() values itself, in the same stream position the
// reference would -- and move()/interact()/flood() return false so the host's while loops fall straight through.
// `WaterParticle::seep(map, vertexpool)` (the full-grid pass the host calls right after the particle loop) flushes:
// smx_tick_water(pending) then smx_grid_pass, both in reference order on the device.
#include "particle.h"

struct WaterParticle : public Particle {
  WaterParticle(Layermap& map) { map.flush_wind(); map.pending_water++; }          // water.h:11-19 (deferred)

  static void init() {                                                             // water.h:21-24
    frequency = new float[SIZEX * SIZEY]{0.0f};
    track = new float[SIZEX * SIZEY]{0.0f};
  }

  double volume = 1.0;
  double sediment = 0.0;
  const double minvol = 0.01;
  double evaprate = 0.001;
  static double volumeFactor;
  int spill = 3;
  ivec2 ipos;
  vec3 n;
  SurfParam param;
  SurfType surface;
  SurfType contains;

  bool move(Layermap&, Vertexpool<Vertex>&) { return false; }                       // water.h:43-73   (runs in the batch)
  bool interact(Layermap&, Vertexpool<Vertex>&) { return false; }                   // water.h:75-121
  bool flood(Layermap&, Vertexpool<Vertex>&) { return false; }                      // water.h:123-145 (always false there too)

  static void cascade(vec2 pos, Layermap& map, Vertexpool<Vertex>&, int spill = 0) {   // water.h:151-283
    map.flush();
    ivec2 ipos = pos;
    if (smx_water_cascade(map.handle(), ipos.x, ipos.y, spill) != 0) { cout << "soilmx: " << smx_last_error(map.handle()) << endl; exit(1); }
    map.touch();
  }
  static void seep(vec2 pos, Layermap& map, Vertexpool<Vertex>&) {                  // water.h:285-333
    map.flush();
    ivec2 ipos = pos;
    if (smx_seep(map.handle(), ipos.x, ipos.y) != 0) { cout << "soilmx: " << smx_last_error(map.handle()) << endl; exit(1); }
    map.touch();
  }
  static void seep(Layermap& map, Vertexpool<Vertex>& vertexpool) {                 // water.h:335-343: flush + grid pass
    map.flush();
    if (smx_grid_pass(map.handle()) != 0) { cout << "soilmx: " << smx_last_error(map.handle()) << endl; exit(1); }
    map.touch();
    map.present(vertexpool);
  }

  static float* frequency;
  static float* track;

  static void resetfrequency(Layermap& map) {                                       // water.h:353-356
    map.flush();
    smx_reset_frequency(map.handle());
    for (int i = 0; i < map.dim.x * map.dim.y; i++) track[i] = 0.0f;
  }
  static void mapfrequency(Layermap& map);                                          // water.h:358-365 (defined in wind.h: refreshes both mirrors)
};

double WaterParticle::volumeFactor = 0.015;
float* WaterParticle::frequency = NULL;
float* WaterParticle::track = NULL;
