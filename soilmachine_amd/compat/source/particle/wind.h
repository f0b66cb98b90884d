// compat/source/particle/wind.h -- drop-in for wind.h:11-140 (WindParticle).
// `WindParticle particle(map)` counts a spawn request; the batch of NWIND particles runs on the device as soon as
// the host's wind loop has issued the last one (SoilMachine.cpp:303-307), in reference order. NWIND is read at every
// request (the GUI can only change it between two passes of the loop, where nothing is pending); whatever is still pending
// is flushed by the next observation of the map anyway (Layermap::flush).
#include "particle.h"

using namespace glm;

struct WindParticle : public Particle {
  WindParticle(Layermap& map) {                                                     // wind.h:13-22 (deferred)
    map.flush_water();
    if (++map.pending_wind >= NWIND) { map.flush_wind(); refresh(map); }
  }
  static void init() { frequency = new float[SIZEX * SIZEY]{0.0f}; }                // wind.h:24-26

  const vec3 pspeed = vec3(-2, 0, 1);
  vec3 speed = pspeed;
  double sediment = 0.0;
  double height = 0.0;
  double sheight = 0.0;
  ivec2 ipos;
  vec3 n;
  SurfType surface;
  SurfType contains;
  SurfParam param;
  const double gravity = 0.25;
  const double winddominance = 0.2;
  const double windfriction = 0.8;
  const double minsed = 0.0001;

  static float* frequency;
  static void refresh(Layermap& map) {                                              // host mirror of wind.h:48 for the GUI texture
    if (frequency) smx_read_frequency(map.handle(), NULL, NULL, frequency);
  }
  bool move(Layermap& map, Vertexpool<Vertex>& vertexpool) { map.present(vertexpool); return false; }   // wind.h:54-92 (runs in the batch; the call after the flush brings the pool up to date)
  bool interact(Layermap&, Vertexpool<Vertex>&) { return false; }                   // wind.h:94-136
};

float* WindParticle::frequency = NULL;

// WaterParticle::mapfrequency (water.h:358-365): device kernel, then refresh the host mirrors the GUI textures read
void WaterParticle::mapfrequency(Layermap& map) {
  map.flush();
  smx_map_frequency(map.handle());
  smx_read_frequency(map.handle(), frequency, track, WindParticle::frequency);
}
