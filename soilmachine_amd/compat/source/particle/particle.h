// compat/source/particle/particle.h -- drop-in for particle.h:14-103 (Particle base + sediment cascade).
#ifndef SOILMX_COMPAT_PARTICLE
#define SOILMX_COMPAT_PARTICLE

using namespace glm;

struct Particle {
  vec2 pos;
  vec2 speed = vec2(0);
  bool isalive = true;

  bool move(Layermap& map);
  bool interact(Layermap& map, Vertexpool<Vertex>& vertexpool);

  // Particle::cascade (particle.h:24-101) as a device point operation
  static void cascade(vec2 pos, Layermap& map, Vertexpool<Vertex>&, int transferloop = 0) {
    map.flush();
    if (smx_particle_cascade(map.handle(), pos.x, pos.y, transferloop) != 0) { cout << "soilmx: " << smx_last_error(map.handle()) << endl; exit(1); }
    map.touch();
  }
};

#endif
