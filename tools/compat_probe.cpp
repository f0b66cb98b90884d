// compat_probe -- a second host program on the drop-in class API (tools/build_host.sh builds it next to the unmodified
// SoilMachine.cpp): the queries and point modifiers that the reference declares (layermap.h:131-150) but its own main()
// never calls, so that the unmodified-host test cannot cover them: Layermap::height(vec2), normal(vec2) and its vertexpool
// variants, surface(ivec2), top(ivec2), add(ivec2, sec*) through pool.get, remove(ivec2, h), Particle::cascade.
// usage: compat_probe -SEED s -soil file -ticks t -o out.f64   (TEST INFRASTRUCTURE; needs a GPU to run)
#include <TinyEngine/TinyEngine>
#include <TinyEngine/camera>
#include <TinyEngine/parse>
#include <TinyEngine/image>

int SIZEX = 256;
int SIZEY = 256;
int SCALE = 80;
int SLICE = 2 * SCALE;
int NWIND = 250;
int NWATER = 250;
#define POOLSIZE 10000000
int SEED;

#include "source/include/vertexpool.h"
#include "source/include/scene.h"
#include "source/layermap.h"
#include "source/particle/water.h"
#include "source/particle/wind.h"
#include "source/io.h"

int main(int argc, char* args[]) {
  parse::get(argc, args);
  SEED = stoi(parse::option["SEED"]);
  srand(SEED);
  loadsoil(parse::option["soil"]);
  WaterParticle::init();
  WindParticle::init();
  Vertexpool<Vertex> vertexpool(SIZEX * SIZEY, 1);
  Layermap map(SEED, glm::ivec2(SIZEX, SIZEY), vertexpool);
  const int ticks = stoi(parse::option["ticks"]);
  for (int t = 0; t < ticks; t++) {                            // the host's tick, written against the same class API
    for (int i = 0; i < NWATER; i++) {
      WaterParticle particle(map);
      while (true) {
        while (particle.move(map, vertexpool) && particle.interact(map, vertexpool));
        if (!particle.flood(map, vertexpool)) break;
      }
    }
    WaterParticle::seep(map, vertexpool);
    for (int i = 0; i < NWIND; i++) {
      WindParticle particle(map);
      while (particle.move(map, vertexpool) && particle.interact(map, vertexpool));
    }
    WaterParticle::mapfrequency(map);                          // SoilMachine.cpp:314,319: the frequency map feeds back into the next tick
    WaterParticle::resetfrequency(map);
  }
  // point modifiers: a layer of type 1 on (5,7), 0.3 removed from (9,3), one Particle::cascade at (20.4, 11.6)
  map.add(glm::ivec2(5, 7), map.pool.get(0.0125, (SurfType)1));
  const double rest = map.remove(glm::ivec2(9, 3), 0.3);
  Particle::cascade(glm::vec2(20.4f, 11.6f), map, vertexpool, 1);
  std::vector<double> out;                                     // (floats widen exactly)
  out.push_back(rest);
  for (int k = 0; k < 500; k++) {
    const glm::vec2 pos(std::fmod(k * 37.77f, (float)(SIZEX - 2)) + 0.25f, std::fmod(k * 11.31f, (float)(SIZEY - 2)) + 0.5f);
    const glm::ivec2 ip = pos;
    const glm::vec3 n = map.normal(pos), nv = map.normal(pos, vertexpool), ni = map.normal(ip, vertexpool);
    const double h = map.height(pos);
    sec* top = map.top(ip);
    const double rec[] = {pos.x, pos.y, h, n.x, n.y, n.z, nv.x, nv.y, nv.z, ni.x, ni.y, ni.z,
                          (double)map.surface(ip), top ? (double)top->type : -1.0, top ? top->size : 0.0, top ? top->floor : 0.0, top ? top->saturation : 0.0};
    out.insert(out.end(), rec, rec + 17);
  }
  FILE* f = fopen(parse::option["o"].c_str(), "wb");
  fwrite(out.data(), 8, out.size(), f);
  fclose(f);
  printf("compat_probe: %zu doubles\n", out.size());
  return 0;
}
