// GL-free stand-ins for the render-side types that appear in every hot-path signature of the
// reference (TEST INFRASTRUCTURE ONLY; used by oracle/ref_driver.cpp to build oracle/_ref).
//
//  * Vertex / Vertexpool<T>: replaces source/include/vertexpool.h:9-59,94-342 (persistently mapped
//    GL VBO + multi-draw-indirect). Only the members the hot path touches are kept:
//    section / unsection / indices / resize / index / update / fill / get (layermap.h:443-613).
//    `fill` keeps the 44-byte placement-new store so Layermap::update() costs what it costs in
//    the real application (SURVEY.md §8d "as-is" CPU line).
//  * image:: and SDL_Surface: only so that source/io.h:234-252 (PNG exporters) compiles.
#pragma once
#include <glm/glm.hpp>
#include <vector>
#include <new>
#include <utility>

struct Vertex {
  Vertex(glm::vec3 p, glm::vec3 n, glm::vec4 c, int i) {
    position[0] = p.x; position[1] = p.y; position[2] = p.z;
    normal[0] = n.x; normal[1] = n.y; normal[2] = n.z;
    color[0] = c.x; color[1] = c.y; color[2] = c.z; color[3] = c.w;
    index = i;
  }
  Vertex() {}
  float position[3];
  float normal[3];
  float color[4];
  float index;
};

template <typename T>
class Vertexpool {
 public:
  std::vector<T> store;
  std::vector<unsigned int> indices;
  unsigned int base = 0;
  Vertexpool() {}
  Vertexpool(int k, int n) { store.resize((size_t)k * (size_t)n); }
  unsigned int* section(const int, const int = 0, glm::vec3 = glm::vec3(0)) { return &base; }
  void unsection(unsigned int*) {}
  T* get(unsigned int* ind, int k) { return store.data() + (*ind + k); }
  template <typename... Args>
  void fill(unsigned int* ind, int k, Args&&... args) {
    T* place = get(ind, k);
    new (place) T(std::forward<Args>(args)...);
  }
  void resize(const unsigned int*, const int) {}
  void index() {}
  void update() {}
};

struct SDL_Surface { int w, h; };
namespace image {
template <typename F> SDL_Surface* make(F, glm::ivec2) { return nullptr; }
inline void save(SDL_Surface*, std::string) {}
}  // namespace image
