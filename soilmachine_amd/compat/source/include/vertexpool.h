// compat/source/include/vertexpool.h -- GL-free stand-in for the reference's vertex pool
// (source/include/vertexpool.h:9-59,94-342). The hot path only passes the pool around; the host reads it back in
// the PNG exporters (io.h:234-252). Members kept: section / unsection / indices / resize / index / update / fill /
// get / render.
#pragma once
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
  void fill(unsigned int* ind, int k, Args&&... args) { new (get(ind, k)) T(std::forward<Args>(args)...); }
  void resize(const unsigned int*, const int) {}
  void index() {}
  void update() {}
  void render(int = 0) {}
};
