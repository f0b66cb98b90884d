// compat/.../lbmwind.h -- the reference's GLSL lattice-Boltzmann wind VISUALISATION (lbmwind.h:12-240) is out of
// scope (it never feeds back into WindParticle). Only the names SoilMachine.cpp touches.
#pragma once
namespace lbmw {
using namespace glm;
const int NX = 64;
const int NY = 40;
const int NZ = 64;
vec4 scale = vec4(1);
float* boundary = nullptr;
Buffer* b = nullptr;
bool updatewind = false;
bool renderwind = false;
inline void initialize() { boundary = new float[NX * NY * NZ]{0.0f}; b = new Buffer(); }
inline void quit() { delete[] boundary; delete b; }
inline void update() {}
inline void render(mat4) {}
}  // namespace lbmw
