// lbm_oracle.cpp -- CPU restatement of the reference's LBM wind: TEST INFRASTRUCTURE ONLY.
//
// The reference runs a D3Q19 two-relaxation-time lattice Boltzmann solver as OpenGL compute shaders
// (source/include/lbmwind/shader/LBM/{lbm,init,collide,stream}.cs, driven by lbmwind.h:75-197) plus a tracer kernel
// (shader/move.cs). This file follows the shaders statement by statement in float32, one fixed evaluation order:
//   dot(a, b) = (a.x*b.x + a.y*b.y) + a.z*b.z, products left to right, no contraction (-ffp-contract=off).
//
// PARITY PINNED (round 3): the reference's own shader text, compiled as C++ (oracle/lbm_ref.cpp + oracle/glsl2cpp.py +
// oracle/shim/glsl_cs.h -> oracle/_ref/liblbm_ref.so), reproduces this restatement bit for bit on every buffer after every step
// (tests/test_lbm.py, live where /root/reference exists; against tests/golden/lbm_ref.npz -- the shaders' own outputs --
// everywhere). What no CPU build can pin: a GLSL compiler may contract or reassociate float arithmetic and fold constants
// at another precision; both builds evaluate every expression as written, in float, no contraction.
// Where the reference's stream.cs races with itself (a face cell's distributions are written by its neighbours' push
// AND by its own driving-force overwrite in the same dispatch, stream.cs:9-34), the overwrite wins here: that is the wet-node
// boundary the shader's comment states (the shim's FBUF takes the same decision for the compiled shaders).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {
constexpr int Q = 19;
// lbm.cs:60-80
const int CX[Q] = {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 0, 0, 1, -1, 1, -1, 0, 0};
const int CY[Q] = {0, 0, 0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 1, -1};
const int CZ[Q] = {0, 0, 0, 0, 0, 1, -1, 0, 0, 1, -1, 1, -1, 0, 0, -1, 1, -1, 1};
const int CP[Q] = {0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15, 18, 17};   // lbm.cs:82-87

struct Consts { float w[Q], cs2, cs4, force[3], omega_plus, omega_minus, grav; };
Consts make_consts() {
  Consts k;
  k.w[0] = 1.0f / 3.0f;                                        // lbm.cs:53-58
  for (int q = 1; q < 7; q++) k.w[q] = 1.0f / 18.0f;
  for (int q = 7; q < Q; q++) k.w[q] = 1.0f / 36.0f;
  const float cs = 1.0f / sqrtf(3.0f);                        // lbm.cs:89-91
  k.cs2 = 1.0f / cs / cs;
  k.cs4 = 1.0f / cs / cs / cs / cs;
  k.force[0] = 0.05f * -2.0f; k.force[1] = 0.05f * 0.0f; k.force[2] = 0.05f * 1.0f;   // lbm.cs:36
  const float tau = 0.56f, lambda = 0.25f;                    // collide.cs:7,31-33
  k.omega_plus = 1.0f / tau;
  k.omega_minus = 1.0f / (lambda / (1.0f / k.omega_plus - 0.5f) + 0.5f);
  k.grav = 1.0f * 0.0001f;                                    // collide.cs:8,19: dt*0.0001
  return k;
}
inline float dot3(const float* a, float bx, float by, float bz) { float t = a[0] * bx; t = t + a[1] * by; t = t + a[2] * bz; return t; }
// lbm.cs:95-105
inline float equilibrium(const Consts& k, int q, float rho, const float* v) {
  const float d = dot3(v, (float)CX[q], (float)CY[q], (float)CZ[q]);
  float eq = 0.0f;
  eq += k.w[q] * rho;
  eq += k.w[q] * rho * d * k.cs2;
  eq += k.w[q] * rho * (d * d) * 0.5f * k.cs4;
  eq -= k.w[q] * rho * dot3(v, v[0], v[1], v[2]) * 0.5f * k.cs2;
  return eq;
}
}  // namespace

struct lo_ctx {
  int nx, ny, nz; size_t n;
  Consts k;
  std::vector<float> F, FPROP, B, RHO, V;                     // F/FPROP [cell*Q+q], V vec4 per cell (lbm.cs:8-26)
};

extern "C" {
lo_ctx* lo_create(int nx, int ny, int nz) {
  lo_ctx* c = new lo_ctx;
  c->nx = nx; c->ny = ny; c->nz = nz; c->n = (size_t)nx * ny * nz; c->k = make_consts();
  c->F.assign(c->n * Q, 0.f); c->FPROP.assign(c->n * Q, 0.f); c->B.assign(c->n, 0.f); c->RHO.assign(c->n, 0.f); c->V.assign(c->n * 4, 0.f);
  return c;
}
void lo_destroy(lo_ctx* c) { delete c; }
void lo_set_boundary(lo_ctx* c, const float* b) { memcpy(c->B.data(), b, c->n * 4); }   // lbmw::b->fill (SoilMachine.cpp:239)

// init.cs:9-24
void lo_initialize(lo_ctx* c) {
  const float zero[3] = {0.f, 0.f, 0.f};
  for (size_t i = 0; i < c->n; i++) {
    for (int q = 0; q < Q; q++) {
      c->F[i * Q + q] = equilibrium(c->k, q, 1.0f, c->k.force);
      if (c->B[i] > 0) c->F[i * Q + q] = equilibrium(c->k, q, 1.0f, zero);
    }
    float rho = 0.f, v[3] = {0.f, 0.f, 0.f};
    for (int q = 0; q < Q; q++) rho += c->F[i * Q + q];                              // getRho lbm.cs:109-116
    for (int q = 0; q < Q; q++) { const float f = c->F[i * Q + q]; v[0] += f * (float)CX[q]; v[1] += f * (float)CY[q]; v[2] += f * (float)CZ[q]; }   // getV :120-127
    c->RHO[i] = rho;
    c->V[4 * i] = v[0] / rho; c->V[4 * i + 1] = v[1] / rho; c->V[4 * i + 2] = v[2] / rho; c->V[4 * i + 3] = 1.0f;
  }
}

// lbmw::update (lbmwind.h:174-188) without the tracers: collide.cs, then stream.cs
void lo_step(lo_ctx* c, int steps) {
  const Consts& k = c->k;
  const float zero[3] = {0.f, 0.f, 0.f};
  const int NX = c->nx, NY = c->ny, NZ = c->nz;
  for (int s = 0; s < steps; s++) {
    for (size_t i = 0; i < c->n; i++) {                       // collide.cs:10-56
      const float* F = &c->F[i * Q];
      float rho = 0.f, v[3] = {0.f, 0.f, 0.f};
      for (int q = 0; q < Q; q++) rho += F[q];
      for (int q = 0; q < Q; q++) { v[0] += F[q] * (float)CX[q]; v[1] += F[q] * (float)CY[q]; v[2] += F[q] * (float)CZ[q]; }
      v[0] = v[0] / rho; v[1] = v[1] / rho; v[2] = v[2] / rho;
      const float den = 2.0f * rho;                           // _v += dt*0.0001*c[4]/(2.0f*_rho), c[4] = (0,-1,0)
      v[0] += (k.grav * 0.0f) / den; v[1] += (k.grav * -1.0f) / den; v[2] += (k.grav * 0.0f) / den;
      c->RHO[i] = rho;
      c->V[4 * i] = v[0]; c->V[4 * i + 1] = v[1]; c->V[4 * i + 2] = v[2]; c->V[4 * i + 3] = 0.0f;
      float ffeq[Q];
      for (int q = 0; q < Q; q++) ffeq[q] = equilibrium(k, q, rho, v);
      for (int q = 0; q < Q; q++) {
        const float f_plus = 0.5f * (F[q] + F[CP[q]]), f_minus = 0.5f * (F[q] - F[CP[q]]);
        const float feq_plus = 0.5f * (ffeq[q] + ffeq[CP[q]]), feq_minus = 0.5f * (ffeq[q] - ffeq[CP[q]]);
        float out = F[q] - k.omega_plus * (f_plus - feq_plus) - k.omega_minus * (f_minus - feq_minus);
        if (c->B[i] > 0.0f) out = equilibrium(k, q, 1.0f, zero);
        c->FPROP[i * Q + q] = out;
      }
    }
    for (int x = 0; x < NX; x++)                              // stream.cs:9-21: push; distributions without a source keep their old value
      for (int y = 0; y < NY; y++)
        for (int z = 0; z < NZ; z++) {
          const size_t i = ((size_t)x * NY + y) * NZ + z;
          for (int q = 0; q < Q; q++) {
            const int ax = x + CX[q], ay = y + CY[q], az = z + CZ[q];
            if (ax < 0 || ax >= NX || ay < 0 || ay >= NY || az < 0 || az >= NZ) continue;
            c->F[(((size_t)ax * NY + ay) * NZ + az) * Q + q] = c->FPROP[i * Q + q];
          }
        }
    for (int x = 0; x < NX; x++)                              // stream.cs:26-36: driving force on five faces (not y == 0)
      for (int y = 0; y < NY; y++)
        for (int z = 0; z < NZ; z++)
          if (y == NY - 1 || x == 0 || x == NX - 1 || z == 0 || z == NZ - 1) {
            const size_t i = ((size_t)x * NY + y) * NZ + z;
            for (int q = 0; q < Q; q++) c->F[i * Q + q] = equilibrium(k, q, 1.0f, k.force);
          }
  }
}
void lo_read(lo_ctx* c, float* rho, float* v4, float* f) {
  if (rho) memcpy(rho, c->RHO.data(), c->n * 4);
  if (v4) memcpy(v4, c->V.data(), c->n * 16);
  if (f) memcpy(f, c->F.data(), c->n * Q * 4);
}
void lo_write_f(lo_ctx* c, const float* f) { memcpy(c->F.data(), f, c->n * Q * 4); }
// move.cs:27-52 on `n` tracers (vec4 each), in place; V as it stands
void lo_move(lo_ctx* c, float* pos4, int n) {
  const int NY = c->ny, NZ = c->nz;
  for (int i = 0; i < n; i++) {
    float* P = pos4 + 4 * (size_t)i;
    int p[4], nn[4]; float w[4];
    for (int d = 0; d < 4; d++) { p[d] = (int)P[d]; nn[d] = p[d] + 1; w[d] = P[d] - (float)p[d]; }
    auto cl = [](int a, int n) { return a < 0 ? 0 : (a > n - 1 ? n - 1 : a); };   // out-of-lattice samples: nearest cell (soilmx.h smx_lbm_move)
    auto V = [&](int x, int y, int z) { return &c->V[4 * (((size_t)cl(x, c->nx) * NY + cl(y, NY)) * NZ + cl(z, NZ))]; };
    const float *v000 = V(p[0], p[1], p[2]), *v100 = V(nn[0], p[1], p[2]), *v010 = V(p[0], nn[1], p[2]), *v001 = V(p[0], p[1], nn[2]),
                *v110 = V(nn[0], nn[1], p[2]), *v101 = V(nn[0], p[1], nn[2]), *v011 = V(p[0], nn[1], nn[2]), *v111 = V(nn[0], nn[1], nn[2]);
    for (int d = 0; d < 4; d++) {
      const float v00 = (1.0f - w[0]) * v000[d] + w[0] * v100[d], v01 = (1.0f - w[0]) * v001[d] + w[0] * v101[d];
      const float v10 = (1.0f - w[0]) * v010[d] + w[0] * v110[d], v11 = (1.0f - w[0]) * v011[d] + w[0] * v111[d];
      const float v0 = (1.0f - w[1]) * v00 + w[1] * v10, v1 = (1.0f - w[1]) * v01 + w[1] * v11;
      P[d] += (1.0f - w[2]) * v0 + w[2] * v1;
    }
  }
}
}  // extern "C"
