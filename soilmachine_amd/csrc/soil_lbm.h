// soil_lbm.h -- the reference's LBM wind as HIP kernels for gfx950, behind smx_lbm_* (include/soilmx.h).
//
// Reference: source/include/lbmwind/lbmwind.h:75-197 drives four OpenGL compute shaders --
// shader/LBM/{init,collide,stream}.cs (+ lbm.cs: D3Q19 velocity set, equilibrium, moments) and shader/move.cs (tracers).
// A D3Q19 two-relaxation-time lattice Boltzmann solver on NX x NY x NZ cells; it is visual only (it never feeds
// WindParticle: wind.h:29,43-44 uses a constant prevailing speed).
//
// MI355X design. The reference keeps distributions as F[cell*19+q] (array of structures) and runs collide and stream as
// two dispatches over two buffers: 4 x 76 B of traffic per cell and step, lanes 76 B apart. Here
//   * structure of arrays f[q][cell], cell = (x*NY+y)*NZ+z as in the reference, so the 64 lanes of a wavefront (consecutive
//     z) read 256 contiguous bytes per direction;
//   * ONE kernel per step: read the cell's 19 values once, moments, TRT collision, and push the 19 results straight into the
//     neighbours' slots of the second lattice (ping-pong) -- 19 reads + 19 writes + mask + rho + v = 176 B per cell and step,
//     the minimum for a two-lattice scheme: the kernel is an HBM stream;
//   * the reference's in-place semantics are kept exactly: a distribution whose source lies outside the domain keeps its old
//     value (the cell copies it across itself), the five driven faces are overwritten with the forcing equilibrium AFTER
//     streaming (neighbours do not push into them, the face cell writes all 19 itself) -- the order the shader's race
//     (stream.cs:9-36) is meant to have and the one oracle/lbm_oracle.cpp restates;
//   * two launch variants were measured and are OFF by default (profiles/r02_lbm_bench.log): blockIdx remapped so that each of
//     the 8 XCDs sweeps one contiguous eighth of the lattice (SMX_LBM_XCD=1: no reuse to win in a pure stream -- equal or
//     slower), and non-temporal hints on the 38 lattice accesses (SMX_LBM_NT=1: equal or slower).
// Arithmetic: float32, the shader's statement order, no contraction; constants that involve sqrt/div are computed on the
// host once (make_lbm_consts) so that device and oracle use identical bit patterns.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <string>

namespace smx {
constexpr int LQ = 19;
struct LbmConsts { float w[LQ], cs2, cs4, force[3], omega_plus, omega_minus, grav; };
struct LbmDims { int nx, ny, nz; unsigned long long n; };

__device__ __constant__ const int L_CX[LQ] = {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 0, 0, 1, -1, 1, -1, 0, 0};   // lbm.cs:60-80
__device__ __constant__ const int L_CY[LQ] = {0, 0, 0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 1, -1};
__device__ __constant__ const int L_CZ[LQ] = {0, 0, 0, 0, 0, 1, -1, 0, 0, 1, -1, 1, -1, 0, 0, -1, 1, -1, 1};
// opposite directions come in pairs (2k-1, 2k) (lbm.cs:82-87)
__host__ __device__ constexpr int lcx(int q) { constexpr int t[LQ] = {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 0, 0, 1, -1, 1, -1, 0, 0}; return t[q]; }
__host__ __device__ constexpr int lcy(int q) { constexpr int t[LQ] = {0, 0, 0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 1, -1}; return t[q]; }
__host__ __device__ constexpr int lcz(int q) { constexpr int t[LQ] = {0, 0, 0, 0, 0, 1, -1, 0, 0, 1, -1, 1, -1, 0, 0, -1, 1, -1, 1}; return t[q]; }
__host__ __device__ constexpr int lcp(int q) { return q == 0 ? 0 : (q & 1 ? q + 1 : q - 1); }

inline LbmConsts make_lbm_consts() {
  LbmConsts k;
  k.w[0] = 1.0f / 3.0f;                                        // lbm.cs:53-58
  for (int q = 1; q < 7; q++) k.w[q] = 1.0f / 18.0f;
  for (int q = 7; q < LQ; q++) k.w[q] = 1.0f / 36.0f;
  const float cs = 1.0f / sqrtf(3.0f);                        // lbm.cs:89-91
  k.cs2 = 1.0f / cs / cs;
  k.cs4 = 1.0f / cs / cs / cs / cs;
  k.force[0] = 0.05f * -2.0f; k.force[1] = 0.05f * 0.0f; k.force[2] = 0.05f * 1.0f;   // lbm.cs:36
  const float tau = 0.56f, lambda = 0.25f;                    // collide.cs:7,31-33
  k.omega_plus = 1.0f / tau;
  k.omega_minus = 1.0f / (lambda / (1.0f / k.omega_plus - 0.5f) + 0.5f);
  k.grav = 1.0f * 0.0001f;                                    // collide.cs:8,19
  return k;
}

__device__ __forceinline__ float l_dot(float ax, float ay, float az, float bx, float by, float bz) { float t = ax * bx; t = t + ay * by; t = t + az * bz; return t; }
// lbm.cs:95-105
template <int q>
__device__ __forceinline__ float l_equilibrium(const LbmConsts& k, float rho, float vx, float vy, float vz) {
  const float d = l_dot(vx, vy, vz, (float)lcx(q), (float)lcy(q), (float)lcz(q));
  float eq = 0.0f;
  eq += k.w[q] * rho;
  eq += k.w[q] * rho * d * k.cs2;
  eq += k.w[q] * rho * (d * d) * 0.5f * k.cs4;
  eq -= k.w[q] * rho * l_dot(vx, vy, vz, vx, vy, vz) * 0.5f * k.cs2;
  return eq;
}
template <int q> struct LbmUnroll {
  template <class F> __device__ __forceinline__ static void run(F&& f) { LbmUnroll<q - 1>::run(f); f(std::integral_constant<int, q>{}); }
};
template <> struct LbmUnroll<-1> { template <class F> __device__ __forceinline__ static void run(F&&) {} };

// one contiguous eighth of the workgroups per XCD (workgroups are dealt to the 8 XCDs round-robin)
__device__ __forceinline__ unsigned long long lbm_block() {
  const unsigned long long nb = gridDim.x, b = blockIdx.x;
  const unsigned long long per = (nb + 7ull) / 8ull, xcd = b & 7ull, k = b >> 3;
  const unsigned long long r = xcd * per + k;
  return r;                                                   // (>= nb for the padding of the last eighth: the caller's bound check drops it)
}
__device__ __forceinline__ bool lbm_driven(const LbmDims& d, int x, int y, int z) {   // stream.cs:26-31
  return y == d.ny - 1 || x == 0 || x == d.nx - 1 || z == 0 || z == d.nz - 1;
}

// init.cs:9-24
__global__ void __launch_bounds__(256) k_lbm_init(LbmDims d, LbmConsts k, const float* __restrict__ B, float* __restrict__ f, float* __restrict__ rho, float4* __restrict__ v) {
  const unsigned long long i = lbm_block() * 256ull + threadIdx.x;
  if (i >= d.n) return;
  const bool solid = B[i] > 0;
  float r = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
  float F[LQ];
  LbmUnroll<LQ - 1>::run([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    F[q] = solid ? l_equilibrium<q>(k, 1.0f, 0.f, 0.f, 0.f) : l_equilibrium<q>(k, 1.0f, k.force[0], k.force[1], k.force[2]);
  });
  LbmUnroll<LQ - 1>::run([&](auto qc) { constexpr int q = decltype(qc)::value; r += F[q]; });
  LbmUnroll<LQ - 1>::run([&](auto qc) { constexpr int q = decltype(qc)::value; vx += F[q] * (float)lcx(q); vy += F[q] * (float)lcy(q); vz += F[q] * (float)lcz(q); });
  LbmUnroll<LQ - 1>::run([&](auto qc) { constexpr int q = decltype(qc)::value; f[(unsigned long long)q * d.n + i] = F[q]; });
  rho[i] = r;
  v[i] = make_float4(vx / r, vy / r, vz / r, 1.0f);
}

// collide.cs + stream.cs fused: lattice `a` (F) -> lattice `b`. NT: the 38 lattice accesses carry the non-temporal hint (each
// byte is touched once per step); XCD: the blockIdx remap above. Both are measured switches (SMX_LBM_NT / SMX_LBM_XCD), off.
template <bool NT> __device__ __forceinline__ float lbm_ld(const float* p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
template <bool NT> __device__ __forceinline__ void lbm_st(float* p, float x) { if constexpr (NT) __builtin_nontemporal_store(x, p); else *p = x; }
template <bool NT, bool XCD>
__global__ void __launch_bounds__(256) k_lbm_step(LbmDims d, LbmConsts k, const float* __restrict__ B, const float* __restrict__ a, float* __restrict__ b,
                                                  float* __restrict__ rho, float4* __restrict__ v) {
  const unsigned long long i = (XCD ? lbm_block() : (unsigned long long)blockIdx.x) * 256ull + threadIdx.x;
  if (i >= d.n) return;
  const int z = (int)(i % (unsigned)d.nz), y = (int)((i / (unsigned)d.nz) % (unsigned)d.ny), x = (int)(i / ((unsigned long long)d.nz * d.ny));
  float F[LQ];
  LbmUnroll<LQ - 1>::run([&](auto qc) { constexpr int q = decltype(qc)::value; F[q] = lbm_ld<NT>(a + ((unsigned long long)q * d.n + i)); });
  const bool solid = B[i] > 0.0f;
  float r = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
  LbmUnroll<LQ - 1>::run([&](auto qc) { constexpr int q = decltype(qc)::value; r += F[q]; });                     // getRho lbm.cs:109-116
  LbmUnroll<LQ - 1>::run([&](auto qc) { constexpr int q = decltype(qc)::value; vx += F[q] * (float)lcx(q); vy += F[q] * (float)lcy(q); vz += F[q] * (float)lcz(q); });   // getV :120-127
  vx = vx / r; vy = vy / r; vz = vz / r;
  const float den = 2.0f * r;                                 // collide.cs:19: _v += dt*0.0001*c[4]/(2.0f*_rho)
  vx += (k.grav * 0.0f) / den; vy += (k.grav * -1.0f) / den; vz += (k.grav * 0.0f) / den;
  rho[i] = r;
  v[i] = make_float4(vx, vy, vz, 0.0f);
  float E[LQ];
  LbmUnroll<LQ - 1>::run([&](auto qc) { constexpr int q = decltype(qc)::value; E[q] = l_equilibrium<q>(k, r, vx, vy, vz); });
  const bool me_driven = lbm_driven(d, x, y, z);
  LbmUnroll<LQ - 1>::run([&](auto qc) {
    constexpr int q = decltype(qc)::value; constexpr int p = lcp(q);
    const float f_plus = 0.5f * (F[q] + F[p]), f_minus = 0.5f * (F[q] - F[p]);                                    // collide.cs:42-50
    const float e_plus = 0.5f * (E[q] + E[p]), e_minus = 0.5f * (E[q] - E[p]);
    float out = F[q] - k.omega_plus * (f_plus - e_plus) - k.omega_minus * (f_minus - e_minus);
    if (solid) out = l_equilibrium<q>(k, 1.0f, 0.f, 0.f, 0.f);                                                     // :52-53
    // stream.cs:11-20: push into the neighbour unless it lies outside -- or is a driven face, whose own overwrite wins
    const int ax = x + lcx(q), ay = y + lcy(q), az = z + lcz(q);
    const bool inside = !(ax < 0 || ax >= d.nx || ay < 0 || ay >= d.ny || az < 0 || az >= d.nz);
    if (inside && !lbm_driven(d, ax, ay, az))
      lbm_st<NT>(b + ((unsigned long long)q * d.n + (unsigned long long)((long long)i + ((long long)lcx(q) * d.ny + lcy(q)) * d.nz + lcz(q))), out);
    // my own slot q: driven face -> forcing equilibrium (stream.cs:33-34); no source for direction q -> the old value stays
    const int sx = x - lcx(q), sy = y - lcy(q), sz = z - lcz(q);
    const bool has_source = !(sx < 0 || sx >= d.nx || sy < 0 || sy >= d.ny || sz < 0 || sz >= d.nz);
    if (me_driven) lbm_st<NT>(b + ((unsigned long long)q * d.n + i), l_equilibrium<q>(k, 1.0f, k.force[0], k.force[1], k.force[2]));
    else if (!has_source) lbm_st<NT>(b + ((unsigned long long)q * d.n + i), F[q]);
  });
}

// move.cs:27-52: tracers advected by the trilinear velocity
__global__ void __launch_bounds__(256) k_lbm_move(LbmDims d, const float4* __restrict__ v, float4* __restrict__ P, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float4 pp = P[i];
  const float Pa[4] = {pp.x, pp.y, pp.z, pp.w};
  int p[4], nn[4]; float w[4];
  for (int c = 0; c < 4; c++) { p[c] = (int)Pa[c]; nn[c] = p[c] + 1; w[c] = Pa[c] - (float)p[c]; }
  // A tracer in [N-1, N) -- or below 0, or NaN -- has p or p+1 outside the lattice. The reference moves tracers BEFORE its
  // range check (lbmwind.h:193-215) and GL's robust buffer access tolerates the stray read; here the index is clamped to
  // the lattice (the nearest cell is sampled): in-range tracers are untouched, out-of-range ones cannot fault the context.
  auto cl = [](int a, int n) { return a < 0 ? 0 : (a > n - 1 ? n - 1 : a); };
  auto V = [&](int x, int y, int z) { return v[((unsigned long long)cl(x, d.nx) * d.ny + cl(y, d.ny)) * d.nz + cl(z, d.nz)]; };
  const float4 q000 = V(p[0], p[1], p[2]), q100 = V(nn[0], p[1], p[2]), q010 = V(p[0], nn[1], p[2]), q001 = V(p[0], p[1], nn[2]),
               q110 = V(nn[0], nn[1], p[2]), q101 = V(nn[0], p[1], nn[2]), q011 = V(p[0], nn[1], nn[2]), q111 = V(nn[0], nn[1], nn[2]);
  auto comp = [](const float4& t, int c) { return c == 0 ? t.x : (c == 1 ? t.y : (c == 2 ? t.z : t.w)); };
  float o[4];
  for (int c = 0; c < 4; c++) {
    const float v00 = (1.0f - w[0]) * comp(q000, c) + w[0] * comp(q100, c), v01 = (1.0f - w[0]) * comp(q001, c) + w[0] * comp(q101, c);
    const float v10 = (1.0f - w[0]) * comp(q010, c) + w[0] * comp(q110, c), v11 = (1.0f - w[0]) * comp(q011, c) + w[0] * comp(q111, c);
    const float v0 = (1.0f - w[1]) * v00 + w[1] * v10, v1 = (1.0f - w[1]) * v01 + w[1] * v11;
    o[c] = Pa[c] + ((1.0f - w[2]) * v0 + w[2] * v1);
  }
  P[i] = make_float4(o[0], o[1], o[2], o[3]);
}
// SoilMachine.cpp:235-238: boundary[x][y][z] = map.height(ivec2(scale.x*x, scale.z*z)) > (scale.y*y)/(float)SCALE
__global__ void __launch_bounds__(256) k_lbm_boundary_from_heights(LbmDims d, const double* __restrict__ heights, int dimy_map, float sx, float sy, float sz, int scale, float* __restrict__ B) {
  const unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
  if (i >= d.n) return;
  const int z = (int)(i % (unsigned)d.nz), y = (int)((i / (unsigned)d.nz) % (unsigned)d.ny), x = (int)(i / ((unsigned long long)d.nz * d.ny));
  const int mx = (int)(sx * (float)x), mz = (int)(sz * (float)z);
  B[i] = heights[(size_t)mx * dimy_map + mz] > (double)((sy * (float)y) / (float)scale) ? 1.0f : 0.0f;
}
// AoS [cell*19+q] (the reference's buffer layout, for callers and tests) <-> SoA
__global__ void __launch_bounds__(256) k_lbm_to_aos(LbmDims d, const float* __restrict__ f, float* __restrict__ out) {
  const unsigned long long j = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
  if (j >= d.n * LQ) return;
  out[j] = f[(j % LQ) * d.n + j / LQ];
}
__global__ void __launch_bounds__(256) k_lbm_from_aos(LbmDims d, const float* __restrict__ in, float* __restrict__ f) {
  const unsigned long long j = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
  if (j >= d.n * LQ) return;
  f[(j % LQ) * d.n + j / LQ] = in[j];
}
}  // namespace smx

struct smx_lbm {
  smx::LbmDims d{};
  smx::LbmConsts k{};
  int device = 0;
  hipStream_t stream = nullptr;
  float *f[2] = {nullptr, nullptr}, *B = nullptr, *rho = nullptr;
  float4* v = nullptr;
  int cur = 0;                                                // which lattice holds F
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double ms_steps = 0.0; unsigned long long steps = 0;
  std::string err;
};
