// lbm_ref.cpp -- runs the REFERENCE'S OWN LBM compute shaders on the CPU (TEST INFRASTRUCTURE ONLY; oracle/_ref/liblbm_ref.so).
//
// The shader text of source/include/lbmwind/shader/LBM/{lbm,init,collide,stream}.cs and shader/move.cs is compiled as C++:
// oracle/glsl2cpp.py rewrites the storage DECLARATIONS (layout/buffer/uniform/#version/#include -- GLSL spellings without a
// C++ meaning) into the macros of oracle/shim/glsl_cs.h at build time, into oracle/_ref/lbm/ (never committed); every
// statement and expression is the reference's. This driver plays lbmw::initialize / lbmw::update (lbmwind.h:98-109,176-197):
// one main() call per invocation of the dispatch, for exactly NX x NY x NZ cells (the reference dispatches NX/32 x NY x NZ/32
// groups of 32x1x32 and only uses multiples of 32). Invocations run in lexicographic order; the one order-dependent spot
// (stream.cs's driven faces) is resolved in the shim (FBUF). oracle/lbm_oracle.cpp and the HIP kernels are held to this.
#include "glsl_cs.h"
#include <cstring>

static glsl::uvec3 gl_GlobalInvocationID;
static size_t g_owner = 0;                                    // linear cell index of the running invocation

#define main glsl_main
namespace sh_init { using namespace glsl;
#include "init.cs.h"
}
namespace sh_collide { using namespace glsl;
#include "collide.cs.h"
}
namespace sh_stream { using namespace glsl;
#include "stream.cs.h"
}
namespace sh_move { using namespace glsl;
#include "move.cs.h"
}
#undef main

struct lr_ctx {
  int nx, ny, nz; size_t n;
  std::vector<float> F, FPROP, B, RHO;
  std::vector<glsl::vec4> V;
  std::vector<unsigned char> fin;
};
#define LR_BIND(NS, c, with_fin)                                                                      \
  NS::F.d = c->F.data(); NS::F.fin = (with_fin) ? c->fin.data() : nullptr; NS::F.owner_cell = &g_owner; NS::F.q = NS::Q; \
  NS::FPROP.d = c->FPROP.data(); NS::B.d = c->B.data(); NS::RHO.d = c->RHO.data(); NS::V.d = c->V.data();                \
  NS::NX = c->nx; NS::NY = c->ny; NS::NZ = c->nz;
template <class MAIN> static void dispatch(lr_ctx* c, MAIN run) {
  for (int x = 0; x < c->nx; x++)
    for (int y = 0; y < c->ny; y++)
      for (int z = 0; z < c->nz; z++) {
        gl_GlobalInvocationID.x = (glsl::uint)x; gl_GlobalInvocationID.y = (glsl::uint)y; gl_GlobalInvocationID.z = (glsl::uint)z;
        g_owner = ((size_t)x * c->ny + y) * c->nz + z;
        run();
      }
}

extern "C" {
lr_ctx* lo_create(int nx, int ny, int nz) {
  lr_ctx* c = new lr_ctx();
  c->nx = nx; c->ny = ny; c->nz = nz; c->n = (size_t)nx * ny * nz;
  c->F.assign(c->n * 19, 0.f); c->FPROP.assign(c->n * 19, 0.f); c->B.assign(c->n, 0.f); c->RHO.assign(c->n, 0.f);
  c->V.assign(c->n, glsl::vec4()); c->fin.assign(c->n * 19, 0);
  return c;
}
void lo_destroy(lr_ctx* c) { delete c; }
void lo_set_boundary(lr_ctx* c, const float* b) { memcpy(c->B.data(), b, c->n * 4); }
void lo_initialize(lr_ctx* c) {                               // lbmwind.h:98-109
  LR_BIND(sh_init, c, false)
  dispatch(c, [] { sh_init::glsl_main(); });
}
void lo_step(lr_ctx* c, int steps) {                          // lbmwind.h:176-191
  for (int s = 0; s < steps; s++) {
    { LR_BIND(sh_collide, c, false) dispatch(c, [] { sh_collide::glsl_main(); }); }
    std::fill(c->fin.begin(), c->fin.end(), 0);
    { LR_BIND(sh_stream, c, true) dispatch(c, [] { sh_stream::glsl_main(); }); }
  }
}
void lo_read(lr_ctx* c, float* rho, float* v4, float* f) {
  if (rho) memcpy(rho, c->RHO.data(), c->n * 4);
  if (v4) memcpy(v4, c->V.data(), c->n * 16);
  if (f) memcpy(f, c->F.data(), c->n * 19 * 4);
}
void lo_write_f(lr_ctx* c, const float* f) { memcpy(c->F.data(), f, c->n * 19 * 4); }
void lo_move(lr_ctx* c, float* pos4, int n) {                 // lbmwind.h:193-197 (in-lattice tracers only: the shader does not check)
  std::vector<glsl::vec4> P((size_t)n);
  memcpy(P.data(), pos4, (size_t)n * 16);
  sh_move::B.d = c->B.data(); sh_move::V.d = c->V.data(); sh_move::P.d = P.data();
  sh_move::NX = c->nx; sh_move::NY = c->ny; sh_move::NZ = c->nz;
  for (int i = 0; i < n; i++) { gl_GlobalInvocationID.x = (glsl::uint)i; sh_move::glsl_main(); }
  memcpy(pos4, P.data(), (size_t)n * 16);
}
}  // extern "C"
