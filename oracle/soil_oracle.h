// C interface of the CPU oracle restatement (TEST INFRASTRUCTURE ONLY).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
// product path (soilmachine_amd/, include/soilmx.h) never links, imports or executes it.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct so_soil {          // POD mirror of SurfParam (source/surface.h:11-39), render fields dropped
  uint32_t transports, erodes, cascades, abrades;
  float density, porosity, solubility, equrate, friction, erosionrate, maxdiff, settling, suspension, abrasion;
} so_soil;

typedef struct so_layer {         // POD mirror of SurfLayer (source/surface.h:65-101)
  uint32_t type;
  float min, bias, scale, octaves, lacunarity, gain, frequency;
} so_layer;

typedef struct so_counters {
  uint64_t steps_water_top;       // top-level WaterParticle::move() calls that returned true
  uint64_t steps_water_all;       // same, incl. nested particles spawned by WaterParticle::cascade
  uint64_t steps_wind;
  uint64_t nested_particles;
  uint64_t floods;                // flood() calls that passed the volume/spill gate
  uint64_t cascade_calls;         // Particle::cascade invocations
  uint64_t cascade_transfers;     // sediment transfers inside Particle::cascade
  uint64_t wcascade_calls;        // WaterParticle::cascade invocations
  uint64_t grid_active_cells;     // grid-pass cells for which seep/cascade did any work
  uint64_t rand_calls;
  uint64_t pool_free;
  uint64_t pool_overflow;         // pool.get() calls that found the pool empty (layermap.h:92-95)
} so_counters;

typedef struct so_ctx so_ctx;

so_ctx* so_create(int dimx, int dimy, int scale, uint64_t pool_capacity);
void so_destroy(so_ctx*);
void so_set_soils(so_ctx*, const so_soil* soils, int n);
void so_set_scale(so_ctx*, int scale);
void so_srand(so_ctx*, unsigned seed);                       // glibc srand() restatement
int so_rand(so_ctx*);                                        // glibc rand() restatement (one draw)
// Layermap::initialize (layermap.h:163-216) with our FastNoiseLite OpenSimplex2/FBm restatement
void so_initialize(so_ctx*, int seed, const so_layer* layers, int nlayers);
// state import/export in snapshot layout (oracle/snapshot.h): columns bottom->top in cell order x*dimy+y
int so_import(so_ctx*, const uint32_t* count, const uint32_t* type, const double* size, const double* floor,
              const double* sat, const float* wfreq, const float* wtrack, const float* windfreq, uint64_t rand_calls);
uint64_t so_num_sections(so_ctx*);
void so_export(so_ctx*, uint32_t* count, uint32_t* type, double* size, double* floor, double* sat,
               float* wfreq, float* wtrack, float* windfreq);
void so_heights(so_ctx*, double* out);                       // height(ivec2) for all cells, index x*dimy+y
// one tick of SoilMachine.cpp:283-329 (rendering removed); individual phases for unit tests
void so_tick(so_ctx*, int nwater, int nwind, int dowater, int dowind);
void so_water_particles(so_ctx*, int n);
void so_grid_pass(so_ctx*);
void so_wind_particles(so_ctx*, int n);
void so_map_frequency(so_ctx*);
void so_reset_frequency(so_ctx*);
void so_get_counters(so_ctx*, so_counters* out);
// primitives (unit tests)
double so_height(so_ctx*, int x, int y);
double so_height_bilinear(so_ctx*, float px, float py);
void so_normal(so_ctx*, int x, int y, float* out3);
uint32_t so_surface(so_ctx*, int x, int y);
void so_add(so_ctx*, int x, int y, double size, uint32_t type);
double so_remove(so_ctx*, int x, int y, double h);
void so_particle_cascade(so_ctx*, float px, float py, int transferloop);
void so_water_cascade(so_ctx*, int x, int y, int spill);
void so_seep(so_ctx*, int x, int y);
// so_set_relaxed(1): the water generations of so_batched_* follow the RELAXED schedule (soilmachine_amd/csrc/soil_relax.h) restated from
// its definition (soil_oracle.cpp r_water_generation) -- checker of SMX_ENGINE_RELAXED, bit-exact.
// The BATCHED schedule (throughput engine, soilmachine_amd/csrc/soil_batch.h) restated on the CPU: same per-step
// functions, the engine's own definition of who acts when. Checker for the batched HIP path (bit-exact); NOT a model of
// the reference's particle order. stats: {epochs, generations, cells touched outside the acting particle's tiles (must be 0)}
void so_batched_water(so_ctx*, int n);
void so_batched_grid(so_ctx*);
void so_batched_wind(so_ctx*, int n);
void so_set_water_stagger(so_ctx*, int gap_epochs);   // relaxed schedule: the k generations of a water phase born gap_epochs apart inside one phase (smx_set_water_stagger; 0 = off)
void so_set_water_generations(so_ctx*, int k);   // the water phase of so_batched_tick as k consecutive generations of n/k (default 8, as the engine)
void so_batched_tick(so_ctx*, int nwater, int nwind, int dowater, int dowind);
void so_set_relaxed(so_ctx*, int on);   // SMX_ENGINE_RELAXED restated (csrc/soil_relax.h): water generations step / apply / cascade colours / floods per epoch
void so_batched_set_dilate(so_ctx*, int tiles);
void so_batched_set_cell_above(so_ctx*, uint32_t nlive);   // schedule knob: per-cell reservations while more particles run
void so_batched_set_strips(so_ctx*, int nstrips, int inset, int seam_halfwidth);   // column-strip schedule (INTERIOR / SEAM chunks)
void so_batched_stats(so_ctx*, uint64_t* out3);
// dependency analysis between particles of one phase (evidence for DESIGN.md, not a simulation feature):
// out7 = {tasks, DAG depth (in-place: RAW+WAW+WAR), DAG depth (multi-version: RAW+WAW), conflict-free tasks,
//         step-granular critical path, total steps, sum of levels}; counters restart at every phase.
void so_dep_enable(so_ctx*, int on);
void so_dep_stats(so_ctx*, uint64_t* out7);
float so_noise_fbm(float x, float y, float z, int octaves, float lacunarity, float gain, float frequency);

#ifdef __cplusplus
}
#endif
