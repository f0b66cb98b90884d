/* soilmx -- C-ABI of the MI355X-native SoilMachine particle-transport hot path.
 *
 * The reference (weigert/SoilMachine) has no FFI layer: its boundary is a source-level C++ header API
 * consumed by one translation unit (SoilMachine.cpp:19-28). This header is the seam a maintainer would
 * bind instead: every entry point names the reference interface it replaces (paths relative to the
 * reference root). Plain pointers and sizes only; no C++ or torch types cross this line.
 *
 * Conventions: all functions return 0 on success, <0 on error (smx_last_error() gives the text);
 * the caller is single-threaded per context (as the reference is, SURVEY.md §8b "Threading");
 * buffers are caller-allocated; "cell order" is the Layermap index x*dimy + y (source/layermap.h:151),
 * "frequency order" is y*dimx + x (source/particle/water.h:53,349).
 *
 * Exactness contract: under SMX_ENGINE_SERIAL and SMX_ENGINE_SPECULATIVE smx_tick* executes the reference's sequential
 * semantics (SoilMachine.cpp:283-329) -- results are bit-identical to the reference CPU path for the same soil table,
 * terrain, SCALE and libc rand() stream. SMX_ENGINE_BATCHED / SMX_ENGINE_RELAXED are deterministic throughput schedules that
 * do NOT keep the reference's particle order: statistical parity only, measured against the reference's own sensitivity to
 * its rand() stream.
 */
#ifndef SOILMX_H
#define SOILMX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smx_ctx smx_ctx;

/* POD mirror of SurfParam (source/surface.h:11-39); render-only fields (name, color, phong) dropped. */
typedef struct smx_soil {
  uint32_t transports, erodes, cascades, abrades;
  float density, porosity, solubility, equrate, friction, erosionrate, maxdiff, settling, suspension, abrasion;
} smx_soil;

/* POD mirror of SurfLayer (source/surface.h:65-101). */
typedef struct smx_layer {
  uint32_t type;
  float min, bias, scale, octaves, lacunarity, gain, frequency;
} smx_layer;

enum { SMX_ENGINE_SERIAL = 0,      /* one device walker, reference order (parity anchor)            */
       SMX_ENGINE_SPECULATIVE = 1, /* optimistic parallel particles, ordered commit, same results    */
       SMX_ENGINE_BATCHED = 2,     /* THROUGHPUT mode: all particles of a phase advance concurrently, one step
                                      per epoch, each step atomic and isolated (claim tiles). Deterministic, but NOT
                                      the reference's particle order: statistical parity only        */
       SMX_ENGINE_RELAXED = 3      /* the batched engine with RELAXED isolation (csrc/soil_relax.h): every running particle steps
                                      in every epoch on the epoch's start state; erosion / deposit are queued per cell and applied
                                      in slot order with Layermap::add / remove, Particle::cascade runs once per touched cell in nine
                                      colour passes; floods and nested particles as in BATCHED. Deterministic; statistical parity  */ };

typedef struct smx_config {
  int32_t dimx, dimy;              /* SIZEX, SIZEY (SoilMachine.cpp:9-10)                            */
  int32_t scale;                   /* SCALE (SoilMachine.cpp:11)                                     */
  int32_t device;                  /* HIP device ordinal                                             */
  uint64_t pool_capacity;          /* POOLSIZE (SoilMachine.cpp:16): max live sections               */
  int32_t engine;                  /* SMX_ENGINE_*                                                   */
  int32_t reserved;
} smx_config;

typedef struct smx_counters {
  uint64_t steps_water_top;        /* WaterParticle::move()==true for top-level particles            */
  uint64_t steps_water_all;        /* ... including nested particles (water.h:251-262)               */
  uint64_t steps_wind;
  uint64_t nested_particles;
  uint64_t floods;
  uint64_t cascade_calls;          /* Particle::cascade (particle.h:24)                              */
  uint64_t cascade_transfers;
  uint64_t wcascade_calls;         /* WaterParticle::cascade (water.h:151)                           */
  uint64_t grid_active_cells;      /* cells the exact grid pass had to visit                         */
  uint64_t rand_calls;             /* rand() draws consumed since smx_srand                          */
  uint64_t pool_free;              /* == map.pool.free.size() (layermap.h:69, SoilMachine.cpp:116)   */
  uint64_t pool_overflow;          /* pool.get() on an empty pool (layermap.h:92-95)                 */
  uint64_t spec_rounds;            /* speculative engine: execution rounds                           */
  uint64_t spec_aborts;            /* speculative engine: particle executions thrown away            */
  uint64_t reserved[2];            /* [0] phases the speculative engine handed to the serial walker, [1] particle executions         */
  uint64_t spec_subphases_cut;     /* speculative engine: sub-phases that ended early (nested-slot budget used up) and were re-armed  */
  uint64_t spec_serial_particles;  /* ... single particles run by the serial walker between two sub-phases (more nested particles    */
                                   /*     than a sub-phase has slots)                                                                */
  uint64_t flood_nested_steps;     /* round 6 (sized getter only): steps of the nested particles that ran INSIDE floods of the relaxed  */
  uint64_t grid_nested_steps;      /* schedule / INSIDE grid tiles -- both are part of steps_water_all; bench.py prices those launches */
} smx_counters;

/* Elapsed device time of the phases of the last smx_tick* calls, measured with HIP events on the
 * context's stream (milliseconds, accumulated since smx_timing_reset). */
typedef struct smx_timing {
  double ms_water, ms_grid, ms_wind, ms_freq;
  uint64_t launches_water, launches_grid, launches_wind, launches_freq;
  /* the particle kernels alone (k_water_serial / k_wind_serial, or every k_spec_exec launch of the speculative
   * engine), each launch bracketed by its own pair of HIP events */
  double ms_kernel_water, ms_kernel_wind;
  uint64_t launches_kernel_water, launches_kernel_wind;
  /* the O(cells) streaming kernels on their own: grid-pass classification (k_grid_classify), the batched engine's nine
   * colour passes (k_batch_grid), water.h:358-365 (k_map_frequency) */
  double ms_kernel_classify, ms_kernel_gridtiles, ms_kernel_mapfreq;
  uint64_t launches_kernel_classify, launches_kernel_gridtiles, launches_kernel_mapfreq;
  /* round 5 (smx_get_timing_sized only): ALL launches of the throughput engines' step kernels (k_relax_step / k_batch_epoch), whether
   * bracketed by events or not. The relaxed water step kernel is bracketed every 8th launch (SMX_STEP_EVENT_SAMPLE): ms_kernel_water /
   * launches_kernel_water is then the average duration over the bracketed launches, and the algorithmic bytes per launch are the
   * steps of ALL launches over launches_step_water (bench.py). */
  uint64_t launches_step_water, launches_step_wind;
  /* round 6 (smx_get_timing_sized only): the relaxed water phase's persistent launches, each between its own pair of HIP events --
   * k_relax_epochs (the dense epochs of a generation: step, apply, filter, colour lists and floods behind device-wide barriers) and
   * k_relax_tail (the same phases inside one workgroup once <= 256 particles run): milliseconds, launches, epochs the launches covered;
   * and the generations of suspended nested particles the grid pass leaves behind (part of ms_grid) */
  double ms_kernel_epochs, ms_kernel_tail, ms_kernel_grid_children;
  uint64_t launches_kernel_epochs, epochs_kernel_epochs, launches_kernel_tail, epochs_kernel_tail, launches_kernel_grid_children;
  /* k_relax_floods: every 7th launch of a context is bracketed (ms / launches_kernel_floods); launches_floods_all counts every launch */
  double ms_kernel_floods;
  uint64_t launches_kernel_floods, launches_floods_all;
} smx_timing;

/* ---- life cycle: replaces Layermap::Layermap + secpool::reserve (layermap.h:64-119,218-226) ---- */
int smx_create(const smx_config* cfg, smx_ctx** out);
/* A context that HOLDS only the columns [x_lo, x_hi) of the dimx x dimy map -- one column strip with its halo, for smx_strips_*: cells,
 * flags and the engines' per-cell planes are allocated for that range (indexing stays global), so per-device memory follows the strip,
 * not the map (the frequency planes, 12 B/cell, stay full-size: their index y*dimx + x is not contiguous in x). x_lo * dimy must be a
 * multiple of 64. Whole-map calls (exact engines, save/load, vertices, point operations) refuse such a context; import / export /
 * read_heights / digest act on the held range. pool_capacity is this context's own. */
int smx_create_strip(const smx_config* cfg, int32_t x_lo, int32_t x_hi, smx_ctx** out);
void smx_destroy(smx_ctx* ctx);
const char* smx_last_error(smx_ctx* ctx);

/* ---- tables: replaces the globals soils[] / SCALE read by every particle (surface.h:41, SoilMachine.cpp:11) ---- */
int smx_set_soils(smx_ctx* ctx, const smx_soil* soils, int32_t n);
int smx_set_scale(smx_ctx* ctx, int32_t scale);

/* ---- libc rand() stream hand-off: replaces srand(SEED)/rand() (SoilMachine.cpp:41, water.h:13, wind.h:15) ---- */
int smx_srand(smx_ctx* ctx, uint32_t seed);
int smx_rand(smx_ctx* ctx, int32_t* out);               /* draw one value on behalf of the host */
int smx_rand_advance(smx_ctx* ctx, uint64_t ndraws);    /* discard ndraws values (resume a stream at a known offset) */
/* the generator itself (glibc TYPE_3: the last 31 words, the running index, draws since smx_srand): hand the stream from one
 * context to another (the compat Layermap does when the host re-initialises with a new map size, SoilMachine.cpp:111-114) */
int smx_get_rand_state(smx_ctx* ctx, uint32_t ring31[31], uint32_t* idx, uint64_t* calls);
int smx_set_rand_state(smx_ctx* ctx, const uint32_t ring31[31], uint32_t idx, uint64_t calls);

/* ---- terrain: replaces Layermap::initialize + SurfLayer::get + FastNoiseLite (layermap.h:163-216, surface.h:82-99) ---- */
int smx_initialize(smx_ctx* ctx, int32_t seed, const smx_layer* layers, int32_t nlayers);

/* ---- state hand-over (snapshot layout, columns bottom->top in cell order); replaces direct access to
 *      Layermap::dat / sec lists (layermap.h:37-62,131) and WaterParticle/WindParticle::frequency,track ---- */
int smx_import_columns(smx_ctx* ctx, const uint32_t* count, const uint32_t* type, const double* size,
                       const double* floor, const double* sat);
int smx_import_frequency(smx_ctx* ctx, const float* wfreq, const float* wtrack, const float* windfreq);
int smx_num_sections(smx_ctx* ctx, uint64_t* out);
int smx_export_columns(smx_ctx* ctx, uint32_t* count, uint32_t* type, double* size, double* floor, double* sat);
int smx_read_frequency(smx_ctx* ctx, float* wfreq, float* wtrack, float* windfreq);   /* NULL = skip */
int smx_read_heights(smx_ctx* ctx, double* out);        /* Layermap::height(ivec2) for every cell (layermap.h:422) */
int smx_read_surface(smx_ctx* ctx, uint32_t* out);      /* Layermap::surface(ivec2) for every cell (layermap.h:417) */

/* ---- checkpoint / resume on disk (the reference has none: SURVEY.md 5; io.h:232 "Should be able to also WRITE to file").
 * File = the snapshot layout of soilmachine_amd/snapshot.py (64-byte header "SMXSNAP1", per-cell section counts, sections
 * bottom -> top, the three frequency planes) followed by a trailer "SMXRAND1" with the rand() generator's 31-word ring, its
 * index and the number of draws -- a loaded context continues the libc stream exactly where the saved one stopped.
 * Since round 4 a second trailer "SMXFLAG1" carries the per-column flag plane (one byte per cell; its sticky "has held a saturation"
 * bit is context state the column records cannot express, and the throughput engines fix the active set of their grid pass from
 * it): every engine now continues bit-identically from a file (tests/test_gpu_depth.py). A bare snapshot without the trailers still
 * loads (return 1: re-seed the generator; the flag plane is re-derived from the saturations present). ---- */
int smx_save(smx_ctx* ctx, const char* path);
int smx_load(smx_ctx* ctx, const char* path);           /* dims must match the context; SCALE and soils are the caller's */

/* ---- the hot path, phase by phase: replaces SoilMachine.cpp:287-320 ---- */
int smx_tick_water(smx_ctx* ctx, int32_t nwater);       /* :287-298  NWATER x {WaterParticle ctor; move/interact; flood} */
int smx_grid_pass(smx_ctx* ctx);                        /* :300-301  WaterParticle::seep(map) (water.h:335-343)          */
int smx_tick_wind(smx_ctx* ctx, int32_t nwind);         /* :303-307  NWIND x {WindParticle ctor; move/interact}          */
int smx_map_frequency(smx_ctx* ctx);                    /* :314      WaterParticle::mapfrequency (water.h:358-365)       */
int smx_reset_frequency(smx_ctx* ctx);                  /* :319      WaterParticle::resetfrequency (water.h:353-356)     */
int smx_tick(smx_ctx* ctx, int32_t nwater, int32_t nwind, int32_t dowater, int32_t dowind);   /* all of the above, in order */
int smx_sync(smx_ctx* ctx);                             /* wait for all queued device work */

/* ---- point operations for API fidelity (Layermap::add/remove, Particle::cascade, ... called by host code) ---- */
int smx_add(smx_ctx* ctx, int32_t x, int32_t y, double size, uint32_t type);            /* layermap.h:230 */
int smx_remove(smx_ctx* ctx, int32_t x, int32_t y, double h, double* remainder);        /* layermap.h:310 */
int smx_particle_cascade(smx_ctx* ctx, float px, float py, int32_t transferloop);       /* particle.h:24  */
int smx_water_cascade(smx_ctx* ctx, int32_t x, int32_t y, int32_t spill);               /* water.h:151    */
int smx_seep(smx_ctx* ctx, int32_t x, int32_t y);                                       /* water.h:285    */
int smx_top(smx_ctx* ctx, int32_t x, int32_t y, uint32_t* type, double* size, double* floor, double* sat,
            int32_t* empty);                                                            /* Layermap::top layermap.h:150 (a copy) */
int smx_normals(smx_ctx* ctx, float* out3);             /* Layermap::normal(ivec2) for every cell, xyz interleaved (layermap.h:341) */
/* Layermap::update(Vertexpool&) (layermap.h:551-555 -> :475-549, no SLICE cut): the visible vertex of every column in
 * one pass -- 44-byte records {position[3], normal[3], color[4], index} (source/include/vertexpool.h:9-28) in cell
 * order x*dimy+y: position = (x, SCALE*height, y), normal = Layermap::normal(ivec2), color = colors4[type], index =
 * type; an empty column is (x, 0, y), (0,1,0), colors4[0], 0. colors4 = ncolors x RGBA (SurfParam::color,
 * surface.h:17); out = dimx*dimy*44 bytes, caller-allocated. */
int smx_fill_vertices(smx_ctx* ctx, const float* colors4, int32_t ncolors, void* out_vertices44);
/* the same with a horizontal cut through the sediment: mode 0 = Layermap::update's rule with the global SLICE = (int)cut
 * (layermap.h:477-510: first section starting at or below SLICE/SCALE; a section reaching above it is drawn flat at y = SLICE,
 * blended with Air's colour where its water table reaches the cut), mode 1 = Layermap::slice(s = cut) (layermap.h:557-613). */
int smx_fill_vertices_cut(smx_ctx* ctx, const float* colors4, int32_t ncolors, int32_t mode, double cut, void* out_vertices44);
/* ONE column's vertex under the same rules (mode < 0: no cut) -- Layermap::update(ivec2, Vertexpool&) (layermap.h:475-549) for a
 * host-driven single edit; out = 44 bytes. */
int smx_fill_vertex_cut(smx_ctx* ctx, const float* colors4, int32_t ncolors, int32_t mode, double cut, int32_t x, int32_t y, void* out_vertex44);
int smx_heights_bilinear(smx_ctx* ctx, const float* pos2, int32_t n, double* out);      /* Layermap::height(vec2) (layermap.h:427) */

/* ---- observability ---- */
/* The state digest of SURVEY.md Appendix E, computed from a device->host copy: sum of Layermap::height(ivec2) over the
 * cells in x-outer / y-inner order (sequential double accumulation), number of sections, and the 64-bit hash
 * h = (h ^ type) * 1099511628211 (start 1469598103934665603) over every column walked top -> bottom, same cell order. */
int smx_digest(smx_ctx* ctx, double* sumh, uint64_t* nsec, uint64_t* typehash);
/* The sized getters write min(struct_size, sizeof) bytes: pass sizeof(smx_counters) / sizeof(smx_timing) of the header you compiled
 * against. The unsized ones keep the layouts they were introduced with (smx_counters: the first 16 words, i.e. without
 * spec_subphases_cut / spec_serial_particles; smx_timing: the first 144 bytes) so that older binaries are never written past their
 * struct -- an ABI note for callers of rounds 1-4: INTEGRATION.md "ABI notes". */
int smx_get_counters_sized(smx_ctx* ctx, smx_counters* out, uint64_t struct_size);
int smx_get_timing_sized(smx_ctx* ctx, smx_timing* out, uint64_t struct_size);
int smx_get_counters(smx_ctx* ctx, smx_counters* out);
int smx_get_timing(smx_ctx* ctx, smx_timing* out);
int smx_timing_reset(smx_ctx* ctx);
int smx_set_engine(smx_ctx* ctx, int32_t engine);
/* speculative engine (round 4): a particle phase runs in sub-phases of at most `particles_per_subphase` top-level particles, each owning
 * `nested_slots` rand() slots for the nested particles its floods spawn (water.h:246-264); a sub-phase whose slots are used up commits
 * and is re-armed from the continued stream. 0 = defaults (4096 / 8192). Results never depend on the values. */
int smx_set_spec_limits(smx_ctx* ctx, uint32_t particles_per_subphase, uint32_t nested_slots);
/* batched engine only: widen every reservation by `tiles` 4x4-cell tiles (0 = default; >= the map's tile count makes
 * the engine run the particles strictly one after the other, i.e. in the reference's order) */
int smx_set_batch_dilate(smx_ctx* ctx, int32_t tiles);
/* relaxed schedule (SMX_ENGINE_RELAXED), wind phase: relaxed epochs -- every running particle takes up to `steps_per_epoch` (1..8)
 * steps on the epoch's start state -- while more than `min_running` particles run; the survivors, a few thousand particles that fly
 * on for up to ~13 000 steps, go on under the exclusive schedule (32 steps per launch). Part of the schedule's definition: results
 * depend on it. Default: min_running = 0xFFFFFFFF, i.e. wind always runs the exclusive schedule (measured faster, profiles/r03_relaxed.md);
 * 4 steps per epoch. */
int smx_set_relax_wind(smx_ctx* ctx, uint32_t min_running, int32_t steps_per_epoch);
/* relaxed schedule, water phase: the LAUNCH SHAPE of its epochs -- never the result (tests/test_gpu_relaxed.py runs every shape against the
 * same host-thread states). persistent: 0 = five launches per epoch (step, apply, filter, colour lists, floods: the default), 1 = the dense
 * epochs of a chunk as one cooperative launch with device-wide barriers (k_relax_epochs, round 6: measured 15-45 % slower on the headline,
 * profiles/r06_persistent_epochs.md); tail_at: from this many running particles down one workgroup runs whole epochs back to back
 * (k_relax_tail; 0..256, default 256). -1 = leave as is / back to the default. */
int smx_set_relax_launch(smx_ctx* ctx, int32_t persistent, int32_t tail_at);
/* REMOVED in round 5 (nested particles run inside their parent since then; there is nothing to interleave): kept as a symbol that fails
 * loudly (-2, smx_last_error says so) so that a round-4 caller neither crashes at load time nor silently runs another schedule. */
int smx_set_grid_interleave(smx_ctx* ctx, int32_t k);
/* throughput engines: smx_tick_water(n) (and smx_strips_tick) run the n particles as k consecutive top-level generations of n/k. DEFAULT 8
 * since round 6 (round 5: 4; rounds 3-4: 1). The particles of one generation advance together and do not see the lakes their own generation
 * makes; in the reference particle i sees what particles < i of the same tick did (SoilMachine.cpp:287-298), and every wet cell stops the
 * particles that reach it (water.h:56 with soils["Air"].friction = 0). k generations give a particle (k-1)/k of that view for k epoch chains.
 * At the headline workload, against 33 rand() streams of the reference itself (profiles/r06_p2_reference_4096.json): k = 1 is outside by up
 * to 7.9 sigma, k = 4 outside on three of eight figures (2.3-3.0 sigma), k = 8 inside |z| < 2 on all eight. Callers of round 5 see a
 * behaviour and a cost change (the water phase ~1.3 x) unless they call smx_set_water_generations(4). Identical to k calls of n/k with k = 1. */
int smx_set_water_generations(smx_ctx* ctx, int32_t k);
int smx_get_water_generations(smx_ctx* ctx, int32_t* k);   /* the value in force (bench.py reports it instead of a literal) */
/* relaxed engine: STAGGERED generations. gap_epochs > 0: a water phase of n particles is ONE phase whose k batches of ceil(n / k) particles (by slot) are
 * born gap_epochs apart -- batch b takes its first step in epoch b * gap --; 0: k consecutive generations, each run to its end. A particle of batch b
 * finds what the batches before it did in the epochs they are ahead (the lakes their stopped particles made); the tick's epoch chain is (k - 1) * gap + one
 * generation long instead of k generations. Part of the schedule's definition (results depend on it); default 0 (SMX_WATER_STAGGER in csrc/soil_batch.h).
 * Measured against the same 33 reference streams (profiles/r06_alt1_stagger_sweep_..., r06_alt2_stagger_ensembles_p2_reference_4096.json): k = 16, gap = 140 takes
 * 30 % off the water phase (64 M instead of 50 M particle-steps/s at the headline) and passes the parity gate with SEED 0's stream and under two strips, but two
 * of three further device streams fail it and neighbouring (k, gap) points do too: a measured step away from the reference, hence an option. */
int smx_set_water_stagger(smx_ctx* ctx, int32_t gap_epochs);
int smx_get_water_stagger(smx_ctx* ctx, int32_t* gap_epochs);
/* batched engine: column strips (DESIGN.md "Multi-GPU"). The schedule alternates INTERIOR chunks (a particle acts iff its
 * reservation lies `inset` cells inside one of `nstrips` equal x-strips) and SEAM chunks (iff inside one seam zone of
 * +-seam_halfwidth cells around a strip boundary). nstrips == 1 (default) = the single-strip schedule. The result is a
 * function of (nstrips, inset, seam_halfwidth) only -- not of how many devices run the strips. */
int smx_set_batch_strips(smx_ctx* ctx, int32_t nstrips, int32_t inset, int32_t seam_halfwidth);
/* ---- column strips on SEVERAL devices, step by step (library driver: smx_strips_*; its Python restatement over these entry points: tests/strips_ref.py; one context per strip, each with
 * the full-size map of which it keeps its strip + right halo current). A generation of a particle phase:
 *   smx_d_gen_begin   draws the generation's rand() values (identical on every rank) and, for nested particles
 *                     (`children` = BChild records: u64 key, i32 tx,ty,bx,by,spill,pad, f64 volume), answers what this rank
 *                     knows about each child's `contains` (0xFFFFFFFF = random cell not in my strip; ranks combine by min)
 *   smx_d_gen_spawn   constructs the particles; those outside [own_x0, own_x1) live on another rank
 *   smx_d_chunk       32 epochs; the schedule follows the GLOBAL number of running particles; chunks alternate INTERIOR /
 *                     SEAM (smx_d_next_phase tells which comes next: 0 interior, 1 seam)
 *   smx_d_gen_end     counters + this rank's newly spawned children
 * smx_d_pack_columns / _particles serialise the changed halo columns (dirty 4x4 tiles with x in [x0,x1)) and the running
 * particles with ipos.x in [x0,x1) (they leave this rank) into a caller-provided host buffer; _unpack_ applies them. */
int smx_d_set_own(smx_ctx* ctx, int32_t own_x0, int32_t own_x1);
int smx_d_gen_begin(smx_ctx* ctx, int32_t wind, uint32_t nslots, const void* children, uint32_t* contains_out);
int smx_d_gen_spawn(smx_ctx* ctx, const uint32_t* contains, uint32_t* nlive_local);
int smx_d_next_phase(smx_ctx* ctx);
int smx_d_chunk(smx_ctx* ctx, uint32_t nlive_global, uint32_t* nlive_local);
int smx_d_gen_end(smx_ctx* ctx, void* children_out, uint32_t cap, uint32_t* n);
int smx_d_pack_columns(smx_ctx* ctx, int32_t x0, int32_t x1, void* buf, uint64_t cap, uint64_t* bytes);
int smx_d_unpack_columns(smx_ctx* ctx, const void* buf, uint64_t bytes);
int smx_d_pack_particles(smx_ctx* ctx, int32_t x0, int32_t x1, void* buf, uint64_t cap, uint64_t* bytes);
int smx_d_unpack_particles(smx_ctx* ctx, const void* buf, uint64_t bytes);
int smx_d_grid_begin(smx_ctx* ctx);                                            /* classification of the grid pass       */
/* (rounds 2-4 exported `smx_d_grid_sweep` over 4x4-TILE indices; the arguments are CELL columns since round 5, so the
 *  entry point carries a new name: a caller of the old one fails to link instead of sweeping the wrong range) */
int smx_d_grid_sweep_cols(smx_ctx* ctx, int32_t phase, int32_t x_lo, int32_t x_hi);   /* the grid tiles of the COLUMNS [x_lo, x_hi) that belong to `phase`, colour by colour */
/* ---- column strips driven INSIDE the library (csrc/soil_strips_host.h): the tick of SoilMachine.cpp:283-329 on this rank's strip of ONE
 * map; halo columns and migrating particles go from the pack kernel to the unpack kernel of the neighbour without leaving device memory.
 *   smx_strips_attach_rccl   transport = RCCL (ncclSend / ncclRecv / ncclAllReduce / ncclAllGather over xGMI) on this context's stream;
 *                            rank 0 makes the 128-byte id with smx_strips_rccl_unique_id and hands it to the other ranks (any side channel)
 *   smx_strips_attach        transport = the caller's callbacks on HOST buffers (loop-back threads, gloo, MPI ...; all calls block; send must
 *                            not wait for the matching recv to be posted... recv returns 0 and *bytes, or 1 and the needed *bytes if cap is short)
 *   smx_strips_tick          every rank calls it with the same arguments; requires the batched engine
 * The result is a function of (world, inset, seam_halfwidth) only -- smx_set_batch_strips runs the same schedule in ONE context. */
typedef struct smx_transport {
  void* user;
  int (*send)(void* user, int32_t dst, const void* buf, uint64_t bytes);
  int (*recv)(void* user, int32_t src, void* buf, uint64_t cap, uint64_t* bytes);
  int (*allreduce_sum_u64)(void* user, uint64_t* v);
  int (*allreduce_max_u64)(void* user, uint64_t* v);
  int (*allreduce_min_u32)(void* user, uint32_t* a, uint64_t n);
  /* every rank contributes `bytes` (<= each) bytes; all = world blocks of `each` bytes in rank order, sizes[r] = rank r's bytes */
  int (*allgather)(void* user, const void* mine, uint64_t bytes, void* all, uint64_t each, uint64_t* sizes);
} smx_transport;
int smx_strips_rccl_unique_id(void* out128);
int smx_strips_attach_rccl(smx_ctx* ctx, const void* unique_id128, int32_t rank, int32_t world, int32_t inset, int32_t seam_halfwidth);
int smx_strips_attach(smx_ctx* ctx, const smx_transport* t, int32_t rank, int32_t world, int32_t inset, int32_t seam_halfwidth);
int smx_strips_detach(smx_ctx* ctx);
int smx_strips_tick(smx_ctx* ctx, int32_t nwater, int32_t nwind, int32_t dowater, int32_t dowind);
/* what the transport itself cost since it was attached: stream synchronisations it asked for and ncclGroupStart/End pairs (RCCL: 2 groups and at most
 * 2 synchronisations per seam exchange since round 6 -- pack counts, incoming header --, plus one per all-reduce) */
int smx_strips_sync_stats(smx_ctx* ctx, uint64_t* host_syncs, uint64_t* groups);
/* one synthetic seam message (nrec column records in nbytes bytes, npart particle records) from this rank to this very rank through the attached transport's
 * exchange -- with RCCL: the grouped ncclSend / ncclRecv pairs of a real seam exchange -- compared byte for byte on return (0; -4 = it came back changed).
 * What a single device can check of the path several GPUs depend on; nothing in the reference corresponds to it. */
int smx_strips_selfcheck(smx_ctx* ctx, uint32_t nrec, uint32_t nbytes, uint32_t npart);
int smx_strips_stats(smx_ctx* ctx, uint64_t* chunks, uint64_t* seam_chunks, uint64_t* generations, uint64_t* bytes_sent, uint64_t* messages);
/* batched engine: epochs (kernel rounds), generations and dropped child particles since the context was created */
int smx_get_batch_stats(smx_ctx* ctx, uint64_t* epochs, uint64_t* generations, uint64_t* children_lost);
void* smx_stream(smx_ctx* ctx);                          /* the hipStream_t all work is queued on */

/* ---- LBM wind: the D3Q19 two-relaxation-time lattice Boltzmann solver the reference runs as OpenGL
 * compute shaders (source/include/lbmwind/lbmwind.h:75-197; shader/LBM/{lbm,init,collide,stream}.cs; shader/move.cs). It is
 * visual only in the reference (it never feeds WindParticle). A lattice is its own object; cell index (x*NY + y)*NZ + z and
 * the distribution order F[cell*19 + q] at this boundary are the reference's (lbm.cs:8-26,60-80); on the device the
 * distributions are stored direction-major and collide + stream are ONE kernel per step (csrc/soil_lbm.h).
 *   smx_lbm_create / _destroy      lbmw::initialize's buffers (lbmwind.h:77-96) / lbmw::quit (:152-172)
 *   smx_lbm_set_boundary           lbmw::b->fill(NX*NY*NZ, boundary) (lbmwind.h:89, SoilMachine.cpp:239): > 0 = solid
 *   smx_lbm_boundary_from_map      SoilMachine.cpp:235-238 on the device: solid where map.height(ivec2(sx*x, sz*z)) > (sy*y)/SCALE
 *   smx_lbm_initialize             init.cs (lbmwind.h:98-109)
 *   smx_lbm_step(n)                n x { collide.cs; stream.cs } (lbmw::update, lbmwind.h:176-188); rho and v are those collide.cs
 *                                  writes: the moments of the state the step started from
 *   smx_lbm_read                   RHO (n floats), V (n vec4), F (n*19, reference order); any pointer may be NULL
 *   smx_lbm_write_f                overwrite F (tests, restart)
 *   smx_lbm_move                   move.cs on the caller's tracer array (n vec4, in place); the respawn of lbmwind.h:199-218 stays host code.
 *                                  Tracers whose cell or cell+1 lies outside the lattice (the reference moves BEFORE its range check,
 *                                  lbmwind.h:193-215) sample the nearest lattice cell: indices are clamped, never out of bounds
 *   smx_lbm_get_timing             HIP-event time of the step kernels since the last reset */
typedef struct smx_lbm smx_lbm;
int smx_lbm_create(int32_t nx, int32_t ny, int32_t nz, int32_t device, smx_lbm** out);
void smx_lbm_destroy(smx_lbm* l);
const char* smx_lbm_last_error(smx_lbm* l);
int smx_lbm_set_boundary(smx_lbm* l, const float* boundary);
int smx_lbm_boundary_from_map(smx_lbm* l, smx_ctx* map, float sx, float sy, float sz);
int smx_lbm_initialize(smx_lbm* l);
int smx_lbm_step(smx_lbm* l, int32_t n);
int smx_lbm_read(smx_lbm* l, float* rho, float* v4, float* f);
int smx_lbm_write_f(smx_lbm* l, const float* f);
int smx_lbm_move(smx_lbm* l, float* pos4, int32_t n);
int smx_lbm_get_timing(smx_lbm* l, double* ms_steps, uint64_t* steps, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* SOILMX_H */
