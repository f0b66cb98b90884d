"""ctypes binding of the C-ABI in include/soilmx.h (libsoilmx.so, hand-written HIP for gfx950).

There is NO CPU fallback: importing the library without a built ``libsoilmx.so`` raises, and creating a
context without a visible HIP device fails with the library's own error text.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsoilmx.so")

ENGINE_SERIAL = 0
ENGINE_SPECULATIVE = 1
ENGINE_BATCHED = 2
ENGINE_RELAXED = 3


class Config(C.Structure):
    _fields_ = [("dimx", C.c_int32), ("dimy", C.c_int32), ("scale", C.c_int32), ("device", C.c_int32),
                ("pool_capacity", C.c_uint64), ("engine", C.c_int32), ("reserved", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "steps_water_top", "steps_water_all", "steps_wind", "nested_particles", "floods", "cascade_calls",
        "cascade_transfers", "wcascade_calls", "grid_active_cells", "rand_calls", "pool_free", "pool_overflow",
        "spec_rounds", "spec_aborts")] + [("reserved", C.c_uint64 * 2), ("spec_subphases_cut", C.c_uint64), ("spec_serial_particles", C.c_uint64),
                                           ("flood_nested_steps", C.c_uint64), ("grid_nested_steps", C.c_uint64)]

    def as_dict(self) -> dict:
        d = {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "reserved"}
        d["spec_fallbacks"], d["spec_executed"] = int(self.reserved[0]), int(self.reserved[1])
        return d


# smx_transport (include/soilmx.h): callbacks on host buffers for smx_strips_attach
SEND_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64)
RECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64))
RED64_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64))
MIN32_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_uint64)
GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64))


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("send", SEND_FN), ("recv", RECV_FN), ("allreduce_sum_u64", RED64_FN), ("allreduce_max_u64", RED64_FN),
                ("allreduce_min_u32", MIN32_FN), ("allgather", GATHER_FN)]


class Timing(C.Structure):
    _fields_ = [("ms_water", C.c_double), ("ms_grid", C.c_double), ("ms_wind", C.c_double), ("ms_freq", C.c_double),
                ("launches_water", C.c_uint64), ("launches_grid", C.c_uint64), ("launches_wind", C.c_uint64),
                ("launches_freq", C.c_uint64), ("ms_kernel_water", C.c_double), ("ms_kernel_wind", C.c_double),
                ("launches_kernel_water", C.c_uint64), ("launches_kernel_wind", C.c_uint64),
                ("ms_kernel_classify", C.c_double), ("ms_kernel_gridtiles", C.c_double), ("ms_kernel_mapfreq", C.c_double),
                ("launches_kernel_classify", C.c_uint64), ("launches_kernel_gridtiles", C.c_uint64),
                ("launches_kernel_mapfreq", C.c_uint64),
                ("launches_step_water", C.c_uint64), ("launches_step_wind", C.c_uint64),
                ("ms_kernel_epochs", C.c_double), ("ms_kernel_tail", C.c_double), ("ms_kernel_grid_children", C.c_double),
                ("launches_kernel_epochs", C.c_uint64), ("epochs_kernel_epochs", C.c_uint64), ("launches_kernel_tail", C.c_uint64),
                ("epochs_kernel_tail", C.c_uint64), ("launches_kernel_grid_children", C.c_uint64),
                ("ms_kernel_floods", C.c_double), ("launches_kernel_floods", C.c_uint64), ("launches_floods_all", C.c_uint64)]

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/soilmx.h declares (tests/test_capi_symbols.py checks the library exports them all)
SYMBOLS = [
    "smx_create", "smx_create_strip", "smx_destroy", "smx_last_error", "smx_set_soils", "smx_set_scale", "smx_srand", "smx_rand", "smx_rand_advance", "smx_get_rand_state", "smx_set_rand_state",
    "smx_initialize", "smx_import_columns", "smx_import_frequency", "smx_num_sections", "smx_export_columns",
    "smx_read_frequency", "smx_read_heights", "smx_read_surface", "smx_tick_water", "smx_grid_pass", "smx_tick_wind",
    "smx_map_frequency", "smx_reset_frequency", "smx_tick", "smx_sync", "smx_add", "smx_remove",
    "smx_particle_cascade", "smx_water_cascade", "smx_seep", "smx_top", "smx_normals", "smx_heights_bilinear", "smx_fill_vertices", "smx_fill_vertices_cut", "smx_fill_vertex_cut",
    "smx_digest", "smx_save", "smx_load", "smx_get_counters", "smx_get_timing", "smx_get_counters_sized", "smx_get_timing_sized", "smx_timing_reset", "smx_set_engine", "smx_set_spec_limits", "smx_stream", "smx_set_batch_dilate", "smx_get_batch_stats", "smx_set_batch_strips", "smx_set_relax_wind", "smx_set_relax_launch", "smx_set_grid_interleave", "smx_set_water_generations", "smx_get_water_generations", "smx_set_water_stagger", "smx_get_water_stagger", "smx_strips_rccl_unique_id", "smx_strips_attach_rccl", "smx_strips_attach", "smx_strips_detach", "smx_strips_tick", "smx_strips_stats", "smx_strips_sync_stats", "smx_strips_selfcheck",
    "smx_d_set_own", "smx_d_gen_begin", "smx_d_gen_spawn", "smx_d_next_phase", "smx_d_chunk", "smx_d_gen_end", "smx_d_pack_columns",
    "smx_d_unpack_columns", "smx_d_pack_particles", "smx_d_unpack_particles", "smx_d_grid_begin", "smx_d_grid_sweep_cols",
    "smx_lbm_create", "smx_lbm_destroy", "smx_lbm_last_error", "smx_lbm_set_boundary", "smx_lbm_boundary_from_map", "smx_lbm_initialize",
    "smx_lbm_step", "smx_lbm_read", "smx_lbm_write_f", "smx_lbm_move", "smx_lbm_get_timing",
]

_lib = None


def load() -> C.CDLL:
    """Load libsoilmx.so and declare the prototypes. Fails loudly if the HIP library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build the HIP extension first "
                           f"(python -c 'import __graft_entry__ as g; g.build()'); soilmx has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, u64, dbl, flt = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_double, C.c_float
    L.smx_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.smx_destroy.argtypes = [vp]; L.smx_destroy.restype = None
    L.smx_last_error.argtypes = [vp]; L.smx_last_error.restype = C.c_char_p
    L.smx_set_soils.argtypes = [vp, vp, i32]
    L.smx_set_scale.argtypes = [vp, i32]
    L.smx_srand.argtypes = [vp, u32]
    L.smx_rand.argtypes = [vp, C.POINTER(i32)]
    L.smx_rand_advance.argtypes = [vp, u64]
    L.smx_get_rand_state.argtypes = [vp, vp, C.POINTER(u32), C.POINTER(u64)]
    L.smx_set_rand_state.argtypes = [vp, vp, u32, u64]
    L.smx_initialize.argtypes = [vp, i32, vp, i32]
    L.smx_import_columns.argtypes = [vp] + [vp] * 5
    L.smx_import_frequency.argtypes = [vp] + [vp] * 3
    L.smx_num_sections.argtypes = [vp, C.POINTER(u64)]
    L.smx_export_columns.argtypes = [vp] + [vp] * 5
    L.smx_read_frequency.argtypes = [vp] + [vp] * 3
    L.smx_read_heights.argtypes = [vp, vp]
    L.smx_read_surface.argtypes = [vp, vp]
    L.smx_tick_water.argtypes = [vp, i32]
    L.smx_grid_pass.argtypes = [vp]
    L.smx_tick_wind.argtypes = [vp, i32]
    L.smx_map_frequency.argtypes = [vp]
    L.smx_reset_frequency.argtypes = [vp]
    L.smx_tick.argtypes = [vp, i32, i32, i32, i32]
    L.smx_sync.argtypes = [vp]
    L.smx_add.argtypes = [vp, i32, i32, dbl, u32]
    L.smx_remove.argtypes = [vp, i32, i32, dbl, C.POINTER(dbl)]
    L.smx_particle_cascade.argtypes = [vp, flt, flt, i32]
    L.smx_water_cascade.argtypes = [vp, i32, i32, i32]
    L.smx_seep.argtypes = [vp, i32, i32]
    L.smx_top.argtypes = [vp, i32, i32, C.POINTER(u32), C.POINTER(dbl), C.POINTER(dbl), C.POINTER(dbl), C.POINTER(i32)]
    L.smx_normals.argtypes = [vp, vp]
    L.smx_fill_vertices.argtypes = [vp, vp, C.c_int32, vp]
    L.smx_fill_vertices_cut.argtypes = [vp, vp, C.c_int32, C.c_int32, dbl, vp]
    L.smx_fill_vertex_cut.argtypes = [vp, vp, C.c_int32, C.c_int32, dbl, C.c_int32, C.c_int32, vp]
    L.smx_heights_bilinear.argtypes = [vp, vp, i32, vp]
    L.smx_save.argtypes = [vp, C.c_char_p]
    L.smx_load.argtypes = [vp, C.c_char_p]
    L.smx_digest.argtypes = [vp, C.POINTER(dbl), C.POINTER(u64), C.POINTER(u64)]
    L.smx_get_counters.argtypes = [vp, C.POINTER(Counters)]
    L.smx_get_timing.argtypes = [vp, C.POINTER(Timing)]
    L.smx_get_counters_sized.argtypes = [vp, C.POINTER(Counters), u64]
    L.smx_get_timing_sized.argtypes = [vp, C.POINTER(Timing), u64]
    L.smx_timing_reset.argtypes = [vp]
    L.smx_set_engine.argtypes = [vp, i32]
    L.smx_set_water_generations.argtypes = [vp, i32]
    L.smx_get_water_generations.argtypes = [vp, C.POINTER(i32)]
    L.smx_set_water_stagger.argtypes = [vp, i32]
    L.smx_get_water_stagger.argtypes = [vp, C.POINTER(i32)]
    L.smx_set_spec_limits.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.smx_set_batch_dilate.argtypes = [vp, i32]
    L.smx_set_batch_strips.argtypes = [vp, i32, i32, i32]
    L.smx_set_relax_wind.argtypes = [vp, u32, i32]
    L.smx_set_relax_launch.argtypes = [vp, i32, i32]
    L.smx_create_strip.argtypes = [C.POINTER(Config), i32, i32, C.POINTER(vp)]
    L.smx_strips_rccl_unique_id.argtypes = [vp]
    L.smx_strips_attach_rccl.argtypes = [vp, vp, i32, i32, i32, i32]
    L.smx_strips_attach.argtypes = [vp, C.POINTER(Transport), i32, i32, i32, i32]
    L.smx_strips_detach.argtypes = [vp]
    L.smx_strips_tick.argtypes = [vp, i32, i32, i32, i32]
    L.smx_strips_stats.argtypes = [vp] + [C.POINTER(u64)] * 5
    L.smx_strips_sync_stats.argtypes = [vp] + [C.POINTER(u64)] * 2
    L.smx_strips_selfcheck.argtypes = [vp, u32, u32, u32]
    L.smx_d_set_own.argtypes = [vp, i32, i32]
    L.smx_d_gen_begin.argtypes = [vp, i32, u32, vp, vp]
    L.smx_d_gen_spawn.argtypes = [vp, vp, C.POINTER(u32)]
    L.smx_d_next_phase.argtypes = [vp]
    L.smx_d_chunk.argtypes = [vp, u32, C.POINTER(u32)]
    L.smx_d_gen_end.argtypes = [vp, vp, u32, C.POINTER(u32)]
    L.smx_d_pack_columns.argtypes = [vp, i32, i32, vp, u64, C.POINTER(u64)]
    L.smx_d_unpack_columns.argtypes = [vp, vp, u64]
    L.smx_d_pack_particles.argtypes = [vp, i32, i32, vp, u64, C.POINTER(u64)]
    L.smx_d_unpack_particles.argtypes = [vp, vp, u64]
    L.smx_d_grid_begin.argtypes = [vp]
    L.smx_d_grid_sweep_cols.argtypes = [vp, i32, i32, i32]
    L.smx_get_batch_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.smx_lbm_create.argtypes = [i32, i32, i32, i32, C.POINTER(vp)]
    L.smx_lbm_destroy.argtypes = [vp]; L.smx_lbm_destroy.restype = None
    L.smx_lbm_last_error.argtypes = [vp]; L.smx_lbm_last_error.restype = C.c_char_p
    L.smx_lbm_set_boundary.argtypes = [vp, vp]
    L.smx_lbm_boundary_from_map.argtypes = [vp, vp, flt, flt, flt]
    L.smx_lbm_initialize.argtypes = [vp]
    L.smx_lbm_step.argtypes = [vp, i32]
    L.smx_lbm_read.argtypes = [vp, vp, vp, vp]
    L.smx_lbm_write_f.argtypes = [vp, vp]
    L.smx_lbm_move.argtypes = [vp, vp, i32]
    L.smx_lbm_get_timing.argtypes = [vp, C.POINTER(dbl), C.POINTER(u64), i32]
    L.smx_stream.argtypes = [vp]; L.smx_stream.restype = vp
    for name in SYMBOLS:
        f = getattr(L, name)
        if name not in ("smx_destroy", "smx_last_error", "smx_stream", "smx_lbm_destroy", "smx_lbm_last_error"):
            f.restype = C.c_int
    _lib = L
    return L


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
