"""The C-ABI library loads on a CPU-only box, exports every symbol include/soilmx.h declares, and
refuses -- loudly -- to create a context without a GPU (no CPU fallback on the product path)."""
import ctypes as C
import os
import re

import pytest

from soilmachine_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "soilmx.h")).read()
    return sorted(set(re.findall(r"\b(smx_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = capi.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"libsoilmx.so does not export {n}"
    assert sorted(capi.SYMBOLS) == names, "capi.SYMBOLS and include/soilmx.h disagree"


def test_struct_sizes():
    assert C.sizeof(capi.Config) == 32
    assert C.sizeof(capi.Counters) == 20 * 8
    assert C.sizeof(capi.Timing) == 96 + 48 + 16 + 64 + 24   # + the three streaming-kernel timers of round 2, + the step-launch totals of round 5, + the persistent launches of round 6


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = capi.load()
    cfg = capi.Config(16, 16, 80, 0, 4096, 0, 0)
    h = C.c_void_p()
    rc = L.smx_create(C.byref(cfg), C.byref(h))
    assert rc != 0
    assert b"no HIP device" in L.smx_last_error(h) or b"hip" in L.smx_last_error(h).lower()
    L.smx_destroy(h)
