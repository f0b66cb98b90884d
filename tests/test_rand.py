"""glibc rand()/srand() restatement (TYPE_3 additive feedback) against the container's libc."""
import ctypes as C

import pytest

from common import load_cfg
from oracle_lib import Oracle


@pytest.mark.parametrize("seed", [0, 1, 42, 2147483647, 123456789])
def test_oracle_rand_matches_libc(seed):
    libc = C.CDLL("libc.so.6")
    libc.srand(C.c_uint(seed))
    o = Oracle(load_cfg("default.soil"), 8, 8, seed=seed)
    for _ in range(2000):
        assert o.L.so_rand(o.h) == libc.rand()
