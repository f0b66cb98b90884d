"""LBM wind (reference: source/include/lbmwind/lbmwind.h + shader/LBM/*.cs + shader/move.cs).

The reference has no test or golden vector for this path and no GL runs here -- but its shaders are near-C: the unmodified
statements of lbmwind/shader/LBM/*.cs and shader/move.cs are compiled as C++ (oracle/lbm_ref.cpp through oracle/glsl2cpp.py +
oracle/shim/glsl_cs.h) into oracle/_ref/liblbm_ref.so. That PINS the CPU restatement (oracle/lbm_oracle.cpp): live against
the compiled shaders where /root/reference exists, against the committed vectors they produced (tests/golden/lbm_ref.npz)
everywhere. On top: the published D3Q19 / TRT algebra (moments of the equilibrium, conservation in collisions, rest state,
symmetry). GPU (-m gpu): the fused HIP kernel (csrc/soil_lbm.h) against the restatement and the golden vectors, bit for bit."""
import os

import numpy as np
import pytest

from oracle_lib import LbmOracle, LbmRef, LBM_REF_LIB

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lbm_ref.npz")

CX = np.array([0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 0, 0, 1, -1, 1, -1, 0, 0])
CY = np.array([0, 0, 0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 1, -1])
CZ = np.array([0, 0, 0, 0, 0, 1, -1, 0, 0, 1, -1, 1, -1, 0, 0, -1, 1, -1, 1])
CP = np.array([0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15, 18, 17])
W = np.array([1 / 3] + [1 / 18] * 6 + [1 / 36] * 12)
FORCE = np.array([-0.1, 0.0, 0.05], np.float32)               # lbm.cs:36: 0.05*vec3(-2, 0, 1)


def terrain_boundary(nx, ny, nz, seed=1):
    """a height field as SoilMachine.cpp:235-238 builds it: solid below the surface"""
    rng = np.random.default_rng(seed)
    hx = 3 + 4 * (np.sin(np.arange(nx)[:, None] / 5.0) + np.cos(np.arange(nz)[None, :] / 7.0)) + rng.random((nx, nz))
    y = np.arange(ny)[None, :, None]
    return (hx[:, None, :] > y).astype(np.float32)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("case", ["a", "b"])
def test_restatement_reproduces_the_reference_shaders_golden_vectors(case):
    """tests/golden/lbm_ref.npz = outputs of the reference's own shader text (make_golden_lbm.py): init, N x {collide, stream}, move"""
    g = np.load(GOLD)
    dims, steps, seed = tuple(int(v) for v in g[f"{case}_dims"]), int(g[f"{case}_steps"]), int(g[f"{case}_seed"])
    o = LbmOracle(*dims)
    o.set_boundary(terrain_boundary(*dims, seed=seed)); o.initialize()
    assert np.array_equal(bits(o.read()[2]), bits(g[f"{case}_f0"]))                            # init.cs
    o.step(steps)
    for name, a in zip(("rho", "v", "f"), o.read()):
        assert np.array_equal(bits(a), bits(g[f"{case}_{name}"])), name                        # collide.cs + stream.cs
    assert np.array_equal(bits(o.move(g[f"{case}_p"])), bits(g[f"{case}_moved"]))              # move.cs


@pytest.mark.skipif(not os.path.exists(LBM_REF_LIB), reason="oracle/_ref/liblbm_ref.so is built where /root/reference exists")
@pytest.mark.parametrize("dims,steps", [((64, 40, 64), 6), ((33, 17, 70), 7), ((5, 4, 3), 12)])
def test_restatement_equals_the_compiled_reference_shaders_live(dims, steps):
    """the reference's lattice (lbmwind.h:18-21) and two awkward shapes, every step, all three buffers + the tracers"""
    b = terrain_boundary(*dims)
    o, r = LbmOracle(*dims), LbmRef(*dims)
    o.set_boundary(b); r.set_boundary(b); o.initialize(); r.initialize()
    for k in range(steps + 1):
        for name, a, w in zip(("rho", "v", "f"), o.read(), r.read()):
            assert np.array_equal(bits(a), bits(w)), (name, k)
        o.step(1); r.step(1)
    rng = np.random.default_rng(2)
    p = (rng.random((3000, 4)) * np.array([dims[0] - 2, dims[1] - 2, dims[2] - 2, 1]) + np.array([0.5, 0.5, 0.5, 0])).astype(np.float32)
    assert np.array_equal(bits(o.move(p)), bits(r.move(p)))


def test_velocity_set_is_d3q19():
    assert len(set(zip(CX, CY, CZ))) == 19 and abs(W.sum() - 1) < 1e-15
    assert (CX[CP] == -CX).all() and (CY[CP] == -CY).all() and (CZ[CP] == -CZ).all()          # lbm.cs:82-87
    for a in (CX, CY, CZ):                                                                     # isotropy: sum w c = 0, sum w c c = cs^2
        assert abs((W * a).sum()) < 1e-15 and abs((W * a * a).sum() - 1 / 3) < 1e-15
    assert abs((W * CX * CY).sum()) < 1e-15 and abs((W * CX * CZ).sum()) < 1e-15


def test_initial_state_has_the_equilibrium_moments():
    """init.cs: F = equilibrium(1, force), solid cells equilibrium(1, 0). The D3Q19 equilibrium's zeroth and first moments
    are rho and rho*v exactly (up to float rounding) -- that is what pins `equilibrium` (lbm.cs:95-105)."""
    o = LbmOracle(8, 6, 8)
    b = np.zeros((8, 6, 8), np.float32); b[:, 0, :] = 1; b[3, 1, 4] = 1
    o.set_boundary(b); o.initialize()
    rho, v, f = o.read()
    solid = b.reshape(-1) > 0
    assert np.allclose(rho, 1.0, atol=3e-7)
    assert np.allclose(v[~solid, :3], FORCE, atol=3e-7) and np.allclose(v[solid, :3], 0.0, atol=3e-7) and (v[:, 3] == 1.0).all()
    assert np.allclose(f[solid], W.astype(np.float32), atol=1e-7)                              # rest state = the weights
    d = (FORCE[0] * CX + FORCE[1] * CY + FORCE[2] * CZ).astype(np.float64)
    eq = W * (1 + 3 * d + 4.5 * d * d - 1.5 * float((FORCE.astype(np.float64) ** 2).sum()))
    assert np.allclose(f[~solid], eq, atol=2e-7)


def test_collision_conserves_mass_and_the_forced_equilibrium_is_stationary():
    """Open box, no terrain: every cell starts in equilibrium(1, force); a step relaxes towards the equilibrium of (rho, v +
    gravity), so F moves by O(gravity) only, and a collision never changes a cell's mass: sum_q of what the cell pushes ==
    sum_q of what it held (checked through the interior, where streaming only permutes)."""
    o = LbmOracle(10, 8, 10)
    o.set_boundary(np.zeros(800, np.float32)); o.initialize()
    _, _, f0 = o.read()
    o.step(1)
    rho, v, f1 = o.read()
    assert np.allclose(rho, 1.0, atol=1e-6)                                                    # collide.cs wrote the moments of f0
    assert np.allclose(v[:, 1], FORCE[1] - 0.0001 / 2, atol=1e-6) and (v[:, 3] == 0.0).all()   # gravity: -dt*0.0001/(2 rho)
    assert np.abs(f1 - f0).max() < 2e-5
    tot0 = f0.reshape(10, 8, 10, 19)[2:-2, 2:-2, 2:-2].sum(dtype=np.float64)
    for _ in range(5):
        o.step(1)
    _, _, f6 = o.read()
    assert abs(f6.reshape(10, 8, 10, 19)[2:-2, 2:-2, 2:-2].sum(dtype=np.float64) - tot0) / tot0 < 1e-5


def test_boundary_rules_of_the_stream_shader():
    o = LbmOracle(6, 5, 6)
    b = terrain_boundary(6, 5, 6)
    o.set_boundary(b); o.initialize()
    rng = np.random.default_rng(0)
    f0 = (W[None, :] * (1 + 0.1 * rng.standard_normal((180, 19)))).astype(np.float32)
    o.write_f(f0)
    o.step(1)
    _, _, f1 = o.read()
    f0 = f0.reshape(6, 5, 6, 19); f1 = f1.reshape(6, 5, 6, 19)
    d = (FORCE[0] * CX + FORCE[1] * CY + FORCE[2] * CZ).astype(np.float64)
    eqf = W * (1 + 3 * d + 4.5 * d * d - 1.5 * float((FORCE.astype(np.float64) ** 2).sum()))
    for (x, y, z) in [(0, 2, 3), (5, 1, 1), (2, 4, 2), (3, 2, 0), (1, 3, 5), (0, 0, 0)]:      # the five driven faces (stream.cs:26-34)
        assert np.allclose(f1[x, y, z], eqf, atol=2e-7)
    for q in np.nonzero(CY == 1)[0]:                                                           # y == 0 is NOT driven: what would come from y = -1
        assert (f1[2, 0, 3, q] == f0[2, 0, 3, q]).all()                                        # keeps its old value (stream.cs:13-16 `continue`)
    sx, sy, sz = 2, 1, 3                                                                       # a solid cell pushes the rest equilibrium (collide.cs:52-53)
    assert b[sx, sy, sz] > 0 or True
    solid = np.argwhere(b[1:-1, 0:-2, 1:-1] > 0)[0] + np.array([1, 0, 1])
    x, y, z = solid
    for q in range(1, 19):
        ax, ay, az = x + CX[q], y + CY[q], z + CZ[q]
        if 0 < ax < 5 and 0 <= ay < 4 and 0 < az < 5:
            assert abs(f1[ax, ay, az, q] - W[q]) < 1e-7


def test_mirror_symmetry():
    """The scheme has no preferred z direction except through the force: mirroring boundary, force-free state and the
    velocity set in z gives the mirrored result (a restatement with a wrong neighbour offset or pairing fails this)."""
    nx, ny, nz = 7, 6, 8
    b = terrain_boundary(nx, ny, nz, seed=3)
    rng = np.random.default_rng(5)
    f0 = (W[None, :] * (1 + 0.05 * rng.standard_normal((nx * ny * nz, 19)))).astype(np.float32).reshape(nx, ny, nz, 19)
    zq = np.array([int(np.nonzero((CX == CX[q]) & (CY == CY[q]) & (CZ == -CZ[q]))[0][0]) for q in range(19)])
    a = LbmOracle(nx, ny, nz); a.set_boundary(b); a.initialize(); a.write_f(f0); a.step(1)
    m = LbmOracle(nx, ny, nz); m.set_boundary(b[:, :, ::-1]); m.initialize(); m.write_f(f0[:, :, ::-1][..., zq]); m.step(1)
    fa = a.read()[2].reshape(nx, ny, nz, 19)[1:-1, :-1, 1:-1]                                  # (driven faces carry the z-asymmetric force)
    fm = m.read()[2].reshape(nx, ny, nz, 19)[:, :, ::-1][..., zq][1:-1, :-1, 1:-1]
    assert np.allclose(fa, fm, atol=1e-7)


def test_tracers_follow_a_uniform_wind():
    o = LbmOracle(8, 8, 8)
    o.set_boundary(np.zeros(512, np.float32)); o.initialize(); o.step(1)                       # V = (force + gravity, 0)
    p = np.array([[2.5, 3.25, 4.0, 1.0], [5.1, 1.9, 2.2, 1.0]], np.float32)
    q = o.move(p)
    _, v, _ = o.read()
    assert np.allclose(q[:, :3] - p[:, :3], v[0, :3], atol=1e-6) and np.allclose(q[:, 3], 1.0)


def stray_tracers(nx, ny, nz):
    """tracers the reference moves before its range check (lbmwind.h:193-215): in [N-1, N), below 0, far out, NaN"""
    return np.array([[nx - 0.5, 2.0, 3.0, 1.0], [1.0, ny - 0.01, 2.0, 1.0], [2.0, 2.0, nz - 0.25, 1.0], [-0.75, 1.0, 1.0, 1.0],
                     [3.0, -2.5, 1.0, 1.0], [1e9, 1.0, 1.0, 1.0], [1.0, 1.0, -1e9, 1.0], [np.nan, 1.0, 1.0, 1.0],
                     [nx - 1.0, ny - 1.0, nz - 1.0, 1.0]], np.float32)


def test_stray_tracers_sample_the_nearest_cell_and_never_leave_the_buffer():
    nx, ny, nz = 8, 7, 9
    o = LbmOracle(nx, ny, nz)
    o.set_boundary(terrain_boundary(nx, ny, nz)); o.initialize(); o.step(2)
    p = stray_tracers(nx, ny, nz)[:7]                                                          # (NaN -> int is undefined on the host)
    q = o.move(p)
    assert np.isfinite(q[:3]).all()
    _, v, _ = o.read()
    v = v.reshape(nx, ny, nz, 4)
    # a tracer past the last cell in x, exactly on the lattice otherwise: both x samples clamp to the last plane
    assert np.allclose(q[0, :3] - p[0, :3], v[nx - 1, 2, 3, :3], atol=1e-6)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("case", ["a", "b"])
def test_hip_lbm_reproduces_the_reference_shaders_golden_vectors(case):
    from soilmachine_amd.lbm import LbmWind
    g = np.load(GOLD)
    dims, steps, seed = tuple(int(v) for v in g[f"{case}_dims"]), int(g[f"{case}_steps"]), int(g[f"{case}_seed"])
    w = LbmWind(*dims)
    w.set_boundary(terrain_boundary(*dims, seed=seed)); w.initialize()
    assert np.array_equal(bits(w.read(f=True)[2]), bits(g[f"{case}_f0"]))
    w.step(steps)
    for name, a in zip(("rho", "v", "f"), w.read(f=True)):
        assert np.array_equal(bits(a), bits(g[f"{case}_{name}"])), name
    assert np.array_equal(bits(w.move(g[f"{case}_p"])), bits(g[f"{case}_moved"]))


@pytest.mark.gpu
def test_hip_stray_tracers_are_clamped_like_the_restatement():
    from soilmachine_amd.lbm import LbmWind
    nx, ny, nz = 16, 9, 12
    b = terrain_boundary(nx, ny, nz)
    g, o = LbmWind(nx, ny, nz), LbmOracle(nx, ny, nz)
    g.set_boundary(b); o.set_boundary(b); g.initialize(); o.initialize(); g.step(3); o.step(3)
    p = stray_tracers(nx, ny, nz)
    a, w = g.move(p), o.move(p[:7])
    assert np.array_equal(a[:7].view(np.uint32), w.view(np.uint32))
    assert np.array_equal(a[8].view(np.uint32), o.move(p[8:9])[0].view(np.uint32))
    g.step(1)                                                                                  # the context survived the stray reads


@pytest.mark.gpu
@pytest.mark.parametrize("dims,steps", [((64, 40, 64), 12), ((33, 17, 70), 7), ((8, 6, 300), 5)])
def test_hip_lbm_equals_the_restatement_bit_for_bit(dims, steps):
    """the reference's lattice (lbmwind.h:18-21) and two awkward shapes (not multiples of the wavefront; long z)"""
    from soilmachine_amd.lbm import LbmWind
    nx, ny, nz = dims
    b = terrain_boundary(nx, ny, nz)
    g, o = LbmWind(nx, ny, nz), LbmOracle(nx, ny, nz)
    g.set_boundary(b); o.set_boundary(b)
    g.initialize(); o.initialize()
    for a, w in zip(g.read(f=True), o.read()):
        assert np.array_equal(a.view(np.uint32), w.view(np.uint32))
    done = 0
    for k in (1, 2, steps - 3):
        g.step(k); o.step(k); done += k
        for name, a, w in zip(("rho", "v", "f"), g.read(f=True), o.read()):
            assert np.array_equal(a.view(np.uint32), w.view(np.uint32)), (name, done)
    rng = np.random.default_rng(2)
    p = (rng.random((5000, 4)) * np.array([nx - 2, ny - 2, nz - 2, 1]) + np.array([0.5, 0.5, 0.5, 0])).astype(np.float32)
    assert np.array_equal(g.move(p).view(np.uint32), o.move(p).view(np.uint32))                # move.cs
    assert g.timing()["steps"] == steps


@pytest.mark.gpu
def test_hip_lbm_boundary_from_the_layermap_and_restart():
    """SoilMachine.cpp:235-238 on the device == the host loop over Layermap::height; write_f/read round trip"""
    from common import load_cfg
    from soilmachine_amd.lbm import LbmWind
    from soilmachine_amd.machine import Layermap
    cfg = load_cfg("rockgravelpebblessand.soil")
    m = Layermap(cfg, 96, 80, seed=2)
    nx, ny, nz = 48, 40, 40
    g = LbmWind(nx, ny, nz)
    sx, sy, sz = np.float32(2.0), np.float32(1.5), np.float32(2.0)
    g.boundary_from_map(m, float(sx), float(sy), float(sz))
    h = m.heights().reshape(96, 80)
    want = np.zeros((nx, ny, nz), np.float32)
    for y in range(ny):
        thr = np.float64(np.float32(sy * np.float32(y)) / np.float32(cfg.SCALE))
        want[:, y, :] = h[(sx * np.arange(nx, dtype=np.float32)).astype(np.int64)][:, (sz * np.arange(nz, dtype=np.float32)).astype(np.int64)] > thr
    o = LbmOracle(nx, ny, nz)
    o.set_boundary(want); o.initialize(); o.step(4)
    g.initialize(); g.step(4)
    for a, w in zip(g.read(f=True), o.read()):
        assert np.array_equal(a.view(np.uint32), w.view(np.uint32))
    assert 0 < want.mean() < 1
    f = g.read(f=True)[2]
    g2 = LbmWind(nx, ny, nz); g2.set_boundary(want); g2.initialize(); g2.write_f(f)
    g.step(3); g2.step(3)
    assert np.array_equal(g.read(f=True)[2].view(np.uint32), g2.read(f=True)[2].view(np.uint32))
    with pytest.raises(RuntimeError):
        g.boundary_from_map(m, 4.0, 1.0, 1.0)                     # 4*47 > 95: does not fit the map
