"""The drop-in boundary: the compat headers (soilmachine_amd/compat/source: layermap.h, surface.h, particle/) keep the
reference's C++ API; everything else of the host tree -- SoilMachine.cpp, its own io.h (.soil parser, exporters), vertex
pool, scene, LBM wind -- is used UNMODIFIED.

* CPU (where /root/reference exists): that host compiles and links against the compat headers + libsoilmx.so
  (tools/build_host.sh), with no file of the host tree shadowed except the three the drop-in replaces.
* GPU: the host, run headless for N ticks, exports exactly the height and colour images the reference's own CPU loop
  produces (computed here from the oracle's state with the exporters' float arithmetic, io.h:234-252)."""
import glob
import os
import subprocess

import numpy as np
import pytest

from common import GOLD, ROOT, SOILS, load_cfg

HOST = os.path.join(ROOT, "build", "host", "soilmachine_headless")
REF = "/root/reference"
FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(SOILS, "*.soil")))


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "SoilMachine.cpp")), reason="reference tree absent")
def test_unmodified_host_compiles_against_compat_headers():
    subprocess.check_call([os.path.join(ROOT, "tools", "build_host.sh")])
    assert os.path.exists(HOST)
    src = os.path.join(ROOT, "build", "host", "source")
    ours = os.path.join(ROOT, "soilmachine_amd", "compat", "source")
    for f in ("layermap.h", "surface.h", "particle"):             # the drop-in: exactly these three
        assert os.path.realpath(os.path.join(src, f)) == os.path.realpath(os.path.join(ours, f))
    for f in ("io.h", "include", "shader"):                        # the host tree's own files, not stand-ins
        assert os.path.realpath(os.path.join(src, f)).startswith(os.path.realpath(REF))
    assert sorted(os.listdir(ours)) == ["layermap.h", "particle", "surface.h"]


def _expected_export(heights, scale):
    """exportheight (io.h:245-252) on Layermap::update's vertex (layermap.h:523-528), in its float arithmetic."""
    pos1 = (scale * heights).astype(np.float32)                 # vec3(p.x, SCALE*(floor+size), p.y) -> float
    q = (pos1 / np.float32(scale)).astype(np.float32)           # float / int
    return (q.astype(np.float64) / np.sqrt(2.0)).astype(np.float32)   # / sqrt(2) (double), then vec4(float)


def _soil_colors(path, names):
    """COLOR of every SOIL block, ordered like `names` (the soilmap order of the parsed file; io.h:25-35 hexcol: two hex
    digits per channel / 255.0f, alpha 1); the built-in Air entry is surface.h:41-51; a SOIL block without COLOR keeps
    SurfParam's default grey."""
    cols = {"Air": np.array([0.0, 0.2, 0.4, 1.0], np.float32)}
    cur = None
    for line in open(path):
        line = line.split("#")[0].strip()
        w = line.split()
        if len(w) >= 2 and w[0] == "SOIL":
            cur = np.array([0.5, 0.5, 0.5, 1.0], np.float32)
            name = line[4:].split("{")[0].strip()
        elif line == "}" and cur is not None:
            cols[name] = cur
            cur = None
        elif cur is not None and len(w) >= 2 and w[0] == "COLOR":
            h = w[1]
            rgb = [np.float32(16 * "0123456789ABCDEF".index(h[2 * k]) + "0123456789ABCDEF".index(h[2 * k + 1])) for k in range(3)]
            cur = (np.array(rgb + [np.float32(255.0)], np.float32) / np.float32(255.0)).astype(np.float32)
    return np.stack([cols[n] for n in names])


@pytest.mark.gpu
@pytest.mark.parametrize("soil,ticks", [("default.soil", 20), ("rockgravelpebblessand.soil", 3)])
def test_unmodified_host_runs_and_matches_reference(tmp_path, soil, ticks):
    if not os.path.exists(HOST):
        pytest.skip("build/host/soilmachine_headless not built (needs the reference tree at build time)")
    from oracle_lib import Oracle
    cfg = load_cfg(soil)
    env = dict(os.environ, TICKS=str(ticks))
    out = subprocess.run([HOST, "-SEED", "0", "-soil", os.path.join(SOILS, soil), "-oh", "h", "-oc", "c"], cwd=tmp_path, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    got = np.fromfile(tmp_path / "h.f32", dtype=np.float32)
    o = Oracle(cfg, cfg.SIZEX, cfg.SIZEY, seed=0, pool=10_000_000)
    o.initialize()
    for _ in range(ticks):
        o.tick(cfg.NWATER, cfg.NWIND, True, True)             # dowatercycles = dowindcycles = true (SoilMachine.cpp:66-67)
    want = _expected_export(o.heights(), cfg.SCALE)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # exportcolor (io.h:234-242): the vertex colour of Layermap::update (layermap.h:475-528, SLICE = 2*SCALE: no cut) =
    # the soil colour of the top section, Air's for an empty column; written as (b, g, r, 1)
    rgba = np.fromfile(tmp_path / "c.rgba.f32", dtype=np.float32).reshape(-1, 4)
    snap = o.snapshot()
    end = np.cumsum(snap.count.astype(np.int64))
    top = np.where(snap.count > 0, snap.type[np.maximum(end - 1, 0)], 0).astype(np.int64)
    col = _soil_colors(os.path.join(SOILS, soil), [sp.name for sp in cfg.soils])
    wantc = np.concatenate([col[top][:, [2, 1, 0]], np.ones((top.size, 1), np.float32)], axis=1)
    assert np.array_equal(rgba.view(np.uint32), wantc.view(np.uint32))
    assert len(np.unique(top)) > 1 or soil == "default.soil"


PROBE = os.path.join(ROOT, "build", "host", "compat_probe")


@pytest.mark.gpu
def test_compat_queries_and_point_modifiers_through_the_class_api(tmp_path):
    """The parts of the class API the reference's main() never calls (tools/compat_probe.cpp): Layermap::height(vec2),
    normal(vec2) and the vertexpool variants (layermap.h:379-413), surface, top, add(pool.get(..)), remove, Particle::cascade --
    after two ticks driven through the same classes. Expected values: the same operations through the C-ABI (each of which
    is pinned to the oracle elsewhere) and, for normal(vec2), its float arithmetic restated on the per-cell normals."""
    import ctypes as C
    from soilmachine_amd.machine import SoilMachine
    if not os.path.exists(PROBE):
        pytest.skip("build/host/compat_probe not built (needs the reference tree at build time)")
    soil = "rocksand.soil"
    cfg = load_cfg(soil)
    out = subprocess.run([PROBE, "-SEED", "4", "-soil", os.path.join(SOILS, soil), "-ticks", "2", "-o", "p.f64"], cwd=tmp_path,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    got = np.fromfile(tmp_path / "p.f64", dtype=np.float64)
    rest, rec = got[0], got[1:].reshape(-1, 17)
    sm = SoilMachine(cfg, dimx=cfg.SIZEX, dimy=cfg.SIZEY, seed=4, nwater=cfg.NWATER, nwind=cfg.NWIND, pool=10_000_000)
    sm.tick(2, sync=True)
    m = sm.map
    m.add(5, 7, 0.0125, 1)
    assert m.remove(9, 3, 0.3) == rest
    m._chk(m.L.smx_particle_cascade(m.h, C.c_float(20.4), C.c_float(11.6), 1))
    nx, ny = cfg.SIZEX, cfg.SIZEY
    pos = rec[:, 0:2].astype(np.float32)
    k = np.arange(500, dtype=np.float32)
    assert np.array_equal(pos[:, 0], np.fmod(k * np.float32(37.77), np.float32(nx - 2)) + np.float32(0.25))
    assert np.array_equal(rec[:, 2], m.heights_bilinear(pos))                      # Layermap::height(vec2)
    nrm = m.normals().reshape(nx, ny, 3)
    ip = np.floor(pos).astype(np.int64)
    w = (pos - np.floor(pos)).astype(np.float32)
    one = np.float32(1.0)
    want = np.zeros((500, 3), np.float32)
    for wt, (dx, dy) in (((one - w[:, 0]) * (one - w[:, 1]), (0, 0)), ((one - w[:, 0]) * w[:, 1], (1, 0)),
                         (w[:, 0] * (one - w[:, 1]), (0, 1)), (w[:, 0] * w[:, 1], (1, 1))):   # the reference's crossed weights (layermap.h:385-388)
        want = (want + wt.astype(np.float32)[:, None] * nrm[ip[:, 0] + dx, ip[:, 1] + dy]).astype(np.float32)
    assert np.array_equal(rec[:, 3:6].astype(np.float32).view(np.uint32), want.view(np.uint32))      # normal(vec2)
    assert np.array_equal(rec[:, 6:9], rec[:, 3:6])                                # normal(vec2, vertexpool)
    assert np.array_equal(rec[:, 9:12].astype(np.float32).view(np.uint32), nrm[ip[:, 0], ip[:, 1]].view(np.uint32))   # normal(ivec2, vertexpool)
    s = m.snapshot()
    end = np.cumsum(s.count.astype(np.int64))
    c = ip[:, 0] * ny + ip[:, 1]
    kk = end[c] - 1
    assert (s.count[c] > 0).all()
    assert np.array_equal(rec[:, 12], m.surface()[c].astype(np.float64))           # surface(ivec2)
    assert np.array_equal(rec[:, 13], s.type[kk].astype(np.float64))               # top(ivec2)
    assert np.array_equal(rec[:, 14], s.size[kk]) and np.array_equal(rec[:, 15], s.floor[kk]) and np.array_equal(rec[:, 16], s.sat[kk])
