"""The .soil parser mirror (soilmachine_amd/soilfile.py) against the tables the REFERENCE's own
loadsoil() (source/io.h:7-230) produced for the same files (tests/golden/soils_*.json)."""
import glob
import json
import os

import numpy as np
import pytest

from common import GOLD, SOILS
from soilmachine_amd.soilfile import loadsoil, soils_array, SOIL_DTYPE

FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(SOILS, "*.soil")))


@pytest.mark.parametrize("name", FILES)
def test_parser_matches_reference_loadsoil(name):
    cfg = loadsoil(os.path.join(SOILS, name))
    ref = json.load(open(os.path.join(GOLD, "soils_" + name.replace(".soil", ".json"))))
    assert (cfg.SCALE, cfg.SIZEX, cfg.SIZEY, cfg.NWATER, cfg.NWIND) == (
        ref["scale"], ref["sizex"], ref["sizey"], ref["nwater"], ref["nwind"])
    assert len(cfg.soils) == len(ref["soils"])
    for a, b in zip(cfg.soils, ref["soils"]):
        assert a.name == b["name"]
        for k in ("transports", "erodes", "cascades", "abrades"):
            assert getattr(a, k) == b[k], (a.name, k)
        for k in ("porosity", "solubility", "equrate", "friction", "erosionrate", "maxdiff", "settling", "suspension", "abrasion"):
            assert np.float32(getattr(a, k)) == np.float32(b[k]), (a.name, k)   # density is uninitialised in the reference
    assert len(cfg.layers) == len(ref["layers"])
    for a, b in zip(cfg.layers, ref["layers"]):
        for k in ("type", "min", "bias", "scale", "octaves", "lacunarity", "gain", "frequency"):
            assert np.float32(getattr(a, k)) == np.float32(b[k]), k


def test_first_mention_index_order():
    # forward references allocate the index at first mention
    cfg = loadsoil(os.path.join(SOILS, "rockgravelpebblessand.soil"))
    assert [s.name for s in cfg.soils] == ["Air", "Rock", "Gravel", "Pebbles", "Red Sand"]
    assert [l.type for l in cfg.layers] == [1, 4]
    rock = cfg.soils[1]
    assert (rock.transports, rock.erodes, rock.cascades, rock.abrades) == (2, 2, 2, 2)


def test_soil_struct_layout_matches_c_abi():
    assert SOIL_DTYPE.itemsize == 56          # sizeof(smx_soil): 4 x u32 + 10 x f32
    a = soils_array(loadsoil(os.path.join(SOILS, "default.soil")))
    assert a.shape == (2,) and a[0]["porosity"] == 1.0 and a[1]["transports"] == 1


def test_syntax_errors(tmp_path):
    p = tmp_path / "bad.soil"
    p.write_text("}\n")
    with pytest.raises(SyntaxError):
        loadsoil(str(p))
    p.write_text("LAYER Nope {\n}\n")
    with pytest.raises(SyntaxError):
        loadsoil(str(p))


@pytest.mark.parametrize("name", FILES)
def test_writesoil_round_trip(name, tmp_path):
    """The `.soil` writer the reference only announces (io.h:232): what it writes parses back into the same tables,
    index order included -- with this module's parser and, where the reference is built, with the reference's own."""
    from soilmachine_amd.soilfile import writesoil
    cfg = loadsoil(os.path.join(SOILS, name))
    out = str(tmp_path / ("w_" + name))
    writesoil(cfg, out)
    back = loadsoil(out)
    assert back.to_json() == cfg.to_json() and back.soilmap == cfg.soilmap
    from oracle_lib import have_ref, run_ref
    if have_ref():
        js = str(tmp_path / "ref.json")
        run_ref(out, ticks=0, size=16, soils_out=js, water=False, wind=False)
        ref = json.load(open(js))
        assert [s["name"] for s in ref["soils"]] == [s.name for s in cfg.soils]
        for a, b in zip(cfg.soils, ref["soils"]):
            for k in ("transports", "erodes", "cascades", "abrades"):
                assert getattr(a, k) == b[k], (a.name, k)
            for k in ("porosity", "solubility", "equrate", "friction", "erosionrate", "maxdiff", "settling", "suspension", "abrasion"):
                assert np.float32(getattr(a, k)) == np.float32(b[k]), (a.name, k)
        assert [l["type"] for l in ref["layers"]] == [l.type for l in cfg.layers]
        assert (ref["scale"], ref["nwater"], ref["nwind"]) == (cfg.SCALE, cfg.NWATER, cfg.NWIND)
