"""`.soil` parser -- host-side mirror of the reference's ``loadsoil`` (source/io.h:7-230).

The soil table is an *input* of the hot path: the reference fills the global
``soils`` / ``soilmap`` / ``layers`` tables and the WORLD globals from a text file, then the
particles read them every step.  This module restates that parser, **including its quirks**
(SURVEY.md Appendix D#9), so the POD tables handed to the C-ABI (``smx_set_soils``) are the same
the reference would use:

* one ``SurfParam`` value is reused for every ``SOIL`` block (io.h:35): keys a block does not set
  are inherited from the previous block;
* forward references (``TRANSPORTS X`` before ``SOIL X``) allocate the soil index at first mention
  (io.h:125-152), so index order is first-mention order;
* soil names are ``val[:find('{')-1]`` (io.h:74), values are parsed with ``stof``/``stoi``
  semantics (leading number, trailing junk ignored);
* index 0 is always "Air" (surface.h:41-57), which doubles as *water*.

Pinned by tests/test_soilfile.py against tests/golden/soils_*.json, which were produced by the
reference's own ``loadsoil`` (oracle/ref_driver.cpp ``--soils-out``).
"""
from __future__ import annotations

import copy
import re
from dataclasses import dataclass, field, asdict
from typing import Dict, List

import numpy as np

_NUM = re.compile(r"^\s*[-+]?(\d+\.?\d*([eE][-+]?\d+)?|\.\d+([eE][-+]?\d+)?)")
_INT = re.compile(r"^\s*[-+]?\d+")


def _stof(s: str) -> float:
    m = _NUM.match(s)
    if not m:
        raise ValueError(f"stof: no conversion for {s!r}")
    return float(np.float32(float(m.group(0))))


def _stoi(s: str) -> int:
    m = _INT.match(s)
    if not m:
        raise ValueError(f"stoi: no conversion for {s!r}")
    return int(m.group(0))


@dataclass
class SurfParam:
    """SurfParam (surface.h:11-39). `color` / `phong` are render-only: they never reach the device tables (soils_array), but they
    are parsed and written back so that a file written by writesoil() gives the reference GUI the same look."""
    name: str = ""
    density: float = 0.0      # uninitialised in the reference (surface.h:15); never read on the hot path
    porosity: float = 0.0
    transports: int = 0
    solubility: float = 1.0
    equrate: float = 1.0
    friction: float = 1.0
    erodes: int = 0
    erosionrate: float = 0.0
    cascades: int = 0
    maxdiff: float = 1.0
    settling: float = 0.0
    abrades: int = 0
    suspension: float = 0.0
    abrasion: float = 0.0
    color: str = "808080"     # COLOR rrggbb (io.h:23-33,158); surface.h:17 default vec4(0.5, 0.5, 0.5, 1.0)
    phong: tuple = (0.5, float(np.float32(0.8)), float(np.float32(0.2)), 32.0)   # Ka Kd Ks Kk (io.h:180-183); surface.h:18 (float)

    def color_rgba(self):
        """vec4(R, G, B, 255) / 255 as hexcol computes it (io.h:23-33)"""
        h = self.color
        return tuple(float(np.float32(int(h[i:i + 2], 16)) / np.float32(255.0)) for i in (0, 2, 4)) + (1.0,)


@dataclass
class SurfLayer:
    """SurfLayer (surface.h:65-101)."""
    type: int = 0
    min: float = 0.0
    bias: float = 0.0
    scale: float = 1.0
    octaves: float = 1.0
    lacunarity: float = 1.0
    gain: float = 0.0
    frequency: float = 1.0


@dataclass
class SoilConfig:
    soils: List[SurfParam] = field(default_factory=list)
    soilmap: Dict[str, int] = field(default_factory=dict)
    layers: List[SurfLayer] = field(default_factory=list)
    # WORLD globals with the defaults of SoilMachine.cpp:9-14
    SIZEX: int = 256
    SIZEY: int = 256
    SCALE: int = 80
    NWIND: int = 250
    NWATER: int = 250

    def to_json(self) -> dict:
        return {
            "scale": self.SCALE, "sizex": self.SIZEX, "sizey": self.SIZEY,
            "nwater": self.NWATER, "nwind": self.NWIND,
            "soils": [asdict(s) for s in self.soils],
            "layers": [asdict(l) for l in self.layers],
        }


def _air() -> SurfParam:
    # surface.h:43-49
    return SurfParam(name="Air", density=0.0, porosity=1.0, transports=0, solubility=0.0, equrate=0.0,
                     friction=0.0, erodes=0, erosionrate=0.0, cascades=0, maxdiff=0.0, settling=0.0,
                     abrades=0, suspension=0.0, abrasion=0.0, color="003366")   # vec4(0.0, 0.2, 0.4, 1.0)


def loadsoil(path: str) -> SoilConfig:
    """Restatement of loadsoil (io.h:7-230)."""
    cfg = SoilConfig(soils=[_air()], soilmap={"Air": 0})
    param = SurfParam()
    is_open = False
    soillayer = ""
    with open(path, "r") as f:
        text = f.read()
    for linenr, line in enumerate(text.split("\n"), 1):
        found = line.find("#")
        if found != -1:
            line = line[:found]
        if line == "":
            continue
        if line == "}":
            if not is_open:
                raise SyntaxError(f"Incorrect Syntax in Line {linenr}")
            if soillayer == "SOIL":
                cfg.soils[cfg.soilmap[param.name]] = copy.copy(param)
            is_open = False
            continue
        found = line.find(" ")
        if found == -1:
            raise SyntaxError(f"Incorrect Syntax in Line {linenr}")
        tag, val = line[:found], line[found + 1:]

        if tag in ("SOIL", "LAYER", "WORLD"):
            b = val.find("{")
            if b == -1:
                raise SyntaxError(f"Incorrect Syntax in Line {linenr}")
            if tag == "SOIL":
                param.name = val[:b - 1] if b >= 1 else val[:0]
                if param.name not in cfg.soilmap:
                    cfg.soilmap[param.name] = len(cfg.soils)
                    cfg.soils.append(copy.copy(param))
            elif tag == "LAYER":
                param.name = val[:b - 1] if b >= 1 else val[:0]
                if param.name not in cfg.soilmap:
                    raise SyntaxError(f"Can't find SOIL {param.name}")
                cfg.layers.append(SurfLayer(type=cfg.soilmap[param.name]))
            soillayer = tag
            is_open = True
            continue

        if soillayer == "SOIL":
            if tag in ("TRANSPORTS", "ERODES", "CASCADES", "ABRADES"):
                if val not in cfg.soilmap:
                    cfg.soilmap[val] = len(cfg.soils)
                    cfg.soils.append(copy.copy(param))
                setattr(param, tag.lower(), cfg.soilmap[val])
            elif tag == "DENSITY":
                param.density = _stof(val)
            elif tag == "POROSITY":
                param.porosity = _stof(val)
            elif tag == "SOLUBILITY":
                param.solubility = _stof(val)
            elif tag == "EQUILIBRIUM":
                param.equrate = _stof(val)
            elif tag == "FRICTION":
                param.friction = _stof(val)
            elif tag == "EROSIONRATE":
                param.erosionrate = _stof(val)
            elif tag == "MAXDIFF":
                param.maxdiff = _stof(val)
            elif tag == "SETTLING":
                param.settling = _stof(val)
            elif tag == "SUSPENSION":
                param.suspension = _stof(val)
            elif tag == "ABRASION":
                param.abrasion = _stof(val)
            elif tag == "COLOR":                                  # hexcol (io.h:23-33): >= 6 characters, all of 0-9A-F
                if len(val) < 6 or any(ch not in "0123456789ABCDEF" for ch in val):
                    raise SyntaxError(f"Incorrect Syntax in Line {linenr}")
                param.color = val[:6]
            elif tag in ("Ka", "Kd", "Ks", "Kk"):
                ph = list(param.phong); ph["adsk".index(tag[1])] = _stof(val); param.phong = tuple(ph)
        if soillayer == "LAYER":
            lay = cfg.layers[-1]
            if tag == "MIN":
                lay.min = _stof(val)
            elif tag == "BIAS":
                lay.bias = _stof(val)
            elif tag == "SCALE":
                lay.scale = _stof(val)
            elif tag == "OCTAVES":
                lay.octaves = _stof(val)
            elif tag == "LACUNARITY":
                lay.lacunarity = _stof(val)
            elif tag == "GAIN":
                lay.gain = _stof(val)
            elif tag == "FREQUENCY":
                lay.frequency = _stof(val)
        if soillayer == "WORLD":
            if tag in ("SIZEX", "SIZEY", "SCALE", "NWIND", "NWATER"):
                setattr(cfg, tag, _stoi(val))
    return cfg


SOIL_DTYPE = np.dtype([
    ("transports", "<u4"), ("erodes", "<u4"), ("cascades", "<u4"), ("abrades", "<u4"),
    ("density", "<f4"), ("porosity", "<f4"), ("solubility", "<f4"), ("equrate", "<f4"), ("friction", "<f4"),
    ("erosionrate", "<f4"), ("maxdiff", "<f4"), ("settling", "<f4"), ("suspension", "<f4"), ("abrasion", "<f4"),
])
LAYER_DTYPE = np.dtype([
    ("type", "<u4"), ("min", "<f4"), ("bias", "<f4"), ("scale", "<f4"), ("octaves", "<f4"),
    ("lacunarity", "<f4"), ("gain", "<f4"), ("frequency", "<f4"),
])


def soils_array(cfg: SoilConfig) -> np.ndarray:
    """POD soil table in the layout of ``smx_soil`` (include/soilmx.h)."""
    a = np.zeros(len(cfg.soils), dtype=SOIL_DTYPE)
    for i, s in enumerate(cfg.soils):
        for k in SOIL_DTYPE.names:
            a[i][k] = getattr(s, k)
    return a


def layers_array(cfg: SoilConfig) -> np.ndarray:
    a = np.zeros(len(cfg.layers), dtype=LAYER_DTYPE)
    for i, l in enumerate(cfg.layers):
        for k in LAYER_DTYPE.names:
            a[i][k] = getattr(l, k)
    return a


def writesoil(cfg: SoilConfig, path: str) -> None:
    """The writer the reference only announces (io.h:232 "Should be able to also WRITE to file"): a `.soil` text that
    ``loadsoil`` -- this module's and the reference's -- parses back into exactly these tables, render attributes (COLOR, Ka/Kd/Ks/Kk)
    included: the reference carries `param` over from block to block (io.h:35), so a soil written without them would take the
    previous block's colour in the reference GUI. Because soil indices are
    handed out at first mention (io.h:125-152) the file first declares every soil by an empty block, in index order, and
    then defines each one with ALL its keys (so nothing is inherited from the previous block, io.h:35)."""
    f32 = lambda v: repr(float(np.float32(v)))
    names = [s.name for s in cfg.soils]
    out = ["# written by soilmachine_amd.soilfile.writesoil", "", "WORLD {", f"SCALE {cfg.SCALE}", f"SIZEX {cfg.SIZEX}",
           f"SIZEY {cfg.SIZEY}", f"NWIND {cfg.NWIND}", f"NWATER {cfg.NWATER}", "}", ""]
    start = 0 if cfg.soils and cfg.soils[0] != _air() else 1          # "Air" is built in (surface.h:41-57) unless it was edited
    for s in cfg.soils[1:]:
        out += [f"SOIL {s.name} {{", "}"]
    out.append("")
    for s in cfg.soils[start:]:
        out += [f"SOIL {s.name} {{", f"DENSITY {f32(s.density)}", f"POROSITY {f32(s.porosity)}", f"TRANSPORTS {names[s.transports]}",
                f"SOLUBILITY {f32(s.solubility)}", f"EQUILIBRIUM {f32(s.equrate)}", f"FRICTION {f32(s.friction)}",
                f"ERODES {names[s.erodes]}", f"EROSIONRATE {f32(s.erosionrate)}", f"CASCADES {names[s.cascades]}",
                f"MAXDIFF {f32(s.maxdiff)}", f"SETTLING {f32(s.settling)}", f"ABRADES {names[s.abrades]}",
                f"SUSPENSION {f32(s.suspension)}", f"ABRASION {f32(s.abrasion)}", f"COLOR {s.color}",
                f"Ka {f32(s.phong[0])}", f"Kd {f32(s.phong[1])}", f"Ks {f32(s.phong[2])}", f"Kk {f32(s.phong[3])}", "}", ""]
    for l in cfg.layers:
        out += [f"LAYER {names[l.type]} {{", f"MIN {f32(l.min)}", f"BIAS {f32(l.bias)}", f"SCALE {f32(l.scale)}",
                f"OCTAVES {f32(l.octaves)}", f"LACUNARITY {f32(l.lacunarity)}", f"GAIN {f32(l.gain)}",
                f"FREQUENCY {f32(l.frequency)}", "}", ""]
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
