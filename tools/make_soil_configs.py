#!/usr/bin/env python3
"""Derive the shipped soil configurations (soilmachine_amd/soils/*.soil) and the parser goldens
(tests/golden/soils_*.json) from the reference's soil/ directory.

The `.soil` files are INPUTS of the hot path. /root/reference does not exist on
the GPU box, so the bench/test configurations have to travel with this repository: this script keeps
only the directives (comments and blank lines dropped) and writes, next to them, the table the
reference's own loadsoil() (source/io.h:7-230, run through oracle/_ref/soil_ref --soils-out) produced
for the ORIGINAL file -- tests/test_soilfile.py checks our parser on the shipped file against that table.
Run in the build container only (needs /root/reference and oracle/_ref/soil_ref).
"""
import glob, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("REF_ROOT", "/root/reference")
OUT = os.path.join(ROOT, "soilmachine_amd", "soils")
GOLD = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "oracle", "_ref", "soil_ref")

for src in sorted(glob.glob(os.path.join(REF, "soil", "*.soil"))):
    name = os.path.basename(src)
    keep = []
    for line in open(src).read().split("\n"):
        h = line.find("#")
        if h != -1:
            line = line[:h]
        if line == "":
            continue
        keep.append(line)
    with open(os.path.join(OUT, name), "w") as f:
        f.write(f"# {name}: directives of the SoilMachine configuration of the same name (comments dropped)\n")
        f.write("\n".join(keep) + "\n")
    with tempfile.TemporaryDirectory() as td:
        js = os.path.join(td, "s.json")
        subprocess.run([BIN, "--soil", src, "--seed", "0", "--ticks", "0", "--soils-out", js],
                       check=True, capture_output=True)
        tab = json.load(open(js))
    json.dump(tab, open(os.path.join(GOLD, "soils_" + name.replace(".soil", ".json")), "w"), indent=1)
    print("wrote", name)
