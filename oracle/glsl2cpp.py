#!/usr/bin/env python3
"""GLSL compute shader -> text a C++ compiler accepts together with oracle/shim/glsl_cs.h (TEST INFRASTRUCTURE).

Only DECLARATIONS that have no C++ spelling are rewritten -- every statement of main() and of the helper functions, every
constant table and every expression goes through untouched:
  #version ...                                   -> dropped
  #include NAME                                  -> #include "NAME.h"        (the converted sibling)
  layout(local_size_x = ..) in;                  -> GLSL_LOCAL_SIZE(..)
  layout (std430, binding = N) buffer n { T A[]; };  -> GLSL_BUFFER(T, A)   (GLSL_BUFFER_F for the lattice F: see the shim)
  uniform T x = v;                               -> GLSL_UNIFORM T x = v;
usage: glsl2cpp.py in.cs out.h"""
import re, sys

src = open(sys.argv[1]).read()
src = re.sub(r"^#version.*$", "", src, flags=re.M)
src = re.sub(r"^#include\s+(\S+)\s*$", r'#include "\1.h"', src, flags=re.M)
src = re.sub(r"layout\s*\((local_size[^)]*)\)\s*in\s*;", r"GLSL_LOCAL_SIZE(\1)", src)


def block(m):
    ty, name = m.group(1), m.group(2)
    return f"GLSL_BUFFER_F({name})" if (ty, name) == ("float", "F") else f"GLSL_BUFFER({ty}, {name})"


src = re.sub(r"layout\s*\(std430[^)]*\)\s*buffer\s+\w+\s*\{\s*(\w+)\s+(\w+)\s*\[\s*\]\s*;\s*\}\s*;", block, src)
src = re.sub(r"^uniform\b", "GLSL_UNIFORM", src, flags=re.M)
open(sys.argv[2], "w").write(src)
