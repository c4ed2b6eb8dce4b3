#!/usr/bin/env python3
"""A second build of libpymodem_amd.so with extra compiler definitions, for A/B measurements of kernel constants on ONE box (two boxes
differ by more than most of the effects looked for).    tools/build_variant.py NAME -DPM_LPF8_WAVES=5 [...]
-> build/variants/libpymodem_amd_NAME.so; run anything with PYMODEM_AMD_LIB=<that path>."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

name, defs = sys.argv[1], sys.argv[2:]
objdir = os.path.join(ROOT, "build", "variants", name)
os.makedirs(objdir, exist_ok=True)
objs = []
for src in G.SOURCES:
    s = os.path.join(G.CSRC, src)
    o = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
    subprocess.check_call([G.HIPCC] + G.FLAGS + defs + (["-mpopcnt"] if src.endswith(".cpp") else []) + ["-c", s, "-o", o])
    objs.append(o)
out = os.path.join(ROOT, "build", "variants", f"libpymodem_amd_{name}.so")
subprocess.check_call([G.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
