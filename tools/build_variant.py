"""Experiment builds: libgoblin_hip with one kernel unit recompiled under extra -D flags.

    python tools/build_variant.py <name> <unit> [-DFLAG=...]...     e.g.  wp_refill8 kernels_wavepool -DWP_REFILL=8

Writes goblin_amd/lib/variants/libgoblin_hip_<name>.so (the other units come from the regular build's objects); select
it with GOBLIN_HIP_LIB=<path>.  Variants are throw-away measurement builds, never what ships.
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from goblin_amd import build as b

name, unit, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build_hip()
vdir = os.path.join(b.LIB, "variants")
os.makedirs(vdir, exist_ok=True)
src = os.path.join(b.CSRC, unit + ".hip")
obj = os.path.join(vdir, "%s_%s.o" % (unit, name))
subprocess.check_call([b.HIPCC] + b.HIP_FLAGS + flags + ["-c", src, "-o", obj])
objs = [os.path.join(b.OBJ, os.path.splitext(os.path.basename(s))[0] + ".o") for s in b.HIP_SOURCES if not s.endswith(unit + ".hip")] + [obj]
out = os.path.join(vdir, "libgoblin_hip_%s.so" % name)
subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-fno-gpu-rdc", "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-lpthread"])
os.remove(obj)
print(out)
