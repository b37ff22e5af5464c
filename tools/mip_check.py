"""Lookup-level comparison of the oracle's MIPMap restatement with the compiled reference (oracle/_ref/ref_harness miplookup):
random texture coordinates and footprints through every filter and address mode; reports the lookups that differ."""
import os, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import oracle_binding as ob
from goblin_amd import scene as gs

scene = gs.load_scene("imagetex", gs.config_overrides(resolution=(16, 16), spp=1, depth=2))
o = ob.Oracle(scene)
exr = os.path.join(REPO, "goblin_amd", "scenes", "images", "tiles.exr")
rng = np.random.default_rng(3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
q = np.zeros((n, 6), np.float32)
q[:, :2] = rng.uniform(-0.5, 1.5, (n, 2))
q[:, 2:] = rng.uniform(-1, 1, (n, 4)) * np.exp2(rng.uniform(-9, -1, (n, 1)))
with tempfile.TemporaryDirectory() as d:
    q.tofile(os.path.join(d, "q.f32"))
    for filt in (0, 1, 2, 3):
        for mode in (0, 1, 2):
            subprocess.check_output([os.path.join(REPO, "oracle", "_ref", "ref_harness"), "miplookup", exr, os.path.join(d, "q.f32"), os.path.join(d, "o.f32"),
                                     str(filt), str(mode), "10"])
            ref = np.fromfile(os.path.join(d, "o.f32"), np.float32).reshape(n, 4)
            mine = o.mip_lookup(0, q, filt, mode, 10.0)
            bad = np.any(ref[:, :3] != mine[:, :3], axis=1)
            print("filter", filt, "address", mode, "differing", int(bad.sum()), "of", n)
            if bad.any() and "-v" in sys.argv:
                i = np.flatnonzero(bad)[0]
                print("   query", [float.hex(float(x)) for x in q[i]], "ref", [float.hex(float(x)) for x in ref[i]], "oracle", [float.hex(float(x)) for x in mine[i]])
