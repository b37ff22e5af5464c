"""Which samples of a pixel block differ between the device and the oracle (native samples), per schedule / tie mode.
    python tools/block_hunt.py cornell 1024 1024 16 100 500 8 8"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import oracle_binding as ob, helpers
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
name, res, spp, depth, bx, by, bw, bh = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
seed = 20261003
scene = gs.load_scene(name, gs.config_overrides(resolution=(res, res), spp=spp, depth=depth))
r = HipPathTracer(scene, 0)
x0, x1, y0, y1 = r.window
sub = (x0 + bx, x0 + bx + bw, y0 + by, y0 + by + bh)
o = ob.Oracle(scene)
samples = o.native_samples(seed, window=sub)
ref, _ = o.li_replay(samples, threads=8)
for label, kw in (("mk lean", dict(schedule="megakernel")), ("mk exact", dict(schedule="megakernel", exact_ties=True)),
                  ("wf lean", dict(schedule="wavefront")), ("wf exact", dict(schedule="wavefront", exact_ties=True)),
                  ("mk replay", dict(schedule="megakernel", replay_samples=samples)), ("wf replay", dict(schedule="wavefront", replay_samples=samples)),
                  ("mk stats", dict(schedule="megakernel", stats=True))):
    li = r.render(seed=seed, window=sub, want_li=True, **kw)["li"].cpu().numpy()
    bad = np.nonzero((li != ref).any(axis=1))[0]
    print(label, "differing", len(bad), [(int(i), li[i, :3].tolist(), ref[i, :3].tolist()) for i in bad[:4]])
