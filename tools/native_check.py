"""Debug: native-mode per-sample radiance, device vs oracle, at a given size."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import oracle_binding as ob, helpers
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
res, spp, depth = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scene = gs.load_scene("bunny", gs.config_overrides(resolution=(res, res), spp=spp, depth=depth))
o = ob.Oracle(scene)
seed = 20261003
samples = o.native_samples(seed)
li_ref, _ = o.li_replay(samples, threads=8)
tr = HipPathTracer(scene, 0)
li = tr.render(seed=seed, want_li=True)["li"].cpu().numpy()
li2 = tr.render(replay_samples=samples, want_li=True)["li"].cpu().numpy()
d = np.abs(li[:, :3] - li_ref[:, :3]).max(axis=1)
d2 = np.abs(li2[:, :3] - li_ref[:, :3]).max(axis=1)
print("native vs oracle: exact", int((d == 0).sum()), "of", d.size, "diff>1e-6", int((d > 1e-6).sum()), "diff>1e-3", int((d > 1e-3).sum()), "relL2", helpers.rel_l2(li[:, :3], li_ref[:, :3]))
print("replay vs oracle: exact", int((d2 == 0).sum()), "diff>1e-6", int((d2 > 1e-6).sum()), "diff>1e-3", int((d2 > 1e-3).sum()))
bad = np.nonzero(d > 1e-3)[0][:5]
for b in bad: print(b, li[b], li_ref[b], li2[b])
