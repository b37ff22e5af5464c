"""exact_ties renders of a whole frame: quad megakernel / one-ray-per-lane megakernel / wavefront against one another, and the
   differing samples' pixels against the oracle.   python tools/exact_hunt.py cornell 1024 64 16"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import oracle_binding as ob
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
name, res, spp, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
seed = 20261003
scene = gs.load_scene(name, gs.config_overrides(resolution=(res, res), spp=spp, depth=depth))
r = HipPathTracer(scene, 0)
x0, x1, y0, y1 = r.window
W = x1 - x0
wf = r.render(seed=seed, want_li=True, schedule="wavefront", exact_ties=True)["li"]
mkq = r.render(seed=seed, want_li=True, schedule="megakernel", exact_ties=True)["li"]
os.environ["GBL_MK_QUAD"] = "0"
mk1 = r.render(seed=seed, want_li=True, schedule="megakernel", exact_ties=True)["li"]
del os.environ["GBL_MK_QUAD"]
lean = r.render(seed=seed, want_li=True, schedule="megakernel")["li"]
print("lean vs wf", int((wf != lean).any(dim=1).sum()), " lean vs mkq", int((lean != mkq).any(dim=1).sum()),
      " mkq wrong where lean is right", int(((wf != mkq).any(dim=1) & ~(lean != wf).any(dim=1)).sum()))
print("wf vs mk1", int((wf != mk1).any(dim=1).sum()), " wf vs mkq", int((wf != mkq).any(dim=1).sum()), "of", wf.shape[0])
bad = torch.nonzero((wf != mkq).any(dim=1)).flatten().cpu().numpy()
o = ob.Oracle(scene)
for i in bad[:6]:
    pix, s = divmod(int(i), spp)
    py, px = divmod(pix, W)
    sub = (x0 + px, x0 + px + 1, y0 + py, y0 + py + 1)
    ref, _ = o.li_replay(o.native_samples(seed, window=sub), threads=1)
    print("pixel", (px, py), "sample", s, "wf", wf[i, :3].tolist(), "mkq", mkq[i, :3].tolist(), "lean", lean[i, :3].tolist(), "oracle", ref[s, :3].tolist())
