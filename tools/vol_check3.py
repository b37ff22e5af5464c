import sys, os, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import oracle_binding as ob
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
base = json.load(open(gs.scene_path("volume")))
for tag, lights in (("mesh area only", [0]), ("sphere area only", [2]), ("all three", [0, 1, 2])):
    ov = gs.config_overrides(resolution=(64, 64), spp=16, depth=3)
    ov["lights"] = [base["lights"][i] for i in lights]
    scene = gs.load_scene("volume", ov)
    o = ob.Oracle(scene); seed = 9
    samples = o.native_samples(seed)
    li_ref, _ = o.li_replay(samples, threads=8)
    li = HipPathTracer(scene, 0).render(seed=seed, want_li=True, schedule="megakernel")["li"].cpu().numpy()
    d = np.abs(li[:, :3] - li_ref[:, :3]).max(axis=1)
    print(tag, "mean dev %.5f ref %.5f" % (li[:, :3].mean(), li_ref[:, :3].mean()), "differing %.3f" % (d > 1e-5).mean(), flush=True)
