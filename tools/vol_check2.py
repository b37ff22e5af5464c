import sys, os, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import oracle_binding as ob
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
base = json.load(open(gs.scene_path("volume")))
def run(tag, vol_edit=None, lights=None):
    ov = gs.config_overrides(resolution=(24, 24), spp=4, depth=3)
    v = dict(base["volume"]); v.update(vol_edit or {})
    ov["volume"] = v
    if lights is not None: ov["lights"] = [base["lights"][i] for i in lights]
    scene = gs.load_scene("volume", ov)
    o = ob.Oracle(scene); seed = 5
    samples = o.native_samples(seed)
    li_ref, _ = o.li_replay(samples, threads=1)
    li = HipPathTracer(scene, 0).render(seed=seed, want_li=True, schedule="megakernel")["li"].cpu().numpy()
    d = np.abs(li[:, :3] - li_ref[:, :3]).max(axis=1)
    print(tag, "max diff %.3g" % d.max(), "n>1e-5", int((d > 1e-5).sum()), "of", d.size, flush=True)
run("no scattering (tr only)", {"albedo": [0.0, 0.0, 0.0]})
run("spot only", None, [1])
run("mesh area only", None, [0])
run("sphere area only", None, [2])
run("one light sample", {"sample_num": 1}, [1])
