"""Debug: medium terms, device vs oracle (hashed draws)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import oracle_binding as ob
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
scene = gs.load_scene("volume", gs.config_overrides(resolution=(24, 24), spp=4, depth=3))
o = ob.Oracle(scene)
seed = 5
samples = o.native_samples(seed)
li_ref, _ = o.li_replay(samples, threads=1)
tr = HipPathTracer(scene, 0)
li = tr.render(seed=seed, want_li=True, schedule="megakernel")["li"].cpu().numpy()
d = np.abs(li[:, :3] - li_ref[:, :3]).max(axis=1)
print("max diff", d.max(), "n>1e-6", int((d > 1e-6).sum()), "of", d.size)
for i in np.argsort(-d)[:6]:
    print(i, li[i], li_ref[i])
out = tr.render(seed=seed, want_li=True, schedule="megakernel", stats=True)
print("stats", out["stats"])
plain = gs.load_scene("volume", dict(gs.config_overrides(resolution=(24, 24), spp=4, depth=3), volume={"type": "homogeneous", "attenuation": [0.0, 0.0, 0.0],
                                     "albedo": [0.0, 0.0, 0.0], "box_min": [-1.0, -1.0, -1.0], "box_max": [1.0, 1.0, 1.0], "sample_num": 3}))
li0 = HipPathTracer(plain, 0).render(seed=seed, want_li=True, schedule="megakernel")["li"].cpu().numpy()
print("device li without medium vs with:", li0[46], li[46], "oracle", li_ref[46])
print("scene volume desc:", scene.desc.volume.type, list(scene.desc.volume.attenuation), scene.desc.volume.sample_num)
