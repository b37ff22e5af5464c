"""Debug: stream-mode film vs the oracle's whole-render film at a given size."""
import sys, os, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import oracle_binding as ob
from goblin_amd import scene as gs, _abi
from goblin_amd.renderer import HipPathTracer
res = int(sys.argv[1]); spp = int(sys.argv[2]); depth = int(sys.argv[3])
scene = gs.load_scene("bunny", gs.config_overrides(resolution=(res, res), spp=spp, depth=depth))
o = ob.Oracle(scene)
ref = o.render(threads=8)["film"]
tr = HipPathTracer(scene, 0)
film = tr.render(sampler="stream")["film"].numpy()
d = np.abs(film[..., 3] - ref[..., 3])
print("fresh tracer: weight max diff", d.max(), "bad pixels", int((d > 1e-3).sum()), "of", d.size)
ys, xs = np.nonzero(d > 1e-3)
if len(ys): print("bad bbox x", xs.min(), xs.max(), "y", ys.min(), ys.max())
big = gs.load_scene("bunny", gs.config_overrides(resolution=(res, res), spp=256, depth=depth))
tr2 = HipPathTracer(big, 0)
s = _abi.gbl_render_setting.from_buffer_copy(big.desc.setting); s.sample_per_pixel = spp
film2 = tr2.render(setting=s, sampler="stream")["film"].numpy()
d2 = np.abs(film2[..., 3] - ref[..., 3])
print("setting override: weight max diff", d2.max(), "bad pixels", int((d2 > 1e-3).sum()))
