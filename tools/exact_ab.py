"""configs[1] kernel ms: lean quad kernel, exact_ties (quad), stream sampler, then the one-ray-per-lane builds (GBL_MK_QUAD=0)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
tr = HipPathTracer(gs.load_scene("bunny", gs.config_overrides(resolution=(512, 512), spp=256, depth=8)), 0)
film = tr.new_film()
for label, kw in (("lean quad", {}), ("exact", dict(exact_ties=True)), ("stream", dict(sampler="stream"))):
    best = 1e9
    for i in range(3):
        film.zero_()
        out = tr.render(film=film, seed=1, timed=True, schedule="megakernel", **kw)
        torch.cuda.synchronize()
        best = min(best, out["stats"]["kernel_ms"])
    print(label, round(best, 2))
os.environ["GBL_MK_QUAD"] = "0"
tr2 = HipPathTracer(gs.load_scene("bunny", gs.config_overrides(resolution=(512, 512), spp=256, depth=8)), 0)
for label, kw in (("lean one-ray-per-lane", {}), ("exact one-ray-per-lane", dict(exact_ties=True))):
    best = 1e9
    for i in range(3):
        film.zero_()
        out = tr2.render(film=film, seed=1, timed=True, schedule="megakernel", **kw)
        torch.cuda.synchronize()
        best = min(best, out["stats"]["kernel_ms"])
    print(label, round(best, 2))
