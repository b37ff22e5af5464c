import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
"""exact_ties against the lean kernels: configs[1] megakernel, Cornell box on both schedules (kernel ms)."""
for name, kw0, sched in (("bunny", dict(resolution=(512, 512), spp=256, depth=8), "megakernel"), ("cornell", dict(resolution=(512, 512), spp=64, depth=16), "megakernel"),
                         ("cornell", dict(resolution=(512, 512), spp=64, depth=16), "wavefront")):
  tr = HipPathTracer(gs.load_scene(name, gs.config_overrides(**kw0)), 0)
  film = tr.new_film()
  for label, kw in (("lean", {}), ("exact", dict(exact_ties=True))):
    best = 1e9
    for i in range(4):
        film.zero_()
        out = tr.render(film=film, seed=1, timed=True, schedule=sched, **kw)
        torch.cuda.synchronize()
        best = min(best, out["stats"]["kernel_ms"])
    print(name, sched, label, round(best, 2), flush=True)
