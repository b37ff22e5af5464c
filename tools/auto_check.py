import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
for name, d in (("bunny", 8), ("bunny", 12), ("bunny", 16), ("cornell", 6), ("cornell", 8), ("cornell", 12), ("cornell", 16), ("grid", 5), ("grid", 8)):
    tr = HipPathTracer(gs.load_scene(name, gs.config_overrides(resolution=(512, 512), spp=64, depth=d)), 0)
    film = tr.new_film()
    row = {"scene": name, "depth": d}
    for sch in ("megakernel", "wavefront", "auto"):
        best = 1e30
        for i in range(3):
            film.zero_()
            out = tr.render(film=film, seed=1, timed=True, schedule=sch, stats=False)
            torch.cuda.synchronize()
            best = min(best, out["stats"]["kernel_ms"])
        row[sch] = round(best, 2)
    print(json.dumps(row), flush=True)
