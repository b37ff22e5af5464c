"""One rank's share of BASELINE configs[3] under strong scaling (tiles t = r mod N of the frame): megakernel against wavefront, kernel ms."""
import sys, os, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
tr = HipPathTracer(gs.load_scene("grid", gs.config_overrides(resolution=(1024, 1024), spp=256, depth=8)), 0)
film = tr.new_film()
for n in (1, 2, 4, 8):
    row = {"ranks": n}
    for sch in ("megakernel", "wavefront", "auto"):
        best = 1e30
        for i in range(2):
            film.zero_()
            out = tr.render(film=film, seed=1, timed=True, schedule=sch, shard=(0, n))
            torch.cuda.synchronize()
            best = min(best, out["stats"]["kernel_ms"])
        row[sch] = round(best, 2)
    row["paths"] = out["paths"]
    print(json.dumps(row), flush=True)
