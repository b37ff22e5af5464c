"""Throughput of the bit-faithful sampler (GBL_SAMPLES_STREAM) next to the native law, BASELINE configs 1 and 2."""
import sys, os, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
for name, ov in (("bunny", gs.config_overrides(resolution=(256, 256), spp=16, depth=4)),
                 ("bunny", gs.config_overrides(resolution=(512, 512), spp=256, depth=8)),
                 ("cornell", gs.config_overrides(resolution=(256, 256), spp=64, depth=16))):
    scene = gs.load_scene(name, ov)
    tr = HipPathTracer(scene, 0)
    film = tr.new_film()
    for sampler in ("native", "stream"):
        best = 1e30
        for i in range(2):
            film.zero_()
            out = tr.render(film=film, seed=1, timed=True, schedule="megakernel", sampler=sampler)
            torch.cuda.synchronize()
            best = min(best, out["stats"]["kernel_ms"])
        print(json.dumps({"scene": name, "sampler": sampler, "paths": out["paths"], "ms": round(best, 2),
                          "mpaths_s": round(out["paths"] / best / 1e3, 1), "mean": float(film.normalized().mean())}), flush=True)
