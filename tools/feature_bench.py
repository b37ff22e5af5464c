"""Throughput of the EXT kernels on the feature scenes (shapes / textured / masked / subsurface), both schedules where available."""
import sys, os, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from goblin_amd import _abi
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
for name in ("shapes", "textured", "masked", "subsurface", "whitted", "imagetex", "ibl", "bumpy", "volume", "hetero"):
    scene = gs.load_scene(name, gs.config_overrides(resolution=(512, 512), spp=64))
    tr = HipPathTracer(scene, 0)
    film = tr.new_film()
    for sch in ("megakernel", "wavefront"):
        try:
            best = 1e30
            for i in range(3):
                film.zero_()
                out = tr.render(film=film, seed=1, timed=True, schedule=sch)
                torch.cuda.synchronize()
                best = min(best, out["stats"]["kernel_ms"])
            print(json.dumps({"scene": name, "schedule": sch, "paths": out["paths"], "ms": round(best, 2),
                              "mpaths_s": round(out["paths"] / best / 1e3, 1), "mean": float(film.normalized().mean())}), flush=True)
        except _abi.GoblinError as e:
            print(json.dumps({"scene": name, "schedule": sch, "error": str(e)}), flush=True)
