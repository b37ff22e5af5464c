"""Plain megakernel against the quad-per-ray one (GBL_MK_QUAD, kernels/quadtrace.h) on the BASELINE and feature scenes: kernel ms, both ways."""
import sys, os, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
CASES = [("bunny", gs.config_overrides(resolution=(512, 512), spp=256, depth=8)),
         ("cornell", gs.config_overrides(resolution=(512, 512), spp=64, depth=16)),
         ("grid", gs.config_overrides(resolution=(512, 512), spp=64, depth=8))]
CASES += [(n, gs.config_overrides(resolution=(512, 512), spp=64)) for n in ("shapes", "textured", "masked", "subsurface", "imagetex", "ibl", "bumpy", "volume")]
only = sys.argv[1:]
for name, ov in CASES:
    if only and name not in only:
        continue
    tr = HipPathTracer(gs.load_scene(name, ov), 0)
    film = tr.new_film()
    row = {"scene": name}
    for mode in ("0", "1"):
        os.environ["GBL_MK_QUAD"] = mode   # "0": one ray per lane; anything else: the default quad queries
        best = 1e30
        for i in range(4):
            film.zero_()
            out = tr.render(film=film, seed=1, timed=True, schedule="megakernel")
            torch.cuda.synchronize()
            best = min(best, out["stats"]["kernel_ms"])
        row["quad" if mode == "1" else "plain"] = round(best, 2)
        row["mean_" + mode] = float(film.normalized().mean())
    row["ratio"] = round(row["quad"] / row["plain"], 3)
    print(json.dumps(row), flush=True)
