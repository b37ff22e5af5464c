"""The top of the tree in LDS (trace.h HotLdsStack): kernel ms of the lean quad kernels by the number of nodes kept there
(GBL_HOT_LDS: 0 = none, unset = what fits beside the stacks at three workgroups per CU), per-sample radiance compared with the
table off.    python tools/hot_ab.py [scene ...]"""
import sys, os, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
CASES = [("bunny", gs.config_overrides(resolution=(512, 512), spp=256, depth=8), {}),
         ("cornell", gs.config_overrides(resolution=(512, 512), spp=64, depth=16), {}),
         ("grid", gs.config_overrides(resolution=(512, 512), spp=64, depth=8), {}),
         ("bunny_ao", gs.config_overrides(resolution=(1024, 1024), spp=16, method="ao", ao_samples=25), {})]
SIZES = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "", "32", "64", "256", "512"]
only = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] != "all" else None
for name, ov, kw in CASES:
    if only and name not in only:
        continue
    tr = HipPathTracer(gs.load_scene(name.split("_")[0], ov), 0)
    ref = None
    row = {"scene": name}
    for size in SIZES:
        if size == "":
            os.environ.pop("GBL_HOT_LDS", None)
        else:
            os.environ["GBL_HOT_LDS"] = size
        best = 1e30
        for i in range(4):
            out = tr.render(seed=1, timed=True, schedule="megakernel", want_li=(i == 0), **kw)
            torch.cuda.synchronize()
            best = min(best, out["stats"]["kernel_ms"])
            if i == 0:
                li = out["li"]
        if ref is None:
            ref = li
        row["hot_" + (size or "fit")] = round(best, 2)
        row["same_" + (size or "fit")] = bool(torch.equal(li, ref))
    os.environ.pop("GBL_HOT_LDS", None)
    print(json.dumps(row), flush=True)
