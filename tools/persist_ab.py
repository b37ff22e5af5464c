"""The lean path kernel with persistent traversal (kernels/persist.h) against the query-per-iteration kernel (GBL_PERSIST=0), over
GBL_PERSIST_WAIT:   python tools/persist_ab.py [scene] [wait ...]"""
import sys, os, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
CASES = {"bunny": gs.config_overrides(resolution=(512, 512), spp=256, depth=8),
         "cornell": gs.config_overrides(resolution=(512, 512), spp=64, depth=16),
         "grid": gs.config_overrides(resolution=(512, 512), spp=64, depth=8)}
name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
waits = [x for x in sys.argv[2:]] or ["24:16", "32:16", "48:16", "48:32"]   # wait:switch
tr = HipPathTracer(gs.load_scene(name, CASES[name]), 0)


def run(env):
    for k, v in env.items():
        os.environ[k] = v
    best, li = 1e30, None
    for i in range(4):
        out = tr.render(seed=1, timed=True, schedule="megakernel", want_li=(i == 0))
        torch.cuda.synchronize()
        best = min(best, out["stats"]["kernel_ms"])
        if i == 0:
            li = out["li"]
    for k in env:
        if k != "GBL_PERSIST":
            os.environ.pop(k, None)
    return round(best, 2), li


base, ref = run({"GBL_PERSIST": "0"})
os.environ["GBL_PERSIST"] = "1"
print(json.dumps({"scene": name, "query_per_iteration_ms": base}), flush=True)
for w in waits:
    a, b, c = (w.split(":") + ["16", "24"])[:3]
    ms, li = run({"GBL_PERSIST_WAIT": a, "GBL_PERSIST_SWITCH": b, "GBL_PERSIST_TH": c})
    print(json.dumps({"wait": w, "ms": ms, "samples_differing": int((li != ref).any(dim=1).sum())}), flush=True)
    del li
