"""The primary pass (kernels/packet.h) on / off (GBL_PRIMARY): main-kernel ms of the lean quad path kernel incl. the pass,
per-sample radiance compared.   python tools/primary_ab.py [scene ...]"""
import sys, os, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
CASES = [("bunny", gs.config_overrides(resolution=(512, 512), spp=256, depth=8)),
         ("cornell", gs.config_overrides(resolution=(512, 512), spp=64, depth=16)),
         ("grid", gs.config_overrides(resolution=(512, 512), spp=64, depth=8)),
         ("ties", None)]
only = sys.argv[1:]
for name, ov in CASES:
    if only and name not in only:
        continue
    tr = HipPathTracer(gs.load_scene(name, ov), 0)
    row = {"scene": name}
    for exact in (False, True):
        tag = "exact_" if exact else ""
        ref = None
        for mode in ("0", "1"):
            os.environ["GBL_PRIMARY"] = mode
            best = 1e30
            for i in range(4):
                out = tr.render(seed=1, timed=True, schedule="megakernel", want_li=(i == 0), exact_ties=exact)
                torch.cuda.synchronize()
                best = min(best, out["stats"]["kernel_ms"])
                if i == 0:
                    li = out["li"]
            row[tag + "primary_" + mode] = round(best, 2)
            if ref is None:
                ref = li
            else:
                row[tag + "samples_differing"] = int((li != ref).any(dim=1).sum())
                row["samples"] = int(li.shape[0])
        del ref, li
    os.environ.pop("GBL_PRIMARY", None)
    print(json.dumps(row), flush=True)
