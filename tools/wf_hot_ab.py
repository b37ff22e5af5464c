"""Wavefront schedule with the top of the tree in the trace kernels' LDS (GBL_WF_HOT nodes, trace.h HotSplitStack): step ms on
the Cornell box / grid / bunny, per-sample radiance compared with the table off.   python tools/wf_hot_ab.py [sizes]"""
import sys, os, json, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
CASES = [("cornell", gs.config_overrides(resolution=(1024, 1024), spp=64, depth=16)),
         ("grid", gs.config_overrides(resolution=(1024, 1024), spp=64, depth=8)),
         ("bunny", gs.config_overrides(resolution=(512, 512), spp=256, depth=8))]
SIZES = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "64", "128", "256"]
for name, ov in CASES:
    tr = HipPathTracer(gs.load_scene(name, ov), 0)
    film = tr.new_film()
    ref = None
    row = {"scene": name}
    for size in SIZES:
        os.environ["GBL_WF_HOT"] = size
        best = 1e30
        for i in range(3):
            film.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = tr.render(film=film, seed=1, schedule="wavefront", want_li=(i == 0))
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
            if i == 0:
                li = out["li"]
        if ref is None:
            ref = li
        row["hot_" + size] = round(best, 2)
        row["same_" + size] = bool(torch.equal(li, ref))
    os.environ.pop("GBL_WF_HOT", None)
    print(json.dumps(row), flush=True)
