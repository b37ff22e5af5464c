"""One scene rendered a few times under the megakernel schedule (for `rocprofv3 --kernel-trace --stats -- python3 tools/render_once.py ...`):
    python tools/render_once.py [scene res spp depth [exact]]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
a = sys.argv[1:]
name = a[0] if len(a) > 0 else "bunny"
res = int(a[1]) if len(a) > 1 else 512
spp = int(a[2]) if len(a) > 2 else 256
depth = int(a[3]) if len(a) > 3 else 8
exact = len(a) > 4 and a[4] == "exact"
tr = HipPathTracer(gs.load_scene(name, gs.config_overrides(resolution=(res, res), spp=spp, depth=depth)), 0)
for i in range(3):
    out = tr.render(seed=1, timed=True, schedule="megakernel", exact_ties=exact)
    torch.cuda.synchronize()
    print(name, "kernel ms %.2f" % out["stats"]["kernel_ms"], flush=True)
