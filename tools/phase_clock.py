"""Where the lean quad megakernel's wave time goes (measurement build, s_memtime around the phases of its two queries):
    python tools/build_variant.py phase kernels_quad -DGBL_PHASE_CLOCK   (here)
    GOBLIN_HIP_LIB=goblin_amd/lib/variants/libgoblin_hip_phase.so GBL_PHASE_CLOCK=1 python tools/phase_clock.py [scene res spp depth]   (GPU box)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("GBL_PHASE_CLOCK", "1")
import torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
a = sys.argv[1:]
name = a[0] if len(a) > 0 else "bunny"
res = int(a[1]) if len(a) > 1 else 512
spp = int(a[2]) if len(a) > 2 else 256
depth = int(a[3]) if len(a) > 3 else 8
extra = {"method": "ao", "ao_samples": 25} if len(a) > 4 and a[4] == "ao" else {}
tr = HipPathTracer(gs.load_scene(name, gs.config_overrides(resolution=(res, res), spp=spp, depth=depth, **extra)), 0)
for i in range(2):
    out = tr.render(seed=1, timed=True, schedule="megakernel")
    torch.cuda.synchronize()
    print(name, res, spp, depth, "kernel ms %.2f" % out["stats"]["kernel_ms"], flush=True)
