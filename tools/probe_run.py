"""SIMD-efficiency probes of one render (instrumented kernel build): GBL_PROBE=1 python tools/probe_run.py [scene] [res] [spp] [depth] [schedule]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("GBL_PROBE", "1")
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
a = sys.argv[1:]
name = a[0] if len(a) > 0 else "bunny"
res = int(a[1]) if len(a) > 1 else 512
spp = int(a[2]) if len(a) > 2 else 256
depth = int(a[3]) if len(a) > 3 else 8
schedule = a[4] if len(a) > 4 else "megakernel"
tr = HipPathTracer(gs.load_scene(name, gs.config_overrides(resolution=(res, res), spp=spp, depth=depth)), 0)
out = tr.render(seed=1, stats=True, schedule=schedule)
print(out["stats"])
