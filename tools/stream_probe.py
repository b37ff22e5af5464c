"""Phase shares of the stream sampler's tile walk (instrumented build), config 2."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ["GBL_PROBE"] = "1"
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
scene = gs.load_scene("bunny", gs.config_overrides(resolution=(512, 512), spp=256, depth=8))
tr = HipPathTracer(scene, 0)
out = tr.render(sampler="stream", stats=True, schedule="megakernel")
print(out["stats"])
