"""Traversal stack entries per scene (GBL_PROBE prints them at gbl_create): exact BLAS need against the per-level bound."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ["GBL_PROBE"] = "1"
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
for name in sys.argv[1:] or ["bunny", "cornell", "grid", "shapes", "masked"]:
    print("==", name, flush=True)
    HipPathTracer(gs.load_scene(name, gs.config_overrides(resolution=(64, 64), spp=1)), 0)
