import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
os.environ["GBL_WP_ALLOW_IMAGES"] = "1"
for name in ("imagetex", "ibl", "bumpy", "textured"):
    scene = gs.load_scene(name, gs.config_overrides(resolution=(96, 96), spp=16, depth=5))
    r = HipPathTracer(scene, 0)
    a = r.render(seed=3, want_li=True, schedule="megakernel")["li"].cpu().numpy()
    try:
        b = r.render(seed=3, want_li=True, schedule="wavepool")["li"].cpu().numpy()
    except Exception as e:
        print(name, "refused:", str(e)[:80])
        continue
    print(name, "differ", int(np.any(a != b, axis=1).sum()), "of", a.shape[0])
