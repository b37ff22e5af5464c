"""Bit-identity check of the megakernel regrouping experiments against the plain megakernel at several depths:
    python tools/regroup_check.py [1 = kernels/blocktrace.h | 2 = kernels/rayexchange.h]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
for depth in (1, 2, 3, 8):
    scene = gs.load_scene("bunny", gs.config_overrides(resolution=(64, 64), spp=16, depth=depth))
    os.environ.pop("GBL_MK_BLOCKTRACE", None)
    a = HipPathTracer(scene, 0).render(seed=3, want_li=True, schedule="megakernel")["li"].cpu().numpy()
    os.environ["GBL_MK_BLOCKTRACE"] = sys.argv[1] if len(sys.argv) > 1 else "1"
    b = HipPathTracer(scene, 0).render(seed=3, want_li=True, schedule="megakernel")["li"].cpu().numpy()
    bad = np.flatnonzero(np.any(a != b, axis=1))
    print("depth", depth, "differ", bad.size, "of", a.shape[0], "mean", a[:, :3].mean(), b[:, :3].mean())
    for i in bad[:6]:
        print("   ", i, a[i], b[i])
