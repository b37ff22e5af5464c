"""Bit-identity of the quad-per-ray megakernel (GBL_MK_QUAD=1, kernels/quadtrace.h) against the plain one, and its timing."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
for name, ov in (("bunny", gs.config_overrides(resolution=(96, 96), spp=16, depth=8)), ("cornell", gs.config_overrides(resolution=(48, 48), spp=16, depth=12)),
                 ("grid", gs.config_overrides(resolution=(64, 64), spp=4, depth=5)), ("masked", gs.config_overrides(resolution=(48, 48), spp=9, depth=5)), ("shapes", gs.config_overrides(resolution=(48, 48), spp=9, depth=5)),
                 ("bumpy", gs.config_overrides(resolution=(48, 48), spp=9, depth=5))):
    scene = gs.load_scene(name, ov)
    os.environ["GBL_MK_QUAD"] = "0"
    a = HipPathTracer(scene, 0).render(seed=3, want_li=True, schedule="megakernel")["li"].cpu().numpy()
    os.environ["GBL_MK_QUAD"] = "1"
    b = HipPathTracer(scene, 0).render(seed=3, want_li=True, schedule="megakernel")["li"].cpu().numpy()
    os.environ["GBL_MK_QUAD"] = "2"
    c2 = HipPathTracer(scene, 0).render(seed=3, want_li=True, schedule="megakernel")["li"].cpu().numpy()
    print(name, "differ", int(np.any(a != b, axis=1).sum()), "parked form", int(np.any(a != c2, axis=1).sum()), "of", a.shape[0], flush=True)
for name, ov in (("bunny", gs.config_overrides(resolution=(96, 96), spp=4, method="ao", ao_samples=9)), ("grid", gs.config_overrides(resolution=(64, 64), spp=4, method="ao", ao_samples=9)),
                 ("shapes", gs.config_overrides(resolution=(48, 48), spp=4, method="ao", ao_samples=9))):
    scene = gs.load_scene(name, ov)
    os.environ["GBL_MK_QUAD"] = "0"
    a = HipPathTracer(scene, 0).render(seed=3, want_li=True)["li"].cpu().numpy()
    os.environ["GBL_MK_QUAD"] = "1"
    b = HipPathTracer(scene, 0).render(seed=3, want_li=True)["li"].cpu().numpy()
    print("ao", name, "differ", int(np.any(a != b, axis=1).sum()), "of", a.shape[0], flush=True)
for sc_name in ("bunny", "grid"):
    tr = HipPathTracer(gs.load_scene(sc_name, gs.config_overrides(resolution=(1024, 1024), spp=16, method="ao", ao_samples=25)), 0)
    film = tr.new_film()
    for mode in ("0", "1"):
        os.environ["GBL_MK_QUAD"] = mode
        best = 1e9
        for i in range(3):
            film.zero_()
            out = tr.render(film=film, seed=1, timed=True)
            torch.cuda.synchronize()
            best = min(best, out["stats"]["kernel_ms"])
        print("ao %s 1024^2 x 16 spp x 25 rays GBL_MK_QUAD=%s: %.2f ms" % (sc_name, mode, best), flush=True)
scene = gs.load_scene("bunny", gs.config_overrides(resolution=(512, 512), spp=256, depth=8))
for mode in ("0", "1", "2"):
    os.environ["GBL_MK_QUAD"] = mode
    tr = HipPathTracer(scene, 0)
    film = tr.new_film()
    best = 1e9
    for i in range(4):
        film.zero_()
        out = tr.render(film=film, seed=1, timed=True, schedule="megakernel")
        torch.cuda.synchronize()
        best = min(best, out["stats"]["kernel_ms"])
    print("config[1] GBL_MK_QUAD=%s: %.2f ms" % (mode, best), flush=True)
