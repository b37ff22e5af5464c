"""Wave-pool schedule against the megakernel: per-sample radiance must be bit-identical; then timings.

    python tools/wp_check.py [small|full] [config ...]
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer

mode = sys.argv[1] if len(sys.argv) > 1 else "small"
CONFIGS = {
    "bunny_small": ("bunny", gs.config_overrides(resolution=(64, 64), spp=16, depth=5)),
    "cornell_small": ("cornell", gs.config_overrides(resolution=(64, 64), spp=16, depth=8)),
    "grid_small": ("grid", gs.config_overrides(resolution=(96, 96), spp=16, depth=6)),
    "cfg2": ("bunny", gs.config_overrides(resolution=(512, 512), spp=256, depth=8)),
    "cfg3_64spp": ("cornell", gs.config_overrides(resolution=(1024, 1024), spp=64, depth=16)),
    "cfg4": ("grid", gs.config_overrides(resolution=(1024, 1024), spp=256, depth=8)),
}
names = [a for a in sys.argv[2:] if not a.startswith("--")] or (["bunny_small", "cornell_small", "grid_small"] if mode == "small" else ["cfg2", "cfg3_64spp", "cfg4"])
for name in names:
    sc_name, ov = CONFIGS[name]
    scene = gs.load_scene(sc_name, ov)
    tr = HipPathTracer(scene, 0)
    ref = None
    for sch in ("megakernel", "wavepool", "wavefront"):
        if mode != "small" and sch == "wavefront" and "--wf" not in sys.argv:
            continue
        film = tr.new_film()
        out = tr.render(film=film, seed=7, want_li=(mode == "small" or "--li" in sys.argv), timed=True, schedule=sch)
        torch.cuda.synchronize()
        ms = [out["stats"]["kernel_ms"]]
        for _ in range(2 if mode != "small" else 0):
            film.zero_()
            o2 = tr.render(film=film, seed=7, timed=True, schedule=sch)
            torch.cuda.synchronize()
            ms.append(o2["stats"]["kernel_ms"])
        rec = {"config": name, "schedule": sch, "paths": out["paths"], "ms": [round(m, 2) for m in ms],
               "mpaths_s": round(out["paths"] / min(ms) / 1e3, 1), "mean": float(film.normalized().mean())}
        if out["li"] is not None:
            li = out["li"].cpu().numpy()
            if ref is None:
                ref = li
            else:
                same = np.array_equal(li.view(np.uint32), ref.view(np.uint32))
                rec["li_bit_identical_with_megakernel"] = bool(same)
                if not same:
                    diff = np.any(li.view(np.uint32) != ref.view(np.uint32), axis=1)
                    rec["differing_samples"] = int(diff.sum())
                    idx = np.nonzero(diff)[0][:5]
                    rec["first"] = [(int(i), li[i].tolist(), ref[i].tolist()) for i in idx]
        print(json.dumps(rec), flush=True)
        if "--probe" in sys.argv:   # instrumented launch: GBL_PROBE=1 prints lane utilisation to stderr
            film.zero_()
            o3 = tr.render(film=film, seed=7, stats=True, schedule=sch)
            torch.cuda.synchronize()
            st = o3["stats"]
            print(json.dumps({"config": name, "schedule": sch, "instrumented_ms": round(st["kernel_ms"], 2),
                              "ext": st["extension_rays"], "shadow": st["shadow_rays"], "nodes": st["nodes"], "tris": st["tris"]}), flush=True)
    del tr
