"""Quick throughput probe for the other BASELINE configs (not the headline bench)."""
import sys, os, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer

CONFIGS = {
    "cfg1_bunny_256_16_d4": ("bunny", gs.config_overrides(resolution=(256, 256), spp=16, depth=4)),
    "cfg2_bunny_512_256_d8": ("bunny", gs.config_overrides(resolution=(512, 512), spp=256, depth=8)),
    "cfg3_cornell_1024_1024_d16": ("cornell", gs.config_overrides(resolution=(1024, 1024), spp=1024, depth=16)),
    "cfg4_grid_1024_256_d8": ("grid", gs.config_overrides(resolution=(1024, 1024), spp=256, depth=8)),
    "cfg5_ao_2048_64": ("bunny", gs.config_overrides(resolution=(2048, 2048), spp=64, method="ao", ao_samples=25)),
}
which = sys.argv[1:] or list(CONFIGS)
for name in which:
    sc_name, ov = CONFIGS[name]
    scene = gs.load_scene(sc_name, ov)
    t0 = time.time(); tr = HipPathTracer(scene, 0); build_s = time.time() - t0
    film = tr.new_film()
    for sch in (["megakernel", "wavefront"] if scene.desc.setting.integrator == 0 else ["auto"]):
        film.zero_()
        out = tr.render(film=film, seed=1, stats=False, timed=True, schedule=sch)
        torch.cuda.synchronize()
        ms = out["stats"]["kernel_ms"]
        print(json.dumps({"config": name, "schedule": sch, "paths": out["paths"], "ms": round(ms, 2),
                          "mpaths_s": round(out["paths"] / ms / 1e3, 1), "create_s": round(build_s, 2),
                          "scene_MB": round(tr.info.scene_bytes / 1e6, 1),
                          "mean": float(film.normalized().mean())}), flush=True)
    del tr
