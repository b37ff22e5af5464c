"""AUTO's calibration cases: megakernel / wavefront / auto kernel ms (512^2 x 64 spp) and the rays per path a 1-spp pilot sees."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goblin_amd import scene as gs
from goblin_amd import _abi
from goblin_amd.renderer import HipPathTracer
CASES = [("bunny", 8), ("bunny", 12), ("bunny", 16), ("cornell", 4), ("cornell", 6), ("cornell", 8), ("cornell", 12), ("cornell", 16), ("grid", 5), ("grid", 8),
         ("shapes", 6), ("textured", 6), ("imagetex", 6), ("ibl", 6), ("bumpy", 6), ("subsurface", 6)]
for name, d in CASES:
    tr = HipPathTracer(gs.load_scene(name, gs.config_overrides(resolution=(512, 512), spp=64, depth=d)), 0)
    film = tr.new_film()
    row = {"scene": name, "depth": d}
    s1 = _abi.gbl_render_setting.from_buffer_copy(tr.scene.desc.setting)
    s1.sample_per_pixel = 1
    st = tr.render(setting=s1, seed=1, stats=True, schedule="megakernel")["stats"]
    row["pilot_rays_per_path"] = round((st["extension_rays"] + st["shadow_rays"]) / st["paths"], 2)
    for sch in ("megakernel", "wavefront", "auto"):
        best = 1e30
        for i in range(3):
            film.zero_()
            out = tr.render(film=film, seed=1, timed=True, schedule=sch, stats=False)
            torch.cuda.synchronize()
            best = min(best, out["stats"]["kernel_ms"])
        row[sch] = round(best, 2)
    row["wf_over_mk"] = round(row["wavefront"] / row["megakernel"], 3)
    print(json.dumps(row), flush=True)
