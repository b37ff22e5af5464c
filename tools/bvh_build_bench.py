"""Host SAH build vs device LBVH build: construction time (gbl_info.build_ms) and trace time, per scene."""
import sys, os, json, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
for sc_name, ov in [("bunny", gs.config_overrides(resolution=(512, 512), spp=64, depth=8)),
                    ("grid", gs.config_overrides(resolution=(512, 512), spp=64, depth=8))]:
    scene = gs.load_scene(sc_name, ov)
    for bvh in ("host", "device", "host", "device"):
        t0 = time.time()
        tr = HipPathTracer(scene, 0, bvh=bvh)
        wall = time.time() - t0
        film = tr.new_film()
        best = 1e30
        for i in range(3):
            film.zero_()
            out = tr.render(film=film, seed=1, timed=True, schedule="megakernel")
            torch.cuda.synchronize()
            best = min(best, out["stats"]["kernel_ms"])
        print(json.dumps({"scene": sc_name, "bvh": bvh, "build_ms": round(tr.info.build_ms, 2), "create_wall_ms": round(wall * 1e3, 1),
                          "triangles": tr.info.triangles, "blas_nodes": tr.info.blas_nodes, "blas_depth": tr.info.blas_depth,
                          "trace_ms": round(best, 2), "mean": float(film.normalized().mean())}), flush=True)
        del tr
