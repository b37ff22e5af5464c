"""Compare libgoblin_hip build variants (GOBLIN_HIP_LIB) on the megakernel: one subprocess per variant."""
import sys, os, subprocess, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, torch, os
SCHED = os.environ.get('GBL_SCHED', 'megakernel')
sys.path.insert(0, %r)
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
for sc_name, ov in [("bunny", gs.config_overrides(resolution=(512, 512), spp=256, depth=8)),
                    ("cornell", gs.config_overrides(resolution=(512, 512), spp=64, depth=16)),
                    ("grid", gs.config_overrides(resolution=(512, 512), spp=64, depth=8))]:
    tr = HipPathTracer(gs.load_scene(sc_name, ov), 0)
    film = tr.new_film()
    best = 1e30
    for i in range(3):
        film.zero_()
        out = tr.render(film=film, seed=1, stats=False, timed=True, schedule=SCHED)
        torch.cuda.synchronize()
        best = min(best, out["stats"]["kernel_ms"])
    print(json.dumps({"scene": sc_name, "schedule": SCHED, "ms": round(best, 2), "mean": float(film.normalized().mean())}), flush=True)
''' % REPO
for lib in sys.argv[1:]:
    env = dict(os.environ, GOBLIN_HIP_LIB=os.path.join(REPO, "goblin_amd", "lib", lib))
    print("==", lib, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
