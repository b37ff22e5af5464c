"""AO throughput of libgoblin_hip build variants (GOBLIN_HIP_LIB): one subprocess per variant."""
import sys, os, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
for sc_name in ("bunny", "grid"):
    tr = HipPathTracer(gs.load_scene(sc_name, gs.config_overrides(resolution=(1024, 1024), spp=16, method="ao", ao_samples=25)), 0)
    film = tr.new_film()
    best = 1e30
    for i in range(3):
        film.zero_()
        out = tr.render(film=film, seed=1, stats=False, timed=True)
        torch.cuda.synchronize()
        best = min(best, out["stats"]["kernel_ms"])
    print(json.dumps({"scene": sc_name, "ms": round(best, 2), "mpaths_s": round(out["paths"] / best / 1e3, 1), "mean": float(film.normalized().mean())}), flush=True)
''' % REPO
for lib in sys.argv[1:]:
    env = dict(os.environ, GOBLIN_HIP_LIB=os.path.join(REPO, "goblin_amd", "lib", lib))
    print("==", lib, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
