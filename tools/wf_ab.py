import sys, os, subprocess, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from goblin_amd import scene as gs
from goblin_amd.renderer import HipPathTracer
row = {}
for name, kw in (("cornell", dict(resolution=(1024, 1024), spp=32, depth=16)), ("grid", dict(resolution=(1024, 1024), spp=32, depth=8)), ("bunny", dict(resolution=(512, 512), spp=256, depth=8))):
    tr = HipPathTracer(gs.load_scene(name, gs.config_overrides(**kw)), 0)
    film = tr.new_film()
    best = 1e9
    for i in range(3):
        film.zero_()
        out = tr.render(film=film, seed=1, timed=True, schedule="wavefront")
        torch.cuda.synchronize()
        best = min(best, out["stats"]["kernel_ms"])
    row[name] = round(best, 2)
print(json.dumps(row), flush=True)
''' % REPO
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "main": env["GOBLIN_HIP_LIB"] = os.path.join(REPO, "goblin_amd", "lib", "variants", "libgoblin_hip_%s.so" % lib)
    sys.stdout.write("%-10s " % lib); sys.stdout.flush()
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
